mkdir -p gpurun_out/s3
run() { tag=$1; sc=$2; shift; shift; env "$@" timeout 300 python bench.py --scenes $sc --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', $sc, round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for i in 1 2; do
for sc in 65536 8192; do
run default $sc A=1
run devkernarg0 $sc HIP_FORCE_DEV_KERNARG=0
run devkernarg1 $sc HIP_FORCE_DEV_KERNARG=1
run noscratchreclaim $sc HSA_NO_SCRATCH_RECLAIM=1
run activewait $sc ROC_ACTIVE_WAIT_TIMEOUT=100
done
done
timeout 300 python tools/config1_latency.py
