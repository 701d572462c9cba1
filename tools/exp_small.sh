timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or overlap_modes" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_agents.py -x -q -k "deferred or config4 or config5" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --repeat-passes 3 $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for NS in 4096 8192 16384 32768; do
EXTRA="--scenes $NS" run pipe_$NS A=1
EXTRA="--scenes $NS" run nopipe_$NS HOPE_PIPE=0
done
EXTRA="" run s65536 A=1
