run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], {k:round(v,3) for k,v in d['roofline']['ms_per_bench_step_by_kernel'].items() if v})"; }
EXTRA="--scenes 4096" run s4096 A=1
EXTRA="--scenes 8192" run s8192 A=1
EXTRA="--scenes 16384" run s16384 A=1
EXTRA="" run s65536 A=1
EXTRA="" run s65536 A=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or auto_reset or f32_outputs" 2>&1 | tail -2
