python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
python tools/soak.py --seeds 6 --scenes 4096 --steps 12 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], d['parity_check']['rs_mismatch'], d['parity_check']['status_mismatch'], d['parity_check']['mask_mismatch'], d['parity_check']['f32_value_mismatch'])"; }
run cull A=1
run nocull HOPE_DEBUG_STAGES=0x4000
run cull A=1
