run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], {k:round(v,3) for k,v in d['roofline']['ms_per_bench_step_by_kernel'].items() if v})"; }
run base A=1
run no_lidar_pairs HOPE_DEBUG_STAGES=0x1000
run no_mask HOPE_DEBUG_STAGES=0x2000
run no_both HOPE_DEBUG_STAGES=0x3000
run rs_first_path_only HOPE_RS_DEBUG=0x8000
python bench.py --stages norss --steps 30 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('norss', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"
