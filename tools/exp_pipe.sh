python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or witness or pool_refresh or fused" 2>&1 | tail -3
python -m pytest tests/test_gpu_agents.py -x -q -k "deferred" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], d['parity_check']['rs_mismatch'], d['parity_check']['status_mismatch'], d['parity_check']['mask_mismatch'], d['parity_check']['f32_value_mismatch'])"; }
run both A=1
run no_caller HOPE_PIPE_CALLER=0
run no_postsearch HOPE_POST_SEARCH=0
run neither HOPE_PIPE_CALLER=0 HOPE_POST_SEARCH=0
run nopipe HOPE_PIPE=0
run both_p6 HOPE_SIDE_PERM=3,2,1,7,5,6,4
run both_old HOPE_SIDE_PERM=1,2,3,4,5,6,7
