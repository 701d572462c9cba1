mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 > gpurun_out/e_$tag.json 2>/dev/null; python -c "
import json,sys; d=json.load(open('gpurun_out/e_$tag.json')); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
run base A=1
run front1 HOPE_PRIO_FRONT=1
run front2 HOPE_PRIO_FRONT=2
run front3 HOPE_PRIO_FRONT=3
run obsw16 HOPE_OBS_WPC1=16
run obsw20 HOPE_OBS_WPC1=20
run obsw24 HOPE_OBS_WPC1=24
run prio1 HOPE_PRIO=1
run prio2 HOPE_PRIO=2
run prio3 HOPE_PRIO=3
run prio2_front2 HOPE_PRIO=2 HOPE_PRIO_FRONT=2
run rsprio0 HOPE_RS_PRIO=0
run rsprio0_front2 HOPE_RS_PRIO=0 HOPE_PRIO_FRONT=2
