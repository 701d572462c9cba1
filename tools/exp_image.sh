mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --image --steps 20 --warmup 5 --preroll 150 --no-cpu-baseline --witness 0 --repeat-passes 2 --repeat-steps 60 > gpurun_out/i_$tag.json 2>/dev/null; python -c "
import json,sys; d=json.load(open('gpurun_out/i_$tag.json')); print('$tag', round(d['ms_per_step'],4), round(d['value']/1e6,2), [round(x,4) for x in d['repeat']['ms_per_step']], {k:round(v,3) for k,v in d['roofline']['ms_per_bench_step_by_kernel'].items() if v})"; }
run base A=1
python tools/bev_probe.py 2>/dev/null | tail -12
