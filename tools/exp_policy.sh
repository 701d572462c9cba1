mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py --policy hope --algo rollout --no-cpu-baseline "$@" 2>gpurun_out/p_$tag.err | tail -1 > gpurun_out/p_$tag.json; python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/p_$tag.json')); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e6,3), 'M', d['config'].get('policy_graph_replays'))
except Exception as e: print('$tag', 'FAILED', e, open('gpurun_out/p_$tag.err').read()[-600:])"; }
run c4_eager --scenes 8192 --image
run c4_graph --scenes 8192 --image --policy-fast graph
run c4_graphamp --scenes 8192 --image --policy-fast graph+amp
run c4_graphamp_img --scenes 8192 --image --policy-fast graph+amp --policy-amp
run n64k_eager
run n64k_graph --policy-fast graph
run n64k_amp --policy-fast amp
run n64k_graphamp --policy-fast graph+amp
