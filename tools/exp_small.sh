mkdir -p gpurun_out; R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for NS in 8192 4096; do
rm -rf /tmp/tls
rocprofv3 --kernel-trace --output-format csv -d /tmp/tls -- python $R/bench.py --scenes $NS --steps 8 --warmup 8 --preroll 60 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
python $R/tools/timeline.py /tmp/tls 3 > $O/tl_small_$NS.txt 2>&1
done
cd $R
for NS in 4096 8192 16384; do
python bench.py --scenes $NS --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$NS plain', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"
python bench.py --scenes $NS --graph --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$NS graph', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"
done
