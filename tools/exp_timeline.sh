mkdir -p gpurun_out; R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl /tmp/tl2
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --rs-join joined --steps 8 --warmup 8 --preroll 100 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
{ python $R/tools/timeline.py /tmp/tl 3; python $R/tools/timeline.py /tmp/tl 2; } > $O/tl_joined.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -- python $R/bench.py --steps 8 --warmup 8 --preroll 100 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
{ python $R/tools/timeline.py /tmp/tl2 3; python $R/tools/timeline.py /tmp/tl2 2; } > $O/tl_deferred.txt 2>&1
cd $R
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"
