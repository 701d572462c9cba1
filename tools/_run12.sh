set -x
cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 8 --warmup 8 --preroll 100 --refresh-every 0 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
python $R/tools/timeline.py /tmp/tl --tail 60 > $R/gpurun_out/r06/step_timeline_pipelined.txt 2>&1
cat $R/gpurun_out/r06/step_timeline_pipelined.txt
