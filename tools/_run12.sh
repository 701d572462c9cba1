set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --witness 0 --repeat-passes 3 --steps 40 --warmup 10 $BA 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeat') or {}; print('$tag', round(d['value']/1e6,2),'M',round(d['ms_per_step'],4),'ms steady',[round(x,4) for x in r.get('ms_per_step')],'joined',round(r.get('joined_ms_per_step'),4))"; }
{
for NS in 16384 32768; do
  for rep in 1 2; do
  BA="--scenes $NS" run "$NS one-launch" A=1
  BA="--scenes $NS" run "$NS split+pairs" HOPE_SPLIT_MIN=1
  done
done
} > gpurun_out/r06/ab_split_min.txt 2>&1
cat gpurun_out/r06/ab_split_min.txt
