#!/bin/bash
# Regenerates every file under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh r01'
# Outputs land in gpurun_out/final/ (merged back by gpurun); copy them into profiles/ afterwards.
# PMC counters are collected in their own passes, without any trace domain.
set -u
TAG=${1:-r01}
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
py() { python "$@" 2>$O/stderr.log; }

py $R/bench.py | tail -1 > $O/${TAG}_bench_default.json
py $R/bench.py --image --no-cpu-baseline | tail -1 > $O/${TAG}_bench_image.json

rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_img -- python $R/bench.py --image --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_image_under_rocprof.json
for V in "" "_image"; do
  FLAG=""; [ -n "$V" ] && FLAG="--image"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc${V}_$C -- python $R/bench.py $FLAG --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  done
done
# derived busy / utilisation metrics (list VALU-busy and LDS-bank-conflict next to the HBM fraction)
for V in "" "_image"; do
  FLAG=""; [ -n "$V" ] && FLAG="--image"
  I=0
  for C in "VALUBusy SALUBusy" "LDSBankConflict MemUnitBusy" "OccupancyPercent VALUUtilization"; do
    I=$((I+1))
    rocprofv3 --pmc $C --output-format csv -d $O/busy${V}_$I -- python $R/bench.py $FLAG --steps 3 --warmup 12 --no-cpu-baseline > /dev/null 2>&1
  done
done
python $R/tools/reduce_profiles.py $O $TAG
py $R/tools/stage_times.py --scenes 32768 > $O/${TAG}_stage_times.txt
py $R/tools/bev_probe.py > $O/${TAG}_image_stage_times.txt
py $R/examples/rollout_demo.py --scenes 65536 --steps 50 > $O/${TAG}_rollout_demo.txt
rm -rf $O/kstats $O/kstats_img $O/pmc_* $O/pmc*_FETCH_SIZE $O/pmc*_WRITE_SIZE $O/busy*
ls -la $O
