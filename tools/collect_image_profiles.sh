set -u
TAG=${1:-r04}
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --image --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_image.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_img -- python $R/bench.py --image --steps 10 --warmup 3 --no-cpu-baseline --witness 0 --repeat-passes 0 2>/dev/null | tail -1 > $O/${TAG}_bench_image_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_image_$C -- python $R/bench.py --image --steps 6 --warmup 2 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
done
I=0
for C in "VALUBusy SALUBusy" "LDSBankConflict MemUnitBusy" "OccupancyPercent VALUUtilization"; do
  I=$((I+1))
  rocprofv3 --pmc $C --output-format csv -d $O/busy_image_$I -- python $R/bench.py --image --steps 3 --warmup 12 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
done
# the non-image directories the reducer expects
for d in kstats pmc_FETCH_SIZE pmc_WRITE_SIZE busy_1 busy_2 busy_3; do mkdir -p $O/$d; done
python $R/tools/reduce_profiles.py $O $TAG 2>&1 | tail -3
python $R/tools/bev_probe2.py 2>/dev/null > $O/${TAG}_image_stage_times.txt
ls $O | grep image
