#!/bin/bash
# Regenerates every file under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh r06'
# Outputs land in gpurun_out/final/ (merged back by gpurun); copy them into profiles/ afterwards.
# PMC counters are collected in their own passes, without any trace domain.  Every command runs under `timeout`.
set -u
TAG=${1:-r06}
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout 400"
py() { $T python "$@" 2>>$O/stderr.log; }
PRE="--preroll 100 --refresh-every 0"     # profiled runs: a shorter pre-roll to the stationary episode population (the default run uses 300), static pool

py $R/bench.py | tail -1 > $O/${TAG}_bench_default.json
py $R/bench.py | tail -1 > $O/${TAG}_bench_default_2.json
py $R/bench.py --image --no-cpu-baseline | tail -1 > $O/${TAG}_bench_image.json
py $R/bench.py --rs-join joined --no-cpu-baseline | tail -1 > $O/${TAG}_bench_joined.json
py $R/bench.py --preroll 0 --no-cpu-baseline --witness 0 | tail -1 > $O/${TAG}_bench_fresh_episodes.json     # the round-3 form of the number

$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python $R/bench.py --steps 10 --warmup 3 $PRE --no-cpu-baseline --witness 0 --repeat-passes 0 2>/dev/null | tail -1 > $O/${TAG}_bench_under_rocprof.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_img -- python $R/bench.py --image --steps 10 --warmup 3 $PRE --no-cpu-baseline --witness 0 --repeat-passes 0 2>/dev/null | tail -1 > $O/${TAG}_bench_image_under_rocprof.json
for V in "" "_image"; do
  FLAG=""; [ -n "$V" ] && FLAG="--image"
  for C in FETCH_SIZE WRITE_SIZE; do
    $T rocprofv3 --pmc $C --output-format csv -d $O/pmc${V}_$C -- python $R/bench.py $FLAG --steps 6 --warmup 2 --preroll 60 --refresh-every 0 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
  done
done
# derived busy / utilisation metrics (list VALU-busy and LDS-bank-conflict next to the HBM fraction)
for V in "" "_image"; do
  FLAG=""; [ -n "$V" ] && FLAG="--image"
  I=0
  for C in "VALUBusy SALUBusy" "LDSBankConflict MemUnitBusy" "OccupancyPercent VALUUtilization"; do
    I=$((I+1))
    $T rocprofv3 --pmc $C --output-format csv -d $O/busy${V}_$I -- python $R/bench.py $FLAG --steps 3 --warmup 12 --preroll 60 --refresh-every 0 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
  done
done
python $R/tools/reduce_profiles.py $O $TAG
# raw SQ counters (instruction mix, wait / active quad-cycles) -> VALU-issue roofline of bench.py
(cd $R && timeout 900 bash tools/pmc_sq.sh final/sq --refresh-every 0 > /dev/null 2>&1)
cp $O/sq/sq_summary.txt $O/${TAG}_sq_summary.txt; cp $O/sq/sq_counters.json $O/${TAG}_sq_counters.json; rm -rf $O/sq
py $R/tools/stage_times.py --scenes 32768 > $O/${TAG}_stage_times.txt
py $R/tools/bev_probe.py > $O/${TAG}_image_stage_times.txt
# batch-size sweep, launch modes, new-map turnover, BASELINE configs 4 / 5 on one GPU, ranks sharing the GPU
{
  for NS in 4096 8192 16384 32768 65536 131072; do echo "== --scenes $NS"; py $R/bench.py --scenes $NS --no-cpu-baseline --witness 0 --repeat-passes 2 --steps 40 --warmup 10 | tail -1; done
  echo "== --scenes 65536 --overlap off"; py $R/bench.py --overlap off --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 0 --witness 0 | tail -1
  echo "== --same-map"; py $R/bench.py --same-map --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 0 | tail -1
  echo "== --refresh-every 0 (static pool)"; py $R/bench.py --refresh-every 0 --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 --witness 0 | tail -1
  echo "== HOPE_RS_DEBUG=0x20000 (no screen pass: round 4's validation kernel)"; HOPE_RS_DEBUG=0x20000 $T python $R/bench.py --refresh-every 0 --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 --witness 0 2>/dev/null | tail -1
  echo "== HOPE_PIPE=0 (steps not pipelined: the round-3 launch structure with this round's kernels)"; HOPE_PIPE=0 $T python $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 --witness 0 2>/dev/null | tail -1
  for NS in 4096 8192 16384; do echo "== --scenes $NS HOPE_RS_DEBUG=0x20000 (no screen pass)"; HOPE_RS_DEBUG=0x20000 $T python $R/bench.py --scenes $NS --no-cpu-baseline --witness 0 --repeat-passes 2 --steps 40 --warmup 10 2>/dev/null | tail -1; done
  echo "== --action-bank 16 (rounds 1-5: each scene repeats a 16-step action pattern)"; py $R/bench.py --action-bank 16 --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 --witness 0 | tail -1
  echo "== HOPE_MOTION_PAIR=0 HOPE_OBS_PAIR=0 (one scene per wave in both launches of the small-tile class)"; HOPE_MOTION_PAIR=0 HOPE_OBS_PAIR=0 $T python $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 --witness 0 2>/dev/null | tail -1
  echo "== the driver's exact form: --steps 20 --warmup 5"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
  echo "== the driver's exact form, second run"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
  echo "== the driver's exact form, third run"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
  echo "== config 2: --stages motion --scenes 4096"; py $R/bench.py --scenes 4096 --stages motion --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 0 --witness 0 | tail -1
  echo "== config 3: --scenes 16384 (full step)"; py $R/bench.py --scenes 16384 --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 | tail -1
  echo "== config 4 share: --policy hope --algo rollout --scenes 8192 --image"; py $R/bench.py --policy hope --algo rollout --scenes 8192 --image --no-cpu-baseline | tail -1
  echo "== --policy hope --algo rollout (65536 scenes, no image)"; py $R/bench.py --policy hope --algo rollout --no-cpu-baseline | tail -1
  echo "== config 5 share: --policy hope --algo ppo --scenes 16384"; py $R/bench.py --policy hope --algo ppo --scenes 16384 --no-cpu-baseline --steps 32 --warmup 16 | tail -1
  echo "== 2 ranks sharing the GPU (gloo): weak scaling 16384 scenes/rank"
  HOPE_BENCH_SHARE_GPU=1 $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --scenes 16384 --steps 20 --warmup 5 --no-cpu-baseline --witness 0 --repeat-passes 0 2>/dev/null | tail -1
  echo "== 8 ranks sharing the GPU (gloo): config 4's shape, 65536 scenes strong-scaled = 8192 per rank"
  HOPE_BENCH_SHARE_GPU=1 $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 $R/bench.py --gpus 8 --scaling strong --scenes 65536 --steps 20 --warmup 5 --preroll 60 --no-cpu-baseline --witness 0 --repeat-passes 0 2>/dev/null | tail -1
  echo "== 2 ranks sharing the GPU (gloo): strong scaling 16384 scenes total, PPO with gradient all-reduce"
  HOPE_BENCH_SHARE_GPU=1 $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 2 --scaling strong --scenes 16384 --policy hope --algo ppo --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1
} > $O/${TAG}_bench_modes.txt
# timeline of one step: joined form (one step can be cut out of the trace) and the last 60 launches of the pipelined form
$T rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python $R/bench.py --rs-join joined --steps 8 --warmup 8 $PRE --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
{ python $R/tools/timeline.py $O/tl 3; python $R/tools/timeline.py $O/tl 2; } > $O/${TAG}_step_timeline.txt 2>&1
rm -rf $O/tl
$T rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python $R/bench.py --steps 8 --warmup 8 $PRE --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
python $R/tools/timeline.py $O/tl --tail 60 > $O/${TAG}_step_timeline_pipelined.txt 2>&1
rm -rf $O/tl
# cycle accounting (instrumented builds), float32-filter statistics + self-check, randomised equality soak, tie census
HOPE_STEP_TIMING=1 $T python $R/tools/step_timing.py > $O/${TAG}_env_step_cycles.txt 2>/dev/null
HOPE_RS_TIMING=1 $T python $R/tools/rs_timing.py > $O/${TAG}_rs_validate_cycles.txt 2>/dev/null
$T python $R/tools/rs_filter_stats.py --check > $O/${TAG}_rs_filter_stats.txt 2>/dev/null
$T python $R/tools/rs_filter_stats.py 2>/dev/null | head -3 > $O/${TAG}_rs_filter_stats_nocheck.txt
$T python $R/tools/rs_bench.py > $O/${TAG}_rs_bench.txt 2>/dev/null
timeout 900 python $R/tools/soak.py --seeds 10 --scenes 4096 --steps 12 > $O/${TAG}_soak.txt 2>/dev/null
$T python $R/tools/tie_census.py --scene-steps 2e8 > $O/${TAG}_tie_census.txt 2>/dev/null
$T python $R/tools/config1_latency.py > $O/${TAG}_config1_latency.txt 2>/dev/null
# vector instructions of k_env_step by stage (SQ_INSTS_VALU of launches with stages switched off)
rm -rf /tmp/pp_valu
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pp_valu -- python $R/tools/pmc_stage_probe.py --mix mixed --scenes 16384 --seq $O/probe_seq.json > /dev/null 2>&1
python $R/tools/pmc_stage_probe.py --seq $O/probe_seq.json --reduce $(find /tmp/pp_valu -name "*counter_collection.csv" | head -1) SQ_INSTS_VALU > $O/${TAG}_env_step_insts_by_stage.txt 2>&1
rm -f $O/probe_seq.json
# round 6: the observation launch by stage, pair kernel vs one-scene kernel, on the stationary episode population
rm -rf /tmp/op
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d /tmp/op -- python $R/tools/obs_probe.py --seq $O/obs_probe_seq.json > /dev/null 2>&1
python $R/tools/obs_probe.py --seq $O/obs_probe_seq.json --reduce /tmp/op > $O/${TAG}_obs_insts_by_stage.txt 2>&1
rm -f $O/obs_probe_seq.json
# round 6: how many coarse beams the mask stage visits, by action source
{ $T python $R/tools/mask_active_census.py; $T python $R/tools/mask_active_census.py --bank 16; } > $O/${TAG}_mask_active_census.txt 2>/dev/null
# calibration of FETCH_SIZE / WRITE_SIZE on the library's access widths
rm -rf /tmp/cf /tmp/cw
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/cf -- python $R/tools/pmc_calib.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/cw -- python $R/tools/pmc_calib.py > /dev/null 2>&1
python $R/tools/pmc_calib.py --reduce /tmp/cf /tmp/cw > $O/${TAG}_pmc_calibration.json 2>/dev/null
# the independent-math soak (libm oracle under OpenMP) and the CPU oracle's thread scaling on this host
timeout 1500 python $R/tools/libm_soak.py --scenes 8192 --steps 128 > $O/${TAG}_libm_soak.txt 2>&1
# one line per bench-modes entry
python - > $O/${TAG}_bench_modes_summary.txt <<P
import json
lab = None
for l in open('$O/${TAG}_bench_modes.txt'):
    l = l.rstrip()
    if l.startswith('=='):
        lab = l
        continue
    if l.startswith('{'):
        d = json.loads(l); r = d.get('repeat') or {}
        print(lab, '|', round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 4), 'ms | steady', r.get('median') and round(r['median'], 4), '| joined',
              r.get('joined_ms_per_step') and round(r['joined_ms_per_step'], 4), '| refresh', (d.get('pool_refresh') or {}).get('every_steps'), (d.get('pool_refresh') or {}).get('refresher_commits'))
P
rm -rf $O/kstats $O/kstats_img $O/pmc_* $O/pmc*_FETCH_SIZE $O/pmc*_WRITE_SIZE $O/busy*
ls -la $O
