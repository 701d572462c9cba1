#!/bin/bash
O=gpurun_out/final; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --witness 0 --repeat-passes 1 $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'steady', round(d['repeat']['median'],4), d['pool_refresh'] and (d['pool_refresh']['commits_in_timed_region'], d['pool_refresh']['refresher_commits'], round(d['pool_refresh']['generator_seconds'],3)))"; }
for i in 1 2; do
EXTRA="--refresh-threads 8" run threads8 A=1
EXTRA="--refresh-threads 1" run threads1 A=1
EXTRA="--refresh-threads 2" run threads2 A=1
EXTRA="--refresh-threads 8" run threads8_from_end HOPE_GEN_CPUS_FROM_END=1
EXTRA="--refresh-every 0" run static A=1
done
