#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests5.txt 2>&1; tail -3 $O/tests5.txt
BENCH_ARGS="--refresh-every 0" bash tools/exp_env_ab.sh 3 "split8:" "split7:HOPE_RS_SCREEN_OCC=7" "onekernel:HOPE_RS_SPLIT=0" > $O/ab_split.txt 2>&1; cat $O/ab_split.txt
bash tools/exp_env_ab.sh 2 "split8_refresh:" > $O/ab_split_refresh.txt 2>&1; cat $O/ab_split_refresh.txt
