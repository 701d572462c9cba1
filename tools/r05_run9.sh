#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sixteen_lane or parity_f64 or with_rs or config3" > $O/tests9a.txt 2>&1; tail -4 $O/tests9a.txt
BENCH_ARGS="--refresh-every 0 --witness 0" bash tools/exp_env_ab.sh 3 "kin16:" "kin4:HOPE_KIN_LANES=4" > $O/ab_kin16.txt 2>&1; cat $O/ab_kin16.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests9.txt 2>&1; tail -3 $O/tests9.txt
