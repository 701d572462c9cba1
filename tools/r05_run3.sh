#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests3.txt 2>&1; tail -5 $O/tests3.txt
bash tools/exp_env_ab.sh 3 "2trips:" "3trips:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_3trips.so" > $O/ab_trips.txt 2>&1; cat $O/ab_trips.txt
HOPE_RS_TIMING=1 timeout 400 python tools/rs_timing.py > $O/rs_validate_cycles3.txt 2>/dev/null; cat $O/rs_validate_cycles3.txt
HOPE_STEP_TIMING=1 timeout 400 python tools/step_timing.py > $O/env_step_cycles3.txt 2>/dev/null; cat $O/env_step_cycles3.txt
