#!/bin/bash
# round-5 first GPU session: tests, screen A/B, cycle accounting, filter self-check
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
bash tools/exp_env_ab.sh 3 "screen:" "noscreen:HOPE_RS_DEBUG=0x20000" > $O/ab_screen.txt 2>&1; cat $O/ab_screen.txt
HOPE_RS_TIMING=1 timeout 400 python tools/rs_timing.py > $O/rs_validate_cycles.txt 2>/dev/null; cat $O/rs_validate_cycles.txt
timeout 600 python tools/rs_filter_stats.py --check > $O/rs_filter_stats.txt 2>/dev/null; cat $O/rs_filter_stats.txt
