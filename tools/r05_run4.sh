#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python - <<'P' 2>&1 | tee $O/queue_check.txt
import torch, time
from hope_amd import ParkingBatch
for pre in (0, 3):
    extra = [torch.cuda.Stream() for _ in range(pre)]      # streams the process created BEFORE the handle
    t = time.perf_counter()
    env = ParkingBatch(1024, 128)
    print('streams created first:', pre, 'create s:', round(time.perf_counter() - t, 3), env.queue_check())
    env.close()
P
bash tools/exp_env_ab.sh 2 "calib:" "table:HOPE_QUEUE_CHECK=0" > $O/ab_queue.txt 2>&1; cat $O/ab_queue.txt
BENCH_ARGS="--rs-join joined" bash tools/exp_env_ab.sh 1 "calib_joined:" "table_joined:HOPE_QUEUE_CHECK=0" > $O/ab_queue_joined.txt 2>&1; cat $O/ab_queue_joined.txt
BENCH_ARGS="--refresh-every 0" bash tools/exp_env_ab.sh 2 "static_pool:" > $O/ab_static_pool.txt 2>&1; cat $O/ab_static_pool.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_errors.py -x -q > $O/tests4.txt 2>&1; tail -3 $O/tests4.txt
