timeout 600 python -m pytest tests/test_math.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
bash tools/exp_ab.sh
