timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or lidar or witness or cull" 2>&1 | tail -3
bash tools/exp_ab.sh
