timeout 900 python -m pytest tests/test_gpu_image.py -x -q -k "pipelined" 2>&1 | tail -15
