timeout 900 python -m pytest tests/test_gpu_image.py -x -q 2>&1 | tail -2
timeout 600 python tools/bev_probe2.py 2>&1 | tail -12
