timeout 900 python -m pytest tests/test_gpu_agents.py -x -q -k "rccl" 2>&1 | tail -60 > gpurun_out/rccl1.txt
