set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r06/gpu_tests_3.txt
tail -5 gpurun_out/r06/gpu_tests_3.txt
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06/bench_driver_form.json
cat gpurun_out/r06/bench_driver_form.json
