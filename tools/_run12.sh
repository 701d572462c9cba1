set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -5 | tee gpurun_out/r06/gpu_tests_v3.txt
bash tools/exp_env_ab.sh 3 "single:HOPE_MOTION_PAIR=0" "pair_v3:" 2>&1 | tee gpurun_out/r06/ab_motion_pair_v3.txt
