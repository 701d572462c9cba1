set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r06/gpu_tests_lut.txt
tail -3 gpurun_out/r06/gpu_tests_lut.txt
BENCH_ARGS="--action-bank 16" bash tools/exp_env_ab.sh 3 "nolut_bank16:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_nolut.so" "lut_bank16:" 2>&1 | tee gpurun_out/r06/ab_mask_lut.txt
bash tools/exp_env_ab.sh 3 "nolut_fresh:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_nolut.so" "lut_fresh:" 2>&1 | tee -a gpurun_out/r06/ab_mask_lut.txt
