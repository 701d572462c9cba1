set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r06/gpu_tests_2.txt
tail -5 gpurun_out/r06/gpu_tests_2.txt
bash tools/exp_env_ab.sh 3 "r5:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_r5.so" "pair:" "nopair:HOPE_OBS_PAIR=0" 2>&1 | tee gpurun_out/r06/ab_obs_pair.txt
