timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "kinematics or motion or pose or stress or config2" 2>&1 | tail -3
python tools/rs_bench.py 2>/dev/null | tail -12
HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_b0.so python tools/rs_bench.py 2>/dev/null | tail -12
bash tools/exp_ab.sh
