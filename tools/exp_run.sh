echo "== new chain"; HOPE_RS_TIMING=1 timeout 300 python tools/rs_timing.py 2>/dev/null
echo "== old chain"; HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_bold.so HOPE_RS_TIMING=1 timeout 300 python tools/rs_timing.py 2>/dev/null
