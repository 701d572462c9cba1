set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{ timeout 300 python tools/mask_active_census.py; timeout 300 python tools/mask_active_census.py --bank 16; timeout 300 python tools/mask_active_census.py --bank 4; } > gpurun_out/r06/mask_active_census.txt 2>&1
cat gpurun_out/r06/mask_active_census.txt
