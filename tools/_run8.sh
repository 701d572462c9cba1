set -x
cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/cf -- python $R/tools/pmc_calib.py > /tmp/cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/cw -- python $R/tools/pmc_calib.py > /tmp/cw.log 2>&1
python $R/tools/pmc_calib.py --reduce /tmp/cf /tmp/cw > $R/gpurun_out/r06/pmc_calibration.json
cat $R/gpurun_out/r06/pmc_calibration.json
cd $R
python tools/cpu_baseline_threads.py 2>&1 | grep threads > gpurun_out/r06/cpu_baseline_threads.txt
cat gpurun_out/r06/cpu_baseline_threads.txt
