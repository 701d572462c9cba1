#!/bin/bash
O=$PWD/gpurun_out/r05; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "parity or cull or witness or config3 or tie_census or stress" > $O/tests13a.txt 2>&1; tail -3 $O/tests11a.txt
BENCH_ARGS="--refresh-every 0 --witness 256" bash tools/exp_env_ab.sh 3 "new:" "base:HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_base.so" > $O/ab_kin_lds.txt 2>&1; cat $O/ab_kin_lds.txt
cd /tmp && export TMPDIR=/tmp
for C in SQ_INSTS_VALU; do
  rm -rf /tmp/pp_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pp_$C -- python $R/tools/pmc_stage_probe.py --mix mixed --scenes 16384 --seq $O/probe_seq_$C.json > /dev/null 2>&1
  f=$(find /tmp/pp_$C -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_stage_probe.py --seq $O/probe_seq_$C.json --reduce $f $C
done > $O/env_step_insts_by_stage4.txt 2>&1
cat $O/env_step_insts_by_stage4.txt
