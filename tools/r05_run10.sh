#!/bin/bash
O=$PWD/gpurun_out/r05; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in SQ_INSTS_VALU SQ_INSTS_SALU; do
  rm -rf /tmp/pp_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pp_$C -- python $R/tools/pmc_stage_probe.py --mix mixed --scenes 16384 --seq $O/probe_seq_$C.json > /dev/null 2>&1
  f=$(find /tmp/pp_$C -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_stage_probe.py --seq $O/probe_seq_$C.json --reduce $f $C
done > $O/env_step_insts_by_stage.txt 2>&1
cat $O/env_step_insts_by_stage.txt
