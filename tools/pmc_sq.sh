#!/bin/bash
# Raw SQ counters per kernel (instruction mix, wait / active quad-cycles): gpurun -- 'bash tools/pmc_sq.sh <tag> [bench args]'
# One rocprofv3 pass per counter group (8 SQ slots), no trace domain.  Output: gpurun_out/<tag>/sq_<i>.csv + sq_summary.txt
set -u
TAG=${1:-sq}; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
I=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU"; do
  I=$((I+1))
  rm -rf $O/raw_$I
  rocprofv3 --pmc $C --output-format csv -d $O/raw_$I -- python $R/bench.py --steps 4 --warmup 8 --preroll 60 --no-cpu-baseline --witness 0 --repeat-passes 0 "$@" > $O/pass_$I.log 2>&1
  F=$(find $O/raw_$I -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp $F $O/sq_$I.csv
  rm -rf $O/raw_$I
done
python $R/tools/pmc_sq_reduce.py $O 76 > $O/sq_summary.txt      # 60 pre-roll + 8 warm-up + 4 timed + 4 breakdown steps
cat $O/sq_summary.txt
