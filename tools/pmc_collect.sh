#!/bin/bash
# usage: tools/pmc_collect.sh <outdir> <bench args...>   (tuning environment such as BB_QUEUE_NETW passes through)
# Separate passes, counters only (--kernel-trace --pmc), as the MI355X guide prescribes.
set -e
OUT=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $REPO/$OUT/pass$i -- python3 $REPO/${PMC_PROG:-bench.py} "$@" > $REPO/$OUT/pass$i.log 2>&1
  echo "pass $i done"
done
