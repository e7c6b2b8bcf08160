#!/bin/bash
# SQ / traffic counters of one command, separate rocprofv3 --pmc passes, summarised per kernel.
# Usage: tools/pmc_cmd.sh <out.txt> <kernel-substring> <command...>
OUT=$1; ONLY=$2; shift 2
R=$PWD
export TMPDIR=/tmp
mkdir -p $(dirname $OUT)
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           ${PMC_EXTRA:+"$PMC_EXTRA"} \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p_cmd_$i
  (cd /tmp; timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $set --kernel-trace -d /tmp/p_cmd_$i -o r -- "$@" > $R/$OUT.run$i.log 2>&1)
  python $R/tools/rocpd_summary.py --pmc --only $ONLY /tmp/p_cmd_$i/r_results.db >> $OUT 2>&1
done
cat $OUT
