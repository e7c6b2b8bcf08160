#!/bin/bash
# SQ counters of the per-plate stage (mpca_blk4_kernel): separate rocprofv3 --pmc passes
O=${1:-gpurun_out/pmc_blk4}
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp; MPCA_LAB_ONE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/p_blk4_$i -o r -- python $R/tools/mpca_lab.py 2097152 > $R/$O/run_$i.log 2>&1)
  python tools/rocpd_summary.py --pmc --only ${PMC_ONLY:-mpca_blk4} /tmp/p_blk4_$i/r_results.db >> $O/pmc_blk4.txt 2>&1
done
cat $O/pmc_blk4.txt
