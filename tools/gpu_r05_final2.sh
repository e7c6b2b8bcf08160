#!/bin/bash
# closing run after the mixture experiment: counters of both phase-2 forms, the profile collection, then tools/gpu_r05_final.sh
PMC_EXTRA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" bash tools/pmc_cmd.sh gpurun_out/pmc_gmm_mfma4.txt gmm_pass python $PWD/tools/gmm_lab.py --steps 4 > /dev/null 2>&1
bash tools/collect_profiles_r05.sh gpurun_out/prof_r05 > gpurun_out/collect_r05.log 2>&1
bash tools/gpu_r05_final.sh
