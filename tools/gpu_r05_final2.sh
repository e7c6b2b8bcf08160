#!/bin/bash
# The closing run of round 5 on ONE box: counters of both phase-2 forms of the mixture pass, the profile collection
# of every leg, the summaries copied where bench.py looks for them (profiles/r05/, keyed on the build id), then
# tools/gpu_r05_final.sh (GPU suite, default bench line + verbose records, world-1 RCCL launch, sweep traces).
PMC_EXTRA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" bash tools/pmc_cmd.sh gpurun_out/pmc_gmm_mfma4.txt gmm_pass python $PWD/tools/gmm_lab.py --steps 4 > /dev/null 2>&1
bash tools/collect_profiles_r05.sh gpurun_out/prof_r05 > gpurun_out/collect_r05.log 2>&1
cp gpurun_out/prof_r05/*.txt gpurun_out/prof_r05/under_rocprof_*.log profiles/r05/
bash tools/gpu_r05_final.sh
