#!/bin/bash
O=gpurun_out/r06_lssm
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for D in 8 16; do
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_l$D -o r -- python -c "
import sys; sys.path.insert(0,'$R')
from tools import workloads
workloads.run_lssm(B=100000, T=1000, M=8, D=$D, steps=10, warmup=2, cpu_baseline=False)" > /dev/null 2>&1)
timeout 120 python tools/rocpd_summary.py /tmp/p_l$D/r_results.db 2>&1 | grep -v "synthetic-data" > $O/kernel_stats_lssm_d${D}_b1e5.txt
head -14 $O/kernel_stats_lssm_d${D}_b1e5.txt
done
