#!/bin/bash
O=gpurun_out/r06_lssm
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_lssm_gpu.py -q -k "rotation_of_the_means" 2>&1 | tail -2
python - <<'PY'
import sys, json
sys.path.insert(0,'.')
from tools import workloads
for (B,D,M) in ((20000,16,8),(20000,12,8),(20000,8,8),(20000,4,8)):
    r = workloads.run_lssm(B=B, T=1000, M=M, D=D, steps=5, warmup=2, cpu_baseline=False)
    print('B=%d D=%d M=%d  ms/iter %.3f' % (B, D, M, r['ms_per_step']))
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_l16 -o r -- python -c "
import sys; sys.path.insert(0,'$R')
from tools import workloads
workloads.run_lssm(B=20000, T=1000, M=8, D=16, steps=5, warmup=2, cpu_baseline=False)" > /dev/null 2>&1)
timeout 120 python tools/rocpd_summary.py /tmp/p_l16/r_results.db > $O/kernel_stats_lssm_d16.txt 2>&1
head -14 $O/kernel_stats_lssm_d16.txt
