#!/bin/bash
O=gpurun_out/r06_c2
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tl1 -o r -- python $R/bench.py --config pca_c2 --steps 40 --no-cpu-baseline > $R/$O/under_c2.log 2>&1)
python tools/rocpd_summary.py --timeline 60 /tmp/p_tl1/r_results.db > $O/timeline_c2.txt 2>&1
python tools/rocpd_summary.py /tmp/p_tl1/r_results.db > $O/kernel_stats_c2.txt 2>&1
for i in 1 2 3; do timeout 300 python bench.py --config pca_c2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done > $O/c2_ms.txt
cat $O/c2_ms.txt
