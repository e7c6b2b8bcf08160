#!/bin/bash
export TMPDIR=/tmp
R=$PWD
for lds in 0 1; do
(cd /tmp; BAYESPY_AMD_GRAPH_QUEUE=1 BAYESPY_AMD_SMALL_QUEUE=all VMP_TUNE_small_queue_lds=$lds timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_q$lds -o r -- python $R/bench.py --config generic_pca --exact-steps --steps 50 --warmup 4 --no-cpu-baseline > /dev/null 2>&1)
echo "== lds=$lds"
timeout 120 python tools/rocpd_summary.py /tmp/p_q$lds/r_results.db 2>&1 | grep -v "synthetic-data" | head -16
done
timeout 120 python tools/rocpd_summary.py --timeline 400 /tmp/p_q1/r_results.db 2>/dev/null | tail -60 | cut -c1-100
