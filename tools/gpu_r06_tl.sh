#!/bin/bash
O=${O:-gpurun_out/r06_tl}
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for c in generic_pca generic_gmm; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_$c -o r -- python $R/bench.py --config $c --exact-steps --steps 20 --no-cpu-baseline > /dev/null 2>&1)
  timeout 120 python tools/rocpd_summary.py --timeline 150 /tmp/p_$c/r_results.db > $O/timeline_$c.txt 2>&1
done
