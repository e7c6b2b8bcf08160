#!/bin/bash
# round 6, step B: kernel test of the fused update, graph tests, kernel traces of the generic legs
O=gpurun_out/r06_b
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_generic_gpu.py tests/test_graph_sweep_gpu.py -q > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for c in generic_pca generic_gmm; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o r -- python $R/bench.py --config $c --exact-steps --steps 50 --no-cpu-baseline > $R/$O/under_rocprof_$c.log 2>&1)
  timeout 120 python tools/rocpd_summary.py /tmp/p_$c/r_results.db > $O/kernel_stats_$c.txt 2>&1
  tail -c 300 $O/under_rocprof_$c.log
done
