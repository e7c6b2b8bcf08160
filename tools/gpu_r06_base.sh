#!/bin/bash
# round-6 baseline evidence of the generic engine: op traces (pca, gmm), bench legs, kernel stats of the mixture
O=gpurun_out/r06_base
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 300 python tools/generic_op_trace.py pca > $O/op_trace_pca.log 2>&1
timeout 300 python tools/generic_op_trace.py gmm > $O/op_trace_gmm.log 2>&1
cp gpurun_out/gen_trace_seq_*.txt $O/ 2>/dev/null
timeout 300 python bench.py --config generic_pca --no-cpu-baseline > $O/bench_generic_pca.json 2> $O/bench_generic_pca.err
timeout 300 python bench.py --config generic_gmm --no-cpu-baseline > $O/bench_generic_gmm.json 2> $O/bench_generic_gmm.err
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_gg -o r -- python $R/bench.py --config generic_gmm --exact-steps --steps 20 --no-cpu-baseline > $R/$O/under_rocprof_generic_gmm.log 2>&1)
timeout 120 python tools/rocpd_summary.py /tmp/p_gg/r_results.db > $O/kernel_stats_generic_gmm.txt 2>&1
ls -la $O
