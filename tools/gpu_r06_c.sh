#!/bin/bash
# round 6, step C: generic + sharded + comm suites, op traces and kernel stats of the generic legs
O=gpurun_out/r06_c
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests/test_generic_gpu.py tests/test_generic_engine_gpu.py tests/test_graph_sweep_gpu.py tests/test_sharded_generic_gpu.py tests/test_comm_gpu.py tests/test_update_order_gpu.py tests/test_hyperparameters_gpu.py tests/test_input_forms_gpu.py -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 300 python tools/generic_op_trace.py pca > $O/op_trace_pca.log 2>&1
timeout 300 python tools/generic_op_trace.py gmm > $O/op_trace_gmm.log 2>&1
cp gpurun_out/gen_trace_seq_*.txt $O/ 2>/dev/null
for c in generic_pca generic_gmm; do
  timeout 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o r -- python $R/bench.py --config $c --exact-steps --steps 50 --no-cpu-baseline > $R/$O/under_rocprof_$c.log 2>&1)
  timeout 120 python tools/rocpd_summary.py /tmp/p_$c/r_results.db > $O/kernel_stats_$c.txt 2>&1
done
python - <<'PY'
import json
for n in ('generic_pca','generic_gmm'):
    d=json.loads(open('gpurun_out/r06_c/bench_%s.json'%n).read().strip().splitlines()[-1])
    print(n, d['ms_per_step'], d['config']['sweep_graph'])
PY
