#!/bin/bash
O=gpurun_out/r06_d
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd tools/experiments && timeout 120 ./graph_branch_lab) > $O/graph_branch_lab.txt 2>&1
cat $O/graph_branch_lab.txt
timeout 2400 python -m pytest tests/test_generic_gpu.py tests/test_generic_engine_gpu.py tests/test_graph_sweep_gpu.py tests/test_sharded_generic_gpu.py -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for c in generic_pca generic_gmm; do
  timeout 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o r -- python $R/bench.py --config $c --exact-steps --steps 50 --no-cpu-baseline > $R/$O/under_rocprof_$c.log 2>&1)
  timeout 120 python tools/rocpd_summary.py /tmp/p_$c/r_results.db > $O/kernel_stats_$c.txt 2>&1
done
python - <<'PY'
import json
for n in ('generic_pca','generic_gmm'):
    d=json.loads(open('gpurun_out/r06_d/bench_%s.json'%n).read().strip().splitlines()[-1])
    print(n, d['ms_per_step'], d['config']['sweep_graph'])
PY
