#!/bin/bash
timeout 2700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
for cfg in generic_pca generic_gmm; do
for env in "BAYESPY_AMD_GRAPH_QUEUE=0 BAYESPY_AMD_SMALL_QUEUE=ew" "X=1"; do
  echo "== $cfg $env"
  env $env timeout 600 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms_per_step %.4f  elbo_last %r' % (d['ms_per_step'], d.get('elbo_last')))"
done
done
