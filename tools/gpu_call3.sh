#!/bin/bash
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_lssm_gpu.py tests/test_chain_gpu.py tests/test_masked_pca_gpu.py tests/test_generic_engine_gpu.py tests/test_sharded_generic_gpu.py; do
  b=$(basename $f .py)
  ( timeout 900 python -m pytest $f -m gpu -q --durations=5 > $O/pytest_$b.txt 2>&1 ); echo "rc=$?" >> $O/pytest_$b.txt
  echo "== $b: $(tail -2 $O/pytest_$b.txt | tr '\n' ' ')"
done
( timeout 600 python bench.py --config lssm > $O/bench_lssm.json 2> $O/bench_lssm.err ); cut -c1-1500 $O/bench_lssm.json; tail -5 $O/bench_lssm.err
( timeout 300 python tools/bench_lssm.py --b 1000 > $O/bench_lssm_b1000.json 2>> $O/bench_lssm.err ); cut -c1-400 $O/bench_lssm_b1000.json
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_*.txt | head -60
