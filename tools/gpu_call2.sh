#!/bin/bash
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_masked_pca_gpu.py tests/test_sharded_generic_gpu.py; do
  b=$(basename $f .py)
  ( timeout 900 python -m pytest $f -m gpu -q --durations=5 > $O/pytest_$b.txt 2>&1 ); echo "rc=$?" >> $O/pytest_$b.txt
  echo "== $b: $(tail -2 $O/pytest_$b.txt | tr '\n' ' ')"
done
( timeout 600 python tools/mpca_lab.py > $O/mpca_lab.txt 2>&1 ); cat $O/mpca_lab.txt | tail -20
( timeout 600 python bench.py --config masked > $O/bench_masked.json 2> $O/bench_masked.err ); cut -c1-300 $O/bench_masked.json; tail -3 $O/bench_masked.err
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_*.txt | head -40
