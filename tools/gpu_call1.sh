#!/bin/bash
# first GPU call of round 2: tests (one pytest process per file: a device fault in one file must
# not take the others with it), microbenchmarks, plate-pass A/B, bench lines
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
( timeout 120 tools/microbench.bin > $O/microbench.txt 2>&1 )
( timeout 300 tools/xpass_lab.bin 10000000 128 32 5 > $O/xpass_lab.txt 2>&1 )
( timeout 120 tools/xpass_lab.bin 1250000 128 32 5 > $O/xpass_lab_shard.txt 2>&1 )
for f in tests/test_masked_pca_gpu.py tests/test_pca_gpu.py tests/test_comm_gpu.py tests/test_gmm_gpu.py tests/test_generic_gpu.py tests/test_generic_engine_gpu.py tests/test_chain_gpu.py tests/test_sharded_generic_gpu.py tests/test_cabi.py; do
  b=$(basename $f .py)
  ( timeout 900 python -m pytest $f -m gpu -q --durations=8 > $O/pytest_$b.txt 2>&1 ); echo "rc=$?" >> $O/pytest_$b.txt
  echo "== $b: $(tail -2 $O/pytest_$b.txt | tr '\n' ' ')"
done
( timeout 600 python bench.py --steps 20 --warmup 3 --no-extra > $O/bench_n1.json 2> $O/bench_n1.err )
( timeout 300 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --layout rows > $O/bench_n1_rows.json 2>> $O/bench_n1.err )
( timeout 300 python bench.py --config gmm > $O/bench_gmm.json 2> $O/bench_gmm.err )
( timeout 600 python bench.py --config masked > $O/bench_masked.json 2> $O/bench_masked.err )
( timeout 300 python tools/bench_masked_pca.py --n 1000000 > $O/bench_masked_1e6.json 2>> $O/bench_masked.err )
cat $O/xpass_lab.txt; tail -12 $O/microbench.txt; cut -c1-1200 $O/bench_n1.json; echo; cut -c1-900 $O/bench_masked.json; tail -5 $O/bench_masked.err
