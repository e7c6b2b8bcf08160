#!/bin/bash
# first GPU call of round 2: tests, microbenchmarks, plate-pass A/B, bench lines
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
( timeout 120 tools/microbench.bin > $O/microbench.txt 2>&1 ) 
( timeout 300 tools/xpass_lab.bin 10000000 128 32 5 > $O/xpass_lab.txt 2>&1 )
( timeout 120 tools/xpass_lab.bin 1250000 128 32 5 > $O/xpass_lab_shard.txt 2>&1 )
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1 ); echo "pytest rc=$?" >> $O/pytest_gpu.txt
( timeout 600 python bench.py --steps 20 --warmup 3 --no-extra > $O/bench_n1.json 2> $O/bench_n1.err )
( timeout 300 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --layout rows > $O/bench_n1_rows.json 2>> $O/bench_n1.err )
( timeout 300 python bench.py --config gmm > $O/bench_gmm.json 2> $O/bench_gmm.err )
tail -3 $O/pytest_gpu.txt; cat $O/xpass_lab.txt; tail -12 $O/microbench.txt; cat $O/bench_n1.json | cut -c1-1500
