#!/bin/bash
O=gpurun_out/r02f
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_generic_gpu.py tests/test_generic_engine_gpu.py -m gpu -q -x --durations=5 > $O/pytest_generic.txt 2>&1 ); echo "rc=$?" >> $O/pytest_generic.txt
tail -8 $O/pytest_generic.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_generic.txt | head -30
( timeout 600 python tools/bench_generic.py > $O/bench_generic.json 2> $O/bench_generic.err ); cat $O/bench_generic.json; tail -3 $O/bench_generic.err
