#!/bin/bash
O=gpurun_out/r02g
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gmm_gpu.py tests/test_generic_gpu.py -m gpu -q -x --durations=5 > $O/pytest_gmm.txt 2>&1 ); echo "rc=$?" >> $O/pytest_gmm.txt
tail -8 $O/pytest_gmm.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_gmm.txt | head -30
( timeout 600 python bench.py --config gmm --no-cpu-baseline --no-extra > $O/bench_gmm.json 2> $O/bench_gmm.err ); cut -c1-600 $O/bench_gmm.json; tail -3 $O/bench_gmm.err
( timeout 600 python tools/bench_generic.py > $O/bench_generic.json 2> $O/bench_generic.err ); python -c "
import json; d=json.load(open('$O/bench_generic.json'))
for k,v in d.items(): print(k, {a: round(b,2) for a,b in v.items()})"
