#!/bin/bash
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
for f in tests/test_lssm_gpu.py tests/test_chain_gpu.py; do
  b=$(basename $f .py)
  ( timeout 900 python -m pytest $f -m gpu -q --durations=5 > $O/pytest_$b.txt 2>&1 ); echo "rc=$?" >> $O/pytest_$b.txt
  echo "== $b: $(tail -2 $O/pytest_$b.txt | tr '\n' ' ')"
done
( timeout 600 python bench.py --config lssm > $O/bench_lssm.json 2> $O/bench_lssm.err ); cut -c1-300 $O/bench_lssm.json; python -c "import json; d=json.load(open('$O/bench_lssm.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['peak_mem_GB'])"
( timeout 300 python tools/bench_lssm.py --b 1000 > $O/bench_lssm_b1000.json 2>> $O/bench_lssm.err ); python -c "import json; d=json.load(open('$O/bench_lssm_b1000.json')); print('b1000', d['ms_per_step'])"
bash tools/collect_profiles_r02.sh $O/prof > $O/collect.log 2>&1
tail -5 $O/collect.log
head -30 $O/prof/kernel_stats_lssm.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_*.txt | head -30
