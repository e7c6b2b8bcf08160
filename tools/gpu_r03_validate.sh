#!/bin/bash
# One GPU-box call of round 3: full `pytest -m gpu`, the default bench line, rocprofv3 summaries
# (tools/collect_profiles_r03.sh).  Usage:
#   gpurun --timeout 1700 -- bash tools/gpu_r03_validate.sh ; then copy gpurun_out/r03/prof/* to profiles/r03/
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
s=$(date +%s)
( timeout 700 python -m pytest tests -m gpu -q -x --durations=10 > $O/pytest_all.txt 2>&1 ); echo "rc=$?" >> $O/pytest_all.txt
echo "pytest secs: $(( $(date +%s) - s ))"
tail -16 $O/pytest_all.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_all.txt | head -30
( timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); cut -c1-700 $O/bench_default.json; tail -2 $O/bench_default.err
python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
for e in d.get('extra', []):
    print(e.get('metric'), e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), e.get('error'))
PY
echo "bench done secs: $(( $(date +%s) - s ))"
bash tools/collect_profiles_r03.sh $O/prof > $O/collect.log 2>&1
tail -3 $O/collect.log
echo "total secs: $(( $(date +%s) - s ))"
