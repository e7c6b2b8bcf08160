#!/bin/bash
# One GPU-box call: full `pytest -m gpu`, the default bench line, rocprofv3 summaries of every block
# (tools/collect_profiles_r02.sh) and the dispatch timelines of the small-size steps (tools/timeline.sh).  Usage:
#   gpurun --timeout 2400 -- bash tools/gpu_validate_and_profile.sh ; then copy gpurun_out/r02h/prof/* to profiles/r02/
O=gpurun_out/r02h
mkdir -p $O
export TMPDIR=/tmp
s=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_all.txt 2>&1 ); echo "rc=$?" >> $O/pytest_all.txt
echo "pytest secs: $(( $(date +%s) - s ))"
tail -16 $O/pytest_all.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_all.txt | head -30
( timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); cut -c1-700 $O/bench_default.json; tail -2 $O/bench_default.err
python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
for e in d.get('extra', []):
    print(e.get('metric'), e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), e.get('error'))
PY
bash tools/collect_profiles_r02.sh $O/prof > $O/collect.log 2>&1
tail -3 $O/collect.log
bash tools/timeline.sh > $O/timeline.log 2>&1; cp gpurun_out/tl/timeline_*.txt $O/prof/ 2>/dev/null
echo "total secs: $(( $(date +%s) - s ))"
