#!/bin/bash
O=gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
s=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_all.txt 2>&1 ); echo "rc=$?" >> $O/pytest_all.txt
echo "pytest secs: $(( $(date +%s) - s ))"
tail -25 $O/pytest_all.txt
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_all.txt | head -30
