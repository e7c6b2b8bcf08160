#!/bin/bash
# round-6 evidence of ONE library build: profiles (kernel stats + counters), the bench line with all
# legs, the GPU suite.  Usage: gpurun --timeout 5400 -- bash tools/gpu_r06_final.sh
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 bash tools/collect_profiles_r06.sh $O > $O/collect.log 2>&1
mkdir -p profiles/r06
cp $O/kernel_stats_*.txt $O/pmc_*.txt profiles/r06/ 2>/dev/null
timeout 900 python tools/lssm_wide_ab.py > profiles/r06/lssm_wide_ab.txt 2> $O/lssm_wide_ab.err
timeout 1800 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
timeout 2700 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
