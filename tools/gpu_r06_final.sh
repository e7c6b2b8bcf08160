#!/bin/bash
# round-6 evidence of ONE library build: profiles (kernel stats + counters), the bench line with all
# legs, the GPU suite.  Usage: gpurun --timeout 5400 -- bash tools/gpu_r06_final.sh ; then copy the
# summaries from gpurun_out/r06_final/ into profiles/r06/ (only gpurun_out/ travels back)
O=gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 bash tools/collect_profiles_r06.sh $O > $O/collect.log 2>&1
timeout 900 python tools/lssm_wide_ab.py > $O/lssm_wide_ab.txt 2> $O/lssm_wide_ab.err
timeout 300 python tools/queue_record_cost.py 2> $O/queue_record_cost.err | grep -v amdgpu.ids > $O/queue_record_cost.txt
O=$O timeout 600 bash tools/gpu_r06_tl.sh > $O/tl.log 2>&1
timeout 1800 python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
timeout 2700 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
