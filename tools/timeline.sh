#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace; start / duration / gap per dispatch, stream ids) of the PCA
# iteration where the replicated-node chain, not the plate pass, sets the step: BASELINE config 2 and the
# shard one of 8 ranks holds at the headline size.  Usage: gpurun -- bash tools/timeline.sh ; copy
# gpurun_out/tl/timeline_*.txt to profiles/r02/.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/tl
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tl1 -o r -- python $R/bench.py --config pca_c2 --steps 20 --no-cpu-baseline > $R/gpurun_out/tl/under_c2.log 2>&1)
python tools/rocpd_summary.py --timeline 40 /tmp/p_tl1/r_results.db > gpurun_out/tl/timeline_c2.txt 2>&1
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tl2 -o r -- python $R/bench.py --n 1250000 --steps 20 --no-cpu-baseline --no-extra > $R/gpurun_out/tl/under_shard.log 2>&1)
python tools/rocpd_summary.py --timeline 40 /tmp/p_tl2/r_results.db > gpurun_out/tl/timeline_shard_n1250000.txt 2>&1
tail -14 gpurun_out/tl/timeline_shard_n1250000.txt
python $R/bench.py --n 1250000 --steps 100 --no-cpu-baseline --no-extra 2>/dev/null > gpurun_out/tl/bench_shard_n1250000.json; cut -c1-300 gpurun_out/tl/bench_shard_n1250000.json
