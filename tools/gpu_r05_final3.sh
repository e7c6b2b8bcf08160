#!/bin/bash
# after the last (Python-only) change of round 5: the default bench line and the GPU suite once more, same library build
O=gpurun_out
python bench.py --full-out $O/bench_full.json > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_summary.txt
tail -2 $O/pytest_gpu_summary.txt; tail -c 400 $O/bench_default.json
