#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 3 (GMM N=1e7, D=8, K=64, 1 GPU).
Prints one JSON line with the same fields as bench.py (tools/workloads.py:run_gmm; also
``python bench.py --config gmm`` and the "extra" list of the default bench.py run)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=10_000_000)
    p.add_argument('--d', type=int, default=8)
    p.add_argument('--k', type=int, default=64)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-sample-n', type=int, default=100_000)
    a = p.parse_args()
    from tools import workloads
    print(json.dumps(workloads.run_gmm(a.n, a.d, a.k, a.steps, a.warmup, not a.no_cpu_baseline,
                                       a.cpu_sample_n)))


if __name__ == '__main__':
    main()
