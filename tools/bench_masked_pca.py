#!/usr/bin/env python
"""Secondary measurement: PCA with missing data (array mask, demos/pca.py:80-82 default usage).
BASELINE.md section 2 measured the reference at N=2e4, D=64, K=16, 10% missing: 2.7 s/iter.
(tools/workloads.py:run_masked; also ``python bench.py --config masked``.)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=10_000_000)
    p.add_argument('--d', type=int, default=128)
    p.add_argument('--k', type=int, default=32)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--missing', type=float, default=0.1)
    p.add_argument('--engine', default=None)
    a = p.parse_args()
    from tools import workloads
    print(json.dumps(workloads.run_masked(a.n, a.d, a.k, a.steps, a.warmup, a.missing, a.engine)))


if __name__ == '__main__':
    main()
