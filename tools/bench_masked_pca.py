#!/usr/bin/env python
"""Secondary measurement: PCA with missing data (array mask, demos/pca.py:80-82 default usage)
on the generic device engine -- per-plate K x K posteriors.  BASELINE.md section 2 measured the
reference at N=2e4, D=64, K=16, 10% missing: 2.7 s/iter."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=20000)
    p.add_argument('--d', type=int, default=64)
    p.add_argument('--k', type=int, default=16)
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    N, D, K = a.n, a.d, a.k
    rs = np.random.RandomState(42)
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < 0.9
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(rs.normal(size=(1, N, K)))
    Y.observe(y, mask=mask)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    Q.update(repeat=1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q.update(repeat=a.steps, verbose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({'metric': 'VB iterations/sec, masked PCA N=%d D=%d K=%d (10%% missing)'
                      % (N, D, K), 'value': 1.0 / dt, 's_per_iter': dt,
                      'engine': type(Q.plans[0]).__name__,
                      'elbo': [float(v) for v in Q.L[:Q.iter]],
                      'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == '__main__':
    main()
