#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 5 shape (linear state-space model,
GaussianMarkovChain + SumMultiply, T time steps x B sequences) on the generic device
engine with the batched smoother kernels.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--b', type=int, default=1000)
    p.add_argument('--t', type=int, default=1000)
    p.add_argument('--m', type=int, default=8)
    p.add_argument('--d', type=int, default=4)
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    B, T, M, D = a.b, a.t, a.m, a.d
    rs = np.random.RandomState(0)
    a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
    x = np.zeros((B, T, D))
    x[:, 0] = rs.normal(size=(B, D))
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(B, D))
    c_true = rs.normal(size=(M, D))
    y = np.einsum('md,btd->mbt', c_true, x) + 0.3 * rs.normal(size=(M, B, T))
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, plates=(B,),
                            name='X')
    X.initialize_from_value(rs.normal(size=(B, T, D)))
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(rs.normal(size=(M, 1, 1, D)))
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    Q.update(repeat=1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q.update(repeat=a.steps, verbose=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({'metric': 'VB iterations/sec, LSSM B=%d T=%d M=%d D=%d' % (B, T, M, D),
                      'value': 1.0 / dt, 's_per_iter': dt, 'elbo': [float(v) for v in Q.L[:Q.iter]],
                      'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == '__main__':
    main()
