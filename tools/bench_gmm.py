#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 3 (GMM N=1e7, D=8, K=64, 1 GPU).
Prints one JSON line; not the driver's bench (bench.py is the PCA headline)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=10_000_000)
    p.add_argument('--d', type=int, default=8)
    p.add_argument('--k', type=int, default=64)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    a = p.parse_args()
    import numpy as np
    import torch
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    N, D, K = a.n, a.d, a.k
    dev = torch.device('cuda')
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_random()
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    Q.update(repeat=a.warmup, verbose=False)
    plan.enable_timing(True)
    ms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        Q.update(repeat=1, verbose=False)
        ms.append(plan.last_pass_ms())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    avg = sum(m[0] for m in ms) / len(ms)
    FS = D * D + D + 1
    flops = 4.0 * N * K * FS                       # 4 N K (D^2 + D + 1)
    byts = 8.0 * N * (D + K)
    print(json.dumps({
        'metric': 'VB iterations/sec, GMM N=%d D=%d K=%d' % (N, D, K), 'value': a.steps / dt,
        'ms_per_step': 1e3 * dt / a.steps, 'pass_ms': avg, 'reduce_ms': sum(m[1] for m in ms) / len(ms),
        'alg_TFLOPs': flops / (avg * 1e-3) / 1e12, 'alg_GBs': byts / (avg * 1e-3) / 1e9,
        'elbo': [float(Q.L[0]), float(Q.L[Q.iter - 1])]}))


if __name__ == '__main__':
    main()
