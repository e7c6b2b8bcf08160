#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 3 (GMM N=1e7, D=8, K=64, 1 GPU).
Prints one JSON line with the same fields as bench.py (which is the driver's bench, the PCA
headline): metric / value / ms_per_step / roofline / cpu_baseline."""
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
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-sample-n', type=int, default=100_000)
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
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q.update(repeat=a.steps, verbose=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = plan.pass_times_ms(64)                    # HIP events recorded inside the timed region
    avg = sum(m[0] for m in ms) / len(ms)
    FS = D * D + D + 1
    flops = 4.0 * N * K * FS                       # 4 N K (D^2 + D + 1)
    byts = 8.0 * N * (D + K)
    tflops = flops / (avg * 1e-3) / 1e12
    out = {
        'metric': 'VB iterations/sec, GMM N=%d D=%d K=%d' % (N, D, K), 'value': a.steps / dt,
        'unit': 'VB iterations/s', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'Gaussian mixture (Mixture+Categorical+GaussianARD+Wishart+'
                               'Dirichlet), N=%d D=%d K=%d, one VB iteration = mu, Lambda, z, '
                               'alpha updates + full ELBO' % (N, D, K)},
        'elbo_first': float(Q.L[0]), 'elbo_last': float(Q.L[Q.iter - 1]),
        'roofline': {'kernel': 'gmm_pass_kernel', 'bound': 'mfma', 'achieved': tflops,
                     'peak': 78.6, 'unit': 'TFLOP/s', 'frac': tflops / 78.6, 'traffic': None,
                     'avg_launch_ms': avg, 'reduce_ms': sum(m[1] for m in ms) / len(ms),
                     'hbm_achieved_GBs': byts / (avg * 1e-3) / 1e9,
                     'alg_flops_per_launch': flops, 'alg_bytes_per_launch': byts},
    }
    if not a.no_cpu_baseline:
        from oracle.gmm import GMMOracle, make_gmm_data
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get('num_threads', 1) for i in threadpool_info()] or [1])
        except Exception:
            cores = os.cpu_count() or 1
        ns = min(a.cpu_sample_n, N)
        ys, lab0 = make_gmm_data(ns, D, K, seed=42)
        o = GMMOracle(ys, lab0, K)
        o.iterate(1, keep_r=False)
        t = time.time()
        o.iterate(2, keep_r=False)
        dtc = (time.time() - t) / 2
        out['cpu_baseline'] = {
            'value': 1.0 / (dtc * (N / float(ns))), 'unit': 'VB iterations/s', 'cores': int(cores),
            'kind': 'port',
            'sample': 'oracle/gmm.py (NumPy fp64, chunked) on N=%d rows of the same D=%d,K=%d '
                      'workload, 2 timed iterations at %.3f s/iter, extrapolated linearly to N=%d'
                      % (ns, D, K, dtc, N)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
