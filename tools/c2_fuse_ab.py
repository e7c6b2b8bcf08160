"""Same-process A/B of the replicated-node chain of the fused PCA block at BASELINE config 2 (and at
the shard one of eight ranks holds at the headline size): the Gram-form messages to W formed by the
tail kernel itself (tune key pca_fuse_gram = 1) against the separate pca_gram_stats + reduce launches
(= 0), alternating on ONE model (same data, same placement), R rounds of S iterations each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
from bayespy_amd.inference import VB
from bayespy_amd.device import get_runtime

def model(N, D, K):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(42)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    y = torch.empty(D, N, device=dev, dtype=torch.float64)
    step = 1 << 20
    for s in range(0, N, step):
        e = min(N, s + step)
        x = torch.randn(K, e - s, generator=g, device=dev, dtype=torch.float64)
        y[:, s:e] = w @ x + 0.1 * torch.randn(D, e - s, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha'); W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X'); F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau'); Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None]); Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha); Q.ignore_bound_checks = True
    return Q

for N, D, K in ((1_000_000, 64, 16), (1_250_000, 128, 32)):
    Q = model(N, D, K)
    rt = get_runtime()
    Q.update(repeat=10, verbose=False)
    res = {0: [], 1: []}
    for r in range(6):
        for fuse in (0, 1):
            rt.lib.vmp_tune_set(b'pca_fuse_gram', fuse)
            Q.update(repeat=5, verbose=False)
            torch.cuda.synchronize(); t = time.perf_counter()
            Q.update(repeat=200, verbose=False)
            torch.cuda.synchronize(); res[fuse].append((time.perf_counter() - t) / 200 * 1e3)
    # third arm: the separate launches with the per-pass HIP event triples of bench.py's pass timing
    rt.lib.vmp_tune_set(b'pca_fuse_gram', 0)
    Q.plans[0].enable_timing(True)
    ev = []
    for r in range(6):
        Q.update(repeat=5, verbose=False)
        torch.cuda.synchronize(); t = time.perf_counter()
        Q.update(repeat=60, verbose=False)
        torch.cuda.synchronize(); ev.append((time.perf_counter() - t) / 60 * 1e3)
        Q.plans[0].pass_times_ms(64)
    Q.plans[0].enable_timing(False)
    print('N=%d D=%d K=%d  separate launches, pass timing (event triples) ON: %s  (median %.4f)' % (N, D, K, ' '.join('%.4f' % v for v in ev), np.median(ev)))
    print('N=%d D=%d K=%d  separate launches: %s  ms/iter (median %.4f)' % (N, D, K, ' '.join('%.4f' % v for v in res[0]), np.median(res[0])))
    print('N=%d D=%d K=%d  fused tail:        %s  ms/iter (median %.4f)' % (N, D, K, ' '.join('%.4f' % v for v in res[1]), np.median(res[1])))
    del Q
