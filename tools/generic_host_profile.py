"""Host cost of one generic-engine PCA iteration: tiny plates (GPU time negligible) under cProfile."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
from bayespy_amd.inference import VB
def build(N, D=64, K=16):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev); g.manual_seed(42)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha'); W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X'); F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau'); Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None]); Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha, engine='generic'); Q.ignore_bound_checks = True
    return Q
for N in (2048, 1_000_000):
    Q = build(N)
    Q.update(repeat=3, verbose=False); torch.cuda.synchronize()
    t = time.perf_counter(); Q.update(repeat=20, verbose=False); torch.cuda.synchronize()
    print('N', N, 'ms/iter', (time.perf_counter() - t) / 20 * 1e3)
    if N == 2048:
        pr = cProfile.Profile(); pr.enable()
        Q.update(repeat=20, verbose=False); torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(35)
        st.sort_stats('cumulative').print_stats(45)
    del Q
