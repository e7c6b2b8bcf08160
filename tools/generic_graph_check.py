"""Graph-replayed sweeps against eager sweeps of the same model: bound traces and moments must be
bit-identical; then timing at N = 1e6."""
import os, sys, time
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
    return Q, (W, X, tau, alpha)
res = {}
for mode in ('0', '1'):
    os.environ['BAYESPY_AMD_GRAPH'] = mode
    Q, nodes = build(100_000)
    Q.update(repeat=8, verbose=False)
    # a read-only operation and a partial update between sweeps, then more sweeps
    m_mid = nodes[0].get_moments()[0].copy()
    Q.update(repeat=4, verbose=False)
    nodes[1].update()
    Q.update(repeat=5, verbose=False)
    res[mode] = (Q.L[:Q.iter].copy(), [np.asarray(m) for n in nodes for m in n.get_moments()], m_mid)
    print('mode', mode, 'info', nodes[0]._plan.graph_info(), flush=True)
    del Q, nodes
L0, M0, mid0 = res['0']; L1, M1, mid1 = res['1']
print('bound traces identical:', np.array_equal(L0, L1), 'max abs diff', np.max(np.abs(L0 - L1)))
print('moments identical:', all(np.array_equal(a, b) for a, b in zip(M0, M1)), np.array_equal(mid0, mid1))
for mode in ('0', '1'):
    os.environ['BAYESPY_AMD_GRAPH'] = mode
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    Q, nodes = build(1_000_000)
    Q.update(repeat=5, verbose=False); torch.cuda.synchronize()
    t = time.perf_counter(); Q.update(repeat=50, verbose=False); torch.cuda.synchronize()
    print('mode', mode, 'N=1e6 ms/iter', (time.perf_counter() - t) / 50 * 1e3, 'peak GB',
          torch.cuda.max_memory_allocated() / 1e9, 'reserved GB', torch.cuda.memory_reserved() / 1e9,
          nodes[0]._plan.graph_info(), flush=True)
    del Q, nodes
