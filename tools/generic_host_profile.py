"""Host-side profile (cProfile) of replayed sweeps of the generic engine at BASELINE config 2: where the
time between the read of one iteration's bound and the launch of the next goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
from bayespy_amd.inference import VB
N, D, K = 1_000_000, 64, 16
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(42)
w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
F = SumMultiply('i,i', W, X, name='F')
tau = Gamma(1e-2, 1e-2, name='tau')
Y = GaussianARD(F, tau, name='Y')
X.initialize_from_value(x0[None])
Y.observe(y)
Q = VB(Y, F, W, X, tau, alpha, engine='generic')
Q.ignore_bound_checks = True
Q.update(repeat=6, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
Q.update(repeat=300, verbose=False)
pr.disable()
torch.cuda.synchronize()
print('ms per iteration (under cProfile): %.4f' % ((time.perf_counter() - t0) / 300 * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
