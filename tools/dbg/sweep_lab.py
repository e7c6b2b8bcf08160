import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
from bayespy_amd.inference import VB
from bayespy_amd.device import get_runtime
N, D, K = 1 << 21, 128, 32
rt = get_runtime(); dev = rt.device
g = torch.Generator(device=dev); g.manual_seed(1)
w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
mask = torch.rand(D, N, generator=g, device=dev) >= 0.1
x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
del x
alpha = Gamma(1e-2, 1e-2, plates=(K,))
W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1))
X = GaussianARD(0, 1, shape=(K,), plates=(1, N))
F = SumMultiply('i,i', W, X)
tau = Gamma(1e-2, 1e-2)
Y = GaussianARD(F, tau)
X.initialize_from_value(x0[None])
Y.observe(y, mask=mask)
Q = VB(Y, F, W, X, tau, alpha)
Q.ignore_bound_checks = True
Q.update(repeat=2, verbose=False)
plan = Q.plans[0]
plan.enable_timing(True)
for dbg in (1, 2, 1, 2):
    rt.lib.vmp_tune_set(b'mpca_sweep_nm', dbg)
    for _ in range(2):
        X.update()
    torch.cuda.synchronize()
    print('dbg', dbg, plan.kernel_times_ms())
