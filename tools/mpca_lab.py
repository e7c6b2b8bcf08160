#!/usr/bin/env python
"""A/B harness of the three plate kernels of the fused missing-data PCA block (one chunk of plates,
same data, every variant in ONE process): python tools/mpca_lab.py [N=1048576] [D=128] [K=32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    import torch
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    dev = rt.device
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    mask = torch.rand(D, N, generator=g, device=dev) >= 0.1
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y, mask=mask)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    Q.update(repeat=2, verbose=False)
    L_ref = Q.L[1]
    plan.enable_timing(True)
    variants = [dict(mpca_sweep_nm=1, mpca_sweep_occ=2), dict(mpca_sweep_nm=1, mpca_sweep_occ=3),
                dict(mpca_sweep_nm=1, mpca_sweep_occ=4), dict(mpca_sweep_nm=2, mpca_sweep_occ=1),
                dict(mpca_sweep_nm=2, mpca_sweep_occ=2), dict(mpca_sweep_nm=2, mpca_sweep_occ=3),
                dict(mpca_sweep_nm=4, mpca_sweep_occ=1), dict(mpca_sweep_nm=4, mpca_sweep_occ=2)]
    extra = [dict(mpca_stats_v=1), dict(mpca_stats_v=2, mpca_stats_ncw=2),
             dict(mpca_stats_v=2, mpca_stats_ncw=3)]
    print('N=%d D=%d K=%d one chunk; ms per chunk' % (N, D, K))
    for rnd in range(2):
        for v in variants + extra:
            for k, val in v.items():
                rt.lib.vmp_tune_set(k.encode(), val)
            for _ in range(3):
                X.update()
            torch.cuda.synchronize()
            t = plan.kernel_times_ms()
            if rnd:
                print('%-46s lambda %.3f  sweep %.3f  stats %.3f' % (v, t['mpca_lambda'],
                                                                    t['mpca_sweep'], t['mpca_stats']))
    # every variant must leave the same state behind: the bound after one more iteration
    Q.update(repeat=1, verbose=False)
    print('L after the variants: %r (second iteration was %r)' % (Q.L[2], L_ref))


if __name__ == '__main__':
    main()
