#!/usr/bin/env python
"""A/B harness of the plate pass of the fused missing-data PCA block (same data, every variant in
ONE process): python tools/mpca_lab.py [N=4194304] [D=128] [K=32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
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
    del x

    def build(chunk):
        os.environ['BAYESPY_AMD_MPCA_CHUNK'] = str(chunk)
        os.environ['BAYESPY_AMD_MPCA_PIPELINE'] = '1'
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
        return Q, X

    variants = [
        (1 << 20, dict(mpca_streams=0)),
        (1 << 20, dict(mpca_streams=1, mpca_lambda_wgs=2, mpca_sweep_wgs=2, mpca_stats_wgs=2)),
        (1 << 19, dict(mpca_streams=1, mpca_lambda_wgs=2, mpca_sweep_wgs=2, mpca_stats_wgs=2)),
        (1 << 19, dict(mpca_streams=0)),
        (1 << 20, dict(mpca_streams=1, mpca_lambda_wgs=1, mpca_sweep_wgs=2, mpca_stats_wgs=1)),
    ]
    if os.environ.get('MPCA_LAB_ONE'):
        variants = variants[:1]
    if os.environ.get('MPCA_LAB_FUSE'):
        # the fused form (precision GEMM inside the per-plate kernel, no Lam~ in HBM) against the
        # separate one; MPCA_LAB_STAGGER = start delays of the odd workgroups to try (units of 3.5 us)
        variants = [(1 << 20, dict(mpca_streams=0, mpca_fuse=0))]
        for st in [int(v) for v in os.environ.get('MPCA_LAB_STAGGER', '0,6').split(',')]:
            variants.append((1 << 20, dict(mpca_streams=0, mpca_fuse=1, mpca_fuse_stagger=st)))
    print('N=%d D=%d K=%d; ms per X.update(), per-chunk kernel times (HIP events), bound after two '
          'iterations' % (N, D, K))
    for chunk, knobs in variants:
        for k, val in knobs.items():
            rt.lib.vmp_tune_set(k.encode(), val)
        Q, X = build(chunk)
        Q.update(repeat=2, verbose=False)
        plan = Q.plans[0]
        plan.enable_timing(True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            X.update()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        km = plan.kernel_times_ms()
        plan.enable_timing(False)
        print('chunk %8d %-40s %8.2f ms  %s  L=%r' % (
            chunk, knobs, 1e3 * dt,
            ' '.join('%s=%.3f' % (k, v) for k, v in (km or {}).items() if k != 'chunk_plates'), Q.L[1]))
        del Q, X, plan


if __name__ == '__main__':
    main()
