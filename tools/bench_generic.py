#!/usr/bin/env python
"""Roofline check of the generic plate-broadcast kernels (all HBM-bound): prints
achieved GB/s of algorithmic traffic for vmp_ewise, vmp_sum_multiply (plate sum and
long reduction), vmp_softmax_moments and the matrices/s of vmp_spd_batched."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=5):
    import torch
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    import numpy as np
    import torch
    from bayespy_amd import darray as da
    from bayespy_amd.utils import misc, linalg
    dev = torch.device('cuda')
    out = {}
    N, D = 10_000_000, 16
    a = da.DArray(torch.randn(N, D, device=dev, dtype=torch.float64))
    b = da.DArray(torch.rand(N, D, device=dev, dtype=torch.float64) + 0.5)
    t = timeit(lambda: da.fuse(lambda x, y: x * y + 0.5, a, b))
    out['ewise x*y+c (N,16)'] = {'GB/s': 3 * 8 * N * D / t / 1e9, 'ms': t * 1e3}
    t = timeit(lambda: da.fuse(lambda x, y: da.digamma(y) - da.log(y) + x, a, b))
    out['ewise digamma/log'] = {'GB/s': 3 * 8 * N * D / t / 1e9, 'ms': t * 1e3}
    w = da.DArray(torch.rand(N, 1, device=dev, dtype=torch.float64))
    t = timeit(lambda: misc.sum_multiply(a, w, axis=0))
    out['sum_multiply plate-sum (N,16)*(N,1)->(16,)'] = {'GB/s': 8 * N * (D + 1) / t / 1e9,
                                                        'ms': t * 1e3}
    t = timeit(lambda: misc.sum_multiply(a, b, axis=-1))
    out['sum_multiply inner (N,16).(N,16)->(N,)'] = {'GB/s': 8 * N * (2 * D + 1) / t / 1e9,
                                                    'ms': t * 1e3}
    phi = da.DArray(torch.randn(N, 64, device=dev, dtype=torch.float64))
    t = timeit(lambda: misc.normalized_exp(phi))
    out['softmax (N,64)'] = {'GB/s': 2 * 8 * N * 64 / t / 1e9, 'ms': t * 1e3}
    for n, batch in ((8, 1_000_000), (32, 100_000)):
        A = torch.randn(batch, n, n, device=dev, dtype=torch.float64)
        C = da.DArray(A @ A.transpose(1, 2) + n * torch.eye(n, device=dev, dtype=torch.float64))
        t = timeit(lambda: linalg.chol(C), reps=3)
        out['spd_batched n=%d batch=%d' % (n, batch)] = {
            'Mmatrices/s': batch / t / 1e6, 'GB/s': 2 * 8 * batch * n * n / t / 1e9, 'ms': t * 1e3}
    # dense contraction on the matrix cores (masked-PCA message shape)
    Dd, Nn, Kk = 128, 1_000_000, 32
    m = da.DArray(torch.randn(Dd, Nn, 1, 1, device=dev, dtype=torch.float64))
    xx = da.DArray(torch.randn(1, Nn, Kk, Kk, device=dev, dtype=torch.float64))
    t = timeit(lambda: misc.sum_multiply(m, xx, axis=(1,)), reps=3)
    out['gemm (128,N)x(N,32,32) N=1e6'] = {'TFLOP/s': 2.0 * Dd * Nn * Kk * Kk / t / 1e12,
                                            'GB/s': 8.0 * Nn * (Dd + Kk * Kk) / t / 1e9,
                                            'ms': t * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
