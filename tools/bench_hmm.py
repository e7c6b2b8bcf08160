#!/usr/bin/env python
"""Throughput of the forward-backward kernel (vmp_alpha_beta_recursion) and of one VB
iteration of a batched Gaussian HMM.  Prints one JSON line per measurement."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=20000)
    ap.add_argument('--steps', type=int, default=1000, help='transitions per chain')
    ap.add_argument('--k', type=int, default=8)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    import torch
    from bayespy_amd.darray import DArray
    from bayespy_amd.device import get_runtime
    from bayespy_amd.utils import random as drandom
    rt = get_runtime()
    B, N, K = a.chains, a.steps, a.k
    gen = torch.Generator(device=rt.device).manual_seed(1)
    logp0 = DArray(torch.randn(B, K, dtype=torch.float64, device=rt.device, generator=gen))
    logP = DArray(torch.randn(B, N, K, K, dtype=torch.float64, device=rt.device, generator=gen))
    for _ in range(2):
        z0, zz, g = drandom.alpha_beta_recursion(logp0, logP)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(a.reps):
        z0, zz, g = drandom.alpha_beta_recursion(logp0, logP)
    torch.cuda.synchronize()
    dt = (time.time() - t) / a.reps
    alg = 8.0 * B * N * K * K * 3            # logP read twice, zz written once
    nexp = 3.0 * B * N * K * K
    print(json.dumps({'kernel': 'alpha_beta_kernel', 'chains': B, 'transitions': N, 'K': K,
                      'ms': dt * 1e3, 'alg_GBps': alg / dt / 1e9,
                      'exp_per_s': nexp / dt, 'chain_steps_per_s': B * N / dt}))


if __name__ == '__main__':
    main()
