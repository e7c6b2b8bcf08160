"""Lab of the masked state-space block (csrc/vmp_lssmm.hip): the forms of the sweeps side by side on
one box -- four lanes per sequence with the statistics carried by the backward sweep (default),
the same without the fusion, one thread per sequence -- at the bench size and at B = 1e5.

    python tools/lssmm_lab.py [--B 10000 100000] [--D 4] [--M 8] [--T 1000] [--iters 20]

Prints one JSON line per (B, form): ms per iteration, the kernel times from the library's events
(forward | backward + statistics), the bound after the iterations (the forms must agree)."""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, nargs='+', default=[10_000, 100_000])
    ap.add_argument('--T', type=int, default=1000)
    ap.add_argument('--D', type=int, default=4)
    ap.add_argument('--M', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--forms', nargs='+', default=['l4f1', 'l4f0', 'l1'])
    args = ap.parse_args()
    import torch
    from bayespy_amd import _lib
    from tools import workloads
    lib = _lib.load()
    forms = {'l4f1': (4, 1), 'l4f0': (4, 0), 'l1': (1, 0)}
    for B in args.B:
        for name in args.forms:
            lanes, fuse = forms[name]
            if lanes == 1 and args.D > 4:
                continue
            lib.vmp_tune_set(b'lssmm_lanes', lanes)
            lib.vmp_tune_set(b'lssmm_fuse', fuse)
            Q, info = workloads.build_lssm_masked(B, args.T, args.M, args.D)
            del info
            plan = Q.plans[0]
            Q.update(repeat=2, verbose=False)
            plan.enable_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Q.update(repeat=args.iters, verbose=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
            NS = args.D * (args.D + 1) // 2
            byts = 8.0 * B * args.T * (args.M + 1 + args.D + NS)
            print(json.dumps({'B': B, 'T': args.T, 'M': args.M, 'D': args.D, 'form': name,
                              'ms_per_iter': 1e3 * dt, 'kernel_ms': plan.kernel_times_ms(),
                              'frac_hbm': byts / dt / 8e12, 'L_last': float(Q.L[Q.iter - 1]),
                              'peak_GB': torch.cuda.max_memory_allocated() / 1e9}), flush=True)
            del Q, plan
            gc.collect()
            torch.cuda.empty_cache()
    lib.vmp_tune_set(b'lssmm_lanes', 0)
    lib.vmp_tune_set(b'lssmm_fuse', 1)


if __name__ == '__main__':
    main()
