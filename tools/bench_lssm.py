#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 5 shape (linear state-space model,
GaussianMarkovChain + SumMultiply, T time steps x B sequences) on the generic device
engine with the batched smoother kernels.  Prints one JSON line.

Multi-GPU (config 5 is quoted on 8 GPUs): launch with ``python -m torch.distributed.run
--nproc-per-node N --master-addr 127.0.0.1 tools/bench_lssm.py --b 100000``; the B sequences
are split over the ranks (``X.shard(-1)``), A, C, tau stay replicated, and the engine completes
the plate sums with RCCL all-reduces."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# At the config-5 size (B = 1e5 sequences, 99 GB of live arrays) the default caching allocator
# fragments: 0.36 s/iter; with expandable segments 0.19 s/iter (same numbers at B <= 5e4).
os.environ.setdefault('PYTORCH_HIP_ALLOC_CONF', 'expandable_segments:True')


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--b', type=int, default=1000)
    p.add_argument('--t', type=int, default=1000)
    p.add_argument('--m', type=int, default=8)
    p.add_argument('--d', type=int, default=4)
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        local = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        backend = os.environ.get('VMP_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    B_total, T, M, D = a.b, a.t, a.m, a.d
    B = B_total * (rank + 1) // world - B_total * rank // world     # this rank's sequences
    rs = np.random.RandomState(rank)
    a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
    x = np.zeros((B, T, D))
    x[:, 0] = rs.normal(size=(B, D))
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(B, D))
    c_true = rs.normal(size=(M, D))
    y = np.einsum('md,btd->mbt', c_true, x) + 0.3 * rs.normal(size=(M, B, T))
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, plates=(B,),
                            name='X')
    if world > 1:
        X.shard(-1)
    X.initialize_from_value(rs.normal(size=(B, T, D)))
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(rs.normal(size=(M, 1, 1, D)))
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    Q.update(repeat=1, verbose=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    Q.update(repeat=a.steps, verbose=False)
    barrier()
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        print(json.dumps({'metric': 'VB iterations/sec, LSSM B=%d T=%d M=%d D=%d'
                                    % (B_total, T, M, D),
                          'value': 1.0 / dt, 's_per_iter': dt, 'n_gpus': world,
                          'sequences_per_rank': B,
                          'elbo': [float(v) for v in Q.L[:Q.iter]],
                          'peak_mem_GB': torch.cuda.max_memory_allocated() / 1e9}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
