#!/usr/bin/env python
"""Secondary measurement: BASELINE.json config 5 shape (linear state-space model,
GaussianMarkovChain + SumMultiply, T time steps x B sequences) on the fused state-space block
(vmp_lssm_*; BAYESPY_AMD_ENGINE=generic selects the generic engine with the batched smoother
kernels).  Prints one JSON line; `python bench.py --config lssm` is the same measurement.

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




def main():
    p = argparse.ArgumentParser()
    p.add_argument('--b', type=int, default=100000)
    p.add_argument('--t', type=int, default=1000)
    p.add_argument('--m', type=int, default=8)
    p.add_argument('--d', type=int, default=4)
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
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
    from tools import workloads
    out = workloads.run_lssm(a.b, a.t, a.m, a.d, a.steps)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
