"""
CPU: the N>1 path (one process per rank, torch.distributed) with the gloo
backend and world_size 2.  Each rank owns a contiguous shard of the observation
plate; the plan's collectives (Gram matrix / Syy / N once, or the streamed
statistics every iteration) must make the sharded run reproduce the UNSHARDED
reference trace bit-for-bit up to fp64 summation order.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, stats, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bayespy_amd.nodes as nodes
    from bayespy_amd.device import Runtime
    from bayespy_amd.inference import VB
    from fake_kernels import CPURuntimeKernels
    from models import build_pca
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    y, x0 = g['y'], g['x0']
    N = y.shape[1]
    lo, hi = N * rank // world, N * (rank + 1) // world
    Q = build_pca(nodes, VB, np.ascontiguousarray(y[:, lo:hi]), x0[lo:hi], x0.shape[1])
    rt = Runtime(device='cpu')
    assert rt.world == world and rt.rank == rank
    plan = Q.plans[0]
    plan._rt, plan._kernels, plan.stats = rt, CPURuntimeKernels(rt), stats
    Q.update(repeat=int(g['n_iter']), verbose=False)
    assert plan.n_total == N
    x = Q['X'].u[0][0]
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:Q.iter], x=x, lo=lo, hi=hi,
             W=Q['W'].u[0], tau=np.array(Q['tau'].u))
    dist.destroy_process_group()


@pytest.mark.parametrize('stats', ['gram', 'stream'])
def test_two_rank_shard_matches_unsharded_reference(tmp_path, stats):
    name = 'pca_n777_d20_k5'          # odd N: ragged shards 388 / 389
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, stats, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    for ri in r:
        # every rank holds the same replicated state and the same (global) bound
        np.testing.assert_allclose(ri['L'], g['L'], rtol=1e-10)
        np.testing.assert_allclose(ri['W'], g['W_u0'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(ri['tau'][0], g['tau_u0'], rtol=1e-10)
        np.testing.assert_allclose(ri['x'], g['X_u0'][0, int(ri['lo']):int(ri['hi'])],
                                   rtol=1e-8, atol=1e-10)
    assert np.array_equal(r[0]['L'], r[1]['L'])
