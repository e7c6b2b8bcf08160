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
    Q = build_pca(nodes, VB, np.ascontiguousarray(y[:, lo:hi]), x0[lo:hi], x0.shape[1],
                  shard=True)
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


def _replica_worker(rank, world, port, name, out_dir):
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
    # NOT declared sharded: every rank fits its own, complete model (rank 1 a different one)
    y, x0 = g['y'], g['x0']
    if rank == 1:
        y, x0 = y[:, ::2], x0[::2]
    Q = build_pca(nodes, VB, np.ascontiguousarray(y), x0, x0.shape[1])
    rt = Runtime(device='cpu')
    plan = Q.plans[0]
    plan._rt, plan._kernels = rt, CPURuntimeKernels(rt)
    Q.update(repeat=int(g['n_iter']), verbose=False)
    assert plan.sharded is False and plan.n_total == y.shape[1]
    # checkpoints of replicas do not collide: each rank writes the name it was given
    fn = os.path.join(out_dir, 'ckpt_rank%d' % rank)
    Q.save(filename=fn)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:Q.iter])
    dist.destroy_process_group()


def test_undeclared_model_under_distributed_is_an_independent_replica(tmp_path):
    """One sharding contract: without Node.shard() a fused plan does NOT
    all-reduce, also when torch.distributed is initialised (ranks may even differ)."""
    name = 'pca_n500_d6_k3'
    port = _free_port()
    mp.spawn(_replica_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'rank1.npz'))
    np.testing.assert_allclose(r0['L'], g['L'], rtol=1e-10)
    assert not np.allclose(r1['L'], g['L'])


def _ckpt_worker(rank, world, port, name, out_dir):
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

    def build():
        Q = build_pca(nodes, VB, np.ascontiguousarray(y[:, lo:hi]), x0[lo:hi], x0.shape[1],
                      shard=True)
        rt = Runtime(device='cpu')
        Q.plans[0]._rt, Q.plans[0]._kernels = rt, CPURuntimeKernels(rt)
        return Q
    Q = build()
    Q.update(repeat=2, verbose=False)
    fn = os.path.join(out_dir, 'shared_name')
    Q.save(filename=fn)                 # same name on every rank: per-rank files
    Q.update(repeat=2, verbose=False)
    Q2 = build()
    Q2.load(filename=fn)
    Q2.update(repeat=2, verbose=False)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:Q.iter], L2=Q2.L[:Q2.iter],
             x=Q['X'].u[0], x2=Q2['X'].u[0])
    dist.destroy_process_group()


def test_sharded_checkpoint_is_written_per_rank(tmp_path):
    """VB.save of a model with a sharded plate writes <name>.rank<r>of<w> on every rank (no
    concurrent writes of one file, no restoring another rank's shard); the resumed run
    continues bit for bit."""
    port = _free_port()
    mp.spawn(_ckpt_worker, args=(2, port, 'pca_n777_d20_k5', str(tmp_path)), nprocs=2, join=True)
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('shared_name'))
    assert len(files) == 2 and all('.rank' in f for f in files), files
    for k in range(2):
        r = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % k))
        np.testing.assert_array_equal(r['L'], r['L2'])
        np.testing.assert_array_equal(r['x'], r['x2'])


def _mean_worker(rank, world, port, out_dir):
    """The prior-mean model of tests/golden/pca_prior_mean.npz (case m3) with the observation plate
    split over two ranks: mu is replicated state, the collectives are those of the zero-mean block."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bayespy_amd.nodes as nodes
    from bayespy_amd.device import Runtime
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.pca import PCAPlan
    from fake_kernels import CPURuntimeKernels
    g = np.load(os.path.join(GOLDEN, 'pca_prior_mean.npz'))
    y, x0, mu = g['m3_y'], g['m3_x0'], g['m3_mu']
    D, N = y.shape
    K = x0.shape[1]
    lo, hi = N * rank // world, N * (rank + 1) // world
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = nodes.GaussianARD(mu, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, hi - lo), name='X').shard(-1)
    F = nodes.SumMultiply('i,i', W, X, name='F')
    tau = nodes.Gamma(1e-2, 1e-2, name='tau')
    Y = nodes.GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None, lo:hi, :])
    Y.observe(np.ascontiguousarray(y[:, lo:hi]))
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    rt = Runtime(device='cpu')
    plan = Q.plans[0]
    assert isinstance(plan, PCAPlan) and plan.mu0 is not None
    plan._rt, plan._kernels = rt, CPURuntimeKernels(rt)
    n = len(g['m3_L'])
    Q.update(repeat=n, verbose=False)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:n], W=Q['W'].u[0],
             alpha=Q['alpha'].u[0])
    dist.destroy_process_group()


def test_two_rank_shard_with_a_prior_mean_of_w(tmp_path):
    world = 2
    mp.spawn(_mean_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, 'pca_prior_mean.npz'))
    for r in range(world):
        out = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        np.testing.assert_allclose(out['L'], g['m3_L'], rtol=1e-10)
        np.testing.assert_allclose(out['W'], g['m3_W_u0'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(out['alpha'], g['m3_alpha_u0'], rtol=1e-9)
