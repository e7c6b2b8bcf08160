"""CPU: host logic of the fused missing-data PCA plan (plans/masked_pca.py) with the kernel test
double tests/fake_kernels.py (CPUMaskedKernels), against the live-reference traces of
tests/golden/masked_pca.npz / masked_pca_erasures.npz (NaN at the missing entries) -- single
process and world_size-2 gloo: the plan all-reduces sum y^2, the observation counts per dimension
(a row of W is an ignored plate only if NO rank observes it, node.py:457-526), the packed
statistics M_d | r_d and the scalar sums after every X.update()."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _inputs(name):
    g = np.load(os.path.join(GOLDEN, name))
    return g, {k[3:]: g[k] for k in g.files if k.startswith('in_')}


def _build(y, mask, x0, shard=False):
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import Runtime
    from fake_kernels import CPUMaskedKernels
    from models import build_masked_pca
    Q = build_masked_pca(nodes, VB, y, mask, x0, shard=shard)
    plan = Q.plans[0]
    assert type(plan).__name__ == 'MaskedPCAPlan'
    rt = Runtime(device='cpu')
    plan._rt, plan._kernels = rt, CPUMaskedKernels(rt)
    return Q


def _compare(Q, g, tag, lo=None, hi=None):
    n = len(g[tag + '_L'])
    np.testing.assert_allclose(Q.L[:n], g[tag + '_L'], rtol=1e-9, err_msg=tag)
    for nm in ('Y', 'W', 'X', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[nm]][:n], g['%s_L_%s' % (tag, nm)], rtol=1e-8, atol=1e-7,
                                   err_msg=nm)
    np.testing.assert_allclose(Q['W'].u[0], g[tag + '_W_u0'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Q['W'].u[1], g[tag + '_W_u1'], rtol=1e-7, atol=1e-9)
    sl = slice(lo, hi)
    np.testing.assert_allclose(Q['X'].u[0], g[tag + '_X_u0'][:, sl], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array([np.asarray(u) for u in Q['tau'].u], dtype=np.float64),
                               g[tag + '_tau_u'], rtol=1e-9)
    np.testing.assert_allclose(Q['alpha'].u[0], g[tag + '_alpha_u0'], rtol=1e-8)


@pytest.mark.parametrize('name,tag,n_iter', [('masked_pca.npz', 'm0', 5), ('masked_pca.npz', 'm1', 5),
                                             ('masked_pca.npz', 'm2', 4),
                                             ('masked_pca_erasures.npz', 'e0', 4),
                                             ('masked_pca_erasures.npz', 'e1', 4)])
def test_plan_reproduces_reference_trace(name, tag, n_iter):
    g, inp = _inputs(name)
    Q = _build(inp[tag + '_y'], inp[tag + '_mask'], inp[tag + '_x0'])
    Q.update(repeat=n_iter, verbose=False)
    _compare(Q, g, tag)
    np.testing.assert_allclose(Q['X'].u[1][0, :8], g[tag + '_X_u1_first'], rtol=1e-7, atol=1e-9)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g, inp = _inputs('masked_pca_erasures.npz')
    tag = 'e0'                       # dimensions without data, one seen on a single plate (rank 0)
    y, mask, x0 = inp[tag + '_y'], inp[tag + '_mask'], inp[tag + '_x0']
    N = y.shape[1]
    lo, hi = (0, 23) if rank == 0 else (23, N)          # ragged shards
    Q = _build(np.ascontiguousarray(y[:, lo:hi]), np.ascontiguousarray(mask[:, lo:hi]), x0[lo:hi],
               shard=True)
    Q.update(repeat=len(g[tag + '_L']), verbose=False)
    assert Q.plans[0].n_total == N
    _compare(Q, g, tag, lo, hi)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:Q.iter])
    dist.destroy_process_group()


def test_two_rank_shard_matches_unsharded_reference(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    assert np.array_equal(r[0]['L'], r[1]['L'])


def test_checkpoint_round_trip(tmp_path):
    """VB.save / VB.load on the missing-data block: a restored model continues bit for bit."""
    g, inp = _inputs('masked_pca.npz')
    args = (inp['m1_y'], inp['m1_mask'], inp['m1_x0'])
    Q = _build(*args)
    Q.update(repeat=2, verbose=False)
    fn = str(tmp_path / 'mpca.bin')
    Q.save(filename=fn)
    Q.update(repeat=2, verbose=False)
    Q2 = _build(*args)
    Q2.load(filename=fn)
    assert Q2.iter == 2 and np.array_equal(Q2.L[:2], Q.L[:2])
    Q2.update(repeat=2, verbose=False)
    assert np.array_equal(Q2.L[:4], Q.L[:4])
    np.testing.assert_array_equal(Q2['X'].u[0], Q['X'].u[0])
    np.testing.assert_array_equal(Q2['W'].u[0], Q['W'].u[0])
