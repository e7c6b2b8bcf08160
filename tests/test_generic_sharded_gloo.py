"""
CPU: the generic engine with the observation plate split over TWO ranks (gloo, one process per
rank) on the NumPy double of the generic entry points: the fused shared-covariance update keeps
LOCAL plate sums, the messages to the replicated nodes and the bound terms are completed by
all-reduce -- both ranks reproduce the unsharded live-reference traces, replicated nodes bitwise
equal across the ranks.  Ragged split (a third / two thirds).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    for k in ('BAYESPY_AMD_LAZY_DOT_MIN', 'BAYESPY_AMD_PAIRWISE_MIN', 'BAYESPY_AMD_HOIST_MIN',
              'BAYESPY_AMD_MEMO_MIN'):
        os.environ[k] = '1'
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import host_generic
    rt = host_generic.install()
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    res = {}
    if case == 'pca':
        from models import build_pca
        g = np.load(os.path.join(GOLDEN, 'pca_n777_d20_k5.npz'))
        y, x0 = g['y'], g['x0']
        N = y.shape[1]
        cut = [0, N // 3, N]
        lo, hi = cut[rank], cut[rank + 1]
        Q = build_pca(nodes, VB, np.ascontiguousarray(y[:, lo:hi]), x0[lo:hi], x0.shape[1],
                      shard=True, engine='generic')
        Q.update(repeat=int(g['n_iter']), verbose=False)
        res['L'] = np.array(Q.L[:Q.iter])
        res['W'] = np.asarray(Q['W'].u[0])
        res['x'] = np.asarray(Q['X'].u[0])
        res['fused'] = np.array([rt.lib.calls.get('vmp_gaussian_shared_update', 0)])
    else:
        from bayespy_amd.nodes import (GaussianARD, Gaussian, Wishart, Dirichlet, Categorical,
                                       Mixture)
        g = np.load(os.path.join(GOLDEN, 'gmm_n400_d3_k4.npz'))
        y, lab0 = g['y'], g['lab0']
        N, D = y.shape
        K = g['alpha_u0'].shape[-1]
        cut = [0, N // 3, N]
        lo, hi = cut[rank], cut[rank + 1]
        alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
        z = Categorical(alpha, plates=(hi - lo,), name='z').shard(-1)
        mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
        Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
        Y = Mixture(z, Gaussian, mu, Lam, plates=(hi - lo,), name='Y')
        z.initialize_from_value(lab0[lo:hi])
        Y.observe(y[lo:hi])
        Q = VB(Y, mu, Lam, z, alpha, engine='generic')
        Q.ignore_bound_checks = True
        Q.update(repeat=int(g['n_iter']), verbose=False)
        res['L'] = np.array(Q.L[:Q.iter])
        res['W'] = np.asarray(mu.u[0])
        res['x'] = np.asarray(z.u[0])
    res['calls'] = np.array([rt.collective_calls['torch']])
    res['lo'], res['hi'] = lo, hi
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **res)
    dist.destroy_process_group()


@pytest.mark.parametrize('case', ['pca', 'gmm'])
def test_two_ranks_of_the_generic_engine_match_the_unsharded_reference(tmp_path, case):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % k)) for k in range(2)]
    if case == 'pca':
        g = np.load(os.path.join(GOLDEN, 'pca_n777_d20_k5.npz'))
        ref_w, ref_x = g['W_u0'], g['X_u0'][0]
        for k in range(2):
            assert int(r[k]['fused'][0]) >= 2 * int(g['n_iter'])
    else:
        g = np.load(os.path.join(GOLDEN, 'gmm_n400_d3_k4.npz'))
        ref_w, ref_x = g['mu_u0'], g['z_u0']
    for k in range(2):
        np.testing.assert_allclose(r[k]['L'], g['L'], rtol=1e-9)
        np.testing.assert_allclose(r[k]['W'], ref_w, rtol=1e-7, atol=1e-10)
        x = r[k]['x'][0] if case == 'pca' else r[k]['x']
        np.testing.assert_allclose(x, ref_x[int(r[k]['lo']):int(r[k]['hi'])], rtol=1e-6, atol=1e-10)
        assert int(r[k]['calls'][0]) > 0
    assert np.array_equal(r[0]['L'], r[1]['L']) and np.array_equal(r[0]['W'], r[1]['W'])
