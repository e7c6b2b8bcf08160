"""CPU (gloo, world_size 2): a rank whose shard of the observation plate is EMPTY -- what happens
when the plate has fewer elements than ranks, or the split leaves a rank nothing.  Rank 0 holds the
whole data set, rank 1 none; every plan must take part in the collectives with zero-size arrays and
both ranks must end with the live-reference trace of the unsharded model.  Kernel test doubles of
tests/fake_kernels.py (the GPU versions of the single-rank N = 0 cases: test_empty_plates_gpu.py)."""
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


def _worker(rank, world, port, which, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from fake_kernels import attach_cpu
    from models import build_pca, build_masked_pca
    own = rank == 0
    if which == 'pca':
        g = np.load(os.path.join(GOLDEN, 'pca_n500_d6_k3.npz'))
        y, x0 = g['y'], g['x0']
        n = y.shape[1] if own else 0
        Q = build_pca(nodes, VB, np.ascontiguousarray(y[:, :n]), x0[:n], x0.shape[1], shard=True)
        ref, iters = g['L'], int(g['n_iter'])
    elif which == 'masked':
        f = np.load(os.path.join(GOLDEN, 'masked_pca.npz'))
        y, m, x0 = f['in_m1_y'], f['in_m1_mask'], f['in_m1_x0']
        n = y.shape[1] if own else 0
        Q = build_masked_pca(nodes, VB, np.ascontiguousarray(y[:, :n]),
                             np.ascontiguousarray(m[:, :n]), x0[:n], shard=True)
        ref, iters = f['m1_L'], len(f['m1_L'])
    elif which == 'gmm':
        from test_gmm_plan_host import _build
        g = np.load(os.path.join(GOLDEN, 'gmm_n400_d3_k4.npz'))
        n = g['y'].shape[0] if own else 0
        Q = _build(np.ascontiguousarray(g['y'][:n]), g['lab0'][:n], 4, shard=True)
        ref, iters = g['L'], int(g['n_iter'])
    elif which == 'lssm_masked':
        from test_lssm_masked_host import build
        g = np.load(os.path.join(GOLDEN, 'lssm_masked.npz'))
        b = 5 if own else 0
        Q, _ = build(np.ascontiguousarray(g['mb_y'][:, :b]), np.ascontiguousarray(g['mb_mask'][:, :b]),
                     np.ascontiguousarray(g['mb_x0'][:b]), g['mb_c0'], b, True, shard=True, host=False)
        ref, iters = g['mb_L'], len(g['mb_L'])
    else:
        from test_lssm_plan_host import _build
        g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
        b = 6 if own else 0
        Q, _ = _build(np.ascontiguousarray(g['lssmB_y'][:, :b]),
                      np.ascontiguousarray(g['lssmB_x0'][:b]), g['lssmB_c0'], b, True, shard=True)
        ref, iters = g['lssmB_L'], len(g['lssmB_L'])
    attach_cpu(Q)
    Q.update(repeat=iters, verbose=False)
    np.testing.assert_allclose(Q.L[:iters], ref, rtol=1e-9)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:iters])
    dist.destroy_process_group()


@pytest.mark.parametrize('which', ['pca', 'masked', 'gmm', 'lssm', 'lssm_masked'])
def test_a_rank_with_an_empty_shard(tmp_path, which):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), which, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    assert np.array_equal(r[0]['L'], r[1]['L'])
