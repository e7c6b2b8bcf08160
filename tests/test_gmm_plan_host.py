"""CPU: host logic of the fused Gaussian-mixture plan (pattern match, update order, statistics,
lower-bound terms, sharded plate sums) with the kernel test double tests/fake_kernels.py
(CPUGMMKernels), against the live-reference golden vectors -- single process and world_size-2
gloo (each rank owns a contiguous shard of the observation plate; the plan all-reduces the
statistics T = [R, S1, S2] and the softmax sums after every pass, node.py:650)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _build(y, lab0, K, shard=False):
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    from bayespy_amd.device import Runtime
    from fake_kernels import CPUGMMKernels
    N, D = y.shape
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    if shard:
        z.shard(-1)
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    assert type(plan).__name__ == 'GMMPlan'
    rt = Runtime(device='cpu')
    plan._rt, plan._kernels = rt, CPUGMMKernels(rt)
    return Q


@pytest.mark.parametrize('name', ['gmm_n400_d3_k4', 'gmm_n3000_d8_k16'])
def test_plan_reproduces_reference_trace(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    K = g['mu_u0'].shape[0]
    Q = _build(g['y'], g['lab0'], K)
    Q.update(repeat=int(g['n_iter']), verbose=False)
    np.testing.assert_allclose(Q.L[:Q.iter], g['L'], rtol=1e-10)
    for k in ('Y', 'mu', 'Lambda', 'z', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:Q.iter], g['L_' + k], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(Q['z'].u[0], g['z_u0'], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(Q['mu'].u[0], g['mu_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['mu'].u[1], g['mu_u1'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['Lambda'].u[0], g['Lambda_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['Lambda'].u[1], g['Lambda_u1'], rtol=1e-10)
    np.testing.assert_allclose(Q['alpha'].u[0], g['alpha_u0'], rtol=1e-10)


def test_update_order_one_pass_per_iteration_and_bound_cache():
    g = np.load(os.path.join(GOLDEN, 'gmm_n400_d3_k4.npz'))
    Q = _build(g['y'], g['lab0'], 4)
    Q.update(repeat=2, verbose=False)
    calls = list(Q.plans[0].kernels.calls)
    assert calls[:2] == ['init_state', 'stats_from_labels']
    per_iter = ['update_mu', 'update_lambda', 'prepare_z', 'pass', 'update_alpha', 'lower_bound']
    assert calls[2:] == per_iter * 2
    # the bound is evaluated once per state: asking again launches nothing
    Q.compute_lowerbound()
    assert Q.plans[0].kernels.calls == calls
    # an explicit node order, as VB.update(*nodes) allows (vmp.py:139-141)
    Q.update(Q['z'], Q['mu'], repeat=1, verbose=False)
    assert Q.plans[0].kernels.calls[len(calls):] == ['prepare_z', 'pass', 'update_mu', 'lower_bound']
    R, S1, S2 = Q.plans[0].statistics()
    r = Q['z'].u[0]
    np.testing.assert_allclose(R, r.sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(S1, r.T @ g['y'], rtol=1e-12, atol=1e-12)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    y, lab0 = g['y'], g['lab0']
    N = y.shape[0]
    lo, hi = N * rank // world, N * (rank + 1) // world
    Q = _build(np.ascontiguousarray(y[lo:hi]), lab0[lo:hi], g['mu_u0'].shape[0], shard=True)
    assert Q.plans[0].rt.world == world
    Q.update(repeat=int(g['n_iter']), verbose=False)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:Q.iter], r=Q['z'].u[0], lo=lo, hi=hi,
             mu=Q['mu'].u[0], Lam=Q['Lambda'].u[0], alpha=Q['alpha'].u[0])
    dist.destroy_process_group()


def test_two_rank_shard_matches_unsharded_reference(tmp_path):
    name = 'gmm_n3000_d8_k16'
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    for ri in r:
        # every rank holds the same replicated state and the same (global) bound
        np.testing.assert_allclose(ri['L'], g['L'], rtol=1e-10)
        np.testing.assert_allclose(ri['mu'], g['mu_u0'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(ri['Lam'], g['Lambda_u0'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(ri['alpha'], g['alpha_u0'], rtol=1e-10)
        np.testing.assert_allclose(ri['r'], g['z_u0'][int(ri['lo']):int(ri['hi'])], rtol=1e-8,
                                   atol=1e-12)
    assert np.array_equal(r[0]['L'], r[1]['L'])


def test_checkpoint_round_trip(tmp_path):
    """VB.save / VB.load (vmp.py:237-356) on the mixture block: the packed state and the
    responsibilities travel; a restored model continues bit for bit."""
    g = np.load(os.path.join(GOLDEN, 'gmm_n400_d3_k4.npz'))
    Q = _build(g['y'], g['lab0'], 4)
    Q.update(repeat=2, verbose=False)
    fn = str(tmp_path / 'gmm.bin')
    Q.save(filename=fn)
    Q.update(repeat=2, verbose=False)
    Q2 = _build(g['y'], g['lab0'], 4)
    Q2.load(filename=fn)
    assert Q2.iter == 2 and np.array_equal(Q2.L[:2], Q.L[:2])
    Q2.update(repeat=2, verbose=False)
    assert np.array_equal(Q2.L[:4], Q.L[:4])
    np.testing.assert_array_equal(Q2['z'].u[0], Q['z'].u[0])
    np.testing.assert_array_equal(Q2['mu'].u[0], Q['mu'].u[0])
