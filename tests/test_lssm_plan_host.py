"""CPU: host logic of the fused state-space plan (plans/lssm.py: state set-up, operation queue,
plate sums over a sharded sequence plate, bound terms) with the kernel test double
tests/fake_kernels.py (CPULSSMKernels), against the live-reference golden traces of
tests/golden/lssm.npz -- single process and world_size-2 gloo (BASELINE config 5 shards the plate
of sequences; the plan all-reduces sum y^2 once and the raw plate sums after every X.update())."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _build(y, x0, c0, B, gamma_nu, shard=False):
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    from bayespy_amd.device import Runtime
    from fake_kernels import CPULSSMKernels
    M = y.shape[0]
    T, D = x0.shape[-2], x0.shape[-1]
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    nu = Gamma(1e-3, 1e-3, plates=(D,), name='nu') if gamma_nu else np.ones(D)
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, nu, n=T,
                            plates=() if B is None else (B,), name='X')
    if shard:
        X.shard(-1)
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1) if B is None else (M, 1, 1), name='C')
    C.initialize_from_value(c0)
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    nodes = [Y, F, C, gamma, X, A, alpha, tau] + ([nu] if gamma_nu else [])
    Q = VB(*nodes)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    assert type(plan).__name__ == 'LSSMPlan'
    rt = Runtime(device='cpu')
    plan._rt, plan._kernels = rt, CPULSSMKernels(rt)
    track = dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma)
    if gamma_nu:
        track['nu'] = nu
    return Q, track


@pytest.mark.parametrize('tag,B,gamma_nu', [('lssm1', None, False), ('lssm1g', None, True),
                                            ('lssmBc', 6, False), ('lssmB', 6, True)])
def test_plan_reproduces_reference_trace(tag, B, gamma_nu):
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    Q, track = _build(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    np.testing.assert_allclose(Q.L[:n], g[tag + '_L'], rtol=1e-9)
    for nm, nd in track.items():
        np.testing.assert_allclose(Q.l[nd][:n], g['%s_%s_L' % (tag, nm)], rtol=1e-8, atol=1e-7,
                                   err_msg=nm)
        for i, ui in enumerate(nd.u):
            ref = g['%s_%s_u%d' % (tag, nm, i)]
            np.testing.assert_allclose(np.broadcast_to(ui, ref.shape), ref, rtol=1e-7, atol=1e-9,
                                       err_msg='%s u[%d]' % (nm, i))


def test_operation_queue_one_launch_per_group():
    """The replicated-node updates of an iteration are queued and issued as ONE operation list
    in front of the next X.update() / the bound (STATS .. ELBO are the codes of vmp_lssm_op)."""
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    Q, _ = _build(g['lssmB_y'], g['lssmB_x0'], g['lssmB_c0'], 6, True)
    Q.update(repeat=2, verbose=False)
    calls = list(Q.plans[0].kernels.calls)
    assert calls[:4] == ['relayout_y', 'x_layout', 'smooth_given', 'op1']
    # model order Y, F, C, gamma, X, A, alpha, tau, nu: (C, gamma, XPREP) | x_update, STATS |
    # (A, alpha, tau, nu, ELBO)
    per_iter = ['op2', 'op3', 'op4', 'x_update', 'op1', 'op5', 'op6', 'op7', 'op8', 'op9']
    assert calls[4:] == per_iter * 2
    Q.compute_lowerbound()
    assert Q.plans[0].kernels.calls == calls


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
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    tag, B = 'lssmB', 6
    lo, hi = B * rank // world, B * (rank + 1) // world
    # rank 0 holds two sequences, rank 1 four: ragged shards
    lo, hi = (0, 2) if rank == 0 else (2, 6)
    y, x0 = g[tag + '_y'][:, lo:hi], g[tag + '_x0'][lo:hi]
    Q, track = _build(np.ascontiguousarray(y), np.ascontiguousarray(x0), g[tag + '_c0'], hi - lo,
                      True, shard=True)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    assert Q.plans[0].B_total == B
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:n], x=track['X'].u[0], lo=lo, hi=hi,
             A=track['A'].u[0], C=track['C'].u[0], tau=np.array(track['tau'].u, dtype=np.float64))
    dist.destroy_process_group()


def test_two_rank_shard_matches_unsharded_reference(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    for ri in r:
        np.testing.assert_allclose(ri['L'], g['lssmB_L'], rtol=1e-9)
        np.testing.assert_allclose(ri['A'], g['lssmB_A_u0'], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(np.broadcast_to(ri['C'], g['lssmB_C_u0'].shape), g['lssmB_C_u0'],
                                   rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(ri['x'], g['lssmB_X_u0'][int(ri['lo']):int(ri['hi'])], rtol=1e-7,
                                   atol=1e-9)
    assert np.array_equal(r[0]['L'], r[1]['L'])


def test_checkpoint_round_trip(tmp_path):
    """VB.save / VB.load on the state-space block: a restored model continues bit for bit."""
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    args = (g['lssmB_y'], g['lssmB_x0'], g['lssmB_c0'], 6, True)
    Q, _ = _build(*args)
    Q.update(repeat=2, verbose=False)
    fn = str(tmp_path / 'lssm.bin')
    Q.save(filename=fn)
    Q.update(repeat=2, verbose=False)
    Q2, _ = _build(*args)
    Q2.load(filename=fn)
    assert Q2.iter == 2 and np.array_equal(Q2.L[:2], Q.L[:2])
    Q2.update(repeat=2, verbose=False)
    assert np.array_equal(Q2.L[:4], Q.L[:4])
    np.testing.assert_array_equal(Q2['X'].u[0], Q['X'].u[0])
    np.testing.assert_array_equal(Q2['A'].u[0], Q['A'].u[0])


@pytest.mark.parametrize('tag,B', [('one', None), ('batch', 5)])
def test_rotations_match_reference(tag, B):
    """The rotation speed-up of demos/lssm.py:134-190 on the fused block (the K x K state of the
    plan rotated on the host, the plate-sized means through vmp_lssm_rotate_x) against the
    live-reference golden lssm_rotations.npz: one stand-alone rotation after two iterations, then
    five iterations with the rotation after each (a truncated nonlinear CG: early values tight)."""
    import warnings
    from bayespy_amd.inference import transformations
    g = np.load(os.path.join(GOLDEN, 'lssm_rotations.npz'))
    Q, nd = _build(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], B, False)
    D = g[tag + '_x0'].shape[-1]
    rotA = transformations.RotateGaussianARD(nd['A'], nd['alpha'], axis=0)
    rotX = transformations.RotateGaussianMarkovChain(nd['X'], rotA)
    rotC = transformations.RotateGaussianARD(nd['C'], nd['gamma'], axis=0)
    R = transformations.RotationOptimizer(rotX, rotC, D)
    Q.update(repeat=2, verbose=False)
    np.testing.assert_allclose(Q.compute_lowerbound(), g[tag + '_L_before'], rtol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        R.rotate(maxiter=10)
    np.testing.assert_allclose(Q.compute_lowerbound(), g[tag + '_L_after'], rtol=1e-6)
    for nm in ('A', 'C', 'alpha', 'gamma', 'X'):
        for i, ui in enumerate(nd[nm].u):
            got, ref = np.broadcast_arrays(np.asarray(ui), g['%s_%s_u%d_rot' % (tag, nm, i)])
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max(),
                                       err_msg='%s u[%d] after the rotation' % (nm, i))
    Ls = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(5):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
            R.rotate(maxiter=10)
    Ls = np.array(Ls)
    np.testing.assert_allclose(Ls[:2], g[tag + '_L'][:2], rtol=1e-5)
    np.testing.assert_allclose(Ls, g[tag + '_L'], rtol=2e-2)
    assert np.all(np.diff(Ls) > 0)


def test_reobserving_data_keeps_the_posteriors():
    """Y.observe(new data) after updates changes Y only (stochastic.py:223-273): the fused block
    keeps every posterior and takes a statistics pass of the new data with the current <x>
    (live-reference trace tests/golden/reobserve.npz, lssm_* entries); no restart, no warning."""
    import warnings
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import Runtime
    from fake_kernels import CPULSSMKernels
    from models import run_reobserve_lssm_case
    g = np.load(os.path.join(GOLDEN, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}

    class CPUVB(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            plan = self.plans[0]
            assert type(plan).__name__ == 'LSSMPlan'
            rt = Runtime(device='cpu')
            plan._rt, plan._kernels = rt, CPULSSMKernels(rt)
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        res = run_reobserve_lssm_case(nodes, CPUVB, inp)
    np.testing.assert_allclose(res['lssm_L'], g['lssm_L'], rtol=1e-9)
    np.testing.assert_allclose(res['lssm_L_mid'], g['lssm_L_mid'], rtol=1e-9)
    np.testing.assert_allclose(res['lssm_L_c'], g['lssm_L_c'], rtol=1e-9)
    for key in ('lssm_X_u0', 'lssm_C_u0', 'lssm_A_u0'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)


WIDE = os.path.join(GOLDEN, 'lssm_wide_states.npz')


@pytest.mark.parametrize('tag,B,gamma_nu', [('w8', 5, True), ('w12', 6, True), ('w16', 4, False)])
def test_plan_reproduces_reference_trace_at_8_to_16_states(tag, B, gamma_nu):
    """The host logic of the block does not depend on the number of states (round 6: D <= 16); the
    live-reference traces of tests/golden/lssm_wide_states.npz through the plan and the kernel double."""
    g = np.load(WIDE)
    Q, track = _build(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    np.testing.assert_allclose(Q.L[:n], g[tag + '_L'], rtol=1e-8)
    for nm, nd in track.items():
        for i, ui in enumerate(nd.u):
            ref = g['%s_%s_u%d' % (tag, nm, i)]
            np.testing.assert_allclose(np.broadcast_to(ui, ref.shape), ref, rtol=1e-6,
                                       atol=1e-8 * max(1.0, float(np.abs(ref).max())),
                                       err_msg='%s u[%d]' % (nm, i))


def _wide_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(WIDE)
    tag, B = 'w12', 6
    lo, hi = (0, 4) if rank == 0 else (4, 6)            # ragged shards of the sequence plate
    y, x0 = g[tag + '_y'][:, lo:hi], g[tag + '_x0'][lo:hi]
    Q, track = _build(np.ascontiguousarray(y), np.ascontiguousarray(x0), g[tag + '_c0'], hi - lo,
                      True, shard=True)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    assert Q.plans[0].B_total == B
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), L=Q.L[:n], x=track['X'].u[0], lo=lo, hi=hi,
             A=track['A'].u[0])
    dist.destroy_process_group()


def test_two_rank_shard_at_12_states_matches_unsharded_reference(tmp_path):
    world = 2
    mp.spawn(_wide_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(WIDE)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    for ri in r:
        np.testing.assert_allclose(ri['L'], g['w12_L'], rtol=1e-8)
        np.testing.assert_allclose(ri['A'], g['w12_A_u0'], rtol=1e-6, atol=1e-8)
        ref = g['w12_X_u0'][int(ri['lo']):int(ri['hi'])]
        np.testing.assert_allclose(ri['x'], ref, rtol=1e-6, atol=1e-8 * float(np.abs(ref).max()))
    assert np.array_equal(r[0]['L'], r[1]['L'])
