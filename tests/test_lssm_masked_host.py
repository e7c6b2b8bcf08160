"""CPU: the masked state-space block (plans/lssm_masked.py) on the HOST BUILD of its device code
(tests/host_build.py compiles bayespy_amd/csrc/vmp_lssmm_dev.h -- the text the HIP kernels run --
with g++ behind the vmp_lssmm_* C ABI), against the live-reference traces of
tests/golden/lssm_masked.npz and the oracle.  Covers the arithmetic of the sweeps and of the
replicated-node kernel, the plan's host logic, and the world-2 sharding of the sequence plate."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CASES = [('md', None, False), ('mb', 5, True), ('ms', 3, False), ('me', 4, True), ('m1', 3, False)]
# 5 ... 8 states (two rows of the blocks per lane on the device): lssm_masked_wide.npz
WIDE_CASES = [('w8', None, False), ('w6', 4, True), ('w5', 3, False), ('w7', 4, True)]


def golden_of(tag):
    return np.load(os.path.join(GOLDEN, 'lssm_masked_wide.npz' if tag.startswith('w')
                                else 'lssm_masked.npz'))


def build(y, mask, x0, c0, B, gamma_nu, shard=False, host=True):
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
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
    Y.observe(y, mask=mask)
    nodes = [Y, F, C, gamma, X, A, alpha, tau] + ([nu] if gamma_nu else [])
    Q = VB(*nodes)
    Q.ignore_bound_checks = True
    plan = Q.plans[0]
    assert type(plan).__name__ == 'MaskedLSSMPlan', type(plan).__name__
    if host:
        import host_build
        from bayespy_amd.device import Runtime
        from bayespy_amd.inference.plans.lssm_masked import MaskedLSSMKernels
        rt = Runtime(device='cpu')
        plan._rt, plan._kernels = rt, MaskedLSSMKernels(rt, lib=host_build.lssmm_host())
    track = dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma)
    if gamma_nu:
        track['nu'] = nu
    return Q, track


def check_against_golden(Q, track, g, tag, n):
    np.testing.assert_allclose(Q.L[:n], g[tag + '_L'][:n], rtol=1e-9)
    for nm, nd in track.items():
        np.testing.assert_allclose(Q.l[nd][:n], g['%s_%s_L' % (tag, nm)][:n], rtol=1e-8, atol=1e-7,
                                   err_msg=nm)
        for i, ui in enumerate(nd.u):
            ref = g['%s_%s_u%d' % (tag, nm, i)]
            np.testing.assert_allclose(np.broadcast_to(ui, ref.shape), ref, rtol=1e-7, atol=1e-8,
                                       err_msg='%s u[%d]' % (nm, i))


@pytest.mark.parametrize('tag,B,gamma_nu', CASES + WIDE_CASES)
def test_plan_reproduces_reference_trace(tag, B, gamma_nu):
    g = golden_of(tag)
    Q, track = build(g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    check_against_golden(Q, track, g, tag, n)
    Y = Q['Y']
    np.testing.assert_allclose(Q.l[Y][:n], g[tag + '_Y_L'], rtol=1e-8, atol=1e-7)


@pytest.mark.parametrize('tag,B,gamma_nu', [('mb', 5, True), ('ms', 3, False), ('w6', 4, True),
                                           ('w7', 4, True)])
def test_four_lane_form_of_the_sweeps_on_the_host(tag, B, gamma_nu, monkeypatch):
    """The device's default form -- a sequence dealt over FOUR lanes, operands of other rows by lane
    moves -- run on four host threads in lock step (tests/host/lssmm_host.cpp: lssmm_lanes<4> as an
    exchange between barriers): the live-reference traces, and bit-for-bit the state of the
    one-lane form (the arithmetic of an entry does not depend on the lane count)."""
    g = golden_of(tag)
    n = len(g[tag + '_L'])
    res = []
    for lanes in ('4', '1'):
        monkeypatch.setenv('LSSMM_HOST_LANES', lanes)
        Q, track = build(g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu)
        Q.update(repeat=n, verbose=False)
        if lanes == '4':
            check_against_golden(Q, track, g, tag, n)
        p = Q.plans[0]
        res.append((p.state.numpy().copy(), p.x_means(), p.x_second_moments()))
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_masks_propagate_like_the_reference():
    g = np.load(os.path.join(GOLDEN, 'lssm_masked.npz'))
    Q, track = build(g['me_y'], g['me_mask'], g['me_x0'], g['me_c0'], 4, True)
    m = g['me_mask']
    assert np.array_equal(track['C'].mask.reshape(-1), m.any(axis=(1, 2)))
    assert np.array_equal(track['X'].mask, m.any(axis=(0, 2)))
    assert not track['C'].mask.reshape(-1)[1] and not track['X'].mask[2]


def test_prior_initialised_chain_and_random_sizes_match_the_oracle():
    """No initialize_from_value on X (the prior smoother as the first q(X)), D = 1 ... 4, ragged
    B, a mask per sequence; against oracle/lssm.py (pinned on the live reference)."""
    from oracle.lssm import MaskedLSSMOracle
    rs = np.random.RandomState(5)
    for (M, B, T, D) in [(3, 1, 7, 1), (5, 3, 9, 2), (2, 70, 5, 3), (7, 4, 12, 4), (64, 2, 4, 2)]:
        y = rs.normal(size=(M, B, T))
        mask = rs.rand(M, B, T) < 0.6
        mask[0, 0, 0] = True
        x0 = rs.normal(size=(B, T, D))
        c0 = rs.normal(size=(M, 1, 1, D))
        Q, track = build(np.where(mask, y, np.nan), mask, x0, c0, B, False)
        Q.update(repeat=3, verbose=False)
        o = MaskedLSSMOracle(y, mask, x0, c0.reshape(M, D))
        o.iterate(3)
        np.testing.assert_allclose(Q.L[:3], o.L, rtol=1e-10, err_msg=str((M, B, T, D)))
        np.testing.assert_allclose(track['X'].u[0], o.X, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(track['X'].u[1], o.P, rtol=1e-8, atol=1e-10)
        xpxn = o.Cn + o.X[:, :-1, :, None] * o.X[:, 1:, None, :]
        np.testing.assert_allclose(track['X'].u[2], xpxn, rtol=1e-7, atol=1e-9)


def test_reobserving_keeps_the_posteriors():
    """Y.observe(new data, new mask) after updates: q(X) and the other posteriors stay, the sums
    over data and mask are taken again (stochastic.py:223-273)."""
    from oracle.lssm import MaskedLSSMOracle
    rs = np.random.RandomState(9)
    M, B, T, D = 4, 3, 10, 2
    y = rs.normal(size=(M, B, T))
    mask = rs.rand(M, B, T) < 0.7
    x0, c0 = rs.normal(size=(B, T, D)), rs.normal(size=(M, 1, 1, D))
    Q, track = build(y, mask, x0, c0, B, False)
    Q.update(repeat=2, verbose=False)
    y2 = y + 0.1 * rs.normal(size=y.shape)
    mask2 = mask.copy()
    mask2[1, 1, 3:6] = ~mask2[1, 1, 3:6]
    o = MaskedLSSMOracle(y, mask, x0, c0.reshape(M, D))
    o.iterate(2)
    # the oracle's state with the new data: same q, new sums
    o.y = np.where(mask2, y2, 0.0)
    o.mask = mask2
    o.n_obs = float(mask2.sum())
    o.Syy = float(np.sum(o.y ** 2))
    o._stats(o.V, o.Cn)
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        Q['Y'].observe(y2, mask=mask2)
    assert not rec, [str(w.message) for w in rec]
    Lo, _ = o.lower_bound()
    np.testing.assert_allclose(Q.compute_lowerbound(), Lo, rtol=1e-10)
    Q.update(repeat=1, verbose=False)
    o.iterate(1)
    np.testing.assert_allclose(Q.L[2], o.L[-1], rtol=1e-10)


def _worker(rank, world, port, tag, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = golden_of(tag)
    y, mask, x0, c0 = g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0']
    B = y.shape[1]
    lo, hi = (0, 2) if rank == 0 else (2, B)               # ragged shards of the sequence plate
    mk = mask if mask.shape[1] == 1 else mask[:, lo:hi]
    nu = dict((t, n_) for t, _, n_ in CASES + WIDE_CASES)[tag]
    Q, track = build(y[:, lo:hi], mk, x0[lo:hi], c0, hi - lo, nu, shard=True)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    q.put((rank, np.array(Q.L[:n]), track['C'].u[0], track['X'].u[0]))
    dist.destroy_process_group()


@pytest.mark.parametrize('tag', ['mb', 'me', 'ms', 'w6', 'w7'])
def test_sharded_sequence_plate_world2_gloo(tag):
    """Two ranks, the sequence plate split 2 + (B - 2): the unsharded live-reference trace on both
    ranks (set-up counts and every plate sum all-reduced), each rank's own <x>; w6 / w7: six and
    seven states (two rows of the blocks per lane)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tag, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    g = golden_of(tag)
    for rank, L, cu0, xu0 in res:
        np.testing.assert_allclose(L, g[tag + '_L'], rtol=1e-9)
        np.testing.assert_allclose(cu0, g[tag + '_C_u0'], rtol=1e-7, atol=1e-9)
        lo, hi = (0, 2) if rank == 0 else (2, g[tag + '_y'].shape[1])
        np.testing.assert_allclose(xu0, g[tag + '_X_u0'][lo:hi], rtol=1e-7, atol=1e-9)


def run_masked_rotation_case(host):
    """demos/lssm.py as it ships (array mask with a stretch without data + the rotation speed-up
    after every iteration) against the live-reference golden lssm_masked_rotations.npz."""
    import warnings
    from bayespy_amd.inference import transformations
    g = np.load(os.path.join(GOLDEN, 'lssm_masked_rotations.npz'))
    Q, nd = build(g['y'], g['mask'], g['x0'], g['c0'], None, False, host=host)
    D = g['x0'].shape[-1]
    rotA = transformations.RotateGaussianARD(nd['A'], nd['alpha'], axis=0)
    rotX = transformations.RotateGaussianMarkovChain(nd['X'], rotA)
    rotC = transformations.RotateGaussianARD(nd['C'], nd['gamma'], axis=0)
    R = transformations.RotationOptimizer(rotX, rotC, D)
    Q.update(repeat=2, verbose=False)
    np.testing.assert_allclose(Q.compute_lowerbound(), g['L_before'], rtol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        R.rotate(maxiter=10)
    np.testing.assert_allclose(Q.compute_lowerbound(), g['L_after'], rtol=1e-6)
    for nm in ('A', 'C', 'alpha', 'gamma', 'X'):
        for i, ui in enumerate(nd[nm].u):
            got, ref = np.broadcast_arrays(np.asarray(ui), g['%s_u%d_rot' % (nm, i)])
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
    np.testing.assert_allclose(Ls[:2], g['L'][:2], rtol=1e-5)
    np.testing.assert_allclose(Ls, g['L'], rtol=2e-2)
    assert np.all(np.diff(Ls) > 0)


def test_rotations_on_the_masked_block_match_reference():
    run_masked_rotation_case(host=True)
