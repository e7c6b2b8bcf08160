"""
GPU parity of the Gaussian Markov chain path (BASELINE.json config 5 family):
``vmp_block_banded_solve`` against known answers of the reference's own
``linalg.block_banded_solve`` and against a dense solve, and linear state-space models
(bayespy/demos/lssm.py) against golden traces of the live reference
(tests/golden/lssm.npz).  Tolerances: ELBO rtol 1e-9, moments rtol 1e-7.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_block_banded_solve_known_answers(golden_dir):
    from bayespy_amd.utils import linalg
    g = np.load(os.path.join(golden_dir, 'lssm.npz'))
    for tag in ('bbs_shared', 'bbs_batch'):
        V, C, x, ld = linalg.block_banded_solve(g[tag + '_A'], g[tag + '_B'], g[tag + '_y'])
        np.testing.assert_allclose(np.broadcast_to(V.numpy(), g[tag + '_V'].shape), g[tag + '_V'],
                                   rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.broadcast_to(C.numpy(), g[tag + '_C'].shape), g[tag + '_C'],
                                   rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(x.numpy(), g[tag + '_x'], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.broadcast_to(ld.numpy(), g[tag + '_ld'].shape),
                                   g[tag + '_ld'], rtol=1e-11)


@pytest.mark.parametrize('T,D,nseq', [(1, 1, 1), (2, 1, 3), (7, 2, 1), (50, 4, 10), (33, 8, 5),
                                      (1000, 4, 200)])
def test_block_banded_solve_vs_dense(T, D, nseq):
    from bayespy_amd.utils import linalg
    rs = np.random.RandomState(T * 10 + D)
    A = rs.normal(size=(T, D, 2 * D))
    A = A @ A.transpose(0, 2, 1) + 4 * D * np.eye(D)
    B = 0.5 * rs.normal(size=(max(T - 1, 1), D, D))[:max(T - 1, 0)]
    y = rs.normal(size=(nseq, T, D))
    V, C, x, ld = linalg.block_banded_solve(A, B if T > 1 else np.zeros((0, D, D)), y)
    if T <= 64:
        full = np.zeros((T * D, T * D))
        for t in range(T):
            full[t * D:(t + 1) * D, t * D:(t + 1) * D] = A[t]
            if t < T - 1:
                full[t * D:(t + 1) * D, (t + 1) * D:(t + 2) * D] = B[t]
                full[(t + 1) * D:(t + 2) * D, t * D:(t + 1) * D] = B[t].T
        inv = np.linalg.inv(full)
        xs = np.linalg.solve(full, y.reshape(nseq, T * D).T).T.reshape(nseq, T, D)
        np.testing.assert_allclose(x.numpy(), xs, rtol=1e-9, atol=1e-12)
        Vn, Cn = V.numpy(), C.numpy()
        for t in range(T):
            np.testing.assert_allclose(Vn[t], inv[t * D:(t + 1) * D, t * D:(t + 1) * D],
                                       rtol=1e-8, atol=1e-12)
            if t < T - 1:
                np.testing.assert_allclose(Cn[t], inv[t * D:(t + 1) * D, (t + 1) * D:(t + 2) * D],
                                           rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(ld.numpy(), np.linalg.slogdet(full)[1], rtol=1e-11)
    else:
        # residual check of the solution for long chains
        xn = x.numpy()
        r = np.einsum('tij,stj->sti', A, xn)
        r[:, :-1] += np.einsum('tij,stj->sti', B, xn[:, 1:])
        r[:, 1:] += np.einsum('tji,stj->sti', B, xn[:, :-1])
        np.testing.assert_allclose(r, y, rtol=1e-8, atol=1e-9)


def test_block_banded_solve_errors():
    from bayespy_amd.utils import linalg
    from bayespy_amd import _lib
    with pytest.raises(ValueError):
        linalg.block_banded_solve(np.ones((3, 2, 2)), np.ones((2, 2, 2)), np.ones((4, 2)))
    with pytest.raises(NotImplementedError):
        linalg.block_banded_solve(np.tile(np.eye(17), (3, 1, 1)), np.zeros((2, 17, 17)),
                                  np.ones((3, 17)))
    bad = np.tile(-np.eye(2), (3, 1, 1))
    with pytest.raises(_lib.NotPositiveDefiniteError):
        linalg.block_banded_solve(bad, np.zeros((2, 2, 2)), np.ones((3, 2)))


def test_block_banded_solve_wide_states_against_dense_solves():
    """8 < K <= 16: the workgroup-per-sequence form of the matrix recursions against dense
    NumPy solves of the assembled block-tridiagonal system (shared and per-sequence matrices)."""
    from bayespy_amd.utils import linalg
    rs = np.random.RandomState(17)
    for K, T, nm, ny in ((9, 5, 1, 3), (12, 7, 4, 4), (16, 4, 1, 2)):
        Bm = 0.3 * rs.normal(size=(nm, T - 1, K, K))
        Am = np.empty((nm, T, K, K))
        for b in range(nm):
            for t in range(T):
                q = rs.normal(size=(K, K))
                Am[b, t] = q @ q.T + 3 * K * np.eye(K)
        y = rs.normal(size=(ny, T, K))
        V, C, x, ld = linalg.block_banded_solve(Am if nm > 1 else Am[0], Bm if nm > 1 else Bm[0], y)
        V, C, x, ld = V.numpy(), C.numpy(), x.numpy(), ld.numpy()
        for s in range(ny):
            b = s if nm > 1 else 0
            big = np.zeros((T * K, T * K))
            for t in range(T):
                big[t * K:(t + 1) * K, t * K:(t + 1) * K] = Am[b, t]
                if t < T - 1:
                    big[t * K:(t + 1) * K, (t + 1) * K:(t + 2) * K] = Bm[b, t]
                    big[(t + 1) * K:(t + 2) * K, t * K:(t + 1) * K] = Bm[b, t].T
            inv = np.linalg.inv(big)
            np.testing.assert_allclose(x[s].reshape(-1), inv @ y[s].reshape(-1), rtol=1e-9, atol=1e-12)
            Vb = V[b] if nm > 1 else V
            Cb = C[b] if nm > 1 else C
            for t in range(T):
                np.testing.assert_allclose(Vb[t], inv[t * K:(t + 1) * K, t * K:(t + 1) * K],
                                           rtol=1e-9, atol=1e-12)
                if t < T - 1:
                    np.testing.assert_allclose(Cb[t], inv[t * K:(t + 1) * K, (t + 1) * K:(t + 2) * K],
                                               rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(np.ravel(ld)[b], np.linalg.slogdet(big)[1], rtol=1e-10)


def _build_lssm(g, tag, B, gamma_nu, engine=None):
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    y, x0, c0 = g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0']
    M = y.shape[0]
    T, D = x0.shape[-2], x0.shape[-1]
    plates_x = () if B is None else (B,)
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    nu = Gamma(1e-3, 1e-3, plates=(D,), name='nu') if gamma_nu else np.ones(D)
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, nu, n=T, plates=plates_x,
                            name='X')
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    cplates = (M, 1) if B is None else (M, 1, 1)
    C = GaussianARD(0, gamma, shape=(D,), plates=cplates, name='C')
    C.initialize_from_value(c0)
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    nodes = [Y, F, C, gamma, X, A, alpha, tau]
    if gamma_nu:
        nodes.append(nu)
    Q = VB(*nodes, engine=engine)
    Q.ignore_bound_checks = True
    track = dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma)
    if gamma_nu:
        track['nu'] = nu
    return Q, track


@pytest.mark.parametrize('tag,B,gamma_nu', [('lssm1', None, False), ('lssm1g', None, True),
                                             ('lssmBc', 6, False), ('lssmB', 6, True)])
@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_lssm_matches_reference(golden_dir, tag, B, gamma_nu, engine):
    """Both engines: the fused block (vmp_lssm_*: one shared covariance recursion, per-sequence
    mean recursions, plate sums) and the generic message passing."""
    g = np.load(os.path.join(golden_dir, 'lssm.npz'))
    Q, track = _build_lssm(g, tag, B, gamma_nu, engine=None if engine == 'fused' else 'generic')
    assert type(Q.plans[0]).__name__ == ('LSSMPlan' if engine == 'fused' else 'GenericPlan')
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


@pytest.mark.parametrize('tag,B,gamma_nu', [('w8', 5, True), ('w12', 6, True), ('w16', 4, False),
                                             ('w10single', None, True)])
def test_lssm_with_up_to_16_states_matches_reference(golden_dir, tag, B, gamma_nu):
    """8, 10, 12 and 16 latent states on the FUSED block against live-reference traces
    (tests/golden/lssm_wide_states.npz, oracle/make_golden.py:lssm_wide_state_cases): D = 8 is the
    largest register-resident instance, 8 < D <= 16 the big-state path (sweeps on the projected
    data carrying the state only, plate sums behind them, the shared covariance recursion on one
    workgroup through LDS).  The demo itself (demos/lssm.py:193-247) runs D = 10."""
    g = np.load(os.path.join(golden_dir, 'lssm_wide_states.npz'))
    Q, track = _build_lssm(g, tag, B, gamma_nu, engine=None)
    assert type(Q.plans[0]).__name__ == 'LSSMPlan'
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


@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_lssm_from_prior_initialisation_matches_reference(golden_dir, engine):
    """Every node at its default (prior) initialisation except C: the chain starts from
    p(X | <A>, nu, mu0, Lam0) (expfamily.py:168-184) -- on the fused block the smoother without the
    message of the observations.  Initial moments and the trace against the live reference."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'lssm_prior_init.npz'))
    y, c0 = g['y'], g['c0']
    M, B, T = y.shape
    D = c0.shape[-1]
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(D,), name='alpha')
    A = nodes.GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    X = nodes.GaussianMarkovChain(np.zeros(D), 1e-2 * np.identity(D), A, np.ones(D), n=T,
                                  plates=(B,), name='X')
    gamma = nodes.Gamma(1e-2, 1e-2, plates=(D,), name='gamma')
    C = nodes.GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(c0)
    tau = nodes.Gamma(1e-2, 1e-2, name='tau')
    F = nodes.SumMultiply('i,i', C, X, name='F')
    Y = nodes.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau, engine=None if engine == 'fused' else 'generic')
    Q.ignore_bound_checks = True
    assert type(Q.plans[0]).__name__ == ('LSSMPlan' if engine == 'fused' else 'GenericPlan')
    for i in range(3):
        # the reference keeps the (sequence-independent) prior moments with a unit plate
        got, ref = np.broadcast_arrays(np.asarray(X.u[i]), g['X_u%d_init' % i])
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12, err_msg='initial X.u[%d]' % i)
    n = len(g['L'])
    Q.update(repeat=n, verbose=False)
    np.testing.assert_allclose(Q.L[:n], g['L'], rtol=1e-9)
    for nm, nd in dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma).items():
        for i, ui in enumerate(nd.u):
            got, ref = np.broadcast_arrays(np.asarray(ui), g['%s_u%d' % (nm, i)])
            np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-9, err_msg='%s u[%d]' % (nm, i))


@pytest.mark.parametrize('tag,B', [('one', None), ('batch', 5)])
@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_lssm_rotations_match_reference(golden_dir, tag, B, engine):
    """The rotation speed-up of demos/lssm.py:134-190 -- RotateGaussianMarkovChain(X,
    RotateGaussianARD(A, alpha)) against RotateGaussianARD(C, gamma) -- on both engines: one
    stand-alone rotation after two plain iterations (bound, rotated moments), then five iterations
    with the rotation after each.  The K x K optimisation is a truncated nonlinear CG on the host
    (see models.check_rotation_results): early values tight, the tail loose."""
    import warnings
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.inference import transformations
    g = np.load(os.path.join(golden_dir, 'lssm_rotations.npz'))
    y, x0, c0 = g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0']
    D = x0.shape[-1]
    T = x0.shape[-2]
    M = y.shape[0]
    px = () if B is None else (B,)
    alpha = nodes.Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = nodes.GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = nodes.GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T,
                                  plates=px, name='X')
    X.initialize_from_value(x0)
    gamma = nodes.Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = nodes.GaussianARD(0, gamma, shape=(D,), plates=(M, 1) if B is None else (M, 1, 1), name='C')
    C.initialize_from_value(c0)
    tau = nodes.Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = nodes.SumMultiply('i,i', C, X, name='F')
    Y = nodes.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau, engine=None if engine == 'fused' else 'generic')
    Q.ignore_bound_checks = True
    assert type(Q.plans[0]).__name__ == ('LSSMPlan' if engine == 'fused' else 'GenericPlan')
    rotA = transformations.RotateGaussianARD(A, alpha, axis=0)
    rotX = transformations.RotateGaussianMarkovChain(X, rotA)
    rotC = transformations.RotateGaussianARD(C, gamma, axis=0)
    R = transformations.RotationOptimizer(rotX, rotC, D)
    Q.update(repeat=2, verbose=False)
    np.testing.assert_allclose(Q.compute_lowerbound(), g[tag + '_L_before'], rtol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        R.rotate(maxiter=10)
    np.testing.assert_allclose(Q.compute_lowerbound(), g[tag + '_L_after'], rtol=1e-6)
    for nm, nd in dict(A=A, C=C, alpha=alpha, gamma=gamma, X=X).items():
        for i, ui in enumerate(nd.u):
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
    np.testing.assert_allclose(Q.compute_lowerbound(), g[tag + '_L_final'], rtol=2e-2)


def test_switching_state_space_model_matches_reference(golden_dir):
    """SwitchingGaussianMarkovChain inside the model of bayespy/demos/lssm_sd.py (a categorical
    Markov chain picks the dynamics matrix of every transition): five VB iterations against
    the live reference, bound and every node's moments and bound term."""
    import os
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_switching_case
    f = np.load(os.path.join(golden_dir, 'switching_lssm.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_switching_case(N_, VB, g)
    np.testing.assert_allclose(res['sw_L'], f['sw_L'], rtol=1e-8)
    for k, v in res.items():
        if isinstance(v, list):
            for i, vi in enumerate(v):
                ref = f['%s_%d' % (k, i)]
                np.testing.assert_allclose(np.broadcast_to(vi, ref.shape), ref, rtol=1e-6,
                                           atol=1e-8, err_msg='%s[%d]' % (k, i))
        else:
            np.testing.assert_allclose(v, f[k], rtol=1e-7, atol=1e-6, err_msg=k)


def test_time_varying_state_space_model_matches_reference(golden_dir):
    """VaryingGaussianMarkovChain inside the model of bayespy/demos/lssm_tvd.py (a second
    Gaussian Markov chain, seen as Gaussian variables and sliced [1:], mixes K dynamics
    matrices): five VB iterations against the live reference."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_varying_case
    f = np.load(os.path.join(golden_dir, 'varying_lssm.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_varying_case(N_, VB, g)
    np.testing.assert_allclose(res['tv_L'], f['tv_L'], rtol=1e-8)
    for k, v in res.items():
        if isinstance(v, list):
            for i, vi in enumerate(v):
                ref = f['%s_%d' % (k, i)]
                np.testing.assert_allclose(np.broadcast_to(vi, ref.shape), ref, rtol=1e-6,
                                           atol=1e-8, err_msg='%s[%d]' % (k, i))
        else:
            np.testing.assert_allclose(v, f[k], rtol=1e-7, atol=1e-6, err_msg=k)


def test_reference_lssm_doctest_known_answer(golden_dir):
    """doc/source/examples/lssm.rst: linear state-space model with a TEN-dimensional latent
    chain (the workgroup-per-sequence form of the smoother kernels), 400 instances, 80 % of the
    30 x 400 observations missing: "Iteration 1: loglike=-1.439704e+05 ... Iteration 10:
    loglike=-1.051441e+04" (data, mask and the reference's random initial C in
    tests/golden/lssm_doctest.npz)."""
    from bayespy_amd.nodes import GaussianARD, GaussianMarkovChain, Gamma, Dot
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'lssm_doctest.npz'))
    M, N, D = 30, 400, 10
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=N, name='X')
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1), name='C')
    F = Dot(C, X, name='F')
    assert F.plates == (M, N)
    C.initialize_from_value(g['C_init'])
    tau = Gamma(1e-5, 1e-5, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    Q = VB(X, C, gamma, A, alpha, tau, Y)
    Y.observe(g['y'], mask=g['mask'])
    Q.update(repeat=10, verbose=False)
    L = Q.L[:10]
    assert '%e' % L[0] == '-1.439704e+05' and '%e' % L[9] == '-1.051441e+04'
    np.testing.assert_allclose(L, g['L'], rtol=1e-8)
    np.testing.assert_allclose(tau.u[0], g['tau_u0'], rtol=1e-6)
    np.testing.assert_allclose(A.u[0], g['A_u0'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(X.u[0], g['X_u0'], rtol=1e-5, atol=1e-7)
