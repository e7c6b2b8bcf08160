"""
GPU parity of the generic device message-passing engine
(bayespy_amd/inference/plans/generic.py) against golden traces of the LIVE
reference on identical inputs (tests/golden/*.npz from oracle/make_golden.py):
config 1 (quickstart, incl. the reference's doctest known answer), masked PCA,
vector GaussianARD, Gaussian+Wishart, Dirichlet+Categorical, the Gaussian
mixture of demos/mog.py, and the PCA block forced through the generic engine.

Tolerances: ELBO rtol 1e-9; moments rtol 1e-7; one-hot / counts exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ELBO_RTOL = 1e-9
MOM_RTOL = 1e-7


def _trace(Q, n):
    Q.ignore_bound_checks = True
    Q.update(repeat=n, verbose=False)
    return Q.L[:n]


def test_quickstart_known_answer_on_device(golden_dir):
    """doc/source/user_guide/quickstart.rst:111-118 reproduced by HIP kernels."""
    from bayespy_amd.nodes import GaussianARD, Gamma
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.generic import GenericPlan
    for name in ('quickstart_n10', 'quickstart_n1000'):
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        data = g['data']
        mu = GaussianARD(0, 1e-6, name='mu')
        tau = Gamma(1e-6, 1e-6, name='tau')
        y = GaussianARD(mu, tau, plates=(len(data),), name='y')
        y.observe(data)
        Q = VB(y, mu, tau)
        assert isinstance(Q.plans[0], GenericPlan)
        n = int(g['n_iter'])
        Q.ignore_bound_checks = True
        for i in range(n):
            Q.update(repeat=1, verbose=False)
            np.testing.assert_allclose([float(v) for v in mu.u], g['mu_u'][i], rtol=MOM_RTOL)
            np.testing.assert_allclose([float(v) for v in tau.u], g['tau_u'][i], rtol=MOM_RTOL)
        np.testing.assert_allclose(Q.L[:n], g['L'], rtol=ELBO_RTOL)
        if name == 'quickstart_n10':
            assert ['%e' % v for v in Q.L[:n]] == ['-6.020956e+01', '-5.820527e+01',
                                                   '-5.820290e+01', '-5.820288e+01']


def _check_nodes(g, tag, nodes):
    for nm, nd in nodes.items():
        u = nd.u
        for i, ui in enumerate(u):
            ref = g['%s_%s_u%d' % (tag, nm, i)]
            got = np.broadcast_to(ui, ref.shape) if np.shape(ui) != ref.shape else ui
            np.testing.assert_allclose(got, ref, rtol=MOM_RTOL, atol=1e-9,
                                       err_msg='%s.%s u[%d]' % (tag, nm, i))


def test_masked_pca_matches_reference(golden_dir):
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.generic import GenericPlan
    g = np.load(os.path.join(golden_dir, 'small_models.npz'))
    y, mask, x0 = g['mpca_y'], g['mpca_mask'], g['mpca_x0']
    D, N = y.shape
    K = x0.shape[1]
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y, mask=mask)
    Q = VB(Y, F, W, X, tau, alpha, engine='generic')
    assert isinstance(Q.plans[0], GenericPlan)
    L = _trace(Q, 4)
    np.testing.assert_allclose(L, g['mpca_L'], rtol=ELBO_RTOL)
    _check_nodes(g, 'mpca', dict(W=W, X=X, tau=tau, alpha=alpha))
    for nm, nd in dict(W=W, X=X, tau=tau, alpha=alpha).items():
        np.testing.assert_allclose(Q.l[nd][:4], g['mpca_%s_L' % nm], rtol=1e-8, atol=1e-8)


def test_vector_gaussian_ard_matches_reference(golden_dir):
    from bayespy_amd.nodes import GaussianARD, Gamma
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'small_models.npz'))
    data = g['vard_data']
    mu = GaussianARD(0, 1e-3, shape=(3,), name='mu')
    al = Gamma(1e-3, 1e-3, plates=(3,), name='al')
    yy = GaussianARD(mu, al, shape=(3,), plates=(len(data),), name='yy')
    yy.observe(data)
    Q = VB(yy, mu, al)
    np.testing.assert_allclose(_trace(Q, 4), g['vard_L'], rtol=ELBO_RTOL)
    _check_nodes(g, 'vard', dict(mu=mu, al=al))


def test_gaussian_wishart_matches_reference(golden_dir):
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'small_models.npz'))
    data = g['gw_data']
    mu = GaussianARD(0, 1e-3, shape=(3,), name='mu')
    Lam = Wishart(3, np.identity(3), name='Lam')
    yg = Gaussian(mu, Lam, plates=(len(data),), name='yg')
    yg.observe(data)
    Q = VB(yg, mu, Lam)
    np.testing.assert_allclose(_trace(Q, 4), g['gw_L'], rtol=ELBO_RTOL)
    _check_nodes(g, 'gw', dict(mu=mu, Lam=Lam))


def test_dirichlet_categorical_matches_reference(golden_dir):
    from bayespy_amd.nodes import Dirichlet, Categorical
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'small_models.npz'))
    p = Dirichlet(np.array([1.0, 0.5, 2.0, 1.5]), name='p')
    z = Categorical(p, plates=(30,), name='z')
    z.observe(g['dc_lab'])
    Q = VB(z, p)
    np.testing.assert_allclose(_trace(Q, 2), g['dc_L'], rtol=ELBO_RTOL)
    _check_nodes(g, 'dc', dict(p=p))
    # one-hot moments of the observed labels: integer indexing, bit-exact
    ref = np.zeros((30, 4))
    ref[np.arange(30), g['dc_lab']] = 1
    assert np.array_equal(z.u[0], ref)
    with pytest.raises(ValueError):
        z.observe(np.full(30, 7))
        z.u


@pytest.mark.parametrize('engine', ['generic', 'fused'])
@pytest.mark.parametrize('name', ['gmm_n400_d3_k4', 'gmm_n3000_d8_k16'])
def test_gaussian_mixture_matches_reference(golden_dir, name, engine):
    """demos/mog.py:17-64 (Mixture + Categorical + Gaussian + Wishart + Dirichlet), through
    the generic message-passing engine and through the fused GMM block."""
    from bayespy_amd.nodes import (GaussianARD, Gaussian, Wishart, Dirichlet, Categorical,
                                   Mixture)
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    y, lab0 = g['y'], g['lab0']
    N, D = y.shape
    K = g['alpha_u0'].shape[-1]
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha, engine=None if engine == 'fused' else 'generic')
    assert type(Q.plans[0]).__name__ == ('GMMPlan' if engine == 'fused' else 'GenericPlan')
    n = int(g['n_iter'])
    L = _trace(Q, n)
    np.testing.assert_allclose(L, g['L'], rtol=ELBO_RTOL)
    for k in ('Y', 'mu', 'Lambda', 'z', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:n], g['L_' + k], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(z.u[0], g['z_u0'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(mu.u[0], g['mu_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(mu.u[1], g['mu_u1'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Lam.u[0], g['Lambda_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Lam.u[1], g['Lambda_u1'], rtol=MOM_RTOL)
    np.testing.assert_allclose(alpha.u[0], g['alpha_u0'], rtol=MOM_RTOL)


def test_pca_block_through_generic_engine(golden_dir):
    """The same PCA golden trace, with the fused block switched off: exercises
    SumMultiply moments/messages (dot.py:316-633) and the plate multipliers."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.generic import GenericPlan
    from models import build_pca
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = build_pca(nodes, VB, g['y'], g['x0'], 3, engine='generic')
    assert isinstance(Q.plans[0], GenericPlan)
    n = int(g['n_iter'])
    np.testing.assert_allclose(_trace(Q, n), g['L'], rtol=ELBO_RTOL)
    np.testing.assert_allclose(Q['W'].u[0], g['W_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(np.broadcast_to(Q['W'].u[1], g['W_u1'].shape), g['W_u1'],
                               rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Q['X'].u[0], g['X_u0'], rtol=MOM_RTOL, atol=1e-10)
    F = Q['F'].get_moments()
    np.testing.assert_allclose(F[0][:, :5], g['F_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(F[1][:, :5], g['F_u1'], rtol=MOM_RTOL, atol=1e-10)


@pytest.mark.parametrize('case', ['fused', 'generic', 'masked'])
def test_rotation_parameter_expansion_matches_reference(golden_dir, case):
    """RotationOptimizer / RotateGaussianARD as the VB callback (demos/pca.py:85-94): the fused
    PCA block, the same model on the generic engine, and PCA with missing values."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB, transformations
    from bayespy_amd.inference.plans.generic import GenericPlan
    from bayespy_amd.inference.plans.pca import PCAPlan
    from models import build_pca, run_rotation_sequence, check_rotation_results
    g = np.load(os.path.join(golden_dir, 'rotations.npz'))
    tag = 'rotm' if case == 'masked' else 'rot'
    y, x0 = g[tag + '_y'], g[tag + '_x0']
    K = x0.shape[1]
    Q = build_pca(nodes, VB, y, x0, K, engine=None if case == 'fused' else 'generic')
    if case == 'masked':
        Q['Y'].observe(y, mask=g[tag + '_mask'])
    assert isinstance(Q.plans[0], PCAPlan if case == 'fused' else GenericPlan)
    res = run_rotation_sequence(Q, K, transformations)
    check_rotation_results(res, g, tag)


@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_reference_pca_doctest_known_answer(golden_dir, engine):
    """doc/source/examples/pca.rst:26-118 (ARD + rotation callback, run to convergence) on
    the device: first bound -2.33...e+03, converged bound ~6.50e+02."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB, transformations
    from models import run_pca_doctest, check_pca_doctest
    g = np.load(os.path.join(golden_dir, 'pca_doctest.npz'))
    Q, nd = run_pca_doctest(nodes, VB, transformations, g,
                            engine=None if engine == 'fused' else 'generic')
    check_pca_doctest(Q, nd, g)


def test_stochastic_variational_inference_matches_reference(golden_dir):
    """plates_multiplier + VB.gradient_step: the mini-batch loop of
    demos/stochastic_inference.py:99-133 against the live-reference trace (bound, global
    moments after every step, local responsibilities of the last mini-batch)."""
    from bayespy_amd.nodes import GaussianARD, Gaussian, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.generic import GenericPlan
    g = np.load(os.path.join(golden_dir, 'svi_gmm.npz'))
    data, batches = g['data'], g['batches']
    N, NB = int(g['N']), int(g['NB'])
    K, D = g['mu0'].shape
    mu = GaussianARD(0, 0.001, shape=(D,), plates=(K,), name='means')
    alpha = Dirichlet(np.ones(K), name='class probabilities')
    Z = Categorical(alpha, plates=(NB,), plates_multiplier=(N / NB,), name='classes')
    Y = Mixture(Z, Gaussian, mu, np.identity(D), name='observations')
    assert Y.plates_multiplier == (N / NB,) and mu.plates_multiplier == ()
    mu.initialize_from_value(g['mu0'])
    Q = VB(Y, Z, mu, alpha)
    Q.ignore_bound_checks = True
    assert isinstance(Q.plans[0], GenericPlan)
    for n in range(len(batches)):
        Y.observe(data[batches[n], :])
        Q.update(Z, verbose=False)
        Q.gradient_step(mu, alpha, scale=(n + 1) ** (-0.7))
        np.testing.assert_allclose(Q.compute_lowerbound(), g['L'][n], rtol=ELBO_RTOL)
        np.testing.assert_allclose(mu.u[0], g['mu_u0'][n], rtol=MOM_RTOL, atol=1e-10)
        np.testing.assert_allclose(alpha.u[0], g['alpha_u0'][n], rtol=MOM_RTOL)
    np.testing.assert_allclose(Z.u[0], g['Z_u0_last'], rtol=MOM_RTOL, atol=1e-12)
    terms = [Y.lower_bound_contribution(), Z.lower_bound_contribution(),
             mu.lower_bound_contribution(), alpha.lower_bound_contribution()]
    np.testing.assert_allclose(terms, g['L_terms_last'], rtol=1e-9, atol=1e-9)


def test_stochastic_vi_streams_minibatches_from_host_memory(golden_dir):
    """The same mini-batch loop with the data kept on the HOST (N D may exceed
    HBM): HostBatchStream gathers batch n+1 into pinned memory and copies it host->HBM on its own
    stream while batch n is in use; the trace is that of the live reference, and with a data set
    much larger than a batch the steps overlap the copies."""
    import time
    import torch
    from bayespy_amd.nodes import GaussianARD, Gaussian, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    from bayespy_amd.utils.streaming import HostBatchStream
    g = np.load(os.path.join(golden_dir, 'svi_gmm.npz'))
    data, batches = g['data'], g['batches']
    N, NB = int(g['N']), int(g['NB'])
    K, D = g['mu0'].shape
    mu = GaussianARD(0, 0.001, shape=(D,), plates=(K,), name='means')
    alpha = Dirichlet(np.ones(K), name='class probabilities')
    Z = Categorical(alpha, plates=(NB,), plates_multiplier=(N / NB,), name='classes')
    Y = Mixture(Z, Gaussian, mu, np.identity(D), name='observations')
    mu.initialize_from_value(g['mu0'])
    Q = VB(Y, Z, mu, alpha)
    Q.ignore_bound_checks = True
    for n, (y_dev, idx) in enumerate(HostBatchStream(data, list(batches))):
        assert isinstance(y_dev, torch.Tensor) and y_dev.is_cuda
        np.testing.assert_array_equal(idx, batches[n])
        Y.observe(y_dev)
        Q.update(Z, verbose=False)
        Q.gradient_step(mu, alpha, scale=(n + 1) ** (-0.7))
        np.testing.assert_allclose(Q.compute_lowerbound(), g['L'][n], rtol=ELBO_RTOL)
        np.testing.assert_allclose(mu.u[0], g['mu_u0'][n], rtol=MOM_RTOL, atol=1e-10)
    assert n == len(batches) - 1
    # a host array of 1 GB streamed in 64 MB mini-batches of contiguous rows
    rows, cols, nb = 1 << 21, 64, 1 << 17
    big = np.random.RandomState(0).normal(size=(rows, cols))
    order = [slice(s, s + nb) for s in range(0, rows, nb)]
    tot = torch.zeros((), dtype=torch.float64, device='cuda')
    t0 = time.perf_counter()
    for y_dev, _ in HostBatchStream(big, order):
        tot += (y_dev * y_dev).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    np.testing.assert_allclose(float(tot), float(np.sum(big * big)), rtol=1e-12)
    assert dt < 5.0, 'streaming 1 GB from the host took %.2f s' % dt


@pytest.mark.parametrize('model', ['pca_fused', 'gmm_fused', 'masked_pca_generic'])
def test_checkpoint_round_trip_on_device(golden_dir, tmp_path, model):
    """VB.save / VB.load (vmp.py:237-356): device state -> file -> a fresh model continues
    bit-for-bit (the kernels are deterministic)."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_pca

    def build():
        if model == 'gmm_fused':
            g = np.load(os.path.join(golden_dir, 'gmm_n400_d3_k4.npz'))
            y, lab0 = g['y'], g['lab0']
            N, D = y.shape
            K = g['alpha_u0'].shape[-1]
            alpha = nodes.Dirichlet(1e-3 * np.ones(K), name='alpha')
            z = nodes.Categorical(alpha, plates=(N,), name='z')
            mu = nodes.GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
            Lam = nodes.Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
            Y = nodes.Mixture(z, nodes.Gaussian, mu, Lam, plates=(N,), name='Y')
            z.initialize_from_value(lab0)
            Y.observe(y)
            Q = VB(Y, mu, Lam, z, alpha)
            Q.ignore_bound_checks = True
            return Q, ['mu', 'alpha']
        g = np.load(os.path.join(golden_dir, 'small_models.npz'))
        y, x0 = g['mpca_y'], g['mpca_x0']
        Q = build_pca(nodes, VB, y, x0, x0.shape[1])
        if model == 'masked_pca_generic':
            Q['Y'].observe(y, mask=g['mpca_mask'])
        return Q, ['W', 'X', 'tau', 'alpha']

    Q, track = build()
    Q.update(repeat=2, verbose=False)
    fn = str(tmp_path / 'ckpt.bin')
    Q.save(filename=fn)
    Q.update(repeat=3, verbose=False)
    Q2, _ = build()
    Q2.load(filename=fn)
    assert Q2.iter == 2
    Q2.update(repeat=3, verbose=False)
    assert np.array_equal(Q2.L[:5], Q.L[:5])
    for nm in track:
        for a, b in zip(Q2[nm].u, Q[nm].u):
            np.testing.assert_array_equal(a, b)


def test_dirichlet_multinomial_matches_reference(golden_dir):
    """Multinomial (multinomial.py:62-319): observed count vectors with a different number of
    trials per plate, and the prior moments of a latent node."""
    from bayespy_amd.nodes import Dirichlet, Multinomial
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'multinomial.npz'))
    p = Dirichlet(np.array([1.0, 0.5, 2.0, 1.5, 1.0]), name='p')
    x = Multinomial(g['trials'], p, name='x')
    assert x.plates == (len(g['trials']),) and x.dims == ((5,),)
    x.observe(g['counts'])
    Q = VB(x, p)
    Q.update(repeat=2, verbose=False)
    np.testing.assert_allclose(Q.L[:2], g['L'], rtol=ELBO_RTOL)
    np.testing.assert_allclose(p.u[0], g['p_u0'], rtol=MOM_RTOL)
    np.testing.assert_allclose(Q.l[x][:2], g['L_x'], rtol=1e-9)
    np.testing.assert_allclose(Q.l[p][:2], g['L_p'], rtol=1e-9)
    p2 = Dirichlet(np.array([2.0, 1.0, 3.0]), name='p2')
    z = Multinomial(7, p2, plates=(4,), name='z')
    Q2 = VB(z, p2)
    Q2.ignore_bound_checks = True        # L = 0: the relative change is 0/0 like in the reference
    Q2.update(repeat=2, verbose=False)
    # moments may be broadcast-compressed over the plates on either side
    np.testing.assert_allclose(np.broadcast_to(z.u[0], (4, 3)), np.broadcast_to(g['z_u0'], (4, 3)),
                               rtol=MOM_RTOL)
    np.testing.assert_allclose(Q2.L[:2], g['L2'], atol=1e-9)
    bad = g['counts'].copy()
    bad[0, 0] += 1
    with pytest.raises(ValueError, match='sum to the number of trials'):
        x.observe(bad)
    with pytest.raises(ValueError, match='integer'):
        Multinomial(2.5, p2)


def test_summultiply_general_patterns_match_reference(golden_dir):
    """SumMultiply beyond the PCA pattern (dot.py:19-633): 'ij,j->i' with a matrix-valued
    GaussianARD parent, and a three-factor product with plates broadcast three ways."""
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'summultiply.npz'))

    def check(tag, Q, track):
        n = len(g[tag + '_L'])
        L = _trace(Q, n)
        np.testing.assert_allclose(L, g[tag + '_L'], rtol=ELBO_RTOL)
        for nm, nd in track.items():
            np.testing.assert_allclose(Q.l[nd][:n], g['%s_%s_L' % (tag, nm)], rtol=1e-8, atol=1e-7,
                                       err_msg=nm)
            for i, ui in enumerate(nd.u):
                ref = g['%s_%s_u%d' % (tag, nm, i)]
                np.testing.assert_allclose(np.broadcast_to(ui, ref.shape), ref, rtol=MOM_RTOL,
                                           atol=1e-10, err_msg='%s u%d' % (nm, i))

    y, x0 = g['mv_y'], g['mv_x0']
    N = y.shape[0]
    A = GaussianARD(0, 1e-2, shape=(2, 3), name='A')
    x = GaussianARD(0, 1, shape=(3,), plates=(N,), name='x')
    F = SumMultiply('ij,j->i', A, x, name='F')
    assert F.plates == (N,) and F.dims == ((2,), (2, 2))
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    x.initialize_from_value(x0)
    Y.observe(y)
    check('mv', VB(Y, F, A, x, tau), dict(A=A, x=x, tau=tau))

    y3, b0, c0 = g['pf_y'], g['pf_b0'], g['pf_c0']
    I, J, Kp = y3.shape
    C = b0.shape[-1]
    a = GaussianARD(0, 1e-1, shape=(C,), plates=(I, 1, 1), name='a')
    b = GaussianARD(0, 1e-1, shape=(C,), plates=(1, J, 1), name='b')
    c = GaussianARD(0, 1e-1, shape=(C,), plates=(1, 1, Kp), name='c')
    F3 = SumMultiply('i,i,i', a, b, c, name='F3')
    assert F3.plates == (I, J, Kp)
    tau3 = Gamma(1e-2, 1e-2, name='tau3')
    Y3 = GaussianARD(F3, tau3, name='Y3')
    b.initialize_from_value(b0)
    c.initialize_from_value(c0)
    Y3.observe(y3)
    check('pf', VB(Y3, F3, a, b, c, tau3), dict(a=a, b=b, c=c, tau3=tau3))


def test_mixture_of_gaussian_ard_matches_reference(golden_dir):
    """Mixture over GaussianARD components with a latent mean and a latent Gamma precision per
    cluster and dimension (mixture.py:359-545 over gaussian.py:1559-1774); the trailing
    positional ``1`` is the ``ndim`` the reference forwards to the mixed class."""
    from bayespy_amd.nodes import GaussianARD, Gamma, Dirichlet, Categorical, Mixture
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'mixture_ard.npz'))
    y, lab0 = g['y'], g['lab0']
    N, D = y.shape
    K = g['mu_u0'].shape[0]
    alpha = Dirichlet(np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-2, shape=(D,), plates=(K,), name='mu')
    lam = Gamma(1e-1, 1e-1, plates=(K, D), name='lam')
    Y = Mixture(z, GaussianARD, mu, lam, 1, name='Y')
    assert Y.plates == (N,) and Y.dims == ((D,), (D, D))
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, lam, z, alpha)
    n = len(g['L'])
    L = _trace(Q, n)
    np.testing.assert_allclose(L, g['L'], rtol=ELBO_RTOL)
    for nm, nd in dict(alpha=alpha, z=z, mu=mu, lam=lam).items():
        np.testing.assert_allclose(Q.l[nd][:n], g['%s_L' % nm], rtol=1e-8, atol=1e-7, err_msg=nm)
        for i, ui in enumerate(nd.u):
            ref = g['%s_u%d' % (nm, i)]
            np.testing.assert_allclose(np.broadcast_to(ui, ref.shape), ref, rtol=MOM_RTOL,
                                       atol=1e-10, err_msg='%s u%d' % (nm, i))
    np.testing.assert_allclose(Q.l[Y][:n], g['Y_L'], rtol=1e-8)
    # the keyword form builds the same node
    Y2 = Mixture(z, GaussianARD, mu, lam, ndim=1)
    assert Y2.plates == Y.plates and Y2.dims == Y.dims


def test_parameter_api_gradients_annealing_and_optimizers_match_reference(golden_dir):
    """tests/models.py run_parameter_api_cases on this framework against its run on the
    reference (tests/golden/parameter_api.npz): node.phi / get_parameters / set_parameters,
    Riemannian and Euclidean gradients of Gamma, Gaussian, GaussianARD, Dirichlet and
    Categorical nodes, logpdf, the unmasked lower bound, deterministic annealing, collapsed
    Riemannian conjugate gradients, plain gradient ascent and pattern search."""
    import bayespy_amd.nodes
    from bayespy_amd.inference import VB
    from models import run_parameter_api_cases
    f = np.load(os.path.join(golden_dir, 'parameter_api.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_parameter_api_cases(bayespy_amd.nodes, VB, g, engine='generic')
    for k, v in res.items():
        if k in ('cg_L', 'cg_W', 'cg_tau'):
            continue
        if isinstance(v, list):
            for i, vi in enumerate(v):
                ref = f['%s_%d' % (k, i)]
                np.testing.assert_allclose(np.broadcast_to(vi, ref.shape), ref, rtol=MOM_RTOL,
                                           atol=1e-9, err_msg='%s[%d]' % (k, i))
        else:
            np.testing.assert_allclose(v, f[k], rtol=1e-8, atol=1e-9, err_msg=k)
    # the optimisers take data-dependent branches (step halving, restarts) and a scalar
    # minimiser: the accepted iterates agree to round-off amplified by those searches
    assert len(res['cg_L']) == len(f['cg_L'])
    np.testing.assert_allclose(res['cg_L'], f['cg_L'], rtol=1e-6)
    np.testing.assert_allclose(res['cg_W'], f['cg_W'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res['cg_tau'], f['cg_tau'], rtol=1e-5)


def _compare_shared(res, f, skip=(), rtol=MOM_RTOL):
    for k, v in res.items():
        if k in skip:
            continue
        if isinstance(v, list):
            for i, vi in enumerate(v):
                ref = f['%s_%d' % (k, i)]
                np.testing.assert_allclose(np.broadcast_to(vi, ref.shape), ref, rtol=rtol,
                                           atol=1e-9, err_msg='%s[%d]' % (k, i))
        else:
            np.testing.assert_allclose(v, f[k], rtol=1e-8, atol=1e-8, err_msg=k)


def test_count_probability_and_add_nodes_match_reference(golden_dir):
    """Beta, Bernoulli, Binomial, Poisson, Complement, Add, and Bernoulli / Poisson mixtures
    (tests/models.py run_count_node_cases on both sides; the Bernoulli mixture is the model
    of doc/source/examples/bmm.rst)."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_count_node_cases
    f = np.load(os.path.join(golden_dir, 'count_nodes.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_count_node_cases(N_, VB, g)
    _compare_shared(res, f)
    # argument checks of the reference
    p = N_.Beta([1.0, 1.0])
    with pytest.raises(ValueError, match='integer'):
        N_.Binomial(2.5, p)
    with pytest.raises(ValueError, match='non-negative'):
        N_.Binomial(-1, p)
    with pytest.raises(ValueError, match='two-dimensional'):
        N_.Beta([1.0, 1.0, 1.0])
    x = N_.Binomial(3, p)
    with pytest.raises(ValueError, match='Invalid count'):
        x.observe(4)
    with pytest.raises(ValueError):
        N_.Poisson(2.0).observe(1.5)
    with pytest.raises(NotImplementedError, match='Gamma'):
        N_.Exponential(1.0)
    with pytest.raises(ValueError, match='at least two'):
        N_.Add(N_.GaussianARD(0, 1))
    with pytest.raises(ValueError, match='identical shapes'):
        N_.Add(N_.GaussianARD(0, 1, shape=(2,)), N_.GaussianARD(0, 1, shape=(3,)))


def test_take_concatenate_gate_match_reference(golden_dir):
    """Take (incl. the doctest of take.py:34-39, a two-axis index array on a non-last plate
    axis, masked data), Concatenate (last and second-last plate axis) and Gate (latent gate,
    fixed labels on a non-default plate axis) -- tests/models.py run_plate_node_cases."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_plate_node_cases
    f = np.load(os.path.join(golden_dir, 'plate_nodes.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_plate_node_cases(N_, VB, g)
    np.testing.assert_array_equal(res['tk_doc'], [2., 2., 3., 3., 2., 1.])
    _compare_shared(res, f)
    # argument checks of take.py:47-62, concatenate.py:27-31, gate.py:45-78
    a = N_.Gamma([1, 2, 3], [1, 1, 1])
    with pytest.raises(ValueError, match='negative'):
        N_.Take(a, [0], plate_axis=0)
    with pytest.raises(ValueError, match='out of bounds'):
        N_.Take(a, [0], plate_axis=-2)
    with pytest.raises(ValueError, match='Index out of bounds'):
        N_.Take(a, [3])
    with pytest.raises(ValueError, match='integers'):
        N_.Take(a, [0.5])
    with pytest.raises(ValueError, match='negative'):
        N_.Concatenate(a, a, axis=0)
    with pytest.raises(ValueError, match='dimensionalities'):
        N_.Concatenate(N_.GaussianARD(0, 1, plates=(2,)), N_.GaussianARD(0, 1, shape=(2,), plates=(2,)))
    with pytest.raises(ValueError, match='Inconsistent number of clusters'):
        N_.Gate(N_.Categorical([0.5, 0.5]), N_.GaussianARD(0, 1, plates=(3,)))


def test_take_and_put_kernels_are_exact_and_reproducible():
    """vmp_take_axis is a copy (bit-exact against np.take); vmp_segment_sum_axis adds the
    sources of a row in a fixed order: equal to a sequential NumPy accumulation bit for bit,
    and identical from run to run."""
    from bayespy_amd.darray import DArray
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(5)
    for shape, axis, ishape in (((7, 5), -1, (11,)), ((4, 6, 3), -2, (2, 5)), ((9,), -1, (3, 2, 2)),
                                ((3, 1000, 2), -2, (4000,))):
        x = rs.normal(size=shape)
        L = shape[axis]
        idx = rs.randint(-L, L, size=ishape)
        im = misc.IndexMap(idx, L)
        got = misc.take(DArray.from_host(x), im, axis=axis).numpy()
        np.testing.assert_array_equal(got, np.take(x, idx, axis=axis))
        y = rs.normal(size=got.shape)
        back = misc.put_simple(DArray.from_host(y), im, axis=axis).numpy()
        a = len(shape) + axis
        ref = np.zeros(shape)
        ym = np.moveaxis(y.reshape(shape[:a] + (-1,) + shape[a + 1:]), a, 0)
        rm = np.moveaxis(ref, a, 0)
        for j, l in enumerate(np.where(idx < 0, idx + L, idx).reshape(-1)):
            rm[l] += ym[j]
        np.testing.assert_array_equal(back, ref)
        np.testing.assert_array_equal(
            back, misc.put_simple(DArray.from_host(y), im, axis=axis).numpy())
    parts = [rs.normal(size=(2, 3, 4)), rs.normal(size=(1, 5, 4)), rs.normal(size=(2, 1, 1))]
    got = misc.concatenate([DArray.from_host(p) for p in parts], axis=-2).numpy()
    ref = np.concatenate([np.broadcast_to(p, (2, p.shape[1], 4)) for p in parts], axis=-2)
    np.testing.assert_array_equal(got, ref)


def test_alpha_beta_recursion_matches_reference(golden_dir):
    """vmp_alpha_beta_recursion against random.alpha_beta_recursion of the reference
    (utils/random.py:357-422): K = 3 ... 40 (all lane-group widths), broadcast plates, shared
    transition slices, an impossible state (-inf), a single transition, 300 chains."""
    from bayespy_amd.darray import DArray
    from bayespy_amd.utils import random as drandom
    f = np.load(os.path.join(golden_dir, 'markov_chains.npz'))
    for tag in ('ab_a', 'ab_b', 'ab_c', 'ab_d', 'ab_e', 'ab_f'):
        z0, zz, g = drandom.alpha_beta_recursion(DArray.from_host(f[tag + '_logp0']),
                                                 DArray.from_host(f[tag + '_logP']))
        np.testing.assert_allclose(g.numpy(), f[tag + '_g'], rtol=1e-11, atol=1e-11, err_msg=tag)
        np.testing.assert_allclose(z0.numpy(), f[tag + '_z0'], rtol=1e-9, atol=1e-14, err_msg=tag)
        np.testing.assert_allclose(zz.numpy(), f[tag + '_zz'], rtol=1e-9, atol=1e-14, err_msg=tag)
        np.testing.assert_allclose(zz.numpy().sum(axis=(-1, -2)), 1.0, rtol=1e-13)


def test_hidden_markov_models_match_reference(golden_dir):
    """CategoricalMarkovChain: the two models of doc/source/examples/hmm.rst with their
    doctest known answers, a batch of chains with time-varying transition priors and latent
    emission parameters, and a chain that gates Gaussian means."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_markov_chain_cases
    f = np.load(os.path.join(golden_dir, 'markov_chains.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_markov_chain_cases(N_, VB, g)
    assert '%e' % res['hmm1_L'][0] == '-1.095883e+02'              # hmm.rst:94
    assert '%e' % res['hmm2_L'][0] == '-9.963054e+02'              # hmm.rst:293
    assert len(res['hmm2_L']) == 8 and '%e' % res['hmm2_L'][7] == '-9.235053e+02'   # hmm.rst:295
    _compare_shared(res, f)
    Z = N_.CategoricalMarkovChain([0.5, 0.5], [[0.9, 0.1], [0.2, 0.8]], states=4)
    with pytest.raises(NotImplementedError):
        Z.observe([0, 1, 1, 0])
    with pytest.raises(ValueError, match='infer the length'):
        N_.CategoricalMarkovChain([0.5, 0.5], [[0.9, 0.1], [0.2, 0.8]])
    with pytest.raises(ValueError, match='different size'):
        N_.CategoricalMarkovChain([0.5, 0.3, 0.2], [[0.9, 0.1], [0.2, 0.8]], states=3)


def test_plate_indexing_and_choose_match_reference(golden_dir):
    """``X[1:3, ::2]``, ``X[..., 0]``, ``X[:, None, 1:4]``, ``t[-2]`` as parents of observed
    nodes (node.py:868-1130) and ``Choose`` with fixed labels (doctest of gate.py:215-225) and
    with a latent categorical (tests/models.py run_slice_cases on both sides)."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_slice_cases
    f = np.load(os.path.join(golden_dir, 'slice_nodes.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_slice_cases(N_, VB, g)
    np.testing.assert_array_equal(res['ch_doc'], [0., 0., 20., 10.])
    _compare_shared(res, f)
    X = N_.GaussianARD(0, 1, plates=(4, 5))
    with pytest.raises(IndexError, match='Too many indices'):
        X[0, 0, 0]
    with pytest.raises(IndexError, match='out of range'):
        X[4]
    with pytest.raises(IndexError, match='empty'):
        X[2:2]
    with pytest.raises(TypeError):
        X[[0, 1]]
    # a reversed slice (gathered on the device) against NumPy indexing of the moments
    r = X[::-1, 1::2]
    Q = VB(r, X)
    u = np.asarray(X.get_moments()[0])
    np.testing.assert_array_equal(np.broadcast_to(r.get_moments()[0], (4, 2)),
                                  np.broadcast_to(u, (4, 5))[::-1, 1::2])


def test_alpha_beta_recursion_at_scale_properties_and_oracle():
    """2e4 chains x 500 transitions x 8 states (4 GB of transition slices): a random subset of
    chains against the NumPy oracle (oracle/hmm.py, pinned to the reference), and for ALL
    chains the size-independent properties -- every pairwise marginal sums to one, consecutive
    slices agree on the marginal of the instance they share, q(z_0) is the row sum of the first
    slice, the log-normalisers are finite, and two runs are bit-identical."""
    import torch
    from bayespy_amd.darray import DArray
    from bayespy_amd.device import get_runtime
    from bayespy_amd.utils import random as drandom
    from oracle import hmm
    rt = get_runtime()
    B, N, K = 20000, 500, 8
    gen = torch.Generator(device=rt.device).manual_seed(11)
    logp0 = torch.randn(B, K, dtype=torch.float64, device=rt.device, generator=gen)
    logP = 2.0 * torch.randn(B, N, K, K, dtype=torch.float64, device=rt.device, generator=gen)
    z0, zz, g = drandom.alpha_beta_recursion(DArray(logp0), DArray(logP))
    tot = zz.t.sum(dim=(-1, -2))
    assert float((tot - 1.0).abs().max()) < 1e-12
    out_n = zz.t[:, :-1].sum(dim=-2)             # q(z_{n+1}) from slice n
    in_n = zz.t[:, 1:].sum(dim=-1)               # q(z_{n+1}) from slice n+1
    assert float((out_n - in_n).abs().max()) < 1e-10
    assert float((z0.t - zz.t[:, 0].sum(dim=-1)).abs().max()) < 1e-12
    assert bool(torch.isfinite(g.t).all())
    idx = np.random.RandomState(0).choice(B, size=40, replace=False)
    ti = torch.from_numpy(idx).to(rt.device)
    r0, rzz, rg = hmm.alpha_beta_recursion(logp0[ti].cpu().numpy(), logP[ti].cpu().numpy())
    np.testing.assert_allclose(g.t[ti].cpu().numpy(), rg, rtol=1e-12)
    np.testing.assert_allclose(z0.t[ti].cpu().numpy(), r0, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(zz.t[ti].cpu().numpy(), rzz, rtol=1e-8, atol=1e-15)
    # bit-reproducible from run to run
    z0b, zzb, gb = drandom.alpha_beta_recursion(DArray(logp0), DArray(logP))
    assert torch.equal(zz.t, zzb.t) and torch.equal(g.t, gb.t)


def test_concat_gaussian_matches_reference(golden_dir):
    """ConcatGaussian (concat_gaussian.py:15-116): block moments and the messages with the
    cross terms through the means of the other blocks."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_concat_gaussian_case
    f = np.load(os.path.join(golden_dir, 'concat_gaussian.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    _compare_shared(run_concat_gaussian_case(N_, VB, g), f)
    with pytest.raises(ValueError, match='vectors'):
        N_.ConcatGaussian(N_.GaussianARD(0, 1), N_.GaussianARD(0, 1, shape=(2,)))


def test_nested_mixtures_known_answers():
    """Nested mixtures with fixed and latent selectors: the Dirichlet counts pinned by the
    reference's own test (nodes/tests/test_mixture.py:215-256), the equivalence of a nested
    Mixture and nested Gates (:258-275), and the broadcast-g regression case (:208-212)."""
    from bayespy_amd.nodes import Categorical, Dirichlet, Mixture, Gate
    t1 = [1, 1, 0, 3, 3]
    t2 = [2]
    p = Dirichlet([1, 1], plates=(4, 3))
    X = Mixture(t1, Mixture, t2, Categorical, p)
    assert X.plates == (5,)
    X.observe([1, 1, 0, 0, 0])
    p.update()
    np.testing.assert_allclose(
        np.broadcast_to(p.phi[0], (4, 3, 2)),
        [[[1, 1], [1, 1], [2, 1]],
         [[1, 1], [1, 1], [1, 3]],
         [[1, 1], [1, 1], [1, 1]],
         [[1, 1], [1, 1], [3, 1]]], rtol=1e-12)
    # sample plates in nested mixtures
    t1 = Categorical([0.3, 0.7], plates=(5,))
    t2 = [[1], [1], [0], [3], [3]]
    t3 = 2
    p = Dirichlet([1, 1], plates=(2, 4, 3))
    X = Mixture(t1, Mixture, t2, Mixture, t3, Categorical, p)
    X.observe([1, 1, 0, 0, 0])
    p.update()
    np.testing.assert_allclose(
        np.broadcast_to(p.phi[0], (2, 4, 3, 2)),
        [[[[1, 1], [1, 1], [1.3, 1]],
          [[1, 1], [1, 1], [1, 1.6]],
          [[1, 1], [1, 1], [1, 1]],
          [[1, 1], [1, 1], [1.6, 1]]],
         [[[1, 1], [1, 1], [1.7, 1]],
          [[1, 1], [1, 1], [1, 2.4]],
          [[1, 1], [1, 1], [1, 1]],
          [[1, 1], [1, 1], [2.4, 1]]]], rtol=1e-12)

    # Gate and nested Mixture are equal
    def build(nested):
        a = Categorical([0.3, 0.7], plates=(5,))
        b = Categorical([0.1, 0.3, 0.6], plates=(5, 1))
        q = Dirichlet([1, 2, 3, 4], plates=(2, 3))
        Y = Mixture(a, Mixture, b, Categorical, q) if nested else Categorical(Gate(a, Gate(b, q)))
        Y.observe([3, 3, 1, 2, 2])
        for nd in (a, b, q):
            nd.update()
        return [np.asarray(nd.phi[0]) for nd in (a, b, q)]
    for x, y in zip(build(True), build(False)):
        np.testing.assert_allclose(x, y, rtol=1e-12)
    # MultiMixture = the nesting with trailing unit axes on the selectors (mixture.py:547-566)
    from bayespy_amd.nodes import MultiMixture
    p = Dirichlet([1, 1], plates=(4, 3))
    X = MultiMixture([[1, 1, 0, 3, 3], [2]], Categorical, p)
    assert X.plates == (5,)
    X.observe([1, 1, 0, 0, 0])
    p.update()
    np.testing.assert_allclose(np.broadcast_to(p.phi[0], (4, 3, 2))[:, 2],
                               [[2, 1], [1, 3], [1, 1], [3, 1]], rtol=1e-12)
    # the mixed distribution broadcasts g
    Z = Categorical([0.3, 0.5, 0.2])
    X = Mixture(Z, Categorical, [[0.2, 0.8], [0.1, 0.9], [0.3, 0.7]])
    Z.update()
    assert np.isfinite(X.lower_bound_contribution())
    np.testing.assert_allclose(np.sum(Z.u[0]), 1.0, rtol=1e-12)


def test_latent_mixture_moments_and_messages_known_answers():
    """A latent Mixture node (moments from its parents, a child above it): the known answers of
    the reference's nodes/tests/test_mixture.py:60-160 -- moments 2 and 2^2+1, and the messages
    to the selector and to the cluster means (incl. a precision without a cluster axis)."""
    from bayespy_amd.nodes import GaussianARD, Gamma, Categorical, Mixture
    K = 3
    for mean in ([0, 2, 4], 2):
        mu = GaussianARD(mean, 1, ndim=0, plates=(K,))
        alpha = Gamma(1, 1, plates=(K,))
        z = Categorical(np.ones(K) / K)
        X = Mixture(z, GaussianARD, mu, alpha)
        assert X.plates == () and X.dims == ((), ())
        u = X.get_moments()
        np.testing.assert_allclose(u[0], 2, rtol=1e-12)
        np.testing.assert_allclose(u[1], 2 ** 2 + 1, rtol=1e-12)
    for aplates in ((K,), ()):
        Mu = GaussianARD(2, 1, ndim=0, plates=(K,))
        Alpha = Gamma(3, 1, plates=aplates)
        z = Categorical(np.ones(K) / K)
        X = Mixture(z, GaussianARD, Mu, Alpha)
        Y = GaussianARD(X, 4)
        Y.observe(5)
        m, mm = [np.asarray(v) for v in Mu.get_moments()]
        a, loga = [np.asarray(v) for v in Alpha.get_moments()]
        x, xx = [float(np.asarray(v)) for v in X.get_moments()]
        # message to z: E[log N(x | mu_k, alpha_k^-1)]  (random.gaussian_logpdf, :131-137)
        # with D = 0: the mixture passes f = 0, no log(2 pi) term (mixture.py:92-98)
        logp = -0.5 * xx * a + x * a * m - 0.5 * mm * a + 0.5 * loga
        prior = np.log(np.ones(K) / K)
        z.update()
        np.testing.assert_allclose(np.broadcast_to(z.phi[0], (K,)),
                                   np.broadcast_to(prior + logp, (K,)), rtol=1e-12)
        # message to Mu: [1/K alpha x, -1/2 1/K alpha] on top of the prior phi = [2, -1/2]
        zk = np.asarray(z.get_moments()[0])
        Mu.update()
        np.testing.assert_allclose(np.broadcast_to(Mu.phi[0], (K,)), 2 + zk * a * x, rtol=1e-12)
        np.testing.assert_allclose(np.broadcast_to(Mu.phi[1], (K,)), -0.5 - 0.5 * zk * a,
                                   rtol=1e-12)


def test_default_ndim_under_vector_means_matches_reference(golden_dir):
    """GaussianARD(mu, alpha) with a vector-valued Gaussian mean and no ndim / shape: scalar
    node, the mean's variable axis turns into a plate (the reference's default,
    gaussian.py:1617-1640); and ndim=1 under a matrix-valued mean."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_default_ndim_case
    f = np.load(os.path.join(golden_dir, 'default_ndim.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_default_ndim_case(N_, VB, g)
    assert list(res['dn_X_plates']) == [4, 3] and list(res['dn_X_ndims']) == [0, 0]
    assert list(res['dn_Z_plates']) == [4, 2] and list(res['dn_Z_shape']) == [3]
    _compare_shared(res, f)


def test_mixture_over_a_non_last_cluster_plate():
    """``cluster_plate=-2`` / ``-3``.  The reference only supports this while the parameter
    moments stay constant along the cluster axis (its compute_logpdf mixes the axis orders
    otherwise), so the checks are: its own known answers (nodes/tests/test_mixture.py:163-200
    messages, :291-318 masks), and for asymmetric parameters the equality with the same model
    written with the cluster axis last."""
    from bayespy_amd.nodes import GaussianARD, Gamma, Categorical, Dirichlet, Mixture
    from bayespy_amd.inference import VB
    K, M = 3, 2
    Mu = GaussianARD(2, 1, ndim=0, plates=(K, M))
    Alpha = Gamma(3, 1, plates=(K, M))
    z = Categorical(np.ones(K) / K)
    X = Mixture(z, GaussianARD, Mu, Alpha, cluster_plate=-2)
    assert X.plates == (M,)
    Y = GaussianARD(X, 4)
    Y.observe(5 * np.ones(M))
    m, mm = [np.broadcast_to(v, (K, M)) for v in Mu.get_moments()]
    a, loga = [np.broadcast_to(v, (K, M)) for v in Alpha.get_moments()]
    x, xx = [np.broadcast_to(v, (M,)) for v in X.get_moments()]
    logp = -0.5 * xx * a + x * a * m - 0.5 * mm * a + 0.5 * loga
    z.update()
    np.testing.assert_allclose(np.broadcast_to(z.phi[0], (K,)),
                               np.log(np.ones(K) / K) + logp.sum(-1), rtol=1e-12)
    zk = np.asarray(z.get_moments()[0])
    Mu.update()
    np.testing.assert_allclose(np.broadcast_to(Mu.phi[0], (K, M)), 2 + zk[:, None] * a * x, rtol=1e-12)
    np.testing.assert_allclose(np.broadcast_to(Mu.phi[1], (K, M)), -0.5 - 0.5 * zk[:, None] * a,
                               rtol=1e-12)
    # masks (test_mixture.py:291-318)
    Z = Categorical(np.ones(K) / K, plates=(4, 5, 1))
    Mu3 = GaussianARD(0, 1, shape=(2,), plates=(4, K, 5))
    Alpha3 = Gamma(1, 1, plates=(4, K, 5, 2))
    X3 = Mixture(Z, GaussianARD, Mu3, Alpha3, cluster_plate=-3)
    assert X3.plates == (4, 5, 2)
    Y3 = GaussianARD(X3, 1, ndim=1)
    mask = np.reshape((np.mod(np.arange(4 * 5), 2) == 0), (4, 5))
    Y3.observe(np.ones((4, 5, 2)), mask=mask)
    np.testing.assert_array_equal(np.broadcast_to(Z.mask, (4, 5, 1)), mask[:, :, None])
    np.testing.assert_array_equal(np.broadcast_to(Mu3.mask, (4, 1, 5)), mask[:, None, :])
    np.testing.assert_array_equal(np.broadcast_to(Alpha3.mask, (4, 1, 5, 1)), mask[:, None, :, None])

    # asymmetric parameters: cluster axis second-last vs. last
    rs = np.random.RandomState(3)
    y = rs.normal(size=(6, M)) * 2 + np.array([0.0, 3.0])
    lab0 = np.array([[0], [1], [2], [0], [1], [2]])

    def run(cp):
        shape = (K, M) if cp == -2 else (M, K)
        pi = Dirichlet(np.ones(K))
        zz = Categorical(pi, plates=(6, 1))
        mu = GaussianARD(0, 1e-1, ndim=0, plates=shape)
        al = Gamma(1, 1, plates=shape)
        Xc = Mixture(zz, GaussianARD, mu, al, cluster_plate=cp)
        assert Xc.plates == (6, M)
        zz.initialize_from_value(lab0)
        Xc.observe(y)
        Q = VB(Xc, mu, al, zz, pi)
        Q.ignore_bound_checks = True
        Q.update(repeat=4, verbose=False)
        t = (lambda v: np.swapaxes(np.broadcast_to(v, (K, M)), 0, 1)) if cp == -2 else \
            (lambda v: np.broadcast_to(v, (M, K)))
        return Q.L[:4], t(mu.get_moments()[0]), t(al.get_moments()[0]), \
            np.asarray(zz.get_moments()[0])
    r2, r1 = run(-2), run(-1)
    for p, q in zip(r2, r1):
        np.testing.assert_allclose(p, q, rtol=1e-10)
    with pytest.raises(ValueError, match='negative'):
        Mixture(z, GaussianARD, Mu, Alpha, cluster_plate=0)


def test_reference_bernoulli_mixture_doctest_known_answer(golden_dir):
    """doc/source/examples/bmm.rst: "Iteration 1: loglike=-6.872145e+02 ... Iteration 17:
    loglike=-5.236921e+02", converged at iteration 17 (the data and the random initial value of
    P drawn by the reference under its testsetup seed are in tests/golden/bmm_doctest.npz)."""
    from bayespy_amd.nodes import Categorical, Dirichlet, Beta, Mixture, Bernoulli
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'bmm_doctest.npz'))
    N, D, K = 100, 10, 10
    R = Dirichlet(K * [1e-5], name='R')
    Z = Categorical(R, plates=(N, 1), name='Z')
    P = Beta([0.5, 0.5], plates=(D, K), name='P')
    X = Mixture(Z, Bernoulli, P)
    Q = VB(Z, R, X, P)
    P.initialize_from_value(g['p_init'])
    X.observe(g['x'])
    Q.update(repeat=1000, verbose=False)
    L = Q.L[:Q.iter]
    assert '%e' % L[0] == '-6.872145e+02'
    assert Q.iter == 17 and '%e' % L[-1] == '-5.236921e+02'
    np.testing.assert_allclose(L, g['L'], rtol=1e-9)
    np.testing.assert_allclose(R.u[0], g['R_u0'], rtol=1e-6)
    np.testing.assert_allclose(P.u[0], g['P_u0'], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('engine', [None, 'generic'])
def test_reference_gaussian_mixture_doctest_known_answer(golden_dir, engine):
    """doc/source/examples/gmm.rst: "Iteration 1: loglike=-1.402345e+03 ... Iteration 61:
    loglike=-8.888464e+02" (data and random initial labels of the reference's seeded run in
    tests/golden/gmm_doctest.npz), on the fused mixture block and on the generic engine."""
    from bayespy_amd.nodes import Dirichlet, Categorical, Gaussian, Wishart, Mixture
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'gmm_doctest.npz'))
    y = g['y']
    N, D, K = 200, 2, 10
    alpha = Dirichlet(1e-5 * np.ones(K), name='alpha')
    Z = Categorical(alpha, plates=(N,), name='z')
    mu = Gaussian(np.zeros(D), 1e-5 * np.identity(D), plates=(K,), name='mu')
    Lambda = Wishart(D, 1e-5 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(Z, Gaussian, mu, Lambda, name='Y')
    Z.initialize_from_value(g['z_init'])
    Q = VB(Y, mu, Lambda, Z, alpha, engine=engine)
    Y.observe(y)
    Q.update(repeat=1000, verbose=False)
    L = Q.L[:Q.iter]
    assert '%e' % L[0] == '-1.402345e+03'
    assert Q.iter == 61 and '%e' % L[-1] == '-8.888464e+02'
    np.testing.assert_allclose(L, g['L'], rtol=1e-8)
    np.testing.assert_allclose(alpha.u[0], g['alpha_u0'], rtol=1e-5, atol=1e-6)


def test_reference_user_guide_inference_doctest(golden_dir):
    """doc/source/user_guide/inference.rst:52-235: the PCA model (Dot node) observed with whole
    rows masked, X initialised from parameters, ``Q.update()``, ``Q.update(C, X)``,
    ``Q.update(C, X, C, tau)``, ``repeat=10``, then convergence at the default and at a tighter
    tolerance.  The doctest prints iterations 1-14 and "Converged at iteration 488." / "847."."""
    from bayespy_amd.nodes import GaussianARD, Gamma, Dot
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(golden_dir, 'inference_doctest.npz'))
    D = 3
    X = GaussianARD(0, 1, shape=(D,), plates=(1, 100), name='X')
    alpha = Gamma(1e-3, 1e-3, plates=(D,), name='alpha')
    C = GaussianARD(0, alpha, shape=(D,), plates=(10, 1), name='C')
    F = Dot(C, X)
    tau = Gamma(1e-3, 1e-3, name='tau')
    Y = GaussianARD(F, tau)
    Y.observe(g['data'])
    Y.observe(g['data'], mask=g['mask'])
    Q = VB(Y, C, X, alpha, tau)
    assert Q['X'] is X
    X.initialize_from_parameters(g['x_init'], 10)
    Q.update(verbose=False)
    Q.update(C, X, verbose=False)
    Q.update(C, X, C, tau, verbose=False)
    Q.update(repeat=10, verbose=False)
    want = ['-9.305259e+02', '-8.818976e+02', '-8.071222e+02', '-7.167588e+02', '-6.827873e+02',
            '-6.259477e+02', '-4.725400e+02', '-3.270816e+02', '-2.208865e+02', '-1.658761e+02',
            '-1.469468e+02', '-1.420311e+02', '-1.405139e+02']
    assert ['%e' % v for v in Q.L[:13]] == want
    Q.update(repeat=1000, verbose=False)
    n1 = Q.iter
    assert '%e' % Q.L[13] == '-1.396481e+02'
    # hundreds of iterations on a slowly rising bound: the stopping iteration may move by a
    # few steps with round-off, the value it stops at may not
    assert Q.converged and abs(n1 - 488) <= 3
    np.testing.assert_allclose(Q.L[n1 - 1], -1.224106e+02, rtol=2e-6)
    Q.update(repeat=10000, tol=1e-6, verbose=False)
    assert Q.converged and abs(Q.iter - 847) <= 5
    np.testing.assert_allclose(Q.L[Q.iter - 1], -1.222506e+02, rtol=2e-6)
    m = min(Q.iter, len(g['L']))
    np.testing.assert_allclose(Q.L[:m], g['L'][:m], rtol=1e-7)
    np.testing.assert_allclose(tau.u[0], g['tau_u0'], rtol=1e-4)


def test_add_node_doctest_known_answer():
    """The doctest of add.py:19-33: mean [[1, 1]] and second moment [[[3, 1], [1, 3]]]."""
    from bayespy_amd import nodes
    X = nodes.Gaussian(np.zeros(2), np.identity(2), plates=(3,))
    Y = nodes.Gaussian(np.ones(2), np.identity(2))
    Z = nodes.Add(X, Y)
    u = Z.get_moments()
    np.testing.assert_allclose(np.broadcast_to(u[0], (3, 2)), np.ones((3, 2)), rtol=1e-13)
    np.testing.assert_allclose(np.broadcast_to(u[1], (3, 2, 2)),
                               np.broadcast_to([[3.0, 1.0], [1.0, 3.0]], (3, 2, 2)), rtol=1e-13)


def test_numeric_parents_of_summultiply_match_reference(golden_dir):
    """SumMultiply / Dot with numeric arrays among the parents (the reference wraps them in
    constants with delta moments, dot.py:186-197): Bayesian linear regression with known
    regressors, and a constant inside a three-factor product."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_constant_parent_cases
    g = np.load(os.path.join(golden_dir, 'constant_parents.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    res = run_constant_parent_cases(nodes, VB, inp)
    for k, v in res.items():
        tol = dict(rtol=ELBO_RTOL) if k.endswith('_L') else dict(rtol=MOM_RTOL, atol=1e-10)
        np.testing.assert_allclose(v, g[k], err_msg=k, **tol)


def test_gaussian_gamma_nodes_match_reference(golden_dir):
    """GaussianGammaMoments / the GaussianGamma joint node / WrapToGaussianGamma /
    GaussianToGaussianGamma (gaussian.py:161-229, :892-1136, :1777-1840, :2226-2371).  Live-reference
    traces of tests/models.py:run_gaussian_gamma_cases: (a) the conjugate normal-gamma model
    GaussianARD(GaussianGamma(..., ndim=0), 1) -- exact after one update --, (b) the joint node under
    a latent Gaussian mean and Gamma rate, with a Gamma scale on the child and an array mask, (c) a
    vector-valued joint node under Gaussian / Wishart / Gamma parents."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_gaussian_gamma_cases
    f = np.load(os.path.join(golden_dir, 'gaussian_gamma.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    res = run_gaussian_gamma_cases(N_, VB, g)
    assert list(res['a_plates']) == [3, 1] and list(res['a_ndims']) == [0, 0, 0, 0]
    assert list(res['c_dims']) == [1, 2, 0, 0]
    _compare_shared(res, f)
    assert abs(res['a_L'][1] - res['a_L'][0]) < 1e-9 * abs(res['a_L'][0])


def test_explicit_gaussian_gamma_converter_and_wrapper_nodes():
    """GaussianToGaussianGamma(X) (gaussian.py:2226-2276) and WrapToGaussianGamma(X, alpha)
    (:2299-2371) as nodes of their own give what the folded forms give: GaussianARD(X, s) ==
    GaussianARD(GaussianToGaussianGamma(X), s) == GaussianARD(WrapToGaussianGamma(X, s), 1)."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    rs = np.random.RandomState(3)
    y = rs.normal(size=(4, 30)) + np.arange(4)[:, None]

    def run(kind):
        X = N_.GaussianARD(0, 1e-2, plates=(4, 1), name='X')
        s = N_.Gamma(1e-2, 1e-2, plates=(1, 30), name='s')
        if kind == 'folded':
            Y = N_.GaussianARD(X, s, name='Y')
        elif kind == 'converter':
            Y = N_.GaussianARD(N_.GaussianToGaussianGamma(X), s, name='Y')
        else:
            Y = N_.GaussianARD(N_.WrapToGaussianGamma(X, s), 1, name='Y')
        Y.observe(y)
        Q = VB(Y, X, s, engine='generic')
        Q.ignore_bound_checks = True
        Q.update(repeat=4, verbose=False)
        return np.array(Q.L[:4]), X.u, s.u
    L0, xu, su = run('folded')
    for kind in ('converter', 'wrapper'):
        L1, xu1, su1 = run(kind)
        np.testing.assert_allclose(L1, L0, rtol=1e-12, err_msg=kind)
        np.testing.assert_allclose(xu1[0], xu[0], rtol=1e-11)
        np.testing.assert_allclose(su1[0], su[0], rtol=1e-11)
    G = N_.GaussianToGaussianGamma(N_.GaussianARD(0, 1, shape=(3,), plates=(2,)))
    assert G.dims == ((3,), (3, 3), (), ()) and G.plates == (2,)
    with pytest.raises(ValueError, match='should be Gaussian'):
        N_.GaussianToGaussianGamma(N_.Gamma(1, 1))


@pytest.mark.parametrize('name,K', [('pca_n500_d6_k3', 3), ('pca_n777_d20_k5', 5),
                                    ('pca_n4000_d64_k16', 16)])
def test_generic_pca_with_factored_second_moments(golden_dir, name, K, monkeypatch):
    """A posterior covariance shared over a plate (scalar mask) is kept as (Cov, <x>) instead of
    a plates x K x K array (plans/generic.py:FactoredMoment): same live-reference traces, and the
    dense (N, K, K) array of X is never formed by an iteration; reading X.u[1] forms it."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.generic import GenericPlan, FactoredMoment
    from models import build_pca
    monkeypatch.setenv('BAYESPY_AMD_FACTORED_MIN_PLATES', '2')
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    Q = build_pca(nodes, VB, g['y'], g['x0'], K, engine='generic')
    plan = Q.plans[0]
    assert isinstance(plan, GenericPlan)
    n = int(g['n_iter'])
    np.testing.assert_allclose(_trace(Q, n), g['L'], rtol=ELBO_RTOL)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:n], g['L_' + k], rtol=1e-8, atol=1e-7, err_msg=k)
    for nm in ('X', 'W'):
        u1 = plan.state[id(Q[nm])].u[1]
        assert isinstance(u1, FactoredMoment) and u1._dense is None, nm
    np.testing.assert_allclose(Q['W'].u[0], g['W_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Q['X'].u[0], g['X_u0'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(np.broadcast_to(Q['W'].u[1], g['W_u1'].shape), g['W_u1'],
                               rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Q['X'].u[1][0, :3], g['X_u1_first'], rtol=MOM_RTOL, atol=1e-10)
    np.testing.assert_allclose(Q['tau'].u[0], g['tau_u0'], rtol=1e-9)
    np.testing.assert_allclose(Q['alpha'].u[1], g['alpha_u1'], rtol=1e-8)


def test_factored_second_moments_keep_memory_flat():
    """N = 2e5, K = 16: the (N, K, K) array would be 410 MB; an iteration of the generic engine
    stays far below it, and equals the dense path (threshold raised) to rounding."""
    import torch
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_pca
    from oracle.pca import make_pca_data
    N, D, K = 200_000, 32, 16
    y, x0 = make_pca_data(N, D, K, seed=11)
    out = {}
    for mode, thr in (('factored', '2'), ('dense', str(10 ** 9))):
        os.environ['BAYESPY_AMD_FACTORED_MIN_PLATES'] = thr
        try:
            Q = build_pca(nodes, VB, y, x0, K, engine='generic')
            Q.update(repeat=1, verbose=False)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            Q.update(repeat=2, verbose=False)
            torch.cuda.synchronize()
            out[mode] = (np.array(Q.L[:3]), torch.cuda.max_memory_allocated() - base)
            del Q
        finally:
            os.environ.pop('BAYESPY_AMD_FACTORED_MIN_PLATES', None)
    np.testing.assert_allclose(out['factored'][0], out['dense'][0], rtol=1e-12)
    assert out['factored'][1] < 0.6 * 8 * N * K * K, out['factored'][1]
    assert out['dense'][1] > 8 * N * K * K


def test_hierarchical_wishart_matches_reference(golden_dir):
    """Wishart(n, V) with V a Wishart node (wishart.py:142-150: the message [-<Lambda>/2, n/2] to
    the inverse scale matrix): bound trace, moments of both nodes, per-node bound terms against the
    live-reference golden (tests/models.py run_hierarchical_wishart_case on both sides)."""
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB
    from models import run_hierarchical_wishart_case
    f = np.load(os.path.join(golden_dir, 'hierarchical_wishart.npz'))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    _compare_shared(run_hierarchical_wishart_case(N_, VB, g), f)
    with pytest.raises(NotImplementedError, match='degrees of freedom'):
        N_.Wishart(N_.Gamma(1.0, 1.0), np.eye(2))
