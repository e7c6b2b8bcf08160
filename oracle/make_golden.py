#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ from the LIVE reference.

TEST INFRASTRUCTURE ONLY.  Runs only in the authoring container, where the
reference is mounted read-only at /root/reference; the GPU box never runs this.

    python oracle/make_golden.py            # writes tests/golden/*.npz

The reference is imported unmodified, with two empty stub modules for its
absent optional dependencies ``h5py`` (save/load only) and ``truncnorm``
(observe_limits only) -- SURVEY.md section 8(c).

Every fixture stores the exact inputs (data + injected initial moments) and the
reference's outputs after each VB iteration, so both the numpy oracle
(oracle/*.py) and the HIP path can be replayed on identical inputs.  RNG
streams are not part of the contract: all random draws are made HERE with a
seeded RandomState and stored.
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
REF = os.environ.get('BAYESPY_REFERENCE', '/root/reference')


def _import_reference():
    stubs = tempfile.mkdtemp(prefix='bpstubs_')
    for name in ('h5py', 'truncnorm'):
        os.makedirs(os.path.join(stubs, name))
        open(os.path.join(stubs, name, '__init__.py'), 'w').close()
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    warnings.simplefilter('ignore')
    import bayespy  # noqa: F401
    return bayespy


def pca_case(name, N, D, K, n_iter, seed):
    """bayespy/demos/pca.py:22-61, fully observed, X initialised from value."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    rs = np.random.RandomState(seed)
    w = rs.normal(0, 1, (D, K))
    x = rs.normal(0, 1, (N, K))
    y = w @ x.T + 0.1 * rs.normal(size=(D, N))
    x0 = rs.normal(0, 1, (N, K))

    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None, :, :])
    Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    Ls = []
    terms = {k: [] for k in ('Y', 'X', 'W', 'tau', 'alpha')}
    for _ in range(n_iter):
        Q.update(repeat=1, verbose=False)
        Ls.append(Q.L[Q.iter - 1])
        for k in terms:
            terms[k].append(Q.l[Q[k]][Q.iter - 1])
    out = dict(
        y=y, x0=x0, n_iter=n_iter, a0=1e-2, b0=1e-2,
        L=np.array(Ls),
        **{'L_' + k: np.array(v) for k, v in terms.items()},
        W_u0=W.u[0], W_u1=W.u[1], W_phi0=W.phi[0], W_phi1=W.phi[1], W_g=W.g,
        X_u0=X.u[0], X_u1_first=X.u[1][0, :3], X_phi1=X.phi[1], X_g=X.g,
        tau_u0=tau.u[0], tau_u1=tau.u[1], tau_phi0=tau.phi[0], tau_phi1=tau.phi[1],
        alpha_u0=alpha.u[0], alpha_u1=alpha.u[1],
        alpha_phi0=alpha.phi[0], alpha_phi1=alpha.phi[1],
        F_u0=F.get_moments()[0][:, :5], F_u1=F.get_moments()[1][:, :5],
    )
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'L =', Ls)


def quickstart_case(name, N, n_iter, seed):
    """
    Config 1 / doc/source/user_guide/quickstart.rst:8-13,41,111-118: unknown
    mean and precision of a 1-D Gaussian.  With seed(1), N=10 the doc's known
    answer is -6.020956e+01, -5.820527e+01, -5.820290e+01, -5.820288e+01.
    """
    from bayespy.nodes import GaussianARD, Gamma
    from bayespy.inference import VB
    np.random.seed(seed)
    data = np.random.normal(5, 10, size=(N,))
    mu = GaussianARD(0, 1e-6, name='mu')
    tau = Gamma(1e-6, 1e-6, name='tau')
    y = GaussianARD(mu, tau, plates=(N,), name='y')
    y.observe(data)
    Q = VB(y, mu, tau)
    Q.ignore_bound_checks = True
    Ls = []
    mu_u, tau_u = [], []
    for _ in range(n_iter):
        Q.update(repeat=1, verbose=False)
        Ls.append(Q.L[Q.iter - 1])
        mu_u.append([float(mu.u[0]), float(mu.u[1])])
        tau_u.append([float(tau.u[0]), float(tau.u[1])])
    np.savez_compressed(os.path.join(OUT, name + '.npz'),
                        data=data, L=np.array(Ls), mu_u=np.array(mu_u),
                        tau_u=np.array(tau_u), n_iter=n_iter)
    print(name, 'L =', Ls)


def gmm_case(name, N, D, K, n_iter, seed):
    """bayespy/demos/mog.py:17-64 (full covariance), z initialised from value."""
    from bayespy.nodes import (GaussianARD, Gaussian, Wishart, Dirichlet,
                               Categorical, Mixture)
    from bayespy.inference import VB
    rs = np.random.RandomState(seed)
    centers = 3 * rs.normal(size=(K, D))
    lab = rs.randint(K, size=N)
    y = centers[lab] + 0.5 * rs.normal(size=(N, D))
    lab0 = rs.randint(K, size=N)

    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha)
    Q.ignore_bound_checks = True
    Ls = []
    terms = {k: [] for k in ('Y', 'mu', 'Lambda', 'z', 'alpha')}
    for _ in range(n_iter):
        Q.update(repeat=1, verbose=False)
        Ls.append(Q.L[Q.iter - 1])
        for k in terms:
            terms[k].append(Q.l[Q[k]][Q.iter - 1])
    out = dict(
        y=y, lab0=lab0, n_iter=n_iter,
        L=np.array(Ls),
        **{'L_' + k: np.array(v) for k, v in terms.items()},
        z_u0=z.u[0], z_phi0=z.phi[0], z_g=z.g,
        mu_u0=mu.u[0], mu_u1=mu.u[1],
        Lambda_u0=Lam.u[0], Lambda_u1=Lam.u[1],
        Lambda_phi0=Lam.phi[0], Lambda_phi1=Lam.phi[1],
        alpha_u0=alpha.u[0], alpha_phi0=alpha.phi[0],
    )
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'L =', Ls)


def utils_cases(name):
    """Known answers of the reference's own utility functions on seeded inputs:
    misc.sum_multiply_to_plates / sum_multiply / logsumexp / normalized_exp /
    multidigamma, linalg.chol_inv / chol_logdet / chol_solve."""
    from bayespy.utils import misc, linalg
    rs = np.random.RandomState(123)
    out = {}
    # (shapes of the arrays, to_plates, from_plates, ndim)
    cases = [
        (((5, 4),), (4,), (5, 4), 0),
        (((5, 4),), (1,), (5, 4), 0),
        (((5, 4),), (), (5, 4), 0),
        (((), ), (), (5, 4), 0),
        (((1, 4),), (), (5, 4), 0),
        (((5, 1),), (4,), (5, 4), 0),
        (((5, 4, 3),), (4,), (5, 4), 1),
        (((5, 1, 3),), (5, 1), (5, 4), 1),
        (((3, 3),), (), (6, 5), 2),
        (((6, 5, 3, 3),), (5,), (6, 5), 2),
        (((6, 1, 3, 3), (6, 5, 1, 1)), (5,), (6, 5), 2),
        (((6, 5, 1), (6, 5, 3)), (6, 1), (6, 5), 1),
        (((7, 1, 2), (1, 4, 2), (4, 1)), (4,), (7, 4), 1),
        (((2, 3, 4, 5),), (3, 1, 5), (2, 3, 4, 5), 0),
        (((2, 3, 4, 5), (4, 1)), (2, 1, 1, 1), (2, 3, 4, 5), 0),
    ]
    for i, (shapes, to_plates, from_plates, ndim) in enumerate(cases):
        arrs = [rs.normal(size=s) for s in shapes]
        y = misc.sum_multiply_to_plates(*arrs, to_plates=to_plates, from_plates=from_plates,
                                        ndim=ndim)
        for j, a in enumerate(arrs):
            out['smtp%d_in%d' % (i, j)] = a
        out['smtp%d_out' % i] = np.asarray(y)
        out['smtp%d_meta' % i] = np.array([len(arrs), ndim])
        out['smtp%d_to' % i] = np.array(to_plates, dtype=np.int64)
        out['smtp%d_from' % i] = np.array(from_plates, dtype=np.int64)
    out['smtp_n'] = len(cases)
    x = rs.normal(size=(7, 5)) * 10
    x[2, :] = -np.inf
    x[3, 1] = -np.inf
    with np.errstate(all='ignore'):
        out['lse_in'] = x
        out['lse_out'] = misc.logsumexp(x, axis=-1)
        p, ls = misc.normalized_exp(x)
        out['nexp_p'], out['nexp_lse'] = p, ls
    a = 3.0 + rs.gamma(2.0, size=(4, 3))
    out['mdg_in'] = a
    out['mdg_out5'] = misc.multidigamma(a, 5)
    for n in (1, 3, 8, 20):
        A = rs.normal(size=(4, 2, n, n))
        C = A @ np.swapaxes(A, -1, -2) + n * np.eye(n)
        b = rs.normal(size=(4, 2, n))
        U = linalg.chol(C)
        out['chol%d_C' % n] = C
        out['chol%d_b' % n] = b
        out['chol%d_inv' % n] = linalg.chol_inv(U)
        out['chol%d_logdet' % n] = linalg.chol_logdet(U)
        out['chol%d_solve' % n] = linalg.chol_solve(U, b)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'cases', len(cases))


def small_model_cases(name):
    """Small graphs that exercise every built node family and the message routing
    (plates, masks, plate multipliers); used to pin the generic device engine."""
    from bayespy.nodes import (GaussianARD, Gaussian, Gamma, Wishart, Dirichlet, Categorical,
                               SumMultiply)
    from bayespy.inference import VB
    rs = np.random.RandomState(77)
    out = {}

    def run(tag, Q, nodes, n_iter=4):
        Q.ignore_bound_checks = True
        Ls = []
        for _ in range(n_iter):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
        out[tag + '_L'] = np.array(Ls)
        for nm, nd in nodes.items():
            for i, ui in enumerate(nd.u):
                out['%s_%s_u%d' % (tag, nm, i)] = np.asarray(ui)
            out['%s_%s_L' % (tag, nm)] = np.array(Q.l[nd][:Q.iter])

    # (a) masked PCA (demos/pca.py:80-82 default usage: randomly missing values)
    D, N, K = 5, 60, 2
    w, x = rs.normal(size=(D, K)), rs.normal(size=(N, K))
    y = w @ x.T + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < 0.8
    x0 = rs.normal(size=(N, K))
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None])
    Y.observe(y, mask=mask)
    out['mpca_y'], out['mpca_mask'], out['mpca_x0'] = y, mask, x0
    run('mpca', VB(Y, F, W, X, tau, alpha), dict(W=W, X=X, tau=tau, alpha=alpha))

    # (b) vector GaussianARD with a Gaussian mean parent and per-component precision
    N = 50
    data = rs.normal(size=(N, 3)) * np.array([1.0, 2.0, 0.5]) + np.array([1.0, -2.0, 0.0])
    mu = GaussianARD(0, 1e-3, shape=(3,), name='mu')
    al = Gamma(1e-3, 1e-3, plates=(3,), name='al')
    yy = GaussianARD(mu, al, shape=(3,), plates=(N,), name='yy')
    yy.observe(data)
    out['vard_data'] = data
    run('vard', VB(yy, mu, al), dict(mu=mu, al=al))

    # (c) Gaussian with Wishart precision
    N = 40
    data = rs.multivariate_normal([1.0, 0.0, -1.0], [[1.0, 0.5, 0.0], [0.5, 2.0, 0.3],
                                                      [0.0, 0.3, 0.5]], size=N)
    mu = GaussianARD(0, 1e-3, shape=(3,), name='mu')
    Lam = Wishart(3, np.identity(3), name='Lam')
    yg = Gaussian(mu, Lam, plates=(N,), name='yg')
    yg.observe(data)
    out['gw_data'] = data
    run('gw', VB(yg, mu, Lam), dict(mu=mu, Lam=Lam))

    # (d) Dirichlet + Categorical with observed labels
    lab = rs.randint(4, size=30)
    p = Dirichlet(np.array([1.0, 0.5, 2.0, 1.5]), name='p')
    z = Categorical(p, plates=(30,), name='z')
    z.observe(lab)
    out['dc_lab'] = lab
    run('dc', VB(z, p), dict(p=p), n_iter=2)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: v for k, v in out.items() if k.endswith('_L') and k.count('_') == 1})


def lssm_cases(name):
    """Linear state-space models (bayespy/demos/lssm.py:33-103): a single chain with fixed
    innovation precision, and a batch of sequences with a Gamma innovation precision;
    plus known answers of linalg.block_banded_solve (utils/linalg.py:468-575)."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
    from bayespy.utils import linalg
    rs = np.random.RandomState(2024)
    out = {}

    def build(tag, M, T, D, B, gamma_nu):
        plates_x = () if B is None else (B,)
        a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
        nseq = 1 if B is None else B
        x = np.zeros((nseq, T, D))
        x[:, 0] = rs.normal(size=(nseq, D))
        for t in range(1, T):
            x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(nseq, D))
        c_true = rs.normal(size=(M, D))
        f = np.einsum('md,btd->mbt', c_true, x)
        y = f + 0.3 * rs.normal(size=f.shape)
        if B is None:
            y = y[:, 0]
        alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
        A.initialize_from_value(np.identity(D))
        nu = Gamma(1e-3, 1e-3, plates=(D,), name='nu') if gamma_nu else np.ones(D)
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, nu, n=T,
                                plates=plates_x, name='X')
        x0 = rs.normal(size=plates_x + (T, D))
        X.initialize_from_value(x0)
        gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
        gamma.initialize_from_value(1e-2 * np.ones(D))
        cplates = (M, 1) if B is None else (M, 1, 1)
        C = GaussianARD(0, gamma, shape=(D,), plates=cplates, name='C')
        c0 = rs.normal(size=cplates + (D,))
        C.initialize_from_value(c0)
        tau = Gamma(1e-5, 1e-5, name='tau')
        tau.initialize_from_value(1e2)
        F = SumMultiply('i,i', C, X, name='F')
        Y = GaussianARD(F, tau, name='Y')
        Y.observe(y)
        nodes = [Y, F, C, gamma, X, A, alpha, tau]
        if gamma_nu:
            nodes.append(nu)
        Q = VB(*nodes)
        Q.ignore_bound_checks = True
        Ls = []
        n_iter = 4
        for _ in range(n_iter):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
        out[tag + '_y'], out[tag + '_x0'], out[tag + '_c0'] = y, x0, c0
        out[tag + '_L'] = np.array(Ls)
        track = dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma)
        if gamma_nu:
            track['nu'] = nu
        for nm, nd in track.items():
            for i, ui in enumerate(nd.u):
                out['%s_%s_u%d' % (tag, nm, i)] = np.asarray(ui)
            out['%s_%s_L' % (tag, nm)] = np.array(Q.l[nd][:Q.iter])
        print(tag, Ls)

    build('lssm1', M=5, T=30, D=3, B=None, gamma_nu=False)
    build('lssmB', M=4, T=25, D=2, B=6, gamma_nu=True)
    build('lssmBc', M=4, T=25, D=2, B=6, gamma_nu=False)
    build('lssm1g', M=5, T=30, D=3, B=None, gamma_nu=True)

    # block_banded_solve known answers: shared and per-sequence matrices
    for tag, pl in (('bbs_shared', ()), ('bbs_batch', (5,))):
        T, D = 12, 3
        Z = rs.normal(size=pl + (T * D, T * D + 3))
        full = Z @ np.swapaxes(Z, -1, -2) + T * D * np.eye(T * D)
        Ab = np.stack([full[..., t * D:(t + 1) * D, t * D:(t + 1) * D] for t in range(T)], axis=-3)
        Bb = np.stack([full[..., t * D:(t + 1) * D, (t + 1) * D:(t + 2) * D]
                       for t in range(T - 1)], axis=-3)
        yv = rs.normal(size=(5, T, D))
        V, Cc, xx, ld = linalg.block_banded_solve(Ab, Bb, yv)
        out[tag + '_A'], out[tag + '_B'], out[tag + '_y'] = Ab, Bb, yv
        out[tag + '_V'], out[tag + '_C'], out[tag + '_x'], out[tag + '_ld'] = V, Cc, xx, ld
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


def rotation_cases(name):
    """PCA with the rotation parameter expansion run as the VB callback, exactly as
    demos/pca.py:85-94 does (RotateGaussianARD(W, alpha), RotateGaussianARD(X),
    RotationOptimizer): (a) fully observed, (b) 20 % missing; plus one stand-alone rotation
    from a fixed state with the optimal R recorded."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    from bayespy.inference.vmp import transformations
    import warnings
    rs = np.random.RandomState(123)
    out = {}
    for tag, D, N, K, masked in (('rot', 8, 300, 4, False), ('rotm', 6, 80, 3, True)):
        w, x = rs.normal(size=(D, K - 1)), rs.normal(size=(N, K - 1))
        y = w @ x.T + 0.1 * rs.normal(size=(D, N))
        mask = rs.rand(D, N) < 0.8
        x0 = rs.normal(size=(N, K))
        alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
        W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
        X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
        F = SumMultiply('i,i', W, X, name='F')
        tau = Gamma(1e-2, 1e-2, name='tau')
        Y = GaussianARD(F, tau, name='Y')
        X.initialize_from_value(x0[None])
        if masked:
            Y.observe(y, mask=mask)
        else:
            Y.observe(y)
        Q = VB(Y, F, W, X, tau, alpha)
        Q.ignore_bound_checks = True
        rotW = transformations.RotateGaussianARD(W, alpha)
        rotX = transformations.RotateGaussianARD(X)
        R = transformations.RotationOptimizer(rotW, rotX, K)
        # stand-alone rotation after two plain iterations
        Q.update(repeat=2, verbose=False)
        L_before = Q.compute_lowerbound()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            R.rotate()
        L_after = Q.compute_lowerbound()
        out[tag + '_L_before'], out[tag + '_L_after'] = np.array(L_before), np.array(L_after)
        # copies: the reference updates moments in place (stochastic.py:248-250)
        out[tag + '_W_u0_rot'], out[tag + '_X_u0_rot'] = np.array(W.u[0]), np.array(X.u[0])
        out[tag + '_alpha_u0_rot'] = np.array(alpha.u[0])
        # then the demo's usage: rotate in the callback of every iteration
        Q.callback = R.rotate
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Q.update(repeat=6, verbose=False)
        out[tag + '_y'], out[tag + '_mask'], out[tag + '_x0'] = y, mask, x0
        out[tag + '_L'] = np.array(Q.L[:Q.iter])
        out[tag + '_W_u0'], out[tag + '_X_u0'] = np.asarray(W.u[0]), np.asarray(X.u[0])
        out[tag + '_tau_u0'], out[tag + '_alpha_u0'] = np.asarray(tau.u[0]), np.asarray(alpha.u[0])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: v for k, v in out.items() if '_L' in k})


def pca_doctest_case(name):
    """doc/source/examples/pca.rst:8-118 verbatim (numpy.random.seed(1) from its testsetup):
    PCA with ARD and the rotation callback run to convergence.  The doctest pins
    "Iteration 1: loglike=-2.33...e+03" and a final "loglike=6.500...e+02"; the drawn initial
    value of C is recorded so that the run can be repeated without sharing the RNG stream."""
    import io
    import contextlib
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    from bayespy.inference.vmp.transformations import RotateGaussianARD, RotationOptimizer
    np.random.seed(1)
    M, N = 20, 100
    x = np.random.randn(N, 2)
    w = np.random.randn(M, 2)
    f = np.einsum('ik,jk->ij', w, x)
    y = f + 0.1 * np.random.randn(M, N)
    D = 10
    X = GaussianARD(0, 1, plates=(1, N), shape=(D,))
    alpha = Gamma(1e-5, 1e-5, plates=(D,))
    C = GaussianARD(0, alpha, plates=(M, 1), shape=(D,))
    F = SumMultiply('d,d->', X, C)
    tau = Gamma(1e-5, 1e-5)
    Y = GaussianARD(F, tau)
    Y.observe(y)
    Q = VB(Y, X, C, alpha, tau)
    C.initialize_from_random()
    C_init = np.array(C.u[0])
    R = RotationOptimizer(RotateGaussianARD(X), RotateGaussianARD(C, alpha), D)
    Q.set_callback(R.rotate)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=1000)
    lines = buf.getvalue().strip().splitlines()
    assert lines[0].startswith('Iteration 1: loglike=-2.33'), lines[0]
    assert 'loglike=6.500' in lines[-2] and lines[-1].startswith('Converged'), lines[-2:]
    np.savez_compressed(os.path.join(OUT, name + '.npz'), y=y, C_init=C_init,
                        L=np.array(Q.L[:Q.iter]), n_iter=Q.iter,
                        alpha_u0=np.array(alpha.u[0]), tau_u0=np.array(tau.u[0]),
                        F_u0=np.array(F.get_moments()[0]))
    print(name, lines[0], '|', lines[-2], '|', lines[-1])


def svi_case(name):
    """Stochastic variational inference on a Gaussian mixture, the loop of
    demos/stochastic_inference.py:85-133 with a recorded mini-batch sequence: the class node
    carries plates_multiplier=(N/N_batch,), each step observes a mini-batch, updates the local
    node and takes a Riemannian gradient step on the global nodes."""
    from bayespy.nodes import GaussianARD, Gaussian, Dirichlet, Categorical, Mixture
    from bayespy.inference import VB
    rs = np.random.RandomState(5)
    N, D, K, NB, steps = 600, 2, 3, 50, 8
    centers = 4 * rs.normal(size=(K, D))
    data = centers[rs.randint(K, size=N)] + rs.normal(size=(N, D))
    mu0 = rs.normal(size=(K, D))
    batches = np.array([rs.choice(N, NB) for _ in range(steps)])
    mu = GaussianARD(0, 0.001, shape=(D,), plates=(K,), name='means')
    alpha = Dirichlet(np.ones(K), name='class probabilities')
    Z = Categorical(alpha, plates=(NB,), plates_multiplier=(N / NB,), name='classes')
    Y = Mixture(Z, Gaussian, mu, np.identity(D), name='observations')
    mu.initialize_from_value(mu0)
    Q = VB(Y, Z, mu, alpha)
    Q.ignore_bound_checks = True
    out = dict(data=data, mu0=mu0, batches=batches, N=N, NB=NB)
    Ls, mus, als = [], [], []
    for n in range(steps):
        Y.observe(data[batches[n], :])
        Q.update(Z, verbose=False)
        step = (n + 1) ** (-0.7)
        Q.gradient_step(mu, alpha, scale=step)
        Ls.append(Q.compute_lowerbound())
        mus.append(np.array(mu.u[0]))
        als.append(np.array(alpha.u[0]))
    out['L'], out['mu_u0'], out['alpha_u0'] = np.array(Ls), np.array(mus), np.array(als)
    out['Z_u0_last'] = np.array(Z.u[0])
    out['L_terms_last'] = np.array([Y.lower_bound_contribution(), Z.lower_bound_contribution(),
                                    mu.lower_bound_contribution(),
                                    alpha.lower_bound_contribution()])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['L'])


def multinomial_case(name):
    """Dirichlet + Multinomial with observed count vectors (different numbers of trials per
    plate), and a latent Multinomial feeding a Mixture-free check of its moments."""
    from bayespy.nodes import Dirichlet, Multinomial
    from bayespy.inference import VB
    rs = np.random.RandomState(21)
    K, N = 5, 40
    trials = rs.randint(1, 30, size=N)
    ptrue = rs.dirichlet(np.ones(K))
    counts = np.array([rs.multinomial(t, ptrue) for t in trials])
    p = Dirichlet(np.array([1.0, 0.5, 2.0, 1.5, 1.0]), name='p')
    x = Multinomial(trials, p, name='x')
    x.observe(counts)
    Q = VB(x, p)
    Q.update(repeat=2, verbose=False)
    out = dict(trials=trials, counts=counts, L=np.array(Q.L[:Q.iter]),
               p_u0=np.array(p.u[0]), L_x=np.array(Q.l[x][:Q.iter]), L_p=np.array(Q.l[p][:Q.iter]))
    # latent multinomial: moments from the prior
    p2 = Dirichlet(np.array([2.0, 1.0, 3.0]), name='p2')
    z = Multinomial(7, p2, plates=(4,), name='z')
    Q2 = VB(z, p2)
    Q2.update(repeat=2, verbose=False)
    out['z_u0'] = np.array(z.u[0])
    out['L2'] = np.array(Q2.L[:Q2.iter])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['L'], out['L2'])


def summultiply_cases(name):
    """SumMultiply beyond the PCA pattern (dot.py:19-633, the shapes of test_dot.py): a
    matrix-vector product with a matrix-shaped Gaussian parent, and a three-factor product
    with plates broadcast three ways (PARAFAC-like)."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    rs = np.random.RandomState(31)
    out = {}

    def run(tag, Q, track, n_iter=4):
        Q.ignore_bound_checks = True
        Q.update(repeat=n_iter, verbose=False)
        out[tag + '_L'] = np.array(Q.L[:Q.iter])
        for nm, nd in track.items():
            for i, ui in enumerate(nd.u):
                out['%s_%s_u%d' % (tag, nm, i)] = np.array(ui)
            out['%s_%s_L' % (tag, nm)] = np.array(Q.l[nd][:Q.iter])

    # (a) y_n = A x_n + noise, A a (2,3) matrix-valued GaussianARD, x_n vectors
    N = 30
    A0 = rs.normal(size=(2, 3))
    x_true = rs.normal(size=(N, 3))
    y = x_true @ A0.T + 0.1 * rs.normal(size=(N, 2))
    x0 = rs.normal(size=(N, 3))
    A = GaussianARD(0, 1e-2, shape=(2, 3), name='A')
    x = GaussianARD(0, 1, shape=(3,), plates=(N,), name='x')
    F = SumMultiply('ij,j->i', A, x, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    x.initialize_from_value(x0)
    Y.observe(y)
    out['mv_y'], out['mv_x0'] = y, x0
    run('mv', VB(Y, F, A, x, tau), dict(A=A, x=x, tau=tau))

    # (b) three factors with plates (4,1,1), (1,5,1), (1,1,6), contracted over the component
    I, J, Kp, C = 4, 5, 6, 2
    a_t, b_t, c_t = rs.normal(size=(I, C)), rs.normal(size=(J, C)), rs.normal(size=(Kp, C))
    y3 = np.einsum('ic,jc,kc->ijk', a_t, b_t, c_t) + 0.1 * rs.normal(size=(I, J, Kp))
    b0, c0 = rs.normal(size=(1, J, 1, C)), rs.normal(size=(1, 1, Kp, C))
    a = GaussianARD(0, 1e-1, shape=(C,), plates=(I, 1, 1), name='a')
    b = GaussianARD(0, 1e-1, shape=(C,), plates=(1, J, 1), name='b')
    c = GaussianARD(0, 1e-1, shape=(C,), plates=(1, 1, Kp), name='c')
    F3 = SumMultiply('i,i,i', a, b, c, name='F3')
    tau3 = Gamma(1e-2, 1e-2, name='tau3')
    Y3 = GaussianARD(F3, tau3, name='Y3')
    b.initialize_from_value(b0)
    c.initialize_from_value(c0)
    Y3.observe(y3)
    out['pf_y'], out['pf_b0'], out['pf_c0'] = y3, b0, c0
    run('pf', VB(Y3, F3, a, b, c, tau3), dict(a=a, b=b, c=c, tau3=tau3))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['mv_L'], out['pf_L'])


def mixture_ard_case(name):
    """Mixture over GaussianARD components (diagonal covariances: latent means AND Gamma
    precisions per cluster and dimension, mixture.py:359-545 with gaussian.py:1559-1774)."""
    from bayespy.nodes import GaussianARD, Gamma, Dirichlet, Categorical, Mixture
    from bayespy.inference import VB
    rs = np.random.RandomState(41)
    N, D, K = 120, 3, 4
    centers = 3 * rs.normal(size=(K, D))
    lab = rs.randint(K, size=N)
    y = centers[lab] + rs.normal(size=(N, D)) * np.array([0.3, 0.6, 1.0])
    lab0 = rs.randint(K, size=N)
    alpha = Dirichlet(np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-2, shape=(D,), plates=(K,), name='mu')
    lam = Gamma(1e-1, 1e-1, plates=(K, D), name='lam')
    Y = Mixture(z, GaussianARD, mu, lam, 1, name='Y')      # positional ndim=1 of GaussianARD
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, lam, z, alpha)
    Q.ignore_bound_checks = True
    Q.update(repeat=4, verbose=False)
    out = dict(y=y, lab0=lab0, L=np.array(Q.L[:Q.iter]))
    for nm, nd in dict(alpha=alpha, z=z, mu=mu, lam=lam).items():
        for i, ui in enumerate(nd.u):
            out['%s_u%d' % (nm, i)] = np.array(ui)
        out['%s_L' % nm] = np.array(Q.l[nd][:Q.iter])
    out['Y_L'] = np.array(Q.l[Y][:Q.iter])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['L'], Y.plates, mu.plates, lam.plates)


def parameter_api_case(name):
    """Variational-parameter access, gradients, annealing, collapsed CG and pattern search:
    tests/models.py run_parameter_api_cases executed on the reference."""
    import bayespy.nodes
    from bayespy.inference import VB
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    import models
    g = models.make_parameter_api_inputs(np.random.RandomState(77))
    res = models.run_parameter_api_cases(bayespy.nodes, VB, g)
    out = {'in_' + k: v for k, v in g.items()}
    for k, v in res.items():
        if isinstance(v, list):
            for i, vi in enumerate(v):
                out['%s_%d' % (k, i)] = np.array(vi)
        else:
            out[k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['cg_L'], out['mix_L'][-3:], out['an0_g_0'], out['an0_g_1'])


def _shared_case(name, make_inputs, run, seed, show):
    """Run a statement-for-statement shared script of tests/models.py on the reference and
    store its inputs (in_*) and results."""
    import bayespy.nodes
    from bayespy.inference import VB
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    import models
    g = getattr(models, make_inputs)(np.random.RandomState(seed))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = getattr(models, run)(bayespy.nodes, VB, g)
    out = {'in_' + k: v for k, v in g.items()}
    for k, v in res.items():
        if isinstance(v, list):
            for i, vi in enumerate(v):
                out['%s_%d' % (k, i)] = np.array(vi)
        else:
            out[k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, *[out[k] for k in show])


def count_nodes_case(name):
    """Beta / Bernoulli / Binomial / Poisson / Complement / Add and count mixtures."""
    _shared_case(name, 'make_count_node_inputs', 'run_count_node_cases', 99,
                 ('bern_L', 'bmm_L', 'pmm_L', 'add_L'))


def plate_nodes_case(name):
    """Take / Concatenate / Gate inside small models."""
    _shared_case(name, 'make_plate_node_inputs', 'run_plate_node_cases', 314,
                 ('tk_doc', 'tk_L', 'tk2_L', 'cc_L', 'cc2_L', 'gt_L', 'gt2_L'))


def slice_nodes_case(name):
    """Plate indexing X[...] and Choose."""
    _shared_case(name, 'make_slice_inputs', 'run_slice_cases', 2718, ('sl_plates', 'sl_L', 'ch_doc', 'ch_L'))


def switching_case(name):
    """Switching linear state-space model (demos/lssm_sd.py)."""
    _shared_case(name, 'make_switching_inputs', 'run_switching_case', 606, ('sw_L',))


def varying_case(name):
    """Linear state-space model with time-varying dynamics (demos/lssm_tvd.py)."""
    _shared_case(name, 'make_varying_inputs', 'run_varying_case', 909, ('tv_L',))


def concat_gaussian_case(name):
    """ConcatGaussian (concat_gaussian.py)."""
    _shared_case(name, 'make_concat_gaussian_inputs', 'run_concat_gaussian_case', 4242, ('cg_L',))


def default_ndim_case(name):
    """GaussianARD's default ndim = 0 under vector- and matrix-valued means."""
    _shared_case(name, 'make_default_ndim_inputs', 'run_default_ndim_case', 1234,
                 ('dn_X_plates', 'dn_L', 'dn_Z_plates', 'dn_Z_shape', 'dn2_L'))


def masked_pca_case(name):
    """PCA with missing values (NaN placeholders), partially observed nodes."""
    _shared_case(name, 'make_masked_pca_inputs', 'run_masked_pca_cases', 5150,
                 ('m0_L', 'm1_L', 'm2_L', 'm3_L', 'po_L'))


def bmm_doctest_case(name):
    """doc/source/examples/bmm.rst:8-95 verbatim (numpy.random.seed(1) from its testsetup): the
    Bernoulli mixture whose doctest pins "Iteration 1: loglike=-6.872145e+02" and
    "Iteration 17: loglike=-5.236921e+02".  The drawn data and the random initial value of P are
    recorded so that the run can be repeated without sharing the RNG stream."""
    from bayespy.utils import random as brandom
    from bayespy.nodes import Categorical, Dirichlet, Beta, Mixture, Bernoulli
    from bayespy.inference import VB
    np.random.seed(1)
    p0 = [0.1, 0.9, 0.1, 0.9, 0.1, 0.9, 0.1, 0.9, 0.1, 0.9]
    p1 = [0.1, 0.1, 0.1, 0.1, 0.1, 0.9, 0.9, 0.9, 0.9, 0.9]
    p2 = [0.9, 0.9, 0.9, 0.9, 0.9, 0.1, 0.1, 0.1, 0.1, 0.1]
    p = np.array([p0, p1, p2])
    z = brandom.categorical([1 / 3, 1 / 3, 1 / 3], size=100)
    x = brandom.bernoulli(p[z])
    N, D, K = 100, 10, 10
    R = Dirichlet(K * [1e-5], name='R')
    Z = Categorical(R, plates=(N, 1), name='Z')
    P = Beta([0.5, 0.5], plates=(D, K), name='P')
    X = Mixture(Z, Bernoulli, P)
    Q = VB(Z, R, X, P)
    P.initialize_from_random()
    p_init = np.exp(np.array(P.u[0])[..., 0])
    X.observe(x)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=1000, verbose=False)
    out = dict(x=np.array(x, dtype=np.int64), p_init=p_init, L=np.array(Q.L[:Q.iter]),
               R_u0=np.array(R.u[0]), P_u0=np.array(P.u[0]))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, '%e' % out['L'][0], '%e' % out['L'][-1], len(out['L']))


def gmm_doctest_case(name):
    """doc/source/examples/gmm.rst:26-117 verbatim (numpy seed 1): the Gaussian mixture whose
    doctest pins "Iteration 1: loglike=-1.402345e+03" and "Iteration 61: loglike=-8.888464e+02".
    The drawn data and the random initial labels of Z are recorded."""
    from bayespy.nodes import Dirichlet, Categorical, Gaussian, Wishart, Mixture
    from bayespy.inference import VB
    np.random.seed(1)
    y0 = np.random.multivariate_normal([0, 0], [[2, 0], [0, 0.1]], size=50)
    y1 = np.random.multivariate_normal([0, 0], [[0.1, 0], [0, 2]], size=50)
    y2 = np.random.multivariate_normal([2, 2], [[2, -1.5], [-1.5, 2]], size=50)
    y3 = np.random.multivariate_normal([-2, -2], [[0.5, 0], [0, 0.5]], size=50)
    y = np.vstack([y0, y1, y2, y3])
    N, D, K = 200, 2, 10
    alpha = Dirichlet(1e-5 * np.ones(K), name='alpha')
    Z = Categorical(alpha, plates=(N,), name='z')
    mu = Gaussian(np.zeros(D), 1e-5 * np.identity(D), plates=(K,), name='mu')
    Lambda = Wishart(D, 1e-5 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(Z, Gaussian, mu, Lambda, name='Y')
    Z.initialize_from_random()
    z_init = np.argmax(np.array(Z.u[0]), axis=-1)
    Q = VB(Y, mu, Lambda, Z, alpha)
    Y.observe(y)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=1000, verbose=False)
    out = dict(y=y, z_init=z_init, L=np.array(Q.L[:Q.iter]), alpha_u0=np.array(alpha.u[0]),
               mu_u0=np.array(mu.u[0]))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, '%e' % out['L'][0], '%e' % out['L'][-1], len(out['L']))


def inference_doctest_case(name):
    """doc/source/user_guide/inference.rst:8-235 verbatim (numpy seed 1): the PCA model observed
    with whole rows masked, X initialised from parameters, updates of node subsets in a given
    order, and convergence at the default and at a tighter tolerance.  The doctest pins
    iterations 1-14 and "Converged at iteration 488." / "... 847."."""
    from bayespy.nodes import GaussianARD, Gamma, Dot
    from bayespy.inference import VB
    np.random.seed(1)
    D = 3
    X = GaussianARD(0, 1, shape=(D,), plates=(1, 100), name='X')
    alpha = Gamma(1e-3, 1e-3, plates=(D,), name='alpha')
    C = GaussianARD(0, alpha, shape=(D,), plates=(10, 1), name='C')
    F = Dot(C, X)
    tau = Gamma(1e-3, 1e-3, name='tau')
    Y = GaussianARD(F, tau)
    c = np.random.randn(10, 2)
    x = np.random.randn(2, 100)
    data = np.dot(c, x) + 0.1 * np.random.randn(10, 100)
    Y.observe(data)
    mask = [[True], [False], [False], [True], [True], [False], [True], [True], [True], [False]]
    Y.observe(data, mask=mask)
    Q = VB(Y, C, X, alpha, tau)
    x_init = np.random.randn(1, 100, D)
    X.initialize_from_parameters(x_init, 10)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(verbose=False)
        Q.update(C, X, verbose=False)
        Q.update(C, X, C, tau, verbose=False)
        Q.update(repeat=10, verbose=False)
        Q.update(repeat=1000, verbose=False)
        n1 = Q.iter
        Q.update(repeat=10000, tol=1e-6, verbose=False)
    out = dict(data=data, mask=np.array(mask), x_init=x_init, L=np.array(Q.L[:Q.iter]),
               n_default=n1, n_tight=Q.iter, tau_u0=np.array(tau.u[0]), alpha_u0=np.array(alpha.u[0]))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, ['%e' % v for v in out['L'][:4]], n1, Q.iter, '%e' % out['L'][n1 - 1], '%e' % out['L'][-1])


def lssm_doctest_case(name):
    """doc/source/examples/lssm.rst:45-202 verbatim (numpy seed 1): linear state-space model with
    a 10-dimensional latent chain, 400 time instances, 30 observed dimensions, 80 % missing
    values.  The doctest pins "Iteration 1: loglike=-1.439704e+05" and "Iteration 10:
    loglike=-1.051441e+04".  Data, mask and the random initial value of C are recorded."""
    from bayespy.nodes import GaussianARD, GaussianMarkovChain, Gamma, Dot
    from bayespy.inference import VB
    from bayespy.utils import random as brandom
    np.random.seed(1)
    M, N, D = 30, 400, 10
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=N, name='X')
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1), name='C')
    F = Dot(C, X, name='F')
    C.initialize_from_random()
    C_init = np.array(C.u[0])
    tau = Gamma(1e-5, 1e-5, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    Q = VB(X, C, gamma, A, alpha, tau, Y)
    w = 0.3
    a = np.array([[np.cos(w), -np.sin(w), 0, 0], [np.sin(w), np.cos(w), 0, 0], [0, 0, 1, 0],
                  [0, 0, 0, 0]])
    c = np.random.randn(M, 4)
    x = np.empty((N, 4))
    f = np.empty((M, N))
    y = np.empty((M, N))
    x[0] = 10 * np.random.randn(4)
    f[:, 0] = np.dot(c, x[0])
    y[:, 0] = f[:, 0] + 3 * np.random.randn(M)
    for n in range(N - 1):
        x[n + 1] = np.dot(a, x[n]) + [1, 1, 10, 10] * np.random.randn(4)
        f[:, n + 1] = np.dot(c, x[n + 1])
        y[:, n + 1] = f[:, n + 1] + 3 * np.random.randn(M)
    mask = brandom.mask(M, N, p=0.2)
    Y.observe(y, mask=mask)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=10, verbose=False)
    out = dict(y=y, mask=mask, C_init=C_init, L=np.array(Q.L[:Q.iter]), tau_u0=np.array(tau.u[0]),
               A_u0=np.array(A.u[0]), X_u0=np.array(X.u[0]))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, '%e' % out['L'][0], '%e' % out['L'][-1], len(out['L']))


def markov_chain_case(name):
    """Categorical Markov chains: raw alpha-beta recursions (utils/random.py:357-422) and the
    models of tests/models.py run_markov_chain_cases.  The data of the two doctest models of
    doc/source/examples/hmm.rst are drawn here exactly as the document does (numpy seed 1)."""
    import bayespy.nodes
    from bayespy.nodes import CategoricalMarkovChain, Categorical, Mixture
    from bayespy.inference import VB
    from bayespy.utils import random as brandom
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    import models
    out = {}
    rs = np.random.RandomState(2024)
    for tag, plates_p0, plates_P, N, K in (('ab_a', (), (), 7, 3), ('ab_b', (5,), (5,), 12, 8),
                                          ('ab_c', (4, 1), (3,), 6, 20), ('ab_d', (2,), (), 5, 40),
                                          ('ab_e', (3,), (3,), 1, 2), ('ab_f', (300,), (), 9, 5)):
        logp0 = np.log(rs.dirichlet(np.ones(K), size=plates_p0)) + rs.normal(size=plates_p0 + (K,))
        logP = np.log(rs.dirichlet(np.ones(K), size=plates_P + (N, K))) \
            + 3 * rs.normal(size=plates_P + (N, 1, K))
        if tag == 'ab_c':
            logP[..., 2, :, 1] = -np.inf            # an impossible state at one instance
        z0, zz, gg = brandom.alpha_beta_recursion(logp0, logP)
        out.update({tag + '_logp0': logp0, tag + '_logP': logP, tag + '_z0': z0, tag + '_zz': zz,
                    tag + '_g': gg})
    # hmm.rst testsetup: numpy.random.seed(1)
    np.random.seed(1)
    Z = CategoricalMarkovChain([0.6, 0.4], [[0.7, 0.3], [0.4, 0.6]], states=100)
    P = [[0.1, 0.4, 0.5], [0.6, 0.3, 0.1]]
    Y = Mixture(Z, Categorical, P)
    weather = Z.random()
    activity = Mixture(weather, Categorical, P).random()
    g = {'hmm1_activity': np.array(activity)}
    mu = np.array([[0, 0], [3, 4], [6, 0]])
    K, N, std = 3, 200, 2.0
    p0 = np.ones(K) / K
    q = 0.9
    r = (1 - q) / (K - 1)
    Pm = q * np.identity(K) + r * (np.ones((3, 3)) - np.identity(3))
    y = np.zeros((N, 2))
    state = np.random.choice(K, p=p0)
    for n in range(N):
        y[n, :] = std * np.random.randn(2) + mu[state]
        state = np.random.choice(K, p=Pm[state])
    g['hmm2_y'] = y
    rs = np.random.RandomState(7)
    B, T, K = 6, 15, 4
    zt = rs.randint(K, size=(B, T))
    g['hmm3_y'] = np.array([-6.0, -2.0, 2.0, 6.0])[zt] + rs.normal(size=(B, T))
    g['hmm3_prior'] = 0.5 + rs.rand(T - 1, K, K)
    g['hmm3_z0'] = rs.randint(K, size=(B, T))
    g['hmm4_y'] = np.where(np.arange(30) // 10 % 2 == 0, -2.0, 2.0) + rs.normal(size=30)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = models.run_markov_chain_cases(bayespy.nodes, VB, g)
    out.update({'in_' + k: v for k, v in g.items()})
    for k, v in res.items():
        if isinstance(v, list):
            for i, vi in enumerate(v):
                out['%s_%d' % (k, i)] = np.array(vi)
        else:
            out[k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['hmm1_L'], out['hmm2_L'], out['hmm3_L'], out['hmm4_L'])


def main():
    os.makedirs(OUT, exist_ok=True)
    _import_reference()
    if len(sys.argv) > 1:
        # regenerate selected fixtures only: python oracle/make_golden.py masked_pca_case:masked_pca
        for spec in sys.argv[1:]:
            fn, _, nm = spec.partition(':')
            globals()[fn](nm)
        return
    quickstart_case('quickstart_n10', N=10, n_iter=4, seed=1)
    quickstart_case('quickstart_n1000', N=1000, n_iter=6, seed=1)
    pca_case('pca_n500_d6_k3', N=500, D=6, K=3, n_iter=5, seed=7)
    pca_case('pca_n777_d20_k5', N=777, D=20, K=5, n_iter=5, seed=8)
    pca_case('pca_n2048_d128_k32', N=2048, D=128, K=32, n_iter=4, seed=9)
    pca_case('pca_n4000_d64_k16', N=4000, D=64, K=16, n_iter=4, seed=10)
    gmm_case('gmm_n400_d3_k4', N=400, D=3, K=4, n_iter=5, seed=11)
    gmm_case('gmm_n3000_d8_k16', N=3000, D=8, K=16, n_iter=4, seed=12)
    utils_cases('utils_known_answers')
    small_model_cases('small_models')
    lssm_cases('lssm')
    rotation_cases('rotations')
    pca_doctest_case('pca_doctest')
    svi_case('svi_gmm')
    multinomial_case('multinomial')
    summultiply_cases('summultiply')
    mixture_ard_case('mixture_ard')
    parameter_api_case('parameter_api')
    count_nodes_case('count_nodes')
    plate_nodes_case('plate_nodes')
    markov_chain_case('markov_chains')
    slice_nodes_case('slice_nodes')
    switching_case('switching_lssm')
    varying_case('varying_lssm')
    concat_gaussian_case('concat_gaussian')
    default_ndim_case('default_ndim')
    masked_pca_case('masked_pca')
    bmm_doctest_case('bmm_doctest')
    gmm_doctest_case('gmm_doctest')
    inference_doctest_case('inference_doctest')
    lssm_doctest_case('lssm_doctest')


if __name__ == '__main__':
    main()
