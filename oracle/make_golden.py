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


def pca_mean_case(name):
    """demos/pca.py's model with a constant NON-ZERO prior mean of W (GaussianARD(mu, alpha):
    gaussian.py:805-880): mu an array of shape (D, 1, K) (m3), of shape (K,) with W initialised from
    a value (mk), a scalar with the default prior initialisation of W (ms); mk and ms in the update
    order X, W, tau, alpha, so that the initial state of W enters the trace."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    out = {}
    for tag, N, D, K, seed in (('m3', 300, 7, 3, 31), ('mk', 257, 12, 4, 32), ('ms', 130, 5, 2, 33)):
        rs = np.random.RandomState(seed)
        if tag == 'm3':
            mu = rs.normal(0, 1, (D, 1, K))
        elif tag == 'mk':
            mu = rs.normal(0, 2, (K,))
        else:
            mu = np.array(0.75)
        w = np.broadcast_to(mu, (D, 1, K)).reshape(D, K) + 0.5 * rs.normal(0, 1, (D, K))
        x = rs.normal(0, 1, (N, K))
        y = w @ x.T + 0.1 * rs.normal(size=(D, N))
        x0 = rs.normal(0, 1, (N, K))
        w0 = rs.normal(0, 1, (D, K))
        alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
        W = GaussianARD(mu, alpha, shape=(K,), plates=(D, 1), name='W')
        X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
        F = SumMultiply('i,i', W, X, name='F')
        tau = Gamma(1e-2, 1e-2, name='tau')
        Y = GaussianARD(F, tau, name='Y')
        X.initialize_from_value(x0[None, :, :])
        if tag == 'mk':
            W.initialize_from_value(w0[:, None, :])
        Y.observe(y)
        order = (W, X, tau, alpha) if tag == 'm3' else (X, W, tau, alpha)
        Q = VB(Y, F, W, X, tau, alpha)
        Q.ignore_bound_checks = True
        Ls, terms = [], {k: [] for k in ('Y', 'X', 'W', 'tau', 'alpha')}
        for _ in range(5):
            Q.update(*order, repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
            for k in terms:
                terms[k].append(Q.l[Q[k]][Q.iter - 1])
        out.update({tag + '_y': y, tag + '_x0': x0, tag + '_w0': w0, tag + '_mu': mu,
                    tag + '_L': np.array(Ls), tag + '_W_u0': W.u[0], tag + '_W_u1': W.u[1],
                    tag + '_X_u0': X.u[0], tag + '_tau_u0': tau.u[0], tag + '_alpha_u0': alpha.u[0],
                    tag + '_alpha_phi0': alpha.phi[0]})
        out.update({tag + '_L_' + k: np.array(v) for k, v in terms.items()})
        print(name, tag, 'L =', Ls)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


def pca_seeded_case(name, N=100000, D=128, K=32, n_iter=3, seed=2718):
    """The headline (D, K) at the largest N the reference runs comfortably here (35 s per
    iteration).  Stores the SEED and the reference's outputs only; the
    inputs are regenerated by tests/models.py:make_seeded_pca."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy.inference import VB
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    import models
    y, x0 = models.make_seeded_pca(seed, N, D, K)
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
    Ls, terms = [], {k: [] for k in ('Y', 'X', 'W', 'tau', 'alpha')}
    for _ in range(n_iter):
        Q.update(repeat=1, verbose=False)
        Ls.append(Q.L[Q.iter - 1])
        for k in terms:
            terms[k].append(Q.l[Q[k]][Q.iter - 1])
    np.savez_compressed(
        os.path.join(OUT, name + '.npz'), seed=seed, N=N, D=D, K=K, n_iter=n_iter,
        L=np.array(Ls), **{'L_' + k: np.array(v) for k, v in terms.items()},
        W_u0=W.u[0], W_u1_first=W.u[1][:2], X_u0_strided=X.u[0][0, ::97],
        X_cov=X.u[1][0, 0] - np.outer(X.u[0][0, 0], X.u[0][0, 0]),
        tau_u0=tau.u[0], tau_u1=tau.u[1], alpha_u0=alpha.u[0], alpha_u1=alpha.u[1],
        cputime=np.array(Q.cputime[:Q.iter]))
    print(name, 'L =', Ls, 'cputime', Q.cputime[:Q.iter])


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


def _lssm_build(rs, out, tag, M, T, D, B, gamma_nu, n_iter=4):
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
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
        _lssm_build(rs, out, tag, M, T, D, B, gamma_nu)

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



def lssm_wide_state_cases(name):
    """The batched state-space model of lssm_cases with 8, 12 and 16 latent states (the demo itself,
    bayespy/demos/lssm.py:193-247, runs D = 10): the fused block's big-state path (8 < D <= 16) and
    its largest register-resident instance, against the live reference."""
    rs = np.random.RandomState(4242)
    out = {}
    _lssm_build(rs, out, 'w8', M=6, T=20, D=8, B=5, gamma_nu=True)
    _lssm_build(rs, out, 'w12', M=7, T=18, D=12, B=6, gamma_nu=True)
    _lssm_build(rs, out, 'w16', M=20, T=15, D=16, B=4, gamma_nu=False)
    _lssm_build(rs, out, 'w10single', M=12, T=25, D=10, B=None, gamma_nu=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


def lssm_masked_cases(name, wide=False):
    """The state-space model of bayespy/demos/lssm.py:33-103 observed through an ARRAY mask
    (demos/lssm.py:239-246: ``mask = random.mask(M, N, p=0.3); mask[:, 30:80] = False``): the demo's
    own shape (no sequence plate, mask (M, T) with a fully missing stretch), a plate of sequences
    with one mask per sequence (M, B, T), and a mask shared by the sequences (M, 1, T).  NaN
    placeholders at the missing entries.  ``wide``: the cases with 5 ... 8 states
    (lssm_masked_wide.npz; the demo's own D = 10 shape scaled to the block's limit)."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
    from bayespy.utils import linalg as ref_linalg
    rs = np.random.RandomState(778 if wide else 777)
    out = {}

    # A DEFECT of the reference on exactly this path (found while pinning the oracle): with one
    # mask per sequence the diagonal blocks of the chain precision carry the sequence plate,
    # (B, T, D, D), while the off-diagonal blocks (shared dynamics) are (1, T-1, D, D).
    # linalg.block_banded_solve (utils/linalg.py:536-538) then calls
    # chol_solve(U (B,D,D), b (1,D,D), matrix=True), and chol_solve's loop over the matrices of U
    # (utils/linalg.py:113-144) finds no common leading axis (jnd_b empty) and assigns EVERY
    # iteration's solution to the whole output: all sequences get S_t^-1 E of the LAST sequence.
    # Posterior and bound of every sequence but the last are then wrong (the traces "*_L_defect"
    # below; the last sequence agrees with the restatement to 1e-13).  For the per-sequence
    # fixtures the reference therefore runs with that one call repaired -- b broadcast to the
    # plates of U before the unchanged chol_solve -- and everything else untouched.
    orig_chol_solve = ref_linalg.chol_solve

    def chol_solve_fixed(U, b, out=None, matrix=False, ndim=1):
        if matrix and ndim == 1 and isinstance(U, np.ndarray) and np.ndim(U) > 2:
            lead = np.broadcast_shapes(np.shape(U)[:-2], np.shape(b)[:-2])
            b = np.broadcast_to(b, lead + np.shape(b)[-2:])
        return orig_chol_solve(U, b, out=out, matrix=matrix, ndim=ndim)

    def build(tag, M, T, D, B, gamma_nu, make_mask, n_iter=4, repair=False):
        if repair:
            st = rs.get_state()
            build(tag + '@defect', M, T, D, B, gamma_nu, make_mask, n_iter)
            out[tag + '_L_defect'] = out.pop(tag + '@defect_L')
            for k in [k for k in out if k.startswith(tag + '@defect')]:
                del out[k]
            rs.set_state(st)
            ref_linalg.chol_solve = chol_solve_fixed
        try:
            _build(tag, M, T, D, B, gamma_nu, make_mask, n_iter)
        finally:
            ref_linalg.chol_solve = orig_chol_solve

    def _build(tag, M, T, D, B, gamma_nu, make_mask, n_iter):
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
        mask = make_mask(y.shape)
        y = np.where(np.broadcast_to(mask, y.shape), y, np.nan)
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
        Y.observe(y, mask=mask)
        nodes = [Y, F, C, gamma, X, A, alpha, tau]
        if gamma_nu:
            nodes.append(nu)
        Q = VB(*nodes)
        Q.ignore_bound_checks = True
        Ls = []
        for _ in range(n_iter):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
        out[tag + '_y'], out[tag + '_mask'] = y, mask
        out[tag + '_x0'], out[tag + '_c0'] = x0, c0
        out[tag + '_L'] = np.array(Ls)
        track = dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma, Y=Y)
        if gamma_nu:
            track['nu'] = nu
        for nm, nd in track.items():
            if nm != 'Y':
                for i, ui in enumerate(nd.u):
                    out['%s_%s_u%d' % (tag, nm, i)] = np.asarray(ui)
            out['%s_%s_L' % (tag, nm)] = np.array(Q.l[nd][:Q.iter])
        print(tag, Ls)

    def demo_mask(shape):
        m = rs.rand(*shape) < 0.3
        m[:, 30:80] = False
        return m

    def per_sequence(shape):
        m = rs.rand(*shape) < 0.7
        m[:, 1, 10:25] = False            # one sequence loses a stretch in every dimension
        m[2, 3, :] = False                # one dimension of one sequence is never seen
        return m

    def shared(shape):
        M, B, T = shape
        m = rs.rand(M, 1, T) < 0.6
        m[:, :, 5:12] = False
        return m

    def erasures(shape):
        m = rs.rand(*shape) < 0.5
        m[1] = False                      # a dimension without any observation
        m[:, 2] = False                   # a sequence without any observation
        m[:, :, 0] = False                # the first and the last time step are never observed
        m[:, :, -1] = False
        return m

    if wide:
        build('w8', M=10, T=100, D=8, B=None, gamma_nu=False, make_mask=demo_mask, n_iter=3)
        build('w6', M=5, T=25, D=6, B=4, gamma_nu=True, make_mask=per_sequence, n_iter=3,
              repair=True)
        build('w5', M=9, T=20, D=5, B=3, gamma_nu=False, make_mask=shared, n_iter=3)
        build('w7', M=12, T=16, D=7, B=4, gamma_nu=True, make_mask=erasures, n_iter=3,
              repair=True)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
        return
    build('md', M=6, T=200, D=3, B=None, gamma_nu=False, make_mask=demo_mask)
    build('mb', M=4, T=40, D=2, B=5, gamma_nu=True, make_mask=per_sequence, repair=True)
    build('ms', M=5, T=30, D=4, B=3, gamma_nu=False, make_mask=shared)
    build('me', M=4, T=20, D=2, B=4, gamma_nu=True, make_mask=erasures, repair=True)
    build('m1', M=3, T=1, D=2, B=3, gamma_nu=False,
          make_mask=lambda shape: np.array([[[True], [False], [True]], [[True], [True], [False]],
                                            [[False], [True], [True]]]))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


def lssm_prior_init_case(name):
    """The state-space model of bayespy/demos/lssm.py:33-103 over a batch of sequences with every
    node left at its default initialisation (from the prior, expfamily.py:168-184) except the
    loading matrix C (a value, to break the symmetry): the initial moments of the chain are those
    of p(X | <A>, nu, mu0, Lam0)."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
    rs = np.random.RandomState(77)
    M, B, T, D = 4, 5, 30, 3
    a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
    x = np.zeros((B, T, D))
    x[:, 0] = rs.normal(size=(B, D))
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(B, D))
    c_true = rs.normal(size=(M, D))
    y = np.einsum('md,btd->mbt', c_true, x) + 0.3 * rs.normal(size=(M, B, T))
    alpha = Gamma(1e-2, 1e-2, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    X = GaussianMarkovChain(np.zeros(D), 1e-2 * np.identity(D), A, np.ones(D), n=T, plates=(B,),
                            name='X')
    gamma = Gamma(1e-2, 1e-2, plates=(D,), name='gamma')
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    c0 = rs.normal(size=(M, 1, 1, D))
    C.initialize_from_value(c0)
    tau = Gamma(1e-2, 1e-2, name='tau')
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    out = dict(y=y, c0=c0)
    out['X_u0_init'], out['X_u1_init'], out['X_u2_init'] = [np.array(u) for u in X.u]
    Ls = []
    for _ in range(4):
        Q.update(repeat=1, verbose=False)
        Ls.append(Q.L[Q.iter - 1])
    out['L'] = np.array(Ls)
    for nm, nd in dict(X=X, A=A, C=C, tau=tau, alpha=alpha, gamma=gamma).items():
        for i, ui in enumerate(nd.u):
            out['%s_u%d' % (nm, i)] = np.asarray(ui)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['L'])


def lssm_rotation_cases(name):
    """The state-space model of bayespy/demos/lssm.py:33-103 with its rotation speed-up
    (demos/lssm.py:134-190: RotateGaussianMarkovChain(X, RotateGaussianARD(A, alpha)) against
    RotateGaussianARD(C, gamma), transformations.py:1096-1450): one stand-alone rotation after two
    plain iterations, then the rotation after every iteration.  A single chain and a batch."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
    from bayespy.inference.vmp import transformations
    import warnings
    rs = np.random.RandomState(909)
    out = {}
    for tag, M, T, D, B in (('one', 5, 60, 3, None), ('batch', 4, 30, 2, 5)):
        plates_x = () if B is None else (B,)
        nseq = 1 if B is None else B
        a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
        x = np.zeros((nseq, T, D))
        x[:, 0] = rs.normal(size=(nseq, D))
        for t in range(1, T):
            x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(nseq, D))
        c_true = rs.normal(size=(M, D))
        y = np.einsum('md,btd->mbt', c_true, x) + 0.3 * rs.normal(size=(M, nseq, T))
        if B is None:
            y = y[:, 0]
        alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
        A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
        A.initialize_from_value(np.identity(D))
        X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T,
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
        Q = VB(Y, F, C, gamma, X, A, alpha, tau)
        Q.ignore_bound_checks = True
        rotA = transformations.RotateGaussianARD(A, alpha, axis=0)
        rotX = transformations.RotateGaussianMarkovChain(X, rotA)
        rotC = transformations.RotateGaussianARD(C, gamma, axis=0)
        R = transformations.RotationOptimizer(rotX, rotC, D)
        Q.update(repeat=2, verbose=False)
        out[tag + '_L_before'] = np.array(Q.compute_lowerbound())
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            R.rotate(maxiter=10)
        out[tag + '_L_after'] = np.array(Q.compute_lowerbound())
        for nm, nd in dict(X=X, A=A, C=C, alpha=alpha, gamma=gamma).items():
            for i, ui in enumerate(nd.u):
                out['%s_%s_u%d_rot' % (tag, nm, i)] = np.array(ui)
        Ls = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for _ in range(5):
                Q.update(repeat=1, verbose=False)
                Ls.append(Q.L[Q.iter - 1])
                R.rotate(maxiter=10)
        out[tag + '_y'], out[tag + '_x0'], out[tag + '_c0'] = y, x0, c0
        out[tag + '_L'] = np.array(Ls)
        out[tag + '_L_final'] = np.array(Q.compute_lowerbound())
        out[tag + '_tau_u0'] = np.asarray(tau.u[0])
        out[tag + '_F_u0'] = np.asarray(F.get_moments()[0])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: v for k, v in out.items() if '_L' in k})



def lssm_masked_rotation_case(name):
    """demos/lssm.py as it ships: an array mask with a stretch without data (:239-246) AND the
    rotation speed-up after every iteration (:134-190), a single chain.  One stand-alone rotation
    after two plain iterations, then the rotation after every iteration."""
    from bayespy.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy.inference import VB
    from bayespy.inference.vmp import transformations
    rs = np.random.RandomState(4242)
    out = {}
    M, T, D = 6, 80, 3
    a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
    x = np.zeros((T, D))
    x[0] = rs.normal(size=D)
    for t in range(1, T):
        x[t] = a_true @ x[t - 1] + rs.normal(size=D)
    c_true = rs.normal(size=(M, D))
    y = c_true @ x.T + 0.3 * rs.normal(size=(M, T))
    mask = rs.rand(M, T) < 0.5
    mask[:, 30:45] = False
    y = np.where(mask, y, np.nan)
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, name='X')
    x0 = rs.normal(size=(T, D))
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1), name='C')
    c0 = rs.normal(size=(M, 1, D))
    C.initialize_from_value(c0)
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y, mask=mask)
    Q = VB(Y, F, C, gamma, X, A, alpha, tau)
    Q.ignore_bound_checks = True
    rotA = transformations.RotateGaussianARD(A, alpha, axis=0)
    rotX = transformations.RotateGaussianMarkovChain(X, rotA)
    rotC = transformations.RotateGaussianARD(C, gamma, axis=0)
    R = transformations.RotationOptimizer(rotX, rotC, D)
    Q.update(repeat=2, verbose=False)
    out['L_before'] = np.array(Q.compute_lowerbound())
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        R.rotate(maxiter=10)
    out['L_after'] = np.array(Q.compute_lowerbound())
    for nm, nd in dict(X=X, A=A, C=C, alpha=alpha, gamma=gamma).items():
        for i, ui in enumerate(nd.u):
            out['%s_u%d_rot' % (nm, i)] = np.array(ui)
    Ls = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(5):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
            R.rotate(maxiter=10)
    out['y'], out['mask'], out['x0'], out['c0'] = y, mask, x0, c0
    out['L'] = np.array(Ls)
    out['L_final'] = np.array(Q.compute_lowerbound())
    out['tau_u0'] = np.asarray(tau.u[0])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: v for k, v in out.items() if k.startswith('L')})


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


def gaussian_gamma_case(name):
    """GaussianGamma joint nodes, as the mean parent of GaussianARD and stand-alone (vector)."""
    _shared_case(name, 'make_gaussian_gamma_inputs', 'run_gaussian_gamma_cases', 808,
                 ('a_L', 'b_L', 'c_L'))


def hierarchical_wishart_case(name):
    """Wishart(n, V) with V a Wishart node (the message of wishart.py:142-150)."""
    _shared_case(name, 'make_hierarchical_wishart_inputs', 'run_hierarchical_wishart_case', 1717,
                 ('hw_L',))


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


def masked_pca_erasures_case(name):
    """PCA with missing values at the edges of the mask logic: plates and dimensions without any
    observation (ignored plates of X and W, node.py:457-526)."""
    _shared_case(name, 'make_erasure_inputs', 'run_erasure_cases', 6021, ('e0_L', 'e1_L', 'e2_L'))


def order_probes_case(name):
    """Node-level update sequences out of constructor order on the PCA, mixture and state-space
    models (vmp.py:132-172: VB.update(*nodes))."""
    _shared_case(name, 'make_order_probe_inputs', 'run_order_probes', 777,
                 ('pca_L', 'mpca_L', 'gmm_L', 'lssm_L'))


def hyper_probes_case(name):
    """The PCA, missing-data PCA, mixture and state-space models with non-default priors, prior
    means / precisions and fixed parameters."""
    _shared_case(name, 'make_hyper_probe_inputs', 'run_hyper_probes', 4243,
                 ('pca_L', 'mpca_L', 'gmm_L', 'lssm_L', 'lssmnu_L'))


def reobserve_case(name):
    """Y.observe(new data) between updates of the PCA model."""
    import bayespy.nodes
    from bayespy.inference import VB
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))
    import models
    g = models.make_reobserve_inputs(np.random.RandomState(31337))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = models.run_reobserve_case(bayespy.nodes, VB, g)
        res.update(models.run_reobserve_lssm_case(bayespy.nodes, VB, g))
        res.update(models.run_reinitialise_case(bayespy.nodes, VB, g))
    out = {'in_' + k: v for k, v in g.items()}
    out.update({k: np.array(v) for k, v in res.items()})
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, out['L'], out['L_mid'], out['lssm_L'], out['lssm_L_mid'], out['lssm_L_c'], out['ri_steps'], out['ri_L'])


def constant_parents_case(name):
    """Numeric arrays as parents of SumMultiply (DeltaMoments constants, dot.py:186-197)."""
    _shared_case(name, 'make_constant_parent_inputs', 'run_constant_parent_cases', 8086,
                 ('lr_L', 'tp_L'))


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
            if fn == 'lssm_masked_wide_cases':
                lssm_masked_cases(nm, wide=True)
                continue
            globals()[fn](nm)
        return
    quickstart_case('quickstart_n10', N=10, n_iter=4, seed=1)
    quickstart_case('quickstart_n1000', N=1000, n_iter=6, seed=1)
    pca_case('pca_n500_d6_k3', N=500, D=6, K=3, n_iter=5, seed=7)
    pca_case('pca_n777_d20_k5', N=777, D=20, K=5, n_iter=5, seed=8)
    pca_case('pca_n2048_d128_k32', N=2048, D=128, K=32, n_iter=4, seed=9)
    pca_case('pca_n4000_d64_k16', N=4000, D=64, K=16, n_iter=4, seed=10)
    pca_seeded_case('pca_seeded_n100000_d128_k32')
    pca_mean_case('pca_prior_mean')
    gmm_case('gmm_n400_d3_k4', N=400, D=3, K=4, n_iter=5, seed=11)
    gmm_case('gmm_n3000_d8_k16', N=3000, D=8, K=16, n_iter=4, seed=12)
    utils_cases('utils_known_answers')
    small_model_cases('small_models')
    lssm_cases('lssm')
    lssm_wide_state_cases('lssm_wide_states')
    lssm_masked_cases('lssm_masked')
    lssm_masked_cases('lssm_masked_wide', wide=True)
    lssm_prior_init_case('lssm_prior_init')
    lssm_rotation_cases('lssm_rotations')
    lssm_masked_rotation_case('lssm_masked_rotations')
    rotation_cases('rotations')
    pca_doctest_case('pca_doctest')
    svi_case('svi_gmm')
    multinomial_case('multinomial')
    summultiply_cases('summultiply')
    mixture_ard_case('mixture_ard')
    parameter_api_case('parameter_api')
    count_nodes_case('count_nodes')
    gaussian_gamma_case('gaussian_gamma')
    hierarchical_wishart_case('hierarchical_wishart')
    plate_nodes_case('plate_nodes')
    markov_chain_case('markov_chains')
    slice_nodes_case('slice_nodes')
    switching_case('switching_lssm')
    varying_case('varying_lssm')
    concat_gaussian_case('concat_gaussian')
    default_ndim_case('default_ndim')
    masked_pca_case('masked_pca')
    masked_pca_erasures_case('masked_pca_erasures')
    order_probes_case('order_probes')
    hyper_probes_case('hyper_probes')
    reobserve_case('reobserve')
    constant_parents_case('constant_parents')
    bmm_doctest_case('bmm_doctest')
    gmm_doctest_case('gmm_doctest')
    inference_doctest_case('inference_doctest')
    lssm_doctest_case('lssm_doctest')


if __name__ == '__main__':
    main()
