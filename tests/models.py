"""Model builders shared by the tests (written against the bayespy API surface)."""
import numpy as np


def build_pca(nodes_mod, vb_cls, y, x0, K, a0=1e-2, b0=1e-2, shard=False, **vb_kwargs):
    """bayespy/demos/pca.py:22-61 with X initialised from ``x0`` (N,K) and Y fully observed.
    ``shard``: the arrays are this rank's part of the observation plate (X.shard(-1))."""
    D, N = y.shape
    alpha = nodes_mod.Gamma(a0, b0, plates=(K,), name='alpha')
    W = nodes_mod.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes_mod.GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    if shard:
        X.shard(-1)
    F = nodes_mod.SumMultiply('i,i', W, X, name='F')
    tau = nodes_mod.Gamma(a0, b0, name='tau')
    Y = nodes_mod.GaussianARD(F, tau, name='Y')
    if x0 is not None:
        X.initialize_from_value(np.asarray(x0)[None, :, :])
    Y.observe(y)
    Q = vb_cls(Y, F, W, X, tau, alpha, **vb_kwargs)
    Q.ignore_bound_checks = True
    return Q


def run_rotation_sequence(Q, K, transformations):
    """The sequence recorded in tests/golden/rotations.npz (oracle/make_golden.py
    rotation_cases): two plain iterations, one stand-alone rotation, then six iterations with
    the rotation as the VB callback (demos/pca.py:85-94).  Returns a dict of results."""
    import warnings
    rotW = transformations.RotateGaussianARD(Q['W'], Q['alpha'])
    rotX = transformations.RotateGaussianARD(Q['X'])
    R = transformations.RotationOptimizer(rotW, rotX, K)
    out = {}
    Q.update(repeat=2, verbose=False)
    out['L_before'] = Q.compute_lowerbound()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        R.rotate()
    out['L_after'] = Q.compute_lowerbound()
    out['W_u0_rot'] = np.asarray(Q['W'].u[0])
    out['X_u0_rot'] = np.asarray(Q['X'].u[0])
    out['alpha_u0_rot'] = np.asarray(Q['alpha'].u[0])
    Q.callback = R.rotate
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=6, verbose=False)
    out['L'] = np.array(Q.L[:Q.iter])
    out['W_u0'], out['X_u0'] = np.asarray(Q['W'].u[0]), np.asarray(Q['X'].u[0])
    out['tau_u0'], out['alpha_u0'] = np.asarray(Q['tau'].u[0]), np.asarray(Q['alpha'].u[0])
    return out


def check_rotation_results(res, g, tag):
    """Rotations go through a truncated nonlinear CG (10 iterations with line searches) on
    the host.  Measured on this case: a 1e-12 relative perturbation of the K x K statistics
    moves the bound of the 7th rotated iteration by 1e-3 (one component is being pruned by
    ARD there and the truncated CG stops on a different iterate), so the stand-alone rotation
    and the first iterations are checked tightly and the tail of the trace loosely."""
    np.testing.assert_allclose(res['L_before'], g[tag + '_L_before'], rtol=1e-10)
    np.testing.assert_allclose(res['L_after'], g[tag + '_L_after'], rtol=1e-7)
    assert res['L_after'] > res['L_before']
    for k in ('W_u0_rot', 'X_u0_rot', 'alpha_u0_rot'):
        np.testing.assert_allclose(res[k], g[tag + '_' + k], rtol=1e-6, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(res['L'][:5], g[tag + '_L'][:5], rtol=1e-7)
    np.testing.assert_allclose(res['L'], g[tag + '_L'], rtol=1e-4)
    assert np.all(np.diff(res['L']) > 0)
    # after six rotated iterations the weakly determined directions (ARD-pruned components)
    # have drifted by up to ~1 % between two runs of the same truncated CG
    np.testing.assert_allclose(res['tau_u0'], g[tag + '_tau_u0'], rtol=1e-4)
    np.testing.assert_allclose(res['alpha_u0'], g[tag + '_alpha_u0'], rtol=5e-2)
    # ... so compare what the rotation leaves invariant: the reconstruction <W><X>^T
    def recon(w, x):
        return np.einsum('dik,ink->dn', w, x)
    np.testing.assert_allclose(recon(res['W_u0'], res['X_u0']),
                               recon(g[tag + '_W_u0'], g[tag + '_X_u0']), rtol=1e-2, atol=5e-3)


def run_pca_doctest(nodes_mod, vb_cls, transformations, g, attach=None, **vb_kwargs):
    """doc/source/examples/pca.rst:26-118 on this framework: same model, same statements,
    the reference's random initial C injected as a value (tests/golden/pca_doctest.npz)."""
    import warnings
    GaussianARD, Gamma, SumMultiply = nodes_mod.GaussianARD, nodes_mod.Gamma, nodes_mod.SumMultiply
    y = g['y']
    M, N = y.shape
    D = 10
    X = GaussianARD(0, 1, plates=(1, N), shape=(D,), name='X')
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    C = GaussianARD(0, alpha, plates=(M, 1), shape=(D,), name='C')
    F = SumMultiply('d,d->', X, C, name='F')
    tau = Gamma(1e-5, 1e-5, name='tau')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, X, C, alpha, tau, **vb_kwargs)
    if attach is not None:
        attach(Q)
    C.initialize_from_value(g['C_init'])
    R = transformations.RotationOptimizer(transformations.RotateGaussianARD(X),
                                          transformations.RotateGaussianARD(C, alpha), D)
    Q.set_callback(R.rotate)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=1000, verbose=False)
    return Q, dict(X=X, C=C, F=F, alpha=alpha, tau=tau)


def check_pca_doctest(Q, nd, g):
    """The doctest prints "Iteration 1: loglike=-2.33...e+03" and a converged
    "loglike=6.500...e+02" (650.0367 after 23 iterations in the reference run recorded in the
    golden file).  The rotated iterations amplify round-off (check_rotation_results), so the
    first value and the first iterations are compared tightly, the converged bound to 1e-3."""
    L = np.array(Q.L[:Q.iter])
    assert ('%e' % L[0]).startswith('-2.33') and ('%e' % L[0]).endswith('e+03')
    np.testing.assert_allclose(L[:7], g['L'][:7], rtol=1e-8)
    # the stopping iteration of a tol=1e-5 test on a slowly rising tail is itself sensitive
    assert Q.converged and 15 <= Q.iter <= 3 * int(g['n_iter']), Q.iter
    assert np.all(np.diff(L) > 0)
    np.testing.assert_allclose(L[-1], g['L'][-1], rtol=1e-3)
    np.testing.assert_allclose(nd['tau'].u[0], g['tau_u0'], rtol=2e-3)
    # two components survive ARD, as in the data (latent dimensionality two)
    a = np.sort(np.ravel(nd['alpha'].u[0]))
    assert a[1] < 10 and a[2] > 100


def make_parameter_api_inputs(rs):
    """Seeded inputs of run_parameter_api_cases (stored in tests/golden/parameter_api.npz)."""
    def spd(d):
        a = rs.normal(size=(d, d))
        return a @ a.T + d * np.eye(d)
    D = 3
    g = dict(
        gam_a=rs.rand(D) + 0.5, gam_b=rs.rand(D) + 0.5, gam_mu=rs.normal(size=D),
        gam_y=rs.normal(size=D), gam_pa=rs.rand(D) + 0.5, gam_pb=rs.rand(D) + 0.5,
        gam_x=rs.rand(D) + 0.2,
        gs_mu=rs.normal(size=D), gs_L=spd(D), gs_mu0=rs.normal(size=D), gs_L0=spd(D),
        gs_V=spd(D), gs_y=rs.normal(size=(5, D)), gs_x=rs.normal(size=D),
        ard_mu=rs.normal(size=(2, D)), ard_al=rs.rand(2, D) + 0.5, ard_m0=rs.normal(size=(2, D)),
        ard_a0=rs.rand(2, D) + 0.5, ard_y=rs.normal(size=(4, 2, D)),
        dir_a=rs.rand(4) + 0.5, dir_a0=rs.rand(4) + 0.5, dir_z=rs.randint(4, size=25),
    )
    # small PCA for the collapsed optimisation (demos/collapsed_cg.py:28-66)
    M, N, K = 6, 30, 3
    g['cg_y'] = (rs.normal(size=(M, 1, K - 1)) * rs.normal(size=(1, N, K - 1))).sum(-1) \
        + 0.1 * rs.normal(size=(M, N))
    g['cg_w0'] = rs.normal(size=(M, 1, K))
    # two-cluster 1-D mixture for the annealing run (demos/annealing.py:33-90)
    z = rs.rand(60) < 0.3
    g['an_y'] = np.where(z, 4.0, -4.0) + rs.normal(size=60)
    return g


def run_parameter_api_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    """Natural-parameter access, Riemannian / Euclidean gradients, densities, deterministic
    annealing, collapsed conjugate gradients and pattern search, statement for statement the
    same on the reference and on this framework (the reference's own checks of these:
    tests/test_annealing.py:34-110, nodes/tests/test_gamma.py:160-260,
    test_gaussian.py:1255-1420, test_categorical.py:227-250; demos/collapsed_cg.py,
    demos/pattern_search.py, demos/annealing.py)."""
    import warnings
    N_ = nodes_mod
    out = {}

    def record(tag, node, Q, x=None):
        out[tag + '_phi'] = [np.array(p) for p in node.phi]
        rg = node.get_riemannian_gradient()
        out[tag + '_rg'] = [np.array(r) for r in rg]
        out[tag + '_g'] = [np.array(v) for v in node.get_gradient(rg)]
        out[tag + '_L'] = Q.compute_lowerbound(ignore_masked=False)
        out[tag + '_u'] = [np.array(v) for v in node.u]
        if x is not None:
            out[tag + '_logpdf'] = np.array(node.logpdf(x))

    # 1. annealed scalar Gaussian (test_annealing.py:34-110)
    X = N_.GaussianARD(3, 4, name='X')
    X.initialize_from_parameters(-1, 6)
    Q = vb_cls(X, **vb_kwargs)
    Q.set_annealing(0.1)
    record('an0', X, Q, x=0.3)
    p = X.get_parameters()
    X.set_parameters([p[0] + 0.25, p[1] - 0.5])
    record('an1', X, Q)
    X.update()
    record('an2', X, Q)

    # 2. Gamma with an observed Gaussian child (test_gamma.py:200-260)
    tau = N_.Gamma(g['gam_a'], g['gam_b'], name='tau')
    Y = N_.GaussianARD(g['gam_mu'], tau, name='Y')
    Y.observe(g['gam_y'])
    Q = vb_cls(Y, tau, **vb_kwargs)
    tau.initialize_from_parameters(g['gam_pa'], g['gam_pb'])
    record('gam', tau, Q, x=g['gam_x'])

    # 3. Gaussian with a full precision and Gaussian observations (test_gaussian.py:1303-1350)
    X = N_.Gaussian(g['gs_mu'], g['gs_L'], name='X')
    Y = N_.Gaussian(X, g['gs_V'], plates=(5,), name='Y')
    Y.observe(g['gs_y'])
    X.initialize_from_parameters(g['gs_mu0'], g['gs_L0'])
    Q = vb_cls(Y, X, **vb_kwargs)
    record('gs', X, Q, x=g['gs_x'])

    # 4. vector GaussianARD with plates and observations (test_gaussian.py:1355-1420)
    X = N_.GaussianARD(g['ard_mu'], g['ard_al'], shape=(3,), name='X')
    Y = N_.GaussianARD(X, 2.0, shape=(3,), plates=(4, 2), name='Y')
    Y.observe(g['ard_y'])
    X.initialize_from_parameters(g['ard_m0'], g['ard_a0'])
    Q = vb_cls(Y, X, **vb_kwargs)
    record('ard', X, Q)

    # 5. Dirichlet with categorical observations; categorical under a Gamma mixture
    #    (test_categorical.py:227-250)
    pi = N_.Dirichlet(g['dir_a'], name='pi')
    Z = N_.Categorical(pi, plates=(25,), name='Z')
    Z.observe(g['dir_z'])
    pi.initialize_from_parameters(g['dir_a0'])
    Q = vb_cls(Z, pi, **vb_kwargs)
    record('dir', pi, Q)
    Z = N_.Categorical([[0.3, 0.5, 0.2], [0.1, 0.6, 0.3]], name='Z')
    Y = N_.Mixture(Z, N_.Gamma, [2, 3, 4], [5, 6, 7], name='Y')
    Y.observe([4.2, 0.2])
    Q = vb_cls(Y, Z, **vb_kwargs)
    Z.set_parameters([np.log([[2, 3, 7], [0.1, 3, 1]])])
    record('cat', Z, Q)

    # 6. deterministic annealing of a two-cluster mixture (demos/annealing.py:33-90)
    mu = N_.GaussianARD(0, 1, plates=(2,), name='means')
    Z = N_.Categorical([0.3, 0.7], plates=(60,), name='classes')
    Y = N_.Mixture(Z, N_.GaussianARD, mu, 1, name='observations')
    Y.observe(g['an_y'])
    mu.initialize_from_value([0, 6])
    Q = vb_cls(Y, Z, mu, **vb_kwargs)
    Q.ignore_bound_checks = True
    for beta in (0.05, 0.2, 0.5, 1.0):
        Q.set_annealing(beta)
        Q.update(repeat=3, verbose=False)
    out['mix_L'] = np.array(Q.L[:Q.iter])
    out['mix_mu'] = np.array(mu.u[0])

    # 7. collapsed Riemannian conjugate gradients, plain gradient ascent and pattern search
    #    on a small PCA model (demos/collapsed_cg.py:28-66, demos/pattern_search.py:40-80)
    y = g['cg_y']
    M, N = y.shape
    K = g['cg_w0'].shape[-1]
    alpha = N_.Gamma(1e-3, 1e-3, plates=(K,), name='alpha')
    W = N_.GaussianARD(0, alpha, plates=(M, 1), shape=(K,), name='W')
    X = N_.GaussianARD(0, 1, plates=(1, N), shape=(K,), name='X')
    tau = N_.Gamma(1e-3, 1e-3, name='tau')
    W.initialize_from_value(g['cg_w0'])
    F = N_.SumMultiply('d,d->', W, X)
    Y = N_.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, X, W, alpha, tau, **vb_kwargs)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Q.update(repeat=1, verbose=False)
        Q.optimize(W, tau, maxiter=5, collapsed=[X, alpha], verbose=False)
        out['cg_W'] = np.array(W.u[0])
        Q.optimize(W, X, riemannian=False, method='gradient', maxiter=3, verbose=False)
        Q.pattern_search(W, tau, maxiter=3, collapsed=[X, alpha])
        Q.update(repeat=2, verbose=False)
    out['cg_L'] = np.array(Q.L[:Q.iter])
    out['cg_tau'] = np.array(tau.u[0])
    return out


def make_count_node_inputs(rs):
    """Seeded inputs of run_count_node_cases (tests/golden/count_nodes.npz)."""
    g = {}
    g['bin_alpha'] = np.array([[1.0, 2.0], [0.5, 0.5], [3.0, 1.0]])
    g['bin_n'] = np.array([10, 7, 12])
    g['bin_x'] = rs.binomial(g['bin_n'], [0.3, 0.6, 0.8], size=(4, 3))
    g['cmp_z'] = rs.randint(2, size=(6, 1))
    g['poi_a'] = rs.rand(3) + 0.5
    g['poi_b'] = rs.rand(3) + 0.5
    g['poi_x'] = rs.poisson([2.0, 7.0, 0.5], size=(20, 3))
    # Bernoulli mixture of doc/source/examples/bmm.rst:40-75
    p = np.array([[0.1, 0.9] * 5, [0.1] * 5 + [0.9] * 5, [0.9] * 5 + [0.1] * 5])
    z = rs.randint(3, size=100)
    g['bmm_x'] = (rs.rand(100, 10) < p[z]).astype(np.int64)
    g['bmm_p0'] = rs.beta(0.5, 0.5, size=(10, 10)).clip(1e-3, 1 - 1e-3)
    # Poisson mixture
    lab = rs.randint(3, size=80)
    g['pmm_x'] = rs.poisson(np.array([1.0, 6.0, 15.0])[lab])
    g['pmm_lab0'] = rs.randint(3, size=80)
    # sums of Gaussians (add.py:19-33 and test_add.py)
    g['add_y'] = rs.normal(size=(5, 3, 2))
    g['add_b0'] = rs.normal(size=(3, 2))
    g['adds_y'] = rs.normal(size=(7,))
    return g


def run_count_node_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    """Beta / Bernoulli / Binomial / Poisson / Complement / Add and mixtures of count
    distributions, the same statements on the reference and on this framework
    (binomial.py:166-195, bernoulli.py:44-62, poisson.py:122-150, beta.py:112-214,
    add.py:15-154, doc/source/examples/bmm.rst:40-95)."""
    N_ = nodes_mod
    out = {}

    def trace(tag, Q, n, track):
        Q.ignore_bound_checks = True
        Q.update(repeat=n, verbose=False)
        out[tag + '_L'] = np.array(Q.L[:n])
        for nm, nd in track.items():
            out['%s_%s_u' % (tag, nm)] = [np.array(v) for v in nd.u]
            out['%s_%s_Lterm' % (tag, nm)] = np.array(Q.l[nd][:n])

    # 1. the doctest of bernoulli.py:55-60
    p = N_.Beta([1e-3, 1e-3], name='p')
    z = N_.Bernoulli(p, plates=(10,), name='z')
    z.observe([0, 1, 1, 1, 0, 1, 1, 1, 0, 1])
    trace('bern', vb_cls(z, p, **vb_kwargs), 2, dict(p=p))
    out['bern_phi'] = [np.array(v) for v in p.phi]
    out['bern_logpdf'] = np.array(p.logpdf(np.array([0.2, 0.7, 0.9])[:, None]))

    # 2. binomial observations with a different number of trials per plate, a latent
    #    binomial under the same probabilities, and the complement of the probability
    p = N_.Beta(g['bin_alpha'], name='p')
    x = N_.Binomial(g['bin_n'], p, plates=(4, 3), name='x')
    x.observe(g['bin_x'])
    h = N_.Binomial(5, p, name='h')
    c = p.complement()
    zc = N_.Bernoulli(c, plates=(6, 3), name='zc')
    zc.observe(np.broadcast_to(g['cmp_z'], (6, 3)))
    trace('bin', vb_cls(x, zc, h, p, **vb_kwargs), 2, dict(p=p, h=h, x=x, zc=zc))
    out['bin_c_u'] = [np.array(v) for v in c.get_moments()]

    # 3. Poisson counts with Gamma rates
    lam = N_.Gamma(g['poi_a'], g['poi_b'], name='lam')
    x = N_.Poisson(lam, plates=(20, 3), name='x')
    x.observe(g['poi_x'])
    xl = N_.Poisson(lam, name='xl')
    trace('poi', vb_cls(x, xl, lam, **vb_kwargs), 2, dict(lam=lam, xl=xl, x=x))

    # 4. Bernoulli mixture (bmm.rst)
    N, D = g['bmm_x'].shape
    K = g['bmm_p0'].shape[1]
    R = N_.Dirichlet(K * [1e-5], name='R')
    Z = N_.Categorical(R, plates=(N, 1), name='Z')
    P = N_.Beta([0.5, 0.5], plates=(D, K), name='P')
    X = N_.Mixture(Z, N_.Bernoulli, P, name='X')
    Q = vb_cls(Z, R, X, P, **vb_kwargs)
    P.initialize_from_value(g['bmm_p0'])
    X.observe(g['bmm_x'])
    trace('bmm', Q, 6, dict(R=R, P=P, Z=Z))

    # 5. Poisson mixture
    N = len(g['pmm_x'])
    al = N_.Dirichlet(np.ones(3), name='al')
    Z = N_.Categorical(al, plates=(N,), name='Z')
    lam = N_.Gamma(1.0, 0.2, plates=(3,), name='lam')
    X = N_.Mixture(Z, N_.Poisson, lam, name='X')
    Z.initialize_from_value(g['pmm_lab0'])
    X.observe(g['pmm_x'])
    trace('pmm', vb_cls(X, lam, Z, al, **vb_kwargs), 5, dict(lam=lam, Z=Z, al=al))

    # 6. sums of independent Gaussian factors as the mean of observations
    a = N_.GaussianARD(0, 1e-1, shape=(2,), plates=(5, 1), name='a')
    b = N_.GaussianARD(0, 1e-1, shape=(2,), plates=(1, 3), name='b')
    c = N_.Gaussian(np.array([0.5, -0.5]), np.array([[2.0, 0.3], [0.3, 1.0]]), name='c')
    s = N_.Add(a, b, c, name='s')
    tau = N_.Gamma(1e-2, 1e-2, name='tau')
    Y = N_.GaussianARD(s, tau, name='Y')
    b.initialize_from_value(g['add_b0'][None])
    Y.observe(g['add_y'])
    trace('add', vb_cls(Y, a, b, c, tau, **vb_kwargs), 4, dict(a=a, b=b, c=c, tau=tau))
    out['add_s_u'] = [np.array(v) for v in s.get_moments()]
    m1 = N_.GaussianARD(0, 1, name='m1')
    m2 = N_.GaussianARD(1, 2, plates=(7,), name='m2')
    Ys = N_.GaussianARD(N_.Add(m1, m2, 0.25), 3.0, name='Ys')
    Ys.observe(g['adds_y'])
    trace('adds', vb_cls(Ys, m1, m2, **vb_kwargs), 3, dict(m1=m1, m2=m2))
    return out


def make_plate_node_inputs(rs):
    """Seeded inputs of run_plate_node_cases (tests/golden/plate_nodes.npz)."""
    g = {}
    g['tk_y'] = rs.normal(size=(6,)) * np.array([1.0, 1.0, 0.5, 0.5, 1.0, 2.0])
    g['tk2_y'] = rs.normal(size=(5, 2, 2, 4, 2))
    g['tk2_mask'] = rs.rand(5, 2, 2, 4) < 0.8
    g['cc_y'] = rs.normal(size=(4, 5)) + np.array([0.0, 0.0, 0.0, 3.0, 3.0])
    g['cc_mask'] = rs.rand(4, 5) < 0.85
    g['cc2_y'] = rs.normal(size=(7, 3, 2))
    lab = rs.randint(3, size=40)
    g['gt_y'] = np.array([[-3.0, 0.0], [0.0, 3.0], [3.0, -1.0]])[lab] + 0.5 * rs.normal(size=(40, 2))
    g['gt_lab0'] = rs.randint(3, size=40)
    g['gt2_y'] = rs.normal(size=(6, 4)) + np.array([0.0, 1.0, -2.0, 4.0])
    return g


def run_plate_node_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    """Take, Concatenate and Gate inside small models, the same statements on the reference
    and on this framework (take.py:34-39 doctest, nodes/tests/test_take.py,
    test_concatenate.py, test_gate.py)."""
    N_ = nodes_mod
    out = {}

    def trace(tag, Q, n, track):
        Q.ignore_bound_checks = True
        Q.update(repeat=n, verbose=False)
        out[tag + '_L'] = np.array(Q.L[:n])
        for nm, nd in track.items():
            out['%s_%s_u' % (tag, nm)] = [np.array(v) for v in nd.get_moments()]

    # 1. the doctest of take.py:34-39, then the taken precisions in a model
    alpha = N_.Gamma([1, 2, 3], [1, 1, 1], name='alpha')
    x = N_.Take(alpha, [1, 1, 2, 2, 1, 0], name='x')
    out['tk_doc'] = np.array(x.get_moments()[0])
    Y = N_.GaussianARD(0, x, name='Y')
    Y.observe(g['tk_y'])
    trace('tk', vb_cls(Y, alpha, **vb_kwargs), 2, dict(alpha=alpha, x=x))

    # 2. an index array with two axes on a non-last plate axis of a vector Gaussian, masked data
    mu = N_.GaussianARD(0, 1e-1, shape=(2,), plates=(3, 4), name='mu')
    t = N_.Take(mu, [[0, 2], [1, 1]], plate_axis=-2, name='t')
    out['tk2_plates'] = np.array(t.plates)
    Y = N_.GaussianARD(t, 2.0, shape=(2,), plates=(5, 2, 2, 4), name='Y')
    Y.observe(g['tk2_y'], mask=g['tk2_mask'])
    trace('tk2', vb_cls(Y, mu, **vb_kwargs), 2, dict(mu=mu, t=t))

    # 3. concatenation of scalar Gaussians along the last plate axis; the mask is constant
    #    along that axis (the reference's Concatenate cannot split a mask, concatenate.py:118-126)
    a = N_.GaussianARD(0, 1, plates=(3,), name='a')
    b = N_.GaussianARD(1, 2, plates=(2,), name='b')
    c = N_.Concatenate(a, b, name='c')
    tau = N_.Gamma(1e-2, 1e-2, name='tau')
    Y = N_.GaussianARD(c, tau, plates=(4, 5), name='Y')
    Y.observe(g['cc_y'], mask=g['cc_mask'][:, :1])
    trace('cc', vb_cls(Y, a, b, tau, **vb_kwargs), 3, dict(a=a, b=b, c=c, tau=tau))
    #    ... and of vector Gaussians along the second-last plate axis
    a = N_.GaussianARD(0, 1, shape=(2,), plates=(2, 1), name='a2')
    b = N_.GaussianARD(0.5, 1, shape=(2,), plates=(1, 1), name='b2')
    c = N_.Concatenate(a, b, axis=-2, name='c2')
    out['cc2_plates'] = np.array(c.plates)
    Y = N_.GaussianARD(c, 1.5, shape=(2,), plates=(7, 3, 1), name='Y')
    Y.observe(g['cc2_y'][:, :, None, :])
    trace('cc2', vb_cls(Y, a, b, **vb_kwargs), 2, dict(a=a, b=b, c=c))

    # 4. gated cluster means: a mixture written with Gate instead of Mixture
    N, K = g['gt_y'].shape[0], 3
    al = N_.Dirichlet(np.ones(K), name='al')
    Z = N_.Categorical(al, plates=(N,), name='Z')
    X = N_.GaussianARD(0, 1e-2, shape=(2,), plates=(K,), name='X')
    G = N_.Gate(Z, X, name='G')
    lam = N_.Gamma(1e-1, 1e-1, name='lam')
    Y = N_.GaussianARD(G, lam, name='Y')
    Z.initialize_from_value(g['gt_lab0'])
    Y.observe(g['gt_y'])
    trace('gt', vb_cls(Y, X, lam, Z, al, **vb_kwargs), 5, dict(X=X, lam=lam, Z=Z, G=G))
    #    ... gating a non-default plate axis with fixed class labels
    X = N_.GaussianARD(0, 1, plates=(3, 4), name='X2')
    G = N_.Gate([[1], [0], [2], [1], [1], [0]], X, gated_plate=-2, name='G2')
    out['gt2_plates'] = np.array(G.plates)
    Y = N_.GaussianARD(G, 2.0, plates=(6, 4), name='Y')
    Y.observe(g['gt2_y'])
    trace('gt2', vb_cls(Y, X, **vb_kwargs), 2, dict(X=X, G=G))
    return out


def run_markov_chain_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    """Categorical Markov chains / hidden Markov models, the same statements on the reference
    and on this framework: the two models of doc/source/examples/hmm.rst (known answers
    "Iteration 1: loglike=-1.095883e+02" and "-9.963054e+02 ... Iteration 8: -9.235053e+02"),
    a batch of chains with time-varying transition priors, and a chain used through Gate."""
    N_ = nodes_mod
    out = {}

    def trace(tag, Q, n, track, tol=None):
        Q.update(repeat=n, verbose=False, **({} if tol is None else dict(tol=tol)))
        out[tag + '_L'] = np.array(Q.L[:Q.iter])
        for nm, nd in track.items():
            out['%s_%s_u' % (tag, nm)] = [np.array(v) for v in nd.get_moments()]

    # 1. discrete HMM with known parameters (hmm.rst:35-94)
    N = len(g['hmm1_activity'])
    Z = N_.CategoricalMarkovChain([0.6, 0.4], [[0.7, 0.3], [0.4, 0.6]], states=N, name='Z')
    P = [[0.1, 0.4, 0.5], [0.6, 0.3, 0.1]]
    Y = N_.Mixture(Z, N_.Categorical, P, name='Y')
    Y.observe(g['hmm1_activity'])
    Q = vb_cls(Y, Z, **vb_kwargs)
    trace('hmm1', Q, 1, dict(Z=Z))
    out['hmm1_g'] = np.array(Z.g)

    # 2. Gaussian HMM with unknown initial state and transition probabilities (hmm.rst:157-295)
    y = g['hmm2_y']
    K = 3
    a0 = N_.Dirichlet(1e-3 * np.ones(K), name='a0')
    A = N_.Dirichlet(1e-3 * np.ones((K, K)), name='A')
    Z = N_.CategoricalMarkovChain(a0, A, states=len(y), name='Z')
    mu = np.array([[0, 0], [3, 4], [6, 0]])
    Lambda = 2.0 ** (-2) * np.identity(2)
    Y = N_.Mixture(Z, N_.Gaussian, mu, Lambda, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, Z, A, a0, **vb_kwargs)
    trace('hmm2', Q, 1000, dict(Z=Z, A=A, a0=a0))

    # 3. a batch of chains, transition prior varying in time, latent emission parameters
    yb = g['hmm3_y']                                  # (B, T)
    B, T = yb.shape
    K = 4
    a0 = N_.Dirichlet(np.ones(K), plates=(B,), name='a0')
    A = N_.Dirichlet(g['hmm3_prior'], name='A')      # plates (T-1, K)
    Z = N_.CategoricalMarkovChain(a0, A, name='Z')
    out['hmm3_plates'] = np.array(Z.plates + (Z.dims[1][0],))
    m = N_.GaussianARD(0, 1e-2, plates=(K,), name='m')
    t = N_.Gamma(1e-1, 1e-1, plates=(K,), name='t')
    Y = N_.Mixture(Z, N_.GaussianARD, m, t, name='Y')
    Z.initialize_from_value(g['hmm3_z0'])
    Y.observe(yb)
    Q = vb_cls(Y, m, t, Z, A, a0, **vb_kwargs)
    Q.ignore_bound_checks = True
    trace('hmm3', Q, 4, dict(Z=Z, A=A, a0=a0, m=m, t=t))

    # 4. a chain that gates Gaussian means
    yg = g['hmm4_y']
    T = len(yg)
    Z = N_.CategoricalMarkovChain([0.5, 0.5], [[0.9, 0.1], [0.2, 0.8]], states=T, name='Z')
    X = N_.GaussianARD(0, 1e-1, plates=(2,), name='X')
    Gt = N_.Gate(Z, X, name='G')
    Y = N_.GaussianARD(Gt, 1.0, name='Y')
    X.initialize_from_value(np.array([-1.0, 1.0]))
    Y.observe(yg)
    Q = vb_cls(Y, Z, X, **vb_kwargs)
    Q.ignore_bound_checks = True
    trace('hmm4', Q, 3, dict(Z=Z, X=X))
    return out


def make_slice_inputs(rs):
    """Seeded inputs of run_slice_cases (tests/golden/slice_nodes.npz)."""
    return dict(sl_y1=rs.normal(size=(2, 3, 2)), sl_y2=rs.normal(size=(4, 2)),
                sl_y3=rs.normal(size=(4, 3, 3, 2)), sl_mask3=rs.rand(4, 3, 3) < 0.8,
                sl_y4=rs.normal(size=(5,)),
                ch_y=np.array([0.5, 9.0, 21.0, 19.0, -1.0, 11.0]) + 0.3 * rs.normal(size=6))


def run_slice_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    """Plate indexing ``X[...]`` (node.py:761-763, :868-1130) and ``Choose`` (gate.py:207-250;
    its doctest gives [0, 0, 20, 10]), the same statements on both sides."""
    N_ = nodes_mod
    out = {}
    X = N_.GaussianARD(0, 1e-1, shape=(2,), plates=(4, 5), name='X')
    a, b, c = X[1:3, ::2], X[..., 0], X[:, None, 1:4]
    out['sl_plates'] = np.array(a.plates + b.plates + c.plates)
    tau = N_.Gamma(1e-1, 1e-1, name='tau')
    Y1 = N_.GaussianARD(a, tau, name='Y1')
    Y2 = N_.GaussianARD(b, 1.0, shape=(2,), name='Y2')
    Y3 = N_.GaussianARD(c, 2.0, shape=(2,), plates=(4, 3, 3), name='Y3')
    Y1.observe(g['sl_y1'])
    Y2.observe(g['sl_y2'])
    Y3.observe(g['sl_y3'], mask=g['sl_mask3'])
    t = N_.Gamma([1.0, 2.0, 3.0, 4.0, 5.0], 1.0, name='t')
    Y4 = N_.GaussianARD(0, t[-2], plates=(5,), name='Y4')          # an integer index
    Y4.observe(g['sl_y4'])
    Q = vb_cls(Y1, Y2, Y3, Y4, X, tau, t, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    out['sl_L'] = np.array(Q.L[:3])
    for nm, nd in dict(X=X, tau=tau, t=t, a=a, c=c).items():
        out['sl_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]

    x0 = N_.GaussianARD(0, 1, name='x0')
    x1 = N_.GaussianARD(10, 1, name='x1')
    x2 = N_.GaussianARD(20, 1, name='x2')
    x = N_.Choose([0, 0, 2, 1], x0, x1, x2)
    out['ch_doc'] = np.array(x.get_moments()[0])
    Z = N_.Categorical([0.3, 0.3, 0.4], plates=(6,), name='Z')
    xz = N_.Choose(Z, x0, x1, x2)
    Y = N_.GaussianARD(xz, 1.0, name='Y')
    Y.observe(g['ch_y'])
    Q = vb_cls(Y, Z, x0, x1, x2, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    out['ch_L'] = np.array(Q.L[:3])
    for nm, nd in dict(Z=Z, x0=x0, x1=x1, x2=x2).items():
        out['ch_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]
    return out


def make_switching_inputs(rs):
    """Seeded inputs of run_switching_case (tests/golden/switching_lssm.npz)."""
    M, N, D, K = 6, 40, 3, 2
    th = 0.4
    A0 = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 0.9]])
    A1 = 0.5 * np.identity(3)
    z = (np.arange(N - 1) // 10) % 2
    x = np.zeros((N, D))
    x[0] = rs.normal(size=D)
    for n in range(N - 1):
        x[n + 1] = (A0 if z[n] == 0 else A1) @ x[n] + 0.3 * rs.normal(size=D)
    C = rs.normal(size=(M, D))
    return dict(sw_y=C @ x.T + 0.1 * rs.normal(size=(M, N)),
                sw_A0=np.identity(D) * np.ones((K, D, D)) + 0.1 * rs.normal(size=(K, D, D)),
                sw_X0=rs.normal(size=(N, D)), sw_C0=rs.normal(size=(M, 1, D)),
                sw_z0=rs.randint(K, size=N - 1))


def run_switching_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """Linear state-space model with switching dynamics, bayespy/demos/lssm_sd.py:36-115 at a
    small size: Dirichlet -> CategoricalMarkovChain -> SwitchingGaussianMarkovChain ->
    SumMultiply -> GaussianARD, ARD priors on the dynamics and loading matrices."""
    N_ = nodes_mod
    y = g['sw_y']
    M, N = y.shape
    K, D = g['sw_A0'].shape[0], g['sw_A0'].shape[-1]
    rho = N_.Dirichlet(1e-3 * np.ones(K), name='rho')
    V = N_.Dirichlet(1e-3 * np.ones(K), plates=(K,), name='V')
    v = 10 * np.identity(K) + 1 * np.ones((K, K))
    V.initialize_from_value(v / np.sum(v, axis=-1, keepdims=True))
    Z = N_.CategoricalMarkovChain(rho, V, states=N - 1, name='Z')
    Z.initialize_from_value(g['sw_z0'])
    alpha = N_.Gamma(1e-5, 1e-5, plates=(K, 1, D), name='alpha')
    A = N_.GaussianARD(0, alpha, shape=(D,), plates=(K, D), name='A')
    A.initialize_from_value(g['sw_A0'])
    X = N_.SwitchingGaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, Z, np.ones(D),
                                        n=N, name='X')
    X.initialize_from_value(g['sw_X0'])
    gamma = N_.Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    C = N_.GaussianARD(0, gamma, shape=(D,), plates=(M, 1), name='C')
    C.initialize_from_value(g['sw_C0'])
    F = N_.SumMultiply('i,i', C, X, name='F')
    tau = N_.Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    Y = N_.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, F, Z, rho, V, C, gamma, X, A, alpha, tau, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=5, verbose=False)
    out = {'sw_L': np.array(Q.L[:5])}
    for nm, nd in dict(X=X, A=A, Z=Z, V=V, rho=rho, C=C, tau=tau, alpha=alpha).items():
        out['sw_%s_u' % nm] = [np.array(u) for u in nd.get_moments()]
        out['sw_%s_Lterm' % nm] = np.array(Q.l[nd][:5])
    return out


def make_varying_inputs(rs):
    """Seeded inputs of run_varying_case (tests/golden/varying_lssm.npz)."""
    M, N, D, K = 5, 30, 2, 2
    x = np.zeros((N, D))
    x[0] = rs.normal(size=D)
    for n in range(N - 1):
        th = 0.3 + 0.4 * np.sin(2 * np.pi * n / N)
        An = 0.95 * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        x[n + 1] = An @ x[n] + 0.2 * rs.normal(size=D)
    C = rs.normal(size=(M, D))
    s0 = 10 * rs.normal(size=(N, K))
    s0[:, 0] = 10
    a0 = np.zeros((D, D, K))
    a0[:, :, 0] = np.identity(D) / s0[0, 0]
    a0[:, :, 1:] = 0.1 / s0[0, 0] * rs.normal(size=(D, D, K - 1))
    return dict(tv_y=C @ x.T + 0.1 * rs.normal(size=(M, N)), tv_s0=s0, tv_a0=a0,
                tv_x0=rs.normal(size=(N, D)), tv_c0=rs.normal(size=(M, 1, D)))


def run_varying_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """Linear state-space model with time-varying dynamics, bayespy/demos/lssm_tvd.py:40-140
    at a small size: the dynamics matrix of X is a combination of K matrices weighted by a
    second Gaussian Markov chain S (its Gaussian view sliced [1:])."""
    N_ = nodes_mod
    y = g['tv_y']
    M, N = y.shape
    D, K = g['tv_a0'].shape[0], g['tv_a0'].shape[-1]
    beta = N_.Gamma(1e-5, 1e-5, plates=(K,), name='beta')
    B = N_.GaussianARD(np.identity(K), beta, shape=(K,), plates=(K,), name='B')
    B.initialize_from_value(np.identity(K))
    S = N_.GaussianMarkovChain(np.ones(K), 1e-6 * np.identity(K), B, np.ones(K), n=N, name='S')
    S.initialize_from_value(g['tv_s0'])
    alpha = N_.Gamma(1e-5, 1e-5, plates=(D, K), name='alpha')
    alpha.initialize_from_value(1 * np.ones((D, K)))
    A = N_.GaussianARD(0, alpha, shape=(D, K), plates=(D,), name='A')
    A.initialize_from_value(g['tv_a0'])
    if hasattr(S, 'as_gaussian'):
        Sg = S.as_gaussian()
    else:
        from bayespy.inference.vmp.nodes.gaussian import GaussianMoments
        Sg = S._ensure_moments(S, GaussianMoments, ndim=1)
    X = N_.VaryingGaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, Sg[1:], np.ones(D),
                                      n=N, name='X')
    X.initialize_from_value(g['tv_x0'])
    gamma = N_.Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    C = N_.GaussianARD(0, gamma, shape=(D,), plates=(M, 1), name='C')
    C.initialize_from_value(g['tv_c0'])
    F = N_.SumMultiply('i,i', C, X, name='F')
    tau = N_.Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    Y = N_.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, F, C, gamma, X, A, alpha, tau, S, B, beta, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=5, verbose=False)
    out = {'tv_L': np.array(Q.L[:5])}
    for nm, nd in dict(X=X, A=A, S=S, B=B, C=C, tau=tau, alpha=alpha, beta=beta).items():
        out['tv_%s_u' % nm] = [np.array(u) for u in nd.get_moments()]
        out['tv_%s_Lterm' % nm] = np.array(Q.l[nd][:5])
    return out


def make_concat_gaussian_inputs(rs):
    """Seeded inputs of run_concat_gaussian_case (tests/golden/concat_gaussian.npz)."""
    def spd(d):
        a = rs.normal(size=(d, d))
        return a @ a.T + d * np.eye(d)
    N, D1, D2, D3 = 5, 3, 4, 2
    return dict(cg_m1=rs.normal(size=(N, D1)), cg_L1=spd(D1), cg_m2=rs.normal(size=(N, D2)),
                cg_L2=spd(D2), cg_x3=rs.normal(size=(N, D3)), cg_V=spd(D1 + D2 + D3),
                cg_y=rs.normal(size=(N, D1 + D2 + D3)))


def run_concat_gaussian_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """ConcatGaussian of two latent Gaussian vectors and a constant under a full-covariance
    Gaussian observation (nodes/tests/test_gaussian.py:1413-1500)."""
    N_ = nodes_mod
    X1 = N_.Gaussian(g['cg_m1'], g['cg_L1'], name='X1')
    X2 = N_.Gaussian(g['cg_m2'], g['cg_L2'], name='X2')
    Z = N_.ConcatGaussian(X1, X2, g['cg_x3'], name='Z')
    Y = N_.Gaussian(Z, g['cg_V'], name='Y')
    Y.observe(g['cg_y'])
    out = {'cg_Z0_u': [np.array(v) for v in Z.get_moments()]}
    Q = vb_cls(Y, X1, X2, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    out['cg_L'] = np.array(Q.L[:3])
    for nm, nd in dict(X1=X1, X2=X2, Z=Z).items():
        out['cg_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]
    return out


def make_default_ndim_inputs(rs):
    """Seeded inputs of run_default_ndim_case (tests/golden/default_ndim.npz)."""
    return dict(dn_y=rs.normal(size=(5, 4, 3)) + np.array([1.0, -1.0, 3.0]),
                dn_mask=rs.rand(5, 4, 3) < 0.8, dn_y2=rs.normal(size=(4, 2, 3)))


def run_default_ndim_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """GaussianARD without ndim / shape is scalar-valued whatever its mean is
    (gaussian.py:1617-1640): the variable axes of a Gaussian mean become plates of the node, the
    posterior factorises over them; ndim=1 under a matrix-valued mean keeps one axis."""
    N_ = nodes_mod
    out = {}
    mu = N_.GaussianARD(0, 1e-1, shape=(3,), plates=(4,), name='mu')
    alpha = N_.Gamma(1e-1, 1e-1, plates=(3,), name='alpha')
    X = N_.GaussianARD(mu, alpha, name='X')
    out['dn_X_plates'], out['dn_X_ndims'] = np.array(X.plates), np.array([len(d) for d in X.dims])
    tau = N_.Gamma(1e-1, 1e-1, name='tau')
    Y = N_.GaussianARD(X, tau, plates=(5, 4, 3), name='Y')
    Y.observe(g['dn_y'], mask=g['dn_mask'])
    Q = vb_cls(Y, X, mu, alpha, tau, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    out['dn_L'] = np.array(Q.L[:3])
    for nm, nd in dict(X=X, mu=mu, alpha=alpha, tau=tau).items():
        out['dn_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]
    M = N_.GaussianARD(0, 1e-1, shape=(2, 3), plates=(4,), name='M')
    Z = N_.GaussianARD(M, 2.0, ndim=1, name='Z')
    out['dn_Z_plates'], out['dn_Z_shape'] = np.array(Z.plates), np.array(Z.dims[0])
    Y2 = N_.GaussianARD(Z, 1.5, ndim=1, name='Y2')
    Y2.observe(g['dn_y2'])
    Q = vb_cls(Y2, Z, M, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    out['dn2_L'] = np.array(Q.L[:3])
    for nm, nd in dict(Z=Z, M=M).items():
        out['dn2_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]
    return out


# ---------------------------------------------------------------------------------------------
# PCA with missing values (demos/pca.py:80-82: the demo's default use) -- the fused masked block,
# NaN placeholders at the missing entries, partially observed nodes.  Shared statement for
# statement by oracle/make_golden.py (live reference) and the device tests.
# ---------------------------------------------------------------------------------------------
MASKED_PCA_SIZES = (('m0', 5, 60, 2, 0.8, 5), ('m1', 12, 300, 4, 0.8, 5),
                    ('m2', 40, 333, 17, 0.6, 4), ('m3', 128, 1024, 32, 0.9, 3))


def make_masked_pca_inputs(rs):
    g = {}
    for tag, D, N, K, keep, n_iter in MASKED_PCA_SIZES:
        w, x = rs.normal(size=(D, K)), rs.normal(size=(N, K))
        y = w @ x.T + 0.1 * rs.normal(size=(D, N))
        mask = rs.rand(D, N) < keep
        mask[0, :3] = False                 # a plate with several holes in one row ...
        mask[:, 5] = mask[:, 5] & (np.arange(D) % 2 == 0)
        if tag == 'm0':
            mask[:, 7] = False              # ... and one plate without any observation
        y = np.where(mask, y, np.nan)       # the usual encoding of missing data
        g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'] = y, mask, rs.normal(size=(N, K))
    # a partially observed node WITH a child: its latent plates take part in the updates
    n = 30
    g['po_z'] = np.where(rs.rand(n) < 0.5, rs.normal(2.0, 1.0, size=n), np.nan)
    g['po_y'] = rs.normal(2.0, 1.5, size=n)
    return g


def build_masked_pca(nodes_mod, vb_cls, y, mask, x0, a0=1e-2, b0=1e-2, shard=False, **vb_kwargs):
    D, N = y.shape
    K = x0.shape[1]
    alpha = nodes_mod.Gamma(a0, b0, plates=(K,), name='alpha')
    W = nodes_mod.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes_mod.GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    if shard:
        X.shard(-1)
    F = nodes_mod.SumMultiply('i,i', W, X, name='F')
    tau = nodes_mod.Gamma(a0, b0, name='tau')
    Y = nodes_mod.GaussianARD(F, tau, name='Y')
    X.initialize_from_value((x0 if hasattr(x0, 'device') else np.asarray(x0))[None, :, :])
    Y.observe(y, mask=mask)
    Q = vb_cls(Y, F, W, X, tau, alpha, **vb_kwargs)
    Q.ignore_bound_checks = True
    return Q


def _run_masked_pca(nodes_mod, vb_cls, g, sizes, res, only=None, **vb_kwargs):
    for tag, D, N, K, keep, n_iter in sizes:
        if only is not None and tag not in only:
            continue
        Q = build_masked_pca(nodes_mod, vb_cls, g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'],
                             **vb_kwargs)
        Q.update(repeat=n_iter, verbose=False)
        res[tag + '_L'] = np.array(Q.L[:Q.iter])
        for nm in ('Y', 'W', 'X', 'tau', 'alpha'):
            res['%s_L_%s' % (tag, nm)] = np.array(Q.l[Q[nm]][:Q.iter])
        W, X, tau, alpha, Y = Q['W'], Q['X'], Q['tau'], Q['alpha'], Q['Y']
        res[tag + '_W_u0'], res[tag + '_W_u1'] = np.array(W.u[0]), np.array(W.u[1])
        res[tag + '_X_u0'] = np.array(X.u[0])
        res[tag + '_X_u1_first'] = np.array(X.u[1][0, :8])
        res[tag + '_tau_u'] = np.array([np.asarray(u) for u in tau.u], dtype=np.float64)
        res[tag + '_alpha_u0'], res[tag + '_alpha_u1'] = np.array(alpha.u[0]), np.array(alpha.u[1])
        if D * N <= 4000:
            # q of the missing entries of the partially observed leaf (the predictive moments)
            res[tag + '_Y_u0'], res[tag + '_Y_u1'] = np.array(Y.u[0]), np.array(Y.u[1])
    return res


# Erasure patterns at the edges (tests/golden/masked_pca_erasures.npz): plates without any observed
# dimension, dimensions without any observed plate (ignored plates of W: left out of the message to
# alpha and of W's bound term, node.py:457-526, :624-650), a dimension seen on one plate only.
ERASURE_SIZES = (('e0', 9, 70, 4, 0.8, 4), ('e1', 40, 300, 32, 0.8, 4), ('e2', 128, 200, 16, 0.5, 3))


def make_erasure_inputs(rs):
    g = {}
    for tag, D, N, K, keep, n_iter in ERASURE_SIZES:
        y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
        mask = rs.rand(D, N) < keep
        mask[:, [0, N // 2, N - 1]] = False          # plates without data (first / middle / last)
        mask[2, :] = False                           # a dimension without data
        mask[D - 1, :] = False
        mask[D - 1, 5] = True                        # a dimension seen once
        if tag == 'e2':
            mask[64:96, :] = False                   # a whole block of dimensions without data
        g[tag + '_y'], g[tag + '_mask'] = np.where(mask, y, np.nan), mask
        g[tag + '_x0'] = rs.normal(size=(N, K))
    return g


def run_erasure_cases(nodes_mod, vb_cls, g, only=None, **vb_kwargs):
    return _run_masked_pca(nodes_mod, vb_cls, g, ERASURE_SIZES, {}, only=only, **vb_kwargs)


def run_masked_pca_cases(nodes_mod, vb_cls, g, only=None, **vb_kwargs):
    res = _run_masked_pca(nodes_mod, vb_cls, g, MASKED_PCA_SIZES, {}, only=only, **vb_kwargs)
    if only is None or 'po' in only:
        zdat, ydat = g['po_z'], g['po_y']
        n = zdat.shape[0]
        mu = nodes_mod.GaussianARD(0, 1e-3, name='mu')
        z = nodes_mod.GaussianARD(mu, 1.0, plates=(n,), name='z')
        tau = nodes_mod.Gamma(1e-2, 1e-2, name='tau')
        y = nodes_mod.GaussianARD(z, tau, plates=(n,), name='y')
        y.observe(ydat)
        z.observe(zdat, mask=~np.isnan(zdat))
        Q = vb_cls(y, z, mu, tau, **vb_kwargs)
        Q.ignore_bound_checks = True
        Q.update(repeat=5, verbose=False)
        res['po_L'] = np.array(Q.L[:Q.iter])
        res['po_z_u0'], res['po_z_u1'] = np.array(z.u[0]), np.array(z.u[1])
        res['po_mu_u'] = np.array([np.asarray(u) for u in mu.u], dtype=np.float64)
        res['po_tau_u'] = np.array([np.asarray(u) for u in tau.u], dtype=np.float64)
    return res


def make_seeded_pca(seed, N, D, K):
    """Inputs of the seeded live-reference fixture (tests/golden/pca_seeded_n100000_d128_k32.npz
    stores the seed and the reference's OUTPUTS only): demos/pca.py:70-74 data + initial <x>."""
    rs = np.random.RandomState(seed)
    w = rs.normal(0, 1, (D, K))
    x = rs.normal(0, 1, (N, K))
    y = w @ x.T + 0.1 * rs.normal(size=(D, N))
    x0 = rs.normal(0, 1, (N, K))
    return y, x0


# ---------------------------------------------------------------------------------------------
# numeric arrays as Gaussian-moment parents of SumMultiply / Dot (dot.py:186-197: the reference
# wraps them in constants with delta moments); shared by oracle/make_golden.py and the tests
# ---------------------------------------------------------------------------------------------
def make_constant_parent_inputs(rs):
    N, K = 40, 3
    c = rs.normal(size=(N, K))                       # known regressors
    w = rs.normal(size=K)
    y = c @ w + 0.2 * rs.normal(size=N)
    d = rs.normal(size=(5, 1, K))
    y2 = np.einsum('mik,nk->mn', d, c) + 0.1 * rs.normal(size=(5, N))
    return dict(c=c, y=y, d=d, y2=y2)


def run_constant_parent_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    res = {}
    c, y = g['c'], g['y']
    N, K = c.shape
    # (a) Bayesian linear regression with known inputs: F = Dot(w, c)
    alpha = nodes_mod.Gamma(1e-3, 1e-3, plates=(K,), name='alpha')
    w = nodes_mod.GaussianARD(0, alpha, shape=(K,), name='w')
    F = nodes_mod.SumMultiply('i,i', w, c, name='F')
    tau = nodes_mod.Gamma(1e-3, 1e-3, name='tau')
    Y = nodes_mod.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, w, alpha, tau, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=5, verbose=False)
    res['lr_L'] = np.array(Q.L[:Q.iter])
    res['lr_w_u0'], res['lr_w_u1'] = np.array(w.u[0]), np.array(w.u[1])
    res['lr_tau_u0'], res['lr_alpha_u0'] = np.array(tau.u[0]), np.array(alpha.u[0])
    res['lr_F_u0'], res['lr_F_u1'] = np.array(F.get_moments()[0]), np.array(F.get_moments()[1])
    # (b) a constant in the middle of a three-factor product with an output key
    d, y2 = g['d'], g['y2']
    M = d.shape[0]
    z = nodes_mod.GaussianARD(0, 1, shape=(K,), plates=(M, 1), name='z')
    s = nodes_mod.GaussianARD(1, 1, plates=(1, N), name='s')
    F2 = nodes_mod.SumMultiply('i,i,', z, c, s, name='F2')
    tau2 = nodes_mod.Gamma(1e-3, 1e-3, name='tau2')
    Y2 = nodes_mod.GaussianARD(F2, tau2, name='Y2')
    z.initialize_from_value(d)
    Y2.observe(y2)
    Q2 = vb_cls(Y2, z, s, tau2, **vb_kwargs)
    Q2.ignore_bound_checks = True
    Q2.update(repeat=4, verbose=False)
    res['tp_L'] = np.array(Q2.L[:Q2.iter])
    res['tp_z_u0'], res['tp_s_u0'] = np.array(z.u[0]), np.array(s.u[0])
    res['tp_s_u1'] = np.array(s.u[1])
    return res


# ---------------------------------------------------------------------------------------------
# Update-order probes (tests/golden/order_probes.npz): node-level update sequences that differ
# from the constructor order -- repeated updates of one node, the bound evaluated in between --
# on the three fused model families.  Every update must see the latest moments of its Markov
# blanket (vmp.py:154-172), whatever a fused block caches.
# ---------------------------------------------------------------------------------------------
PCA_PROBE_SEQ = ('L', 'W', 'L', 'X', 'W', 'L', 'W', 'tau', 'L', 'alpha', 'X', 'X', 'L', 'tau', 'alpha', 'W', 'L',
                 'tau', 'tau', 'X', 'alpha', 'L')
GMM_PROBE_SEQ = ('L', 'mu', 'Lambda', 'L', 'z', 'mu', 'L', 'Lambda', 'alpha', 'L', 'z', 'z', 'L', 'mu', 'Lambda', 'L', 'alpha',
                 'Lambda', 'mu', 'z', 'L')
LSSM_PROBE_SEQ = ('X', 'C', 'L', 'A', 'tau', 'L', 'gamma', 'alpha', 'X', 'L', 'C', 'C', 'tau', 'X',
                  'A', 'L')


def make_order_probe_inputs(rs):
    g = {}
    D, N, K = 6, 50, 3
    g['pca_y'] = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    g['pca_x0'] = rs.normal(size=(N, K))
    g['pca_mask'] = rs.rand(D, N) < 0.8
    g['gmm_y'] = np.concatenate([rs.normal(size=(30, 2)), rs.normal(size=(30, 2)) + 4])
    g['gmm_lab0'] = rs.randint(3, size=60)
    M, B, T, Dx = 3, 4, 12, 2
    g['lssm_y'] = rs.normal(size=(M, B, T))
    g['lssm_x0'] = rs.normal(size=(B, T, Dx))
    g['lssm_c0'] = rs.normal(size=(M, 1, 1, Dx))
    return g


def _run_sequence(Q, seq, out):
    for step in seq:
        if step == 'L':
            out.append(float(Q.compute_lowerbound()))
        else:
            Q.update(Q[step], repeat=1, verbose=False)
    return out


def run_order_probes(nodes_mod, vb_cls, g, **vb_kwargs):
    N_ = nodes_mod
    res = {}
    Q = build_pca(N_, vb_cls, g['pca_y'], g['pca_x0'], g['pca_x0'].shape[1], **vb_kwargs)
    res['pca_L'] = np.array(_run_sequence(Q, PCA_PROBE_SEQ, []))
    res['pca_W_u0'], res['pca_X_u0'] = np.array(Q['W'].u[0]), np.array(Q['X'].u[0])
    Q = build_masked_pca(N_, vb_cls, np.where(g['pca_mask'], g['pca_y'], np.nan), g['pca_mask'],
                         g['pca_x0'], **vb_kwargs)
    res['mpca_L'] = np.array(_run_sequence(Q, PCA_PROBE_SEQ, []))
    res['mpca_W_u0'], res['mpca_X_u0'] = np.array(Q['W'].u[0]), np.array(Q['X'].u[0])
    # mixture (demos/mog.py:17-64)
    y, lab0 = g['gmm_y'], g['gmm_lab0']
    N, D = y.shape
    K = 3
    alpha = N_.Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = N_.Categorical(alpha, plates=(N,), name='z')
    mu = N_.GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = N_.Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = N_.Mixture(z, N_.Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = vb_cls(Y, mu, Lam, z, alpha, **vb_kwargs)
    Q.ignore_bound_checks = True
    res['gmm_L'] = np.array(_run_sequence(Q, GMM_PROBE_SEQ, []))
    res['gmm_z_u0'], res['gmm_mu_u0'] = np.array(z.u[0]), np.array(mu.u[0])
    # linear state-space model (demos/lssm.py:33-103) with a sequence plate
    y, x0, c0 = g['lssm_y'], g['lssm_x0'], g['lssm_c0']
    M, B, T = y.shape
    Dx = x0.shape[-1]
    al = N_.Gamma(1e-5, 1e-5, plates=(Dx,), name='alpha')
    A = N_.GaussianARD(0, al, shape=(Dx,), plates=(Dx,), name='A')
    A.initialize_from_value(np.identity(Dx))
    X = N_.GaussianMarkovChain(np.zeros(Dx), 1e-3 * np.identity(Dx), A, np.ones(Dx), n=T,
                               plates=(B,), name='X')
    X.initialize_from_value(x0)
    gamma = N_.Gamma(1e-5, 1e-5, plates=(Dx,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(Dx))
    C = N_.GaussianARD(0, gamma, shape=(Dx,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(c0)
    tau = N_.Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = N_.SumMultiply('i,i', C, X, name='F')
    Yl = N_.GaussianARD(F, tau, name='Y')
    Yl.observe(y)
    Q = vb_cls(Yl, F, C, gamma, X, A, al, tau, **vb_kwargs)
    Q.ignore_bound_checks = True
    res['lssm_L'] = np.array(_run_sequence(Q, LSSM_PROBE_SEQ, []))
    res['lssm_X_u0'], res['lssm_A_u0'] = np.array(X.u[0]), np.array(A.u[0])
    return res


# ---------------------------------------------------------------------------------------------
# Hyperparameter probes (tests/golden/hyper_probes.npz): the four fused model families with
# priors, prior means / precisions and fixed parameters away from the demos' defaults -- every
# constant a fused block reads from its nodes' parents.
# ---------------------------------------------------------------------------------------------
def make_hyper_probe_inputs(rs):
    g = {}
    D, N, K = 7, 60, 3
    g['pca_y'] = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.3 * rs.normal(size=(D, N))
    g['pca_x0'] = rs.normal(size=(N, K))
    g['pca_mask'] = rs.rand(D, N) < 0.75
    g['gmm_y'] = np.concatenate([rs.normal(size=(40, 3)), rs.normal(size=(30, 3)) * 0.5 + 3])
    g['gmm_lab0'] = rs.randint(3, size=70)
    a = rs.normal(size=(3, 3))
    g['gmm_V0'] = a @ a.T + 3 * np.identity(3)
    M, B, T, Dx = 3, 5, 10, 2
    g['lssm_y'] = rs.normal(size=(M, B, T))
    g['lssm_x0'] = rs.normal(size=(B, T, Dx))
    g['lssm_c0'] = rs.normal(size=(M, 1, 1, Dx))
    g['lssm_a0'] = 0.5 * np.identity(Dx) + 0.1 * rs.normal(size=(Dx, Dx))
    b = rs.normal(size=(Dx, Dx))
    g['lssm_Lam0'] = b @ b.T + np.identity(Dx)
    return g


def _pca_with_priors(N_, vb_cls, y, x0, mask, **vb_kwargs):
    D, N = y.shape
    K = x0.shape[1]
    alpha = N_.Gamma(0.5, 2.0, plates=(K,), name='alpha')
    W = N_.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = N_.GaussianARD(0, 2.5, shape=(K,), plates=(1, N), name='X')
    F = N_.SumMultiply('i,i', W, X, name='F')
    tau = N_.Gamma(3.0, 0.1, name='tau')
    Y = N_.GaussianARD(F, tau, name='Y')
    X.initialize_from_value(np.asarray(x0)[None, :, :])
    if mask is None:
        Y.observe(y)
    else:
        Y.observe(np.where(mask, y, np.nan), mask=mask)
    Q = vb_cls(Y, F, W, X, tau, alpha, **vb_kwargs)
    Q.ignore_bound_checks = True
    return Q


def run_hyper_probes(nodes_mod, vb_cls, g, **vb_kwargs):
    N_ = nodes_mod
    res = {}
    for tag, mask in (('pca', None), ('mpca', g['pca_mask'])):
        Q = _pca_with_priors(N_, vb_cls, g['pca_y'], g['pca_x0'], mask, **vb_kwargs)
        Q.update(repeat=5, verbose=False)
        res[tag + '_L'] = np.array(Q.L[:5])
        res[tag + '_W_u0'], res[tag + '_X_u0'] = np.array(Q['W'].u[0]), np.array(Q['X'].u[0])
        res[tag + '_tau_u0'], res[tag + '_alpha_u0'] = np.array(Q['tau'].u[0]), np.array(Q['alpha'].u[0])
    # mixture: non-uniform Dirichlet, prior precision of the means, Wishart degrees and scale
    y, lab0 = g['gmm_y'], g['gmm_lab0']
    N, D = y.shape
    K = 3
    alpha = N_.Dirichlet(np.array([0.5, 2.0, 1.0]), name='alpha')
    z = N_.Categorical(alpha, plates=(N,), name='z')
    mu = N_.GaussianARD(0, 0.3, shape=(D,), plates=(K,), name='mu')
    Lam = N_.Wishart(D + 2.5, g['gmm_V0'], plates=(K,), name='Lambda')
    Y = N_.Mixture(z, N_.Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0)
    Y.observe(y)
    Q = vb_cls(Y, mu, Lam, z, alpha, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=5, verbose=False)
    res['gmm_L'] = np.array(Q.L[:5])
    res['gmm_z_u0'], res['gmm_mu_u0'] = np.array(z.u[0]), np.array(mu.u[0])
    res['gmm_Lambda_u0'], res['gmm_alpha_u0'] = np.array(Lam.u[0]), np.array(alpha.u[0])
    # state-space model: prior mean / precision of the first state, fixed and Gamma innovation
    # precisions away from one, three different Gamma priors
    y, x0, c0 = g['lssm_y'], g['lssm_x0'], g['lssm_c0']
    M, B, T = y.shape
    Dx = x0.shape[-1]
    for tag, gamma_nu in (('lssm', False), ('lssmnu', True)):
        al = N_.Gamma(0.5, 0.5, plates=(Dx,), name='alpha')
        A = N_.GaussianARD(0, al, shape=(Dx,), plates=(Dx,), name='A')
        A.initialize_from_value(g['lssm_a0'])
        nu = N_.Gamma(2.0, 3.0, plates=(Dx,), name='nu') if gamma_nu else np.array([2.0, 0.5])
        X = N_.GaussianMarkovChain(np.array([1.0, -1.0]), g['lssm_Lam0'], A, nu, n=T, plates=(B,),
                                   name='X')
        X.initialize_from_value(x0)
        gamma = N_.Gamma(2.0, 1.0, plates=(Dx,), name='gamma')
        gamma.initialize_from_value(0.5 * np.ones(Dx))
        C = N_.GaussianARD(0, gamma, shape=(Dx,), plates=(M, 1, 1), name='C')
        C.initialize_from_value(c0)
        tau = N_.Gamma(1.5, 0.2, name='tau')
        tau.initialize_from_value(3.0)
        F = N_.SumMultiply('i,i', C, X, name='F')
        Yl = N_.GaussianARD(F, tau, name='Y')
        Yl.observe(y)
        nodes_ = [Yl, F, C, gamma, X, A, al, tau] + ([nu] if gamma_nu else [])
        Q = vb_cls(*nodes_, **vb_kwargs)
        Q.ignore_bound_checks = True
        Q.update(repeat=5, verbose=False)
        res[tag + '_L'] = np.array(Q.L[:5])
        res[tag + '_X_u0'], res[tag + '_A_u0'] = np.array(X.u[0]), np.array(A.u[0])
        res[tag + '_C_u0'], res[tag + '_tau_u0'] = np.array(C.u[0]), np.array(tau.u[0])
    return res


# ---------------------------------------------------------------------------------------------
# Re-observing data after updates (tests/golden/reobserve.npz): only Y changes, every posterior
# stays (stochastic.py:223-273); the next messages combine the new data with the current <x>.
# ---------------------------------------------------------------------------------------------
def make_reobserve_inputs(rs):
    D, N, K = 6, 80, 3
    w = rs.normal(size=(D, K))
    g = dict(y1=w @ rs.normal(size=(K, N)) + 0.2 * rs.normal(size=(D, N)),
             y2=w @ rs.normal(size=(K, N)) + 0.2 * rs.normal(size=(D, N)),
             x0=rs.normal(size=(N, K)))
    M, B, T, Dx = 3, 4, 9, 2
    g['lssm_y1'], g['lssm_y2'] = rs.normal(size=(M, B, T)), rs.normal(size=(M, B, T)) + 0.5
    g['lssm_x0'], g['lssm_c0'] = rs.normal(size=(B, T, Dx)), rs.normal(size=(M, 1, 1, Dx))
    g['x1'], g['w1'] = rs.normal(size=(N, K)), rs.normal(size=(D, 1, K))
    return g


def run_reinitialise_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """initialize_from_value on X, later on W, between updates: that node becomes a point mass (its
    bound term -inf until updated), every other posterior stays (expfamily.py:193-204)."""
    Q = build_pca(nodes_mod, vb_cls, g['y1'], g['x0'], g['x0'].shape[1], **vb_kwargs)
    Q.update(repeat=3, verbose=False)
    Q['X'].initialize_from_value(g['x1'][None])
    out = [float(Q.compute_lowerbound())]
    Q.update(Q['W'], Q['tau'], repeat=1, verbose=False)
    out.append(float(Q.compute_lowerbound()))
    Q.update(Q['X'], repeat=1, verbose=False)
    out.append(float(Q.compute_lowerbound()))
    Q['W'].initialize_from_value(g['w1'])
    out.append(float(Q.compute_lowerbound()))
    Q.update(Q['X'], Q['alpha'], repeat=1, verbose=False)
    out.append(float(Q.compute_lowerbound()))
    Q.update(repeat=2, verbose=False)
    return dict(ri_steps=np.array(out), ri_L=np.array(Q.L[:Q.iter]), ri_W_u0=np.array(Q['W'].u[0]),
                ri_X_u0=np.array(Q['X'].u[0]), ri_alpha_u0=np.array(Q['alpha'].u[0]))


def run_reobserve_case(nodes_mod, vb_cls, g, **vb_kwargs):
    Q = build_pca(nodes_mod, vb_cls, g['y1'], g['x0'], g['x0'].shape[1], **vb_kwargs)
    Q.update(repeat=3, verbose=False)
    Q['Y'].observe(g['y2'])
    L_mid = float(Q.compute_lowerbound())
    Q.update(Q['W'], repeat=1, verbose=False)
    L_w = float(Q.compute_lowerbound())
    Q.update(repeat=2, verbose=False)
    return dict(L=np.array(Q.L[:Q.iter]), L_mid=L_mid, L_w=L_w, W_u0=np.array(Q['W'].u[0]),
                X_u0=np.array(Q['X'].u[0]), tau_u0=np.array(Q['tau'].u[0]))


def run_reobserve_lssm_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """The same for the linear state-space model: new observations, q(X) and every other posterior
    kept; the messages to C and tau combine the new data with the current <x>."""
    N_ = nodes_mod
    y, x0, c0 = g['lssm_y1'], g['lssm_x0'], g['lssm_c0']
    M, B, T = y.shape
    Dx = x0.shape[-1]
    al = N_.Gamma(1e-5, 1e-5, plates=(Dx,), name='alpha')
    A = N_.GaussianARD(0, al, shape=(Dx,), plates=(Dx,), name='A')
    A.initialize_from_value(np.identity(Dx))
    X = N_.GaussianMarkovChain(np.zeros(Dx), 1e-3 * np.identity(Dx), A, np.ones(Dx), n=T,
                               plates=(B,), name='X')
    X.initialize_from_value(x0)
    gamma = N_.Gamma(1e-5, 1e-5, plates=(Dx,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(Dx))
    C = N_.GaussianARD(0, gamma, shape=(Dx,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(c0)
    tau = N_.Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = N_.SumMultiply('i,i', C, X, name='F')
    Y = N_.GaussianARD(F, tau, name='Y')
    Y.observe(y)
    Q = vb_cls(Y, F, C, gamma, X, A, al, tau, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=3, verbose=False)
    Y.observe(g['lssm_y2'])
    L_mid = float(Q.compute_lowerbound())
    Q.update(C, tau, repeat=1, verbose=False)
    L_c = float(Q.compute_lowerbound())
    Q.update(repeat=2, verbose=False)
    return dict(lssm_L=np.array(Q.L[:Q.iter]), lssm_L_mid=L_mid, lssm_L_c=L_c,
                lssm_X_u0=np.array(X.u[0]), lssm_C_u0=np.array(C.u[0]), lssm_A_u0=np.array(A.u[0]))


# ---------------------------------------------------------------------------------------------
# Gaussian-gamma nodes (GaussianGammaMoments gaussian.py:161-229, the GaussianGamma
# node :1777-1840 with GaussianGammaDistribution :892-1136, WrapToGaussianGamma :2299-2371).
# Shared statement for statement by oracle/make_golden.py (live reference) and the device tests.
# ---------------------------------------------------------------------------------------------
def make_gaussian_gamma_inputs(rs):
    G, N = 3, 40
    return dict(gg_y=rs.normal(size=(G, N)) * np.array([[0.5], [1.0], [2.0]])
                + np.array([[1.0], [-2.0], [4.0]]),
                gg_mask=rs.rand(G, N) < 0.85,
                gg_mu0=np.array([0.5, -0.5, 0.0]), gg_lam0=np.array([0.1, 0.2, 0.3]),
                gg_a0=np.array([0.6, 0.7, 0.8]), gg_b0=np.array([0.9, 1.0, 1.1]),
                gg_V=np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.2], [0.0, 0.2, 1.5]]))


def run_gaussian_gamma_cases(nodes_mod, vb_cls, g, **vb_kwargs):
    N_ = nodes_mod
    out = {}
    G, N = g['gg_y'].shape

    def record(tag, Q, nodes, n_iter):
        Ls = []
        for _ in range(n_iter):
            Q.update(repeat=1, verbose=False)
            Ls.append(Q.L[Q.iter - 1])
        out[tag + '_L'] = np.array(Ls)
        for nm, nd in nodes.items():
            out['%s_%s_u' % (tag, nm)] = [np.array(v) for v in nd.get_moments()]
            out['%s_%s_l' % (tag, nm)] = np.array(Q.l[nd][:Q.iter])

    # (a) the conjugate normal-gamma model of a group's mean and precision, jointly:
    #     y_gn ~ N(x_g, 1 / tau_g), (x_g, tau_g) ~ GaussianGamma: exact after one update
    XT = N_.GaussianGamma(g['gg_mu0'][:, None], g['gg_lam0'][:, None], g['gg_a0'][:, None],
                          g['gg_b0'][:, None], ndim=0, name='XT')
    out['a_plates'], out['a_ndims'] = np.array(XT.plates), np.array([len(d) for d in XT.dims])
    Y = N_.GaussianARD(XT, 1, name='Y')
    out['a_Y_plates'] = np.array(Y.plates)
    Y = N_.GaussianARD(XT, 1, plates=(G, N), name='Y')
    Y.observe(g['gg_y'])
    Q = vb_cls(Y, XT, **vb_kwargs)
    Q.ignore_bound_checks = True
    record('a', Q, dict(XT=XT, Y=Y), 2)

    # (b) latent parents of the joint node (Gaussian mean, Gamma rate), an extra Gamma scale on
    #     the child, an array mask
    m = N_.GaussianARD(0, 1e-2, plates=(G, 1), name='m')
    b = N_.Gamma(1e-1, 1e-1, plates=(G, 1), name='b')
    XT2 = N_.GaussianGamma(m, g['gg_lam0'][:, None], 1.5, b, ndim=0, name='XT2')
    s = N_.Gamma(1e-2, 1e-2, plates=(1, N), name='s')
    Y2 = N_.GaussianARD(XT2, s, name='Y2')
    Y2.observe(g['gg_y'], mask=g['gg_mask'])
    Q = vb_cls(Y2, XT2, s, m, b, **vb_kwargs)
    Q.ignore_bound_checks = True
    record('b', Q, dict(XT2=XT2, s=s, m=m, b=b, Y2=Y2), 4)

    # (c) vector-valued joint node under a Gaussian mean, a Wishart precision and a Gamma rate
    #     (the reference's own construction test, nodes/tests/test_gaussian.py:966-972): moments
    #     and bound term of the prior-initialised node and after an update without children
    mu = N_.Gaussian(np.array([1.0, -1.0, 0.5]), g['gg_V'], name='mu')
    Lam = N_.Wishart(5, g['gg_V'], name='Lam')
    b3 = N_.Gamma(2.0, 3.0, name='b3')
    XV = N_.GaussianGamma(mu, Lam, 2.5, b3, name='XV')
    out['c_dims'] = np.array([len(d) for d in XV.dims])
    Q = vb_cls(XV, mu, Lam, b3, **vb_kwargs)
    Q.ignore_bound_checks = True
    record('c', Q, dict(XV=XV, mu=mu, b3=b3), 2)
    return out


def make_hierarchical_wishart_inputs(rs):
    """Seeded inputs of run_hierarchical_wishart_case (tests/golden/hierarchical_wishart.npz)."""
    D, P, N = 3, 4, 12
    a = rs.normal(size=(D, D))
    lam_true = [np.linalg.inv(0.3 * (i + 1) * (a @ a.T + D * np.eye(D))) for i in range(P)]
    mu = rs.normal(size=(P, 1, D))
    y = np.stack([rs.multivariate_normal(mu[i, 0], np.linalg.inv(lam_true[i]), size=N)
                  for i in range(P)])
    return dict(hw_V0=0.5 * np.eye(D) + 0.1 * (a @ a.T), hw_mu=mu, hw_y=y)


def run_hierarchical_wishart_case(nodes_mod, vb_cls, g, **vb_kwargs):
    """A Wishart node whose inverse scale matrix is a Wishart node (wishart.py:142-150: the
    message [-<Lambda>/2, n/2] to V), precisions of P groups of Gaussian observations."""
    N_ = nodes_mod
    D = g['hw_V0'].shape[0]
    P = g['hw_y'].shape[0]
    V = N_.Wishart(D + 2.0, g['hw_V0'], name='V')
    Lam = N_.Wishart(D + 1.0, V, plates=(P, 1), name='Lam')
    Y = N_.Gaussian(g['hw_mu'], Lam, plates=g['hw_y'].shape[:2], name='Y')
    Y.observe(g['hw_y'])
    Q = vb_cls(Y, Lam, V, **vb_kwargs)
    Q.ignore_bound_checks = True
    Q.update(repeat=4, verbose=False)
    out = {'hw_L': np.array(Q.L[:4]), 'hw_plates': np.array(Lam.plates)}
    for nm, nd in dict(Lam=Lam, V=V).items():
        out['hw_%s_u' % nm] = [np.array(v) for v in nd.get_moments()]
        out['hw_%s_L' % nm] = np.array(Q.l[nd][:4])
    return out
