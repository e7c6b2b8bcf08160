"""Model builders shared by the tests (written against the bayespy API surface)."""
import numpy as np


def build_pca(nodes_mod, vb_cls, y, x0, K, a0=1e-2, b0=1e-2, **vb_kwargs):
    """bayespy/demos/pca.py:22-61 with X initialised from ``x0`` (N,K) and Y fully observed."""
    D, N = y.shape
    alpha = nodes_mod.Gamma(a0, b0, plates=(K,), name='alpha')
    W = nodes_mod.GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes_mod.GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
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
