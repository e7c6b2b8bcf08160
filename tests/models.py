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
