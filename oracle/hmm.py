"""
CPU oracle for the forward-backward recursion of categorical Markov chains.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: PINNED against the live
reference through tests/golden/markov_chains.npz (cases ``ab_*``, made by
oracle/make_golden.py from ``bayespy.utils.random.alpha_beta_recursion``).

Restates ``alpha_beta_recursion`` (utils/random.py:357-422), the moments of
``CategoricalMarkovChain`` (categorical_markov_chain.py:107-117), in NumPy float64, vectorised
over the chains and sequential over time, with a streaming log-sum-exp instead of the
reference's (plates, K, K) temporaries so that it also runs at sizes the reference cannot hold:

    logalpha_0 = logp0
    v_n[i, j]  = logalpha_n[i] + logP_n[i, j];  c_n = lse_ij v_n;  g -= c_n
    logalpha_{n+1}[j] = lse_i (v_n[i, j] - c_n)                       (:390-402)
    logbeta_{N-1} = 0;  logbeta_{n-1}[i] = lse_j (logbeta_n[j] + logP_n[i, j] - c'_n)   (:404-409)
    zz_n = softmax_ij (logalpha_n[i] + logbeta_n[j] + logP_n[i, j]);  z0 = rows of zz_0   (:411-420)
"""
import numpy as np


def _lse(x, axis):
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        return np.log(np.sum(np.exp(x - m), axis=axis)) + np.squeeze(m, axis=axis)


def alpha_beta_recursion(logp0, logP):
    """logp0 (..., K), logP (..., N, K, K) -> (z0 (..., K), zz (..., N, K, K), g (...,))."""
    logp0 = np.asarray(logp0, dtype=np.float64)
    logP = np.asarray(logP, dtype=np.float64)
    K = logp0.shape[-1]
    N = logP.shape[-3]
    plates = np.broadcast_shapes(logp0.shape[:-1], logP.shape[:-3])
    logP = np.broadcast_to(logP, plates + (N, K, K))
    la = np.empty(plates + (N, K))
    la[..., 0, :] = np.broadcast_to(logp0, plates + (K,))
    g = np.zeros(plates)
    for n in range(N):
        v = la[..., n, :, None] + logP[..., n, :, :]
        c = _lse(v.reshape(plates + (K * K,)), -1)
        g -= c
        if n + 1 < N:
            la[..., n + 1, :] = _lse(v - c[..., None, None], -2)
    zz = np.empty(plates + (N, K, K))
    lb = np.zeros(plates + (K,))
    for n in range(N - 1, -1, -1):
        w = la[..., n, :, None] + lb[..., None, :] + logP[..., n, :, :]
        m = np.max(w.reshape(plates + (K * K,)), axis=-1)[..., None, None]
        e = np.exp(w - m)
        zz[..., n, :, :] = e / np.sum(e, axis=(-1, -2), keepdims=True)
        if n > 0:
            v = lb[..., None, :] + logP[..., n, :, :]
            c = _lse(v.reshape(plates + (K * K,)), -1)
            lb = _lse(v - c[..., None, None], -1)
    z0 = np.sum(zz[..., 0, :, :], axis=-1)
    z0 = z0 / np.sum(z0, axis=-1, keepdims=True)
    return z0, zz, g
