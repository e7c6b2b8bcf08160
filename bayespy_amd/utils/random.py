"""
Device counterpart of ``bayespy.utils.random.alpha_beta_recursion`` (utils/random.py:357-422):
the forward-backward recursion of categorical Markov chains, one launch of the
``vmp_alpha_beta_recursion`` HIP kernel for all chains and all time instances.
"""
import ctypes

import numpy as np

from ..darray import DArray, asdarray, contiguous
from ..device import get_runtime
from .shapes import broadcasted_shape


def _shared_over(t, naxes):
    """True when the leading ``naxes`` axes of tensor ``t`` are all broadcast."""
    return all(t.shape[i] == 1 or t.stride(i) == 0 for i in range(naxes))


def alpha_beta_recursion(logp0, logP):
    """
    ``logp0``: (..., K) unnormalised log-probabilities of the first state;
    ``logP``: (..., N, K, K) with  logP[..., n, i, j] = log p(z_{n+1}=j | z_n=i) + the evidence
    of instance n+1.  Returns ``(z0, zz, g)``: q(z_0) (..., K), q(z_n, z_{n+1}) (..., N, K, K)
    and minus the log-normaliser (...,), for the broadcast plates of the two arguments.
    """
    logp0, logP = asdarray(logp0), asdarray(logP)
    if logp0.ndim < 1:
        logp0 = logp0.reshape((1,))
    if logP.ndim < 3:
        logP = logP.reshape((1,) * (3 - logP.ndim) + logP.shape)
    K = logp0.shape[-1]
    N = logP.shape[-3]
    if logP.shape[-2:] != (K, K):
        raise ValueError("Dimension mismatch %s != %s" % (logP.shape[-2:], (K, K)))
    plates = broadcasted_shape(logp0.shape[:-1], logP.shape[:-3])
    B = int(np.prod(plates, dtype=np.int64))
    rt = get_runtime()
    # first-state vector: shared by all chains, or one dense row per chain
    if _shared_over(logp0.t, logp0.ndim - 1):
        p0 = contiguous(DArray(logp0.t.reshape(-1)[:K] if logp0.t.is_contiguous() else
                               logp0.t[(0,) * (logp0.ndim - 1)]))
        p0_bs = 0
    else:
        p0 = contiguous(logp0.broadcast_to(plates + (K,)))
        p0_bs = K
    # transition slices: every K x K slice dense; chain and time axes may be shared
    tP = logP.t
    shared_b = _shared_over(tP, logP.ndim - 3)
    shared_t = N > 1 and tP.stride(-3) == 0
    if shared_b:
        sl = tP[(0,) * (logP.ndim - 3)]                 # (N, K, K)
        if shared_t:
            sl = sl[:1]
        Pd = contiguous(DArray(sl))
        P_bs, P_ts = 0, (0 if shared_t else K * K)
    else:
        Pd = contiguous(logP.broadcast_to(plates + (N, K, K)))
        P_bs, P_ts = N * K * K, K * K
    z0 = DArray.empty(plates + (K,))
    zz = DArray.empty(plates + (N, K, K))
    g = DArray.empty(plates)
    ws = rt.torch.empty(max(B * N * K, 1), dtype=rt.torch.float64, device=rt.device)
    rt.sync_stream()
    rt.note_reads([p0, Pd])
    rt.check(rt.lib.vmp_alpha_beta_recursion(
        rt.ctx, N, K, B, ctypes.c_void_p(p0.t.data_ptr()), p0_bs,
        ctypes.c_void_p(Pd.t.data_ptr()), P_bs, P_ts, ctypes.c_void_p(z0.t.data_ptr()),
        ctypes.c_void_p(zz.t.data_ptr()), ctypes.c_void_p(g.t.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8))
    return z0, zz, g


# ---------------------------------------------------------------------------
# Host-side data-generation helpers of bayespy.utils.random (set-up of demos and tests: masks,
# random covariance matrices, draws from discrete distributions).  NumPy on the host, the
# global numpy.random state like the reference; none of this is on the inference path.
# ---------------------------------------------------------------------------
def mask(*shape, p=0.5):
    """Boolean array, each element True with probability ``p`` (utils/random.py:45-56)."""
    return np.random.rand(*shape) < p


def covariance(D, size=(), nu=None):
    """Random SPD matrix (or a ``size`` stack of them) from an inverse-Wishart distribution
    with ``nu`` degrees of freedom, default ``D`` (utils/random.py:80-113)."""
    nu = D if nu is None else nu
    if nu < D:
        raise ValueError("nu must be greater than or equal to D")
    try:
        size = tuple(size)
    except TypeError:
        size = (size,)
    C = np.random.randn(*(size + (D, nu)))
    return np.linalg.inv(C @ np.swapaxes(C, -1, -2) / nu)


def correlation(D):
    """Random correlation matrix (utils/random.py:116-123)."""
    X = np.random.randn(D, D)
    X = X / np.sqrt(np.sum(X ** 2, axis=-1, keepdims=True))
    return X @ X.T


def orth(D):
    """Random orthogonal matrix (utils/random.py:208-213)."""
    return np.linalg.qr(np.random.randn(D, D))[0]


def svd(s):
    """Random matrix with the given singular values (utils/random.py:216-222)."""
    D = len(s)
    return (orth(D) * s) @ orth(D).T


def sphere(N=1):
    """N points uniform on the unit sphere as (latitude, longitude) in degrees
    (utils/random.py:225-233)."""
    lon = np.random.uniform(-180, 180, N)
    lat = np.arccos(np.random.uniform(-1, 1, N)) * 180 / np.pi - 90
    return lat, lon


def bernoulli(p, size=None):
    """Bernoulli draws (boolean array) with success probability ``p`` (utils/random.py:236-244)."""
    if isinstance(size, int):
        size = (size,)
    if size is None:
        size = np.shape(p)
    return np.random.rand(*size) < p


def categorical(p, size=None):
    """Class labels drawn with (unnormalised) probabilities ``p[..., k]``
    (utils/random.py:247-287)."""
    p = np.asarray(p, dtype=np.float64)
    if size is None:
        size = p.shape[:-1]
    if isinstance(size, int):
        size = (size,)
    size = tuple(size)
    if np.any(p < 0):
        raise ValueError("Array contains negative probabilities")
    try:
        ok = broadcasted_shape(p.shape[:-1], size) == size
    except ValueError:
        ok = False
    if not ok:
        raise ValueError("Probability array shape and requested size are inconsistent")
    c = np.cumsum(p / np.sum(p, axis=-1, keepdims=True), axis=-1)
    x = np.random.rand(*size)
    return np.sum(x[..., None] > np.broadcast_to(c, size + c.shape[-1:]), axis=-1) \
        .clip(0, p.shape[-1] - 1).astype(int)


def multinomial(n, p, size=None):
    """Count vectors of ``n`` trials with probabilities ``p`` (utils/random.py:290-316)."""
    n, p = np.asarray(n), np.asarray(p, dtype=np.float64)
    k = p.shape[-1]
    if size is None:
        size = broadcasted_shape(n.shape, p.shape[:-1])
    size = tuple(size)
    n = np.broadcast_to(n, size)
    p = np.broadcast_to(p, size + (k,))
    x = np.empty(size + (k,))
    for i in np.ndindex(*size):
        x[i] = np.random.multinomial(n[i], p[i])
    return x.astype(int)


def gamma(a, b, size=None):
    """Gamma draws with shape ``a`` and SCALE ``b`` (utils/random.py:319-326)."""
    x = np.random.gamma(a, b, size=size)
    if np.any(x == 0):
        raise RuntimeError("Numerically zero samples. Try using a larger shape parameter in "
                           "the gamma distribution.")
    return x


def dirichlet(alpha, size=None):
    """Dirichlet draws with concentration ``alpha`` (utils/random.py:329-347)."""
    alpha = np.asarray(alpha, dtype=np.float64)
    if isinstance(size, int):
        size = (size,)
    size = alpha.shape if size is None else tuple(size) + alpha.shape[-1:]
    p = np.random.gamma(alpha, size=size)
    s = np.sum(p, axis=-1, keepdims=True)
    if np.any(s == 0):
        raise RuntimeError("Numerically zero samples. Try using a larger Dirichlet "
                           "concentration parameter value.")
    return p / s


def logodds_to_probability(x):
    """1 / (1 + exp(-x))  (utils/random.py:350-354)."""
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))
