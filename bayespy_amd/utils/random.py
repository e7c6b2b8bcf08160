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
    rt.check(rt.lib.vmp_alpha_beta_recursion(
        rt.ctx, N, K, B, ctypes.c_void_p(p0.t.data_ptr()), p0_bs,
        ctypes.c_void_p(Pd.t.data_ptr()), P_bs, P_ts, ctypes.c_void_p(z0.t.data_ptr()),
        ctypes.c_void_p(zz.t.data_ptr()), ctypes.c_void_p(g.t.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8))
    return z0, zz, g
