"""
Batched linear algebra on device arrays.

Device counterparts of ``bayespy.utils.linalg``: ``chol`` (:31), ``chol_solve``
(:66), ``chol_inv`` (:174), ``chol_logdet`` (:209), ``inner`` (:299), ``outer``
(:309), ``mvdot`` (:407), ``mmdot`` (:430), ``transpose`` (:444).  The reference
loops in Python over the plates and calls SciPy once per matrix; here one
``vmp_spd_batched`` launch handles every plate (one workgroup or wavefront per
matrix).  ``chol`` returns an opaque factor handle, like the reference's ``U``.
"""
import ctypes

import numpy as np

from .. import _lib
from ..darray import DArray, asdarray, contiguous, fuse
from ..device import get_runtime
from .misc import sum_multiply


class CholFactor:
    """Handle returned by :func:`chol`; inverse and log-determinant are computed
    together by one batched kernel on first use."""

    def __init__(self, C):
        self.C = contiguous(asdarray(C))
        if self.C.ndim < 2 or self.C.shape[-1] != self.C.shape[-2]:
            raise ValueError('chol needs (..., n, n) arrays')
        self._inv = None
        self._logdet = None
        self._factor()

    def _factor(self):
        rt = get_runtime()
        C = self.C
        n = C.shape[-1]
        batch = C.size // (n * n) if n else 0
        inv = DArray.empty(C.shape)
        logdet = DArray.empty(C.shape[:-2])
        # (every flag is written by the kernel; an empty batch never reads it)
        info = (rt.torch.empty(batch, dtype=rt.torch.int32, device=rt.device) if batch else
                rt.torch.zeros(1, dtype=rt.torch.int32, device=rt.device))
        rt.sync_stream()
        rt.note_reads([C])
        rt.check(rt.lib.vmp_spd_batched(rt.ctx, n, batch, ctypes.c_void_p(C.t.data_ptr()),
                                        ctypes.c_void_p(inv.t.data_ptr()),
                                        ctypes.c_void_p(logdet.t.data_ptr()),
                                        ctypes.c_void_p(info.data_ptr())))
        rt.keep_until_flush([C], (inv, logdet, info), kind='sm')   # a few small matrices are queued
        # same failure the reference reports (utils/linalg.py:58-59); inside a plan
        # operation the flag is read together with the operation's other checks
        rt.defer_check(info, _lib.NotPositiveDefiniteError, "Matrix not positive definite")
        self._inv, self._logdet = inv, logdet


def gaussian_moments(phi0, phi1):
    """Moments and log-normaliser of Gaussians with a full covariance per plate from their
    natural parameters, in ONE launch when the fused kernel applies (dense (..., n) / (..., n, n)
    arrays of the same plates, 8 < n <= 32, a chip-filling number of plates); returns
    ``(u0, u1, g)`` or ``None`` when the caller should use chol / chol_solve / outer
    (GaussianARDDistribution.compute_moments_and_cgf, gaussian.py:680-706)."""
    phi0, phi1 = asdarray(phi0), asdarray(phi1)
    n = phi1.shape[-1]
    if not (8 < n <= 32) or phi1.shape[-2] != n or phi0.shape[-1] != n:
        return None
    plates = phi1.shape[:-2]
    if phi0.shape[:-1] != plates:
        return None
    rt = get_runtime()
    batch = int(np.prod(plates)) if plates else 1
    if batch < 4 * 256 or rt.lib is None:
        return None
    p0, p1 = contiguous(phi0), contiguous(phi1)
    u0, u1, g = DArray.empty(p0.shape), DArray.empty(p1.shape), DArray.empty(plates)
    info = rt.torch.zeros(batch, dtype=rt.torch.int32, device=rt.device)
    rt.sync_stream()
    rt.note_reads([p0, p1])
    rt.check(rt.lib.vmp_gaussian_moments(rt.ctx, n, batch, ctypes.c_void_p(p0.t.data_ptr()),
                                         ctypes.c_void_p(p1.t.data_ptr()),
                                         ctypes.c_void_p(u0.t.data_ptr()),
                                         ctypes.c_void_p(u1.t.data_ptr()),
                                         ctypes.c_void_p(g.t.data_ptr()),
                                         ctypes.c_void_p(info.data_ptr())))
    rt.defer_check(info, _lib.NotPositiveDefiniteError, "Matrix not positive definite")
    return u0, u1, g


def chol(C, ndim=1):
    if ndim != 1:
        raise NotImplementedError('chol with ndim != 1')
    return CholFactor(C)


def chol_inv(U, ndim=1):
    return U._inv


def chol_logdet(U, ndim=1):
    return U._logdet


def chol_solve(U, b, ndim=1, out=None, matrix=False):
    """Solve C x = b for every plate (broadcasting like the reference)."""
    b = asdarray(b)
    if matrix:
        return mmdot(U._inv, b)
    return mvdot(U._inv, b)


def inner(*args, ndim=1):
    """Sum of the elementwise product over the trailing ``ndim`` axes (linalg.py:299-306)."""
    if ndim == 0:
        return sum_multiply(*args, axis=None, sumaxis=False) if len(args) > 1 else asdarray(args[0])
    return sum_multiply(*args, axis=tuple(range(-ndim, 0)))


def outer(A, B, ndim=1):
    """Outer product over the trailing ``ndim`` axes (linalg.py:309-334)."""
    A, B = asdarray(A), asdarray(B)
    a = A.reshape(A.shape + (1,) * ndim)
    sb = B.shape
    b = B.reshape(sb[:len(sb) - ndim] + (1,) * ndim + sb[len(sb) - ndim:])
    return fuse(lambda x, y: x * y, a, b)


def mvdot(A, b, ndim=1):
    """(..., M, N) x (..., N) -> (..., M)  (linalg.py:407-427)."""
    A, b = asdarray(A), asdarray(b)
    if ndim == 0:
        return fuse(lambda x, y: x * y, A, b)
    if ndim != 1:
        # several variable axes: the same product on the flattened axes
        shape = b.shape[b.ndim - ndim:]
        D = 1
        for d in shape:
            D *= d
        Af = A.reshape(A.shape[:A.ndim - 2 * ndim] + (D, D))
        bf = b.reshape(b.shape[:b.ndim - ndim] + (D,))
        r = mvdot(Af, bf)
        return r.reshape(r.shape[:-1] + tuple(shape))
    bb = b.reshape(b.shape[:-1] + (1, b.shape[-1]))
    return sum_multiply(A, bb, axis=-1)


def mmdot(A, B, ndim=1):
    """(..., M, K) x (..., K, N) -> (..., M, N)  (linalg.py:430-441)."""
    if ndim != 1:
        raise NotImplementedError
    A, B = asdarray(A), asdarray(B)
    a = A.reshape(A.shape + (1,))                         # (..., M, K, 1)
    b = B.reshape(B.shape[:-2] + (1,) + B.shape[-2:])     # (..., 1, K, N)
    return sum_multiply(a, b, axis=-2)


def dot(*arrays):
    out = asdarray(arrays[0])
    for a in arrays[1:]:
        out = mmdot(out, a)
    return out


def transpose(X, ndim=1):
    X = asdarray(X)
    if ndim == 0:
        return X
    if ndim != 1:
        shape = X.shape[X.ndim - ndim:]
        D = 1
        for d in shape:
            D *= d
        lead = X.shape[:X.ndim - 2 * ndim]
        return X.reshape(lead + (D, D)).swapaxes(-1, -2).reshape(lead + tuple(shape) * 2)
    return X.swapaxes(-1, -2)


def logdet_cov(C):
    return chol_logdet(chol(C))


def block_banded_solve(A, B, y):
    """
    Solve a symmetric positive-definite block-tridiagonal system for every plate and
    return the diagonal / super-diagonal blocks of its inverse, the solution and the
    log-determinant -- the Kalman filter + RTS smoother of the Gaussian Markov chain
    (reference: utils/linalg.py:468-575).

    A (..., N, D, D) diagonal blocks, B (..., N-1, D, D) super-diagonal blocks,
    y (..., N, D).  One ``vmp_block_banded_solve`` launch sequence; the matrix
    recursions run once when A and B carry no plates (shared dynamics).
    """
    from .shapes import broadcasted_shape
    rt = get_runtime()
    A, B, y = asdarray(A), asdarray(B), asdarray(y)
    N, D = y.shape[-2], y.shape[-1]
    # a unit time axis means "the same block at every time instance"
    if A.ndim < 3 or A.shape[-3] not in (1, N):
        raise ValueError("The number of diagonal blocks is incorrect")
    if A.shape[-2:] != (D, D):
        raise ValueError("The diagonal blocks have wrong shape")
    if N > 1 and (B.ndim < 3 or B.shape[-3] not in (1, N - 1) or B.shape[-2:] != (D, D)):
        raise ValueError("The super-diagonal blocks have wrong shape")
    plates_m = broadcasted_shape(A.shape[:-3], B.shape[:-3])
    plates_y = broadcasted_shape(plates_m, y.shape[:-2])
    shared = all(p == 1 for p in plates_m)
    pm = plates_m if shared else plates_y
    Ac = contiguous(A.broadcast_to(pm + (N, D, D)))
    Bc = contiguous(B.broadcast_to(pm + (max(N - 1, 1), D, D))) if N > 1 else Ac
    yc = contiguous(y.broadcast_to(plates_y + (N, D)))
    nm = int(np.prod(pm)) if pm else 1
    ny = int(np.prod(plates_y)) if plates_y else 1
    V = DArray.empty(pm + (N, D, D))
    C = DArray.empty(pm + (max(N - 1, 0), D, D))
    x = DArray.empty(plates_y + (N, D))
    ldet = DArray.empty(pm)
    info = rt.torch.zeros(max(nm, 1), dtype=rt.torch.int32, device=rt.device)
    rt.sync_stream()
    rt.note_reads([Ac, Bc, yc])
    rt.check(rt.lib.vmp_block_banded_solve(
        rt.ctx, N, D, nm, ny, ctypes.c_void_p(Ac.t.data_ptr()), ctypes.c_void_p(Bc.t.data_ptr()),
        ctypes.c_void_p(yc.t.data_ptr()), ctypes.c_void_p(V.t.data_ptr()),
        ctypes.c_void_p(C.t.data_ptr()), ctypes.c_void_p(x.t.data_ptr()),
        ctypes.c_void_p(ldet.t.data_ptr()), ctypes.c_void_p(info.data_ptr())))
    # read with the other validity flags of the running plan operation (no sync per solve; inside a
    # recorded sweep the flag is one of the graph's outputs)
    rt.defer_check(info, _lib.NotPositiveDefiniteError, "Matrix not positive definite")
    return V, C, x, ldet
