"""GaussianMarkovChain and its Gaussian view (gaussian_markov_chain.py:270-707, :1988-2098)."""
import os

import numpy as np

from .... import darray as da
from ....darray import DArray, fuse, contiguous
from ....nodes.node import Constant, Stochastic
from ....nodes.gaussian import is_gaussian_gamma
from ....utils import misc, linalg
from ....utils.shapes import broadcasted_shape, is_shape_subset, multiplier_factor
from ..lazy import (DerivedArray,
                    FactoredMoment,
                    LOG2PI,
                    LazyContract,
                    LazySum,
                    PlateSums,
                    Terms,
                    _CONSTS,
                    _Deferred,
                    _LazyList,
                    _arr,
                    _check_device,
                    _const,
                    _diag2,
                    _eye,
                    _factored_min_plates,
                    _gaussian_gradient,
                    _gaussian_q_term,
                    _inner_second,
                    _is_lazy,
                    _lazy_mvdot,
                    _multigammaln,
                    _ones,
                    _shape,
                    _sum_last,
                    _trail,
                    _wsum)
from .base import Family


class GaussianMarkovChainFamily(Family):
    """gaussian_markov_chain.py:270-707 with the joint-parent wrappers folded in; the
    smoother (compute_moments_and_cgf, :89-123) is ``linalg.block_banded_solve``."""

    def __init__(self, node):
        super().__init__(node)
        self.N, self.D = node.N, node.D
        N = self.N
        e0 = np.zeros(N); e0[0] = 1.0
        enl = np.ones(N); enl[-1] = 0.0
        self._e0v = DArray.from_host(e0.reshape(N, 1))
        self._e0 = DArray.from_host(e0.reshape(N, 1, 1))
        self._en0 = DArray.from_host((1.0 - e0).reshape(N, 1, 1))
        self._enl = DArray.from_host(enl.reshape(N, 1, 1))

    def plates_to_parent(self, index):
        if index < 2:
            return self.node.plates
        return self.node.plates + (self.N - 1, self.D)

    def mask_to_parent(self, index, mask):
        if index < 2:
            return mask
        return mask.reshape(mask.shape + (1, 1))

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, linalg.outer(v, v)]
        if index == 1:
            return [v, linalg.chol_logdet(linalg.chol(v))]
        if index == 2:
            return [v, linalg.outer(v, v)]
        return [v, fuse(lambda a: da.log(a), v)]

    def _time_axis(self, x, tail, index):
        """Parent moments of A / nu with full row (and variable) axes ``tail`` and an explicit
        (unit) time axis before the row axis; plate-compressed moments are expanded as views."""
        x = _arr(x)
        par = self.node.parents[index]
        npl = len(par.value.shape) - (1 if index == 2 else 0) if isinstance(par, Constant) \
            else len(par.plates)
        nt = len(tail)
        lead = x.shape[:max(0, x.ndim - nt)]
        x = x.broadcast_to(lead + tuple(tail))
        if npl >= 2 and len(lead) >= 1:
            return x                      # (..., 1, D, ...) already carries the time axis
        return x.reshape(lead + (1,) + tuple(tail)) if x.t.is_contiguous() else \
            DArray(x.t.unsqueeze(len(lead)))

    def _dyn(self, up):
        D = self.D
        Am = self._time_axis(up[2][0], (D, D), 2)          # (..., 1, D, D)
        AA = self._time_axis(up[2][1], (D, D, D), 2)       # (..., 1, D, D, D)
        nu = self._time_axis(up[3][0], (D,), 3)            # (..., 1, D)
        lognu = self._time_axis(up[3][1], (D,), 3)
        return Am, AA, nu, lognu

    def phi_from_parents(self, up):
        m, Lam = up[0][0], up[1][0]
        Am, AA, nu, _ = self._dyn(up)
        Lm = linalg.mvdot(Lam, m)
        phi0 = fuse(lambda e, v: e * v, self._e0v, _arr(Lm).reshape(_shape(Lm)[:-1] + (1, self.D)))
        nuAA = misc.sum_multiply(_trail(nu, 2), AA, axis=-3)                  # (..., 1, D, D)
        dnu = misc.diag(nu, ndim=1)                                           # (..., 1, D, D)
        L = _arr(Lam)
        L = L.reshape(L.shape[:-2] + (1,) + L.shape[-2:])
        phi1 = fuse(lambda a, b, c, l, d, q: -0.5 * (a * l + b * d + c * q),
                    self._e0, self._en0, self._enl, L, dnu, nuAA)
        phi2 = fuse(lambda n, a: n * a, _trail(nu, 1), Am).swapaxes(-1, -2)   # nu_i A_ij -> [j][i]
        return [phi0, phi1, phi2]

    def moments_and_cgf(self, phi):
        A = fuse(lambda p: -2 * p, phi[1])
        B = fuse(lambda p: -p, phi[2])
        V, C, x, ld = linalg.block_banded_solve(A, B, phi[0])
        D = self.D
        xa = x.reshape(x.shape + (1,))
        xb = x.reshape(x.shape[:-1] + (1, D))
        u1 = fuse(lambda a, b, c: a * b + c, xa, xb, V)
        u2 = fuse(lambda a, b, c: a * b + c, xa[..., :-1, :, :], xb[..., 1:, :, :], C)
        g = fuse(lambda s, l: -0.5 * s + 0.5 * l,
                 misc.sum_multiply(x, phi[0], axis=(-1, -2)), ld)
        return [x, u1, u2], g

    def cgf_from_parents(self, up):
        mm = up[0][1]
        Lam, logdet = up[1]
        _, _, _, lognu = self._dyn(up)
        s = misc.sum_multiply(lognu, axis=(-1, -2))
        return fuse(lambda t, ld, ln: -0.5 * t + 0.5 * ld + 0.5 * (self.N - 1) * ln,
                    misc.sum_multiply(Lam, mm, axis=(-1, -2)), logdet, s)

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        if x.shape[-2:] != (self.N, self.D):
            raise ValueError("Invalid shape")
        D = self.D
        xa = x.reshape(x.shape + (1,))
        xb = x.reshape(x.shape[:-1] + (1, D))
        u1 = fuse(lambda a, b: a * b, xa, xb)
        u2 = fuse(lambda a, b: a * b, xa[..., :-1, :, :], xb[..., 1:, :, :])
        return [x, u1, u2], -0.5 * self.N * D * LOG2PI

    def message_to_parent(self, index, u, up):
        x, XX, XpXn = _arr(u[0]), _arr(u[1]), _arr(u[2])
        if index < 2:
            # the initial state is a Gaussian(mu, Lambda) variable (:443-460)
            x0, x0x0 = x[..., 0, :], XX[..., 0, :, :]
            m, mm = up[0]
            L = up[1][0]
            if index == 0:
                return [linalg.mvdot(L, x0), fuse(lambda l: -0.5 * l, L)]
            xm, mx = linalg.outer(x0, m), linalg.outer(m, x0)
            return [fuse(lambda a, b, c, d: -0.5 * (a - b - c + d), x0x0, xm, mx, mm), 0.5]
        Am, AA, nu, _ = self._dyn(up)
        XnXp = XpXn.swapaxes(-1, -2)                     # [i][j] = <x_n[i] x_{n-1}[j]>
        XXp = XX[..., :-1, :, :]
        if index == 2:
            # to the dynamics matrix, weighted by the innovation precision (:462-475,
            # gaussian.py:2354-2360)
            m0 = (XnXp, _trail(nu, 1))
            m1 = (fuse(lambda q: -0.5 * q, XXp.reshape(XXp.shape[:-2] + (1,) + XXp.shape[-2:])),
                  _trail(nu, 2))
            return [m0, m1]
        t1 = misc.sum_multiply(XnXp, Am, axis=-1)
        t2 = misc.sum_multiply(XXp.reshape(XXp.shape[:-2] + (1,) + XXp.shape[-2:]), AA,
                               axis=(-1, -2))
        t3 = misc.get_diag(XX[..., 1:, :, :], ndim=1)
        return [fuse(lambda a, b, c: a - 0.5 * b - 0.5 * c, t1, t2, t3), 0.5]


class ChainToGaussianFamily:
    """``_MarkovChainToGaussian`` (gaussian_markov_chain.py:1988-2098): the time axis of a
    chain becomes the last plate; the cross-time moment is dropped."""
    deterministic = True

    def __init__(self, node):
        self.node = node

    def moments(self, ups):
        return list(ups[0][:2])

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        return np.any(mask, axis=-1) if mask.ndim >= 1 else mask

    def message_to_parent(self, index, m_child, ups, mask=None):
        out = []
        for i, m in enumerate(m_child[:2]):
            if m is None:
                out.append(None)
            elif mask is not None:
                # the last plate turns into a variable axis: apply its mask here
                out.append(fuse(lambda a, w: a * w, _arr(m), _trail(mask, 1 + i)))
            else:
                out.append(m)
        return out + [None]
