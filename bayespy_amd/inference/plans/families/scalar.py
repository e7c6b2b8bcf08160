"""Gamma, Wishart, Dirichlet, Categorical, Multinomial (gamma.py:90-211, wishart.py:118-225,
dirichlet.py:107-231, categorical.py:25-126, multinomial.py:62-231)."""
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


class GammaFamily(Family):
    """gamma.py:90-211."""

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, fuse(lambda a: da.gammaln(a), v)]          # GammaPriorMoments, gamma.py:33-58
        return [v, fuse(lambda b: da.log(b), v)]

    def phi_from_parents(self, up):
        return [fuse(lambda b: -b, up[1][0]), fuse(lambda a: 1.0 * a, up[0][0])]

    def moments_and_cgf(self, phi):
        u0 = fuse(lambda p0, p1: p1 / (-p0), phi[0], phi[1])
        u1 = fuse(lambda p0, p1: da.digamma(p1) - da.log(-p0), phi[0], phi[1])
        g = fuse(lambda p0, p1: p1 * da.log(-p0) - da.gammaln(p1), phi[0], phi[1])
        return [u0, u1], g

    def cgf_from_parents(self, up):
        return fuse(lambda a, lga, logb: a * logb - lga, up[0][0], up[0][1], up[1][1])

    missing_fill = 1.0       # finite log at masked-out entries

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        _check_device(x, 'negative', ValueError, "Values must be positive")
        logx = fuse(lambda v: da.log(v), x)
        return [x, logx], fuse(lambda l: -l, logx)

    def message_to_parent(self, index, u, up):
        if index == 1:
            return [fuse(lambda x: -x, u[0]), up[0][0]]
        raise NotImplementedError('message from Gamma to its shape parameter')

    def gradient(self, rg, u, phi):
        # gamma.py:183-211
        d0 = fuse(lambda a, b, p0, p1: a * p1 / (p0 * p0) - b / p0, rg[0], rg[1], phi[0], phi[1])
        d1 = fuse(lambda a, b, p0, p1: b * da.trigamma(p1) - a / p0, rg[0], rg[1], phi[0], phi[1])
        return [d0, d1]



class WishartFamily(Family):
    """wishart.py:118-225."""

    def __init__(self, node):
        super().__init__(node)
        self.D = node.dims[0][0]

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, _multigammaln(fuse(lambda n: 0.5 * n, v), self.D)]   # wishart.py:96-115
        return [v, linalg.chol_logdet(linalg.chol(v))]

    def phi_from_parents(self, up):
        return [fuse(lambda V: -0.5 * V, up[1][0]), fuse(lambda n: 0.5 * n, up[0][0])]

    def moments_and_cgf(self, phi):
        U = linalg.chol(fuse(lambda p: -p, phi[0]))
        ld = linalg.chol_logdet(U)
        p1 = _arr(phi[1])
        u0 = fuse(lambda n, c: n * c, _trail(p1, 2), linalg.chol_inv(U))
        u1 = fuse(lambda l, md: -l + md, ld, misc.multidigamma(p1, self.D))
        g = fuse(lambda n, l, mg: n * l - mg, p1, ld, _multigammaln(p1, self.D))
        return [u0, u1], g

    def cgf_from_parents(self, up):
        n, gln = up[0]
        ldV = up[1][1]
        k = self.D
        return fuse(lambda n_, l, g_: 0.5 * n_ * l - 0.5 * k * np.log(2.0) * n_ - g_, n, ldV, gln)

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        ld = linalg.chol_logdet(linalg.chol(x))
        return [x, ld], fuse(lambda l: -(self.D + 1) / 2.0 * l, ld)

    def message_to_parent(self, index, u, up):
        # wishart.py:142-150: to the inverse scale matrix V (a Wishart node): [-<Lambda>/2, n/2]
        if index != 1:
            raise NotImplementedError('the degrees of freedom of a Wishart node are numeric')
        return [fuse(lambda l: -0.5 * l, _arr(u[0])), fuse(lambda n: 0.5 * n, _arr(up[0][0]))]


class DirichletFamily(Family):
    """dirichlet.py:107-231."""

    def constant_moments(self, index, value):
        v = _arr(value)
        return [v]

    def phi_from_parents(self, up):
        return [up[0][0]]

    def moments_and_cgf(self, phi):
        p = _arr(phi[0])
        _check_device(p, 'nonpositive', ValueError, "Natural parameters should be positive")
        s = misc.sum_multiply(p, axis=-1, keepdims=True)
        u0 = fuse(lambda a, t: da.digamma(a) - da.digamma(t), p, s)
        lg = misc.sum_multiply(fuse(lambda a: da.gammaln(a), p), axis=-1)
        g = fuse(lambda t, l: da.gammaln(t) - l, s.reshape(s.shape[:-1]), lg)
        return [u0], g

    def cgf_from_parents(self, up):
        a = _arr(up[0][0])
        s = misc.sum_multiply(a, axis=-1)
        lg = misc.sum_multiply(fuse(lambda v: da.gammaln(v), a), axis=-1)
        return fuse(lambda t, l: da.gammaln(t) - l, s, lg)

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        logp = fuse(lambda v: da.log(v), x)
        return [logp], fuse(lambda s: -s, misc.sum_multiply(logp, axis=-1))

    def message_to_parent(self, index, u, up):
        raise NotImplementedError('Dirichlet concentration is a constant in the built path')

    def gradient(self, rg, u, phi):
        # dirichlet.py:213-231
        p = _arr(phi[0])
        s = misc.sum_multiply(p, axis=-1, keepdims=True)
        return [fuse(lambda g, a, t: g * (da.trigamma(a) - da.trigamma(t)), rg[0], p, s)]


class CategoricalFamily(Family):
    """categorical.py:25-126, multinomial.py:62-231 (one trial)."""

    def __init__(self, node):
        super().__init__(node)
        self.K = node.dims[0][0]

    def constant_moments(self, index, value):
        return [fuse(lambda p: da.log(p), _arr(value))]

    def phi_from_parents(self, up):
        return [up[0][0]]

    def moments_and_cgf(self, phi):
        p, lse = misc.normalized_exp(_arr(phi[0]))
        return [p], fuse(lambda l: -l, lse.reshape(lse.shape[:-1]))

    def cgf_from_parents(self, up):
        return 0.0

    def fixed_moments_and_f(self, x):
        return [misc.onehot(np.asarray(x), self.K)], 0.0

    def message_to_parent(self, index, u, up):
        return [u[0]]

    _trials = 1.0

    def gradient(self, rg, u, phi):
        # multinomial.py:161-212:  u_i (g_i - sum_j g_j u_j / N)
        t = misc.sum_multiply(_arr(rg[0]), _arr(u[0]), axis=-1, keepdims=True)
        n = self._trials if not isinstance(self._trials, DArray) else _trail(self._trials, 1)
        return [fuse(lambda u_, g, t_, n_: u_ * (g - t_ / n_), u[0], rg[0], t, n)]


class MultinomialFamily(CategoricalFamily):
    """multinomial.py:62-231 with N trials (an integer or an integer array over the plates)."""

    def __init__(self, node):
        super().__init__(node)
        self.trials = np.asarray(node.trials, dtype=np.float64)
        self.Nd = DArray.from_host(self.trials)
        self._trials = self.Nd

    def moments_and_cgf(self, phi):
        p, lse = misc.normalized_exp(_arr(phi[0]))
        u0 = fuse(lambda n, q: n * q, _trail(self.Nd, 1), p)
        return [u0], fuse(lambda n, l: -(n * l), self.Nd, lse.reshape(lse.shape[:-1]))

    def fixed_moments_and_f(self, x):
        # f = log N! - sum_k log x_k!   (multinomial.py:153-155)
        x = _arr(np.asarray(x, dtype=np.float64))
        lg = misc.sum_multiply(fuse(lambda c: da.gammaln(c + 1.0), x), axis=-1)
        return [x], fuse(lambda n, s_: da.gammaln(n + 1.0) - s_, self.Nd, lg)
