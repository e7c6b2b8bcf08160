"""GaussianARD, Gaussian and the Gaussian-gamma nodes with the joint-parent wrappers folded in
(gaussian.py:293-1136, :2226-2527)."""
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


class GaussianARDFamily(Family):
    """gaussian.py:576-889 with the wrapper gaussian.py:2299-2371 folded in."""

    def __init__(self, node):
        super().__init__(node)
        self.shape = node.shape
        self.ndim = node.ndim
        mu = node.parents[0]
        # a Gaussian-gamma mean parent (a GaussianGamma node, or the explicit converter /
        # wrapper nodes): its own precision scale tau multiplies this node's alpha
        # (WrapToGaussianGamma, gaussian.py:2299-2371); scalar-valued like in the reference
        # (parent_moments = GaussianGammaMoments(()), gaussian.py:1646)
        self.mu_gg = is_gaussian_gamma(mu)
        if self.mu_gg and (len(mu.dims[0]) != 0 or self.ndim != 0):
            raise NotImplementedError('a Gaussian-gamma mean parent must be scalar-valued (ndim=0) '
                                      'under a scalar-valued GaussianARD')
        # a Gaussian mean parent with k variable axes: they are the LAST k axes of this node's
        # (plates + shape) grid; with k > ndim (e.g. the reference's default ndim = 0 under a
        # vector-valued mean, gaussian.py:1617-1640) the leading k - ndim of them are plates here
        self.mu_ndim = 0 if isinstance(mu, Constant) else len(mu.dims[0])
        self.mu_shape = () if isinstance(mu, Constant) else tuple(mu.dims[0])
        if 0 < self.mu_ndim < self.ndim:
            raise NotImplementedError('mean parent with %d variable axes for a node with %d'
                                      % (self.mu_ndim, self.ndim))

    def plates_to_parent(self, index):
        grid = self.node.plates + self.shape
        if index == 0 and self.mu_ndim > 0:
            return grid[:len(grid) - self.mu_ndim]
        return grid

    def mask_to_parent(self, index, mask):
        if index == 0 and self.mu_ndim > 0:
            j = self.mu_ndim - self.ndim
            mask = np.asarray(mask)
            if j > 0 and mask.ndim > 0:
                # plates of this node that are variable axes of the mean: "sum" over them
                mask = np.any(mask, axis=tuple(range(-min(j, mask.ndim), 0)))
            return mask
        return mask.reshape(mask.shape + (1,) * self.ndim) if self.ndim else mask

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, fuse(lambda m: m * m, v)]
        return [v, fuse(lambda a: da.log(a), v)]

    def _mu(self, up):
        """(m, m2) elementwise over plates + shape."""
        m, mm = up[0]
        if self.mu_ndim > 0:
            return m, _diag2(mm, self.mu_ndim)
        return m, mm

    def phi_from_parents(self, up):
        if self.mu_gg:
            tm, _, t, _ = up[0]
            a = up[1][0]
            return [fuse(lambda a_, m_: a_ * m_, a, tm), fuse(lambda a_, t_: -0.5 * a_ * t_, a, t)]
        m, _ = self._mu(up)
        a = up[1][0]
        if self.ndim == 0:
            return [fuse(lambda a_, m_: a_ * m_, a, m), fuse(lambda a_: -0.5 * a_, a)]
        ones = _ones(self.shape)
        phi0 = fuse(lambda a_, m_, o: a_ * m_ * o, a, m, ones)
        d = fuse(lambda a_, o: -0.5 * a_ * o, a, ones)
        return [phi0, misc.diag(d, ndim=self.ndim)]

    def moments_and_cgf(self, phi):
        if self.ndim == 0:
            u0 = fuse(lambda p0, p1: -p0 / (2 * p1), phi[0], phi[1])
            u1 = fuse(lambda u, p1: u * u - 1.0 / (2 * p1), u0, phi[1])
            g = fuse(lambda u, p0, p1: -0.5 * u * p0 + 0.5 * da.log(-2 * p1), u0, phi[0], phi[1])
            return [u0, u1], g
        D = int(np.prod(self.shape))
        p0 = _arr(phi[0])
        p1 = _arr(phi[1])
        p0f = p0.reshape(p0.shape[:p0.ndim - self.ndim] + (D,))
        p1f = p1.reshape(p1.shape[:p1.ndim - 2 * self.ndim] + (D, D))
        fused = linalg.gaussian_moments(p0f, p1f)     # one launch for per-plate posteriors
        if fused is not None:
            u0, u1, g = fused
            return [u0.reshape(u0.shape[:-1] + self.shape),
                    u1.reshape(u1.shape[:-2] + self.shape + self.shape)], g
        U = linalg.chol(fuse(lambda p: -2 * p, p1f))
        cov = linalg.chol_inv(U)
        u0 = linalg.chol_solve(U, p0f)
        ld = linalg.chol_logdet(U)
        g = fuse(lambda s, ld_: -0.5 * s + 0.5 * ld_, misc.sum_multiply(u0, p0f, axis=-1), ld)
        # one covariance for many plates (a scalar mask: the precision carries no plate axis where
        # the mean does): keep <x x^T> as (Cov, <x>) -- see FactoredMoment
        pl0, pl1 = u0.shape[:-1], cov.shape[:-2]
        pl1 = (1,) * (len(pl0) - len(pl1)) + tuple(pl1)
        shared = [a for a, b in zip(pl0, pl1) if b == 1 and a > 1]
        if len(pl1) == len(pl0) and shared and int(np.prod(shared)) >= _factored_min_plates():
            u0 = u0.reshape(u0.shape[:-1] + self.shape)
            covs = cov.reshape(pl1 + self.shape + self.shape)
            return [u0, FactoredMoment(covs, u0, self.ndim, logdet_prec=ld.reshape(pl1))], g
        u1 = fuse(lambda a, b, c: a * b + c, _trail(u0, 1), u0.reshape(u0.shape[:-1] + (1, D)), cov)
        u0 = u0.reshape(u0.shape[:-1] + self.shape)
        u1 = u1.reshape(u1.shape[:-2] + self.shape + self.shape)
        return [u0, u1], g

    def q_term(self, phi, u, g):
        return _gaussian_q_term(self.ndim, self.shape, phi, u, g)

    def gradient(self, rg, u, phi):
        return _gaussian_gradient(rg, u, self.ndim, self.shape)

    def cgf_from_parents(self, up):
        if self.mu_gg:
            _, tmm, _, lt = up[0]
            a, loga = up[1]
            return fuse(lambda a_, q, la, lt_: -0.5 * a_ * q + 0.5 * (la + lt_), a, tmm, loga, lt)
        m, m2 = self._mu(up)
        a, loga = up[1]
        if self.ndim == 0:
            return fuse(lambda a_, q, la: -0.5 * a_ * q + 0.5 * la, a, m2, loga)
        t = fuse(lambda a_, q, la, o: (-0.5 * a_ * q + 0.5 * la) * o, a, m2, loga,
                 _ones(self.shape))
        return _sum_last(t, self.ndim)

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        if self.ndim > 0 and x.shape[x.ndim - self.ndim:] != self.shape:
            raise ValueError("Invalid shape")
        k = int(np.prod(self.shape)) if self.ndim else 1
        if self.ndim and x.size >= k * _factored_min_plates():
            # delta moments x x^T of many plates: the factored form with a zero covariance
            xx = FactoredMoment(DArray.zeros((1,) * (x.ndim - self.ndim) + self.shape + self.shape),
                                x, self.ndim)
        else:
            xx = linalg.outer(x, x, ndim=self.ndim) if self.ndim else fuse(lambda v: v * v, x)
        return [x, xx], -0.5 * k * LOG2PI

    def message_to_parent(self, index, u, up):
        x = u[0]
        a = up[1][0]
        if self.mu_gg:
            # [x, -1/2, -1/2 x^2, 1/2] (gaussian.py:609-632) through the wrapper (:2348-2369)
            if index == 0:
                return [fuse(lambda a_, x_: a_ * x_, a, x), fuse(lambda a_: -0.5 * a_, a),
                        fuse(lambda a_, q: -0.5 * a_ * q, a, u[1]), 0.5]
            tm, tmm, t, _ = up[0]
            m0 = fuse(lambda x_, tm_, q, x2_, t_: x_ * tm_ - 0.5 * q - 0.5 * x2_ * t_,
                      x, tm, tmm, u[1], t)
            return [m0, 0.5]
        if index == 0:
            if getattr(self, '_terms_ok', False) and isinstance(a, DArray) and isinstance(x, DArray):
                m0 = LazySum([(1.0, [a, x])], broadcasted_shape(a.shape, x.shape),
                             lambda: fuse(lambda a_, x_: a_ * x_, a, x))
            else:
                m0 = fuse(lambda a_, x_: a_ * x_, a, x)
            if self.mu_ndim > 0:
                d = fuse(lambda a_, o: -0.5 * a_ * o, a, _ones(self.mu_shape))
                return [m0, misc.diag(d, ndim=self.mu_ndim)]
            return [m0, fuse(lambda a_: -0.5 * a_, a)]
        m, m2 = self._mu(up)
        x2 = _diag2(u[1], self.ndim) if self.ndim else u[1]
        if self.ndim == 0 and getattr(self, '_terms_ok', False) \
                and all(isinstance(a, DArray) for a in (x, m, m2, x2)):
            # x m - <m^2> / 2 - <x^2> / 2 as three plate sums (no plates-sized temporary); only
            # for the engine's own call -- a wrapping family (mixture, gate) indexes the arrays
            return [Terms([(1.0, [x, m]), (-0.5, [m2]), (-0.5, [x2])]), 0.5]
        m0 = fuse(lambda x_, m_, q, x2_: x_ * m_ - 0.5 * q - 0.5 * x2_, x, m, m2, x2)
        return [m0, 0.5]

    finite_phi = True          # (alpha mu, -alpha / 2): 0 * phi needs no guard (MixtureFamily)

    # the message to a parent does not depend on that parent's own moments (conjugacy): the
    # router may reuse it while everything else it reads is unchanged
    message_independent_of_target = True

    def observed_bound_terms(self, u, up):
        """cgf_from_parents + f + phi_p . u of a fully observed scalar-valued node as a sum of
        products over its plates (expfamily.py:400-480): -a <m^2>/2 + log a / 2 - log(2 pi)/2 +
        a m x - a x^2 / 2.  None when this form does not apply."""
        if self.ndim != 0 or self.mu_gg:
            return None
        m, m2 = self._mu(up)
        a, loga = up[1]
        x, x2 = u
        ops = (m, m2, a, loga, x, x2)
        if not all(isinstance(o, DArray) for o in ops):
            return None
        return [(-0.5, [a, m2]), (0.5, [loga]), (-0.5 * LOG2PI, []), (1.0, [a, m, x]), (-0.5, [a, x2])]


class GaussianFamily(Family):
    """gaussian.py:293-573 with the wrapper gaussian.py:2374-2527 folded in."""

    def __init__(self, node):
        super().__init__(node)
        self.D = node.dims[0][0]
        self.shape = (self.D,)
        self.ndim = 1

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, linalg.outer(v, v)]
        return [v, linalg.chol_logdet(linalg.chol(v))]

    def phi_from_parents(self, up):
        m, L = up[0][0], up[1][0]
        return [linalg.mvdot(L, m), fuse(lambda l: -0.5 * l, L)]

    moments_and_cgf = GaussianARDFamily.moments_and_cgf
    q_term = GaussianARDFamily.q_term
    gradient = GaussianARDFamily.gradient

    def cgf_from_parents(self, up):
        mm = up[0][1]
        L, logdet = up[1]
        return fuse(lambda t, ld: -0.5 * t + 0.5 * ld, misc.sum_multiply(L, mm, axis=(-1, -2)),
                    logdet)

    def fixed_moments_and_f(self, x):
        x = _arr(x)
        if x.shape[-1:] != (self.D,):
            raise ValueError("Invalid shape")
        return [x, linalg.outer(x, x)], -0.5 * self.D * LOG2PI

    def message_to_parent(self, index, u, up):
        x, xx = u
        m, mm = up[0]
        L = up[1][0]
        if index == 0:
            if getattr(self, '_terms_ok', False) and isinstance(L, DArray) and isinstance(x, DArray) \
                    and not isinstance(x, (LazySum, LazyContract)):
                # Lambda x stays a contraction: under a mixture it is weighted by the
                # responsibilities and summed over the plates, sum_n r_nk Lambda_k x_n =
                # Lambda_k (sum_n r_nk x_n) -- the (N, K, D) array of the reference
                # (gaussian.py:2451-2454 under mixture.py:126-158) is never formed
                return [_lazy_mvdot(L, x), fuse(lambda l: -0.5 * l, L)]
            return [linalg.mvdot(L, x), fuse(lambda l: -0.5 * l, L)]
        if getattr(self, '_terms_ok', False) and all(isinstance(a, DArray) for a in (x, xx, m, mm)):
            # -(<xx^T> - <x><m>^T - <m><x>^T + <mm^T>) / 2 as four products: whoever sums it over
            # plates (weighted by responsibilities under a mixture) contracts <xx^T> and <x>
            # directly -- the plates x D x D array (x K clusters under a mixture) is never formed
            xc, xr = x.reshape(x.shape + (1,)), x.reshape(x.shape[:-1] + (1, self.D))
            mc, mr = m.reshape(m.shape + (1,)), m.reshape(m.shape[:-1] + (1, self.D))
            return [Terms([(-0.5, [xx]), (0.5, [xc, mr]), (0.5, [mc, xr]), (-0.5, [mm])]), 0.5]
        xm = linalg.outer(x, m)
        mx = linalg.outer(m, x)
        return [fuse(lambda a, b, c, d: -0.5 * (a - b - c + d), xx, xm, mx, mm), 0.5]

    # natural parameters are finite whatever the moments: 0 * phi needs no guard (MixtureFamily)
    finite_phi = True


class GaussianGammaFamily(Family):
    """GaussianGammaDistribution (gaussian.py:892-1136) with the (mu, Lambda) wrapper
    (WrapToGaussianWishart, gaussian.py:2374-2527) folded in: parents mu, Lambda, a, b;
    moments u = [<tau x>, <tau x x^T>, <tau>, <log tau>]; phi = [Lambda mu, -Lambda / 2,
    -mu^T Lambda mu / 2 - b, a]."""

    def __init__(self, node):
        super().__init__(node)
        self.ndim = node.ndim
        self.shape = node.shape
        self.D = int(np.prod(node.shape)) if node.ndim else 1

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 0:
            return [v, linalg.outer(v, v)] if self.ndim else [v, fuse(lambda m: m * m, v)]
        if index == 1:
            if self.ndim:
                return [v, linalg.chol_logdet(linalg.chol(v))]
            return [v, fuse(lambda l: da.log(l), v)]
        if index == 2:
            return [v, fuse(lambda a: da.gammaln(a), v)]          # GammaPriorMoments, gamma.py:33-58
        return [v, fuse(lambda b: da.log(b), v)]

    def phi_from_parents(self, up):
        (m, mm), (L, _), (a, _), (b, _) = up[0][:2], up[1][:2], up[2], up[3]
        if self.ndim:
            return [linalg.mvdot(L, m), fuse(lambda l: -0.5 * l, L),
                    fuse(lambda t, b_: -0.5 * t - b_, misc.sum_multiply(L, mm, axis=(-1, -2)), b),
                    fuse(lambda a_: 1.0 * a_, a)]
        return [fuse(lambda l, m_: l * m_, L, m), fuse(lambda l: -0.5 * l, L),
                fuse(lambda l, q, b_: -0.5 * l * q - b_, L, mm, b), fuse(lambda a_: 1.0 * a_, a)]

    def moments_and_cgf(self, phi):
        p0, p1, p2, a = (_arr(p) for p in phi)
        if self.ndim == 0:
            mu = fuse(lambda p0_, p1_: -p0_ / (2 * p1_), p0, p1)
            b = fuse(lambda p2_, mu_, p0_: -p2_ - 0.5 * mu_ * p0_, p2, mu, p0)
            u2 = fuse(lambda a_, b_: a_ / b_, a, b)
            u3 = fuse(lambda a_, b_: da.digamma(a_) - da.log(b_), a, b)
            u0 = fuse(lambda mu_, t: mu_ * t, mu, u2)
            u1 = fuse(lambda p1_, mu_, t: -1.0 / (2 * p1_) + mu_ * mu_ * t, p1, mu, u2)
            g = fuse(lambda p1_, a_, b_: 0.5 * da.log(-2 * p1_) + a_ * da.log(b_) - da.gammaln(a_),
                     p1, a, b)
            return [u0, u1, u2, u3], g
        D = self.D
        U = linalg.chol(fuse(lambda p: -2 * p, p1))
        cov = linalg.chol_inv(U)
        mu = linalg.chol_solve(U, p0)
        b = fuse(lambda p2_, s: -p2_ - 0.5 * s, p2, linalg.inner(mu, p0))
        u2 = fuse(lambda a_, b_: a_ / b_, a, b)
        u3 = fuse(lambda a_, b_: da.digamma(a_) - da.log(b_), a, b)
        u0 = fuse(lambda mu_, t: mu_ * t, mu, _trail(u2, 1))
        u1 = fuse(lambda c, x, y, t: c + x * y * t, cov, _trail(mu, 1),
                  mu.reshape(mu.shape[:-1] + (1, D)), _trail(u2, 2))
        g = fuse(lambda ld, a_, b_: 0.5 * ld + a_ * da.log(b_) - da.gammaln(a_),
                 linalg.chol_logdet(U), a, b)
        return [u0, u1, u2, u3], g

    def cgf_from_parents(self, up):
        ld = up[1][1]
        a, gla = up[2]
        logb = up[3][1]
        return fuse(lambda ld_, a_, lb, g_: 0.5 * ld_ + a_ * lb - g_, ld, a, logb, gla)

    def fixed_moments_and_f(self, x):
        raise NotImplementedError('fixed values of a GaussianGamma node')

    def message_to_parent(self, index, u, up):
        tx, txx, t, lt = u
        (m, mm), L = up[0][:2], up[1][0]
        if index == 0:
            # [<tau x>, -<tau>/2, ...] to (mu, Lambda) (gaussian.py:957-972), then the part of mu
            # (gaussian.py:2464-2477): [Lambda <tau x>, -<tau> Lambda / 2]
            if self.ndim:
                return [linalg.mvdot(L, tx), fuse(lambda l, t_: -0.5 * l * t_, L, _trail(t, 2))]
            return [fuse(lambda l, x_: l * x_, L, tx), fuse(lambda l, t_: -0.5 * l * t_, L, t)]
        if index == 1:
            if self.ndim:
                xm = linalg.outer(tx, m)
                mx = linalg.outer(m, tx)
                return [fuse(lambda a, b, c, d, t_: -0.5 * (a - b - c + d * t_), txx, xm, mx, mm,
                             _trail(t, 2)), 0.5]
            return [fuse(lambda a, x_, m_, d, t_: -0.5 * (a - 2 * x_ * m_ + d * t_), txx, tx, m, mm, t),
                    0.5]
        if index == 2:
            raise NotImplementedError('message from GaussianGamma to its shape parameter')
        return [fuse(lambda t_: -t_, t), up[2][0]]


class GaussianToGaussianGammaFamily:
    """gaussian.py:2226-2276: u = [<x>, <x x^T>, 1, 0]; the message keeps the Gaussian part."""
    deterministic = True

    def __init__(self, node):
        self.node = node

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        v = _arr(value)
        nd = self.node.ndim
        return [v, linalg.outer(v, v, ndim=nd) if nd else fuse(lambda m: m * m, v)]

    def moments(self, ups):
        return [ups[0][0], ups[0][1], 1.0, 0.0]

    def message_to_parent(self, index, m_child, ups, mask=None):
        return list(m_child[:2])


class WrapToGaussianGammaFamily:
    """gaussian.py:2299-2371: the joint (X, alpha) parent as a node of its own."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.ndim = node.ndim

    def mask_to_parent(self, index, mask):
        return mask

    def plates_to_parent(self, index):
        return self.node.plates

    def constant_moments(self, index, value):
        v = _arr(value)
        if index == 1:
            return [v, fuse(lambda a: da.log(a), v)]
        raise NotImplementedError('constant Gaussian-gamma parent of WrapToGaussianGamma')

    def moments(self, ups):
        (tx, txx, t, lt), (a, la) = ups[0], ups[1]
        nd = self.ndim
        return [fuse(lambda x, a_: x * a_, _arr(tx), _trail(_arr(a), nd)),
                fuse(lambda x, a_: x * a_, _arr(txx), _trail(_arr(a), 2 * nd)),
                fuse(lambda t_, a_: t_ * a_, _arr(t), _arr(a)),
                fuse(lambda l, la_: l + la_, _arr(lt), _arr(la))]

    def message_to_parent(self, index, m_child, ups, mask=None):
        nd = self.ndim
        (tx, txx, t, lt), (a, la) = ups[0], ups[1]
        mk = (lambda x, k: x) if mask is None else \
            (lambda x, k: fuse(lambda v, w: v * w, _arr(x), _trail(mask, k)))
        if index == 0:
            out = []
            for i, k in enumerate((nd, 2 * nd, 0)):
                m = m_child[i]
                out.append(None if m is None else
                           mk(fuse(lambda v, a_: v * a_, _arr(m), _trail(_arr(a), k)), k))
            m3 = m_child[3]
            out.append(None if m3 is None else mk(m3, 0))
            return out
        m0 = None
        for m, uu, k in ((m_child[0], tx, nd), (m_child[1], txx, 2 * nd), (m_child[2], t, 0)):
            if m is None:
                continue
            term = _sum_last(fuse(lambda v, w: v * w, _arr(m), _arr(uu)), k)
            m0 = term if m0 is None else fuse(lambda p, q: p + q, m0, term)
        m3 = m_child[3]
        return [None if m0 is None else mk(m0, 0), None if m3 is None else mk(m3, 0)]
