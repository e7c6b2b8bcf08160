"""
Generic variational message passing on device arrays.

Any graph of the built node types that no fused plan covers runs here: the
reference's per-node protocol (the five VMP formulas of ``Distribution``,
stochastic.py:16-80 / expfamily.py:17-70; ``_compute_moments`` /
``_compute_message_to_parent`` of ``Deterministic``, deterministic.py:16-96) and
its message routing (``Node._message_to_parent`` node.py:570-655,
``_message_from_children`` :657-688, mask propagation :457-526, the lower bound
expfamily.py:400-480) restated over ``DArray``s, so that every array operation
is a HIP kernel (``vmp_ewise``, ``vmp_sum_multiply``, ``vmp_spd_batched``,
``vmp_softmax_moments``, ``vmp_onehot_i64``).

Differences from the reference's design (not its results):

* the joint-parent wrappers ``WrapToGaussianGamma`` / ``WrapToGaussianWishart``
  (gaussian.py:2299-2527) are folded into the Gaussian families;
* messages may be *products of factors* that are multiplied and plate-summed by
  ONE fused kernel launch, so e.g. the (N, K, D, D) weighted messages of a
  mixture (mixture.py:126-158) are never materialised;
* plate sums, masks and the integer plate multiplier (utils/misc.py:761-844)
  are a single ``sum_multiply_to_plates`` launch per message.
"""
import ctypes
import os

import numpy as np

from ... import darray as da
from ...darray import DArray, fuse, contiguous
from ...nodes.node import Constant, Stochastic
from ...nodes.gamma import Gamma
from ...nodes.gaussian import (GaussianARD, Gaussian, GaussianGamma, GaussianToGaussianGamma,
                               WrapToGaussianGamma, is_gaussian_gamma)
from ...nodes.dot import SumMultiply
from ...nodes.wishart import Wishart
from ...nodes.dirichlet import Dirichlet
from ...nodes.categorical import Categorical
from ...nodes.multinomial import Multinomial
from ...nodes.mixture import Mixture
from ...nodes.gaussian_markov_chain import GaussianMarkovChain, MarkovChainToGaussian
from ...utils import misc, linalg
from ...utils.shapes import broadcasted_shape, is_shape_subset, multiplier_factor
from .graph_iter import GraphIteration

LOG2PI = float(np.log(2 * np.pi))


def _shape(x):
    return x.shape if isinstance(x, DArray) else np.shape(x)


def _arr(x):
    if isinstance(x, DArray):
        return x
    if hasattr(x, 'is_cuda'):
        # a tensor handed to observe() / initialize_from_value(): used in place when it is
        # fp64 and already resident in HBM
        from ...device import get_runtime
        rt = get_runtime()
        return DArray(x.to(device=rt.device, dtype=rt.torch.float64))
    if isinstance(x, (int, float, np.floating, np.integer)) and len(_CONSTS) < 4096:
        # numbers that recur every sweep (a constant prior's log-normaliser, ...): uploaded once
        v = float(x)
        return _const(('scalar', v), lambda: np.asarray(v, dtype=np.float64))
    return DArray.from_host(np.asarray(x, dtype=np.float64))


def _trail(x, n):
    """Append n unit axes."""
    if n == 0 or not isinstance(x, DArray):
        return x
    return x.reshape(x.shape + (1,) * n)


class FactoredMoment(DArray):
    """Second moment of Gaussian factors whose posterior covariance is SHARED over plates:
    <x x^T> = Cov + <x><x>^T kept as the pair (Cov, <x>) instead of a plates x K x K array.

    The reference materialises the array (gaussian.py:672-706: ``u1 = outer(u0, u0) + Cov``; 2 GB
    at N = 1e6, K = 16 and 82 GB at the headline size) and contracts it with einsum (dot.py:355,
    :403, :581).  Here the consumers that matter read the factors -- ``SumMultiplyFamily`` expands
    the product of (Cov + x x^T) terms, the Gamma message takes diag(Cov) + x^2, the bound takes
    phi : Cov + x^T phi x -- and anything else sees an ordinary device array: ``.t`` forms the
    dense array on first use (same values as the reference's)."""
    __slots__ = ('cov', 'mean', 'nd', '_dense', 'logdet_prec', 'sums')

    def __init__(self, cov, mean, nd, logdet_prec=None, sums=None):
        self.cov, self.mean, self.nd = cov, mean, int(nd)
        self._dense = None
        # log|Cov^-1| with the plates of ``cov`` (no variable axes), or None when the maker does
        # not have it (point masses, rotated moments): the bound term then takes the general route
        self.logdet_prec = logdet_prec
        # plate sums of the means made by the pass that wrote them (PlateSums), or None
        self.sums = sums

    @property
    def t(self):
        if self._dense is None:
            o = linalg.outer(self.mean, self.mean, ndim=self.nd)
            self._dense = fuse(lambda c, o_: c + o_, self.cov, o).t
        return self._dense

    @property
    def shape(self):
        nd = self.nd
        mp = self.mean.shape[:self.mean.ndim - nd]
        cp = self.cov.shape[:self.cov.ndim - 2 * nd]
        return tuple(broadcasted_shape(mp, cp)) + tuple(self.cov.shape[self.cov.ndim - 2 * nd:])

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape))


class PlateSums:
    """Sums over the plates of the posterior means <x_n> of a shared-covariance Gaussian node, made
    by the pass that wrote the means (vmp_gaussian_shared_update): ``x`` = sum_n <x_n> (K),
    ``xx`` = sum_n <x_n><x_n>^T (K, K) and, when the pass streamed the data array Y of the Dot
    message, ``yx`` = sum_n y_n <x_n>^T (D, K) with ``ydesc`` = (address, stride along the rows,
    stride along the plates, D) of that array and ``ykeep`` the tensor itself.  They are part of
    the node's state (the next sweep's first message reads them) and are offered to whoever asks
    for the same reductions through the plan's memo (GenericPlan._seed_sums)."""
    __slots__ = ('x', 'xx', 'yx', 'ydesc', 'ykeep', 'n')

    def __init__(self, x, xx, yx=None, ydesc=None, ykeep=None, n=0):
        self.x, self.xx, self.yx, self.ydesc, self.ykeep, self.n = x, xx, yx, ydesc, ykeep, int(n)


class DerivedArray(DArray):
    """A state array that is a function of other state arrays and is formed only if somebody
    reads it: the natural parameter phi0 = Lambda <x> and the log-normaliser
    g = -<x>^T Lambda <x> / 2 + log|Lambda| / 2 of a shared-covariance Gaussian node after the fused
    update (the reference stores both, gaussian.py:649-706; here nothing in a sweep reads them)."""
    __slots__ = ('kind', 'deps', '_shape', '_dense')

    def __init__(self, kind, deps, shape):
        self.kind, self.deps, self._shape = kind, tuple(deps), tuple(shape)
        self._dense = None

    @property
    def t(self):
        if self._dense is None:
            if self.kind == 'gauss_phi0':
                phi1, x = self.deps
                lam = fuse(lambda p: -2.0 * p, phi1)
                self._dense = linalg.mvdot(lam, x).t
            elif self.kind == 'gauss_g':
                phi1, x, ld = self.deps
                lam = fuse(lambda p: -2.0 * p, phi1)
                q = misc.sum_multiply(linalg.mvdot(lam, x), x, axis=-1)
                self._dense = fuse(lambda q_, l: -0.5 * q_ + 0.5 * l, q, ld).t
            else:
                raise ValueError(self.kind)
        return self._dense

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def size(self):
        return int(np.prod(self._shape))


class LazySum(DArray):
    """A plates-sized array known as a SUM of products of smaller or already existing arrays,
    ``[(coef, [factor, ...]), ...]`` -- e.g. <f^2> = <f>^2 + x^T Cov_w x + w^T Cov_x w + tr(Cov_w Cov_x)
    of a dot product of factored parents, or tau * y of an observed node's message.  Consumers that
    only plate-sum it (the message to a precision, the lower-bound term) or contract it
    (SumMultiply messages) read the factors; anything else sees an ordinary device array: ``.t``
    evaluates ``dense()`` on first use."""
    __slots__ = ('terms', '_shape', '_make', '_dense')

    def __init__(self, terms, shape, dense):
        self.terms, self._shape, self._make = list(terms), tuple(shape), dense
        self._dense = None

    @property
    def t(self):
        if self._dense is None:
            self._dense = self._make().t
            self._make = None
        return self._dense

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def size(self):
        return int(np.prod(self._shape))


class LazyContract(DArray):
    """A plates-sized array known as a contraction of smaller arrays: the first moment <f> = W X of
    a dot product.  Whoever plate-sums a product that contains it (sum y <f>, sum <f>^2: the message
    to the precision of the observed child and its bound term) contracts the factors pair by pair
    (``misc.contract_path``: sum_dn y_dn w_dk x_nk = sum_dk w_dk (Y X^T)_dk, a K-sliced GEMM, and
    sum <f>^2 = (W^T W) : (X^T X)) without the (D, N) array; anything else sees an ordinary device
    array: ``.t`` evaluates the contraction on first use."""
    __slots__ = ('ops', 'labs', 'out', 'sizes', 'compress', '_shape', '_dense', '_make')

    def __init__(self, ops, labs, out, sizes, compress, make=None):
        self.ops, self.labs, self.out = list(ops), [list(l) for l in labs], list(out)
        self.sizes, self.compress = dict(sizes), tuple(compress)
        self._make = make          # how to form the dense array, if not as ONE contraction launch
        var = set()
        for a, ls in zip(self.ops, self.labs):
            for ax, lab in enumerate(ls):
                if a.shape[ax] != 1:
                    var.add(lab)
        self._shape = tuple(int(sizes[lab]) if (lab not in self.compress or lab in var) else 1
                            for lab in self.out)
        self._dense = None

    @property
    def t(self):
        if self._dense is None:
            if self._make is not None:
                self._dense = self._make().t
                self._make = None
            else:
                self._dense = misc.contract(self.ops, self.labs, self.out, self.sizes,
                                            compress=self.compress).t
        return self._dense

    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def size(self):
        return int(np.prod(self._shape))


def _is_lazy(x):
    # (whether somebody has evaluated the dense form must not change what a consumer computes)
    return isinstance(x, LazySum)


def _factored_min_plates():
    return int(os.environ.get('BAYESPY_AMD_FACTORED_MIN_PLATES', '2'))


def _diag2(xx, nd):
    """diag over the last 2 nd axes of a second moment, factored or dense."""
    if isinstance(xx, FactoredMoment):
        return fuse(lambda c, x: c + x * x, misc.get_diag(xx.cov, ndim=nd), xx.mean)
    return misc.get_diag(xx, ndim=nd)


def _inner_second(phi, xx, nd):
    """sum over the last 2 nd axes of phi * <x x^T>."""
    axes = tuple(range(-2 * nd, 0))
    if isinstance(xx, FactoredMoment):
        x = xx.mean
        phi = _arr(phi)
        a = misc.sum_multiply(phi, xx.cov, axis=axes)
        # x^T phi x per plate in two steps -- t = phi x (a GEMM over the plates), then the row
        # products t . x -- instead of one three-operand contraction (a thread-group kernel that
        # walks K^2 products per plate: 1.7 ms at N = 1e6, K = 16)
        D = int(np.prod(x.shape[x.ndim - nd:]))
        xf = x.reshape(x.shape[:x.ndim - nd] + (D,))
        pf = phi.reshape(phi.shape[:phi.ndim - 2 * nd] + (D, D))
        b = misc.sum_multiply(linalg.mvdot(pf, xf), xf, axis=-1)
        return fuse(lambda p, q: p + q, a, b)
    return misc.sum_multiply(_arr(phi), _arr(xx), axis=axes)


def _lazy_mvdot(A, b):
    """linalg.mvdot(A, b) -- (..., D, E) . (..., E) -> (..., D) with broadcast plates -- as a
    LazyContract: whoever plate-sums a product that contains it plans the contraction pair by
    pair; anything else sees the array (``.t`` evaluates it)."""
    npl = max(A.ndim - 2, b.ndim - 1)
    q = ['q%d' % i for i in range(npl)]
    la = q[npl - (A.ndim - 2):] + ['d', 'e']
    lb = q[npl - (b.ndim - 1):] + ['e']
    plates = broadcasted_shape(A.shape[:-2], b.shape[:-1])
    sizes = {lab: s for lab, s in zip(q, plates)}
    sizes['d'], sizes['e'] = A.shape[-2], A.shape[-1]
    return LazyContract([A, b], [la, lb], q + ['d'], sizes, q)


_CONSTS = {}


def _const(key, make):
    """Small read-only device constants (ones, identities, ...) are uploaded once."""
    from ...device import get_runtime
    k = (id(get_runtime()),) + key
    if k not in _CONSTS:
        _CONSTS[k] = DArray.from_host(make())
    return _CONSTS[k]


def _ones(shape):
    shape = tuple(shape)
    return _const(('ones', shape), lambda: np.ones(shape))


def _eye(shape):
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    return _const(('eye', shape), lambda: np.eye(n).reshape(shape + shape))


def _check_device(x, bad, exc_type, message):
    """Raise ``exc_type(message)`` if ``bad`` ('negative': x < 0, 'nonpositive': x <= 0 or NaN)
    holds anywhere -- a count formed on the device by the library's own kernels (so that it
    queues with the formulas around it) and read with the other checks of the running plan
    operation (device.Runtime.defer_check)."""
    from ...device import get_runtime
    x = _arr(x)
    if bad == 'negative':
        ind = fuse(lambda v: da.where_nonzero(da.maximum(-v, 0.0), 1.0), x)
    elif bad == 'nonpositive':
        ind = fuse(lambda v: 1.0 - da.where_nonzero(da.maximum(v, 0.0), 1.0), x)
    else:
        raise ValueError(bad)
    get_runtime().defer_check(misc.sum_multiply(ind).t, exc_type, message)


def _wsum(pairs):
    """sum_i coef_i * array_i as ONE fused launch per six operands (a chain of two-operand
    additions is a chain of dependent launches: 3-5 us each on scalars)."""
    pairs = [(float(c), _arr(a)) for c, a in pairs]
    if not pairs:
        return None
    while True:
        chunk, pairs = pairs[:6], pairs[6:]
        cs = tuple(c for c, _ in chunk)
        if len(chunk) == 1 and cs[0] == 1.0:
            acc = chunk[0][1]
        else:
            def f(*xs, cs=cs):
                tot = None
                for c, x in zip(cs, xs):
                    t = x if c == 1.0 else c * x
                    tot = t if tot is None else tot + t
                return tot
            acc = fuse(f, *[a for _, a in chunk])
        if not pairs:
            return acc
        pairs = [(1.0, acc)] + pairs


def _sum_last(x, n):
    return x if n == 0 else misc.sum_multiply(x, axis=tuple(range(-n, 0)))


def _multigammaln(a, d):
    """log Gamma_d(a) (scipy.special.multigammaln call site wishart.py:187)."""
    half = _const(('half_arange', int(d)), lambda: 0.5 * np.arange(d))
    t = fuse(lambda x, h: da.gammaln(x - h), _trail(_arr(a), 1), half)
    return fuse(lambda s: s + d * (d - 1) / 4.0 * np.log(np.pi), misc.sum_multiply(t, axis=-1))


def _gaussian_q_term(family_ndim, shape, phi, u, g):
    """-(g_q + phi_q . u_q) of a Gaussian factor without touching its second-order arrays:
    with Lambda = -2 phi1 and mean x,  phi0.x = x^T Lambda x,  phi1:<xx^T> = -K/2 - x^T Lambda x / 2
    and  g = -x^T Lambda x / 2 + log|Lambda| / 2,  so the sum is  K/2 - g - phi0.x / 2
    (expfamily.py:449-468 evaluates the same quantity as two contractions over plates x K x K)."""
    k = float(np.prod(shape)) if family_ndim else 1.0
    d = _sum_last(fuse(lambda p, x: p * x, _arr(phi[0]), _arr(u[0])), family_ndim)
    return fuse(lambda g_, d_: 0.5 * k - g_ - 0.5 * d_, _arr(g), d)



def _gaussian_gradient(rg, u, ndim, shape):
    """Euclidean gradient of a Gaussian factor given the Riemannian one (the chain rule of
    gaussian.py:489-556 / :824-892) -- with Cov = <xx> - <x><x>^T:
    d0 = Cov g0 + 2 Cov g1 x,   d1 = Cov g0 x^T + x (Cov g0)^T + 2 <xx> g1 <xx> - 2 (x^T g1 x) x x^T."""
    x, xx, g0, g1 = _arr(u[0]), _arr(u[1]), _arr(rg[0]), _arr(rg[1])
    if ndim == 0:
        d0 = fuse(lambda x_, q, a, b: (q - x_ * x_) * (a + 2 * b * x_), x, xx, g0, g1)
        d1 = fuse(lambda x_, q, a, b: 2 * (q - x_ * x_) * a * x_ + 2 * q * b * q
                  - 2 * x_ * x_ * b * x_ * x_, x, xx, g0, g1)
        return [d0, d1]
    D = int(np.prod(shape))

    def flat(a, k):
        return a.reshape(a.shape[:a.ndim - k * ndim] + (D,) * k)
    x, xx, g0, g1 = flat(x, 1), flat(xx, 2), flat(g0, 1), flat(g1, 2)
    cov = fuse(lambda q, a, b: q - a * b, xx, _trail(x, 1), x.reshape(x.shape[:-1] + (1, D)))
    cov_g0 = linalg.mvdot(cov, g0)
    g1_x = linalg.mvdot(g1, x)
    d0 = fuse(lambda a, b: a + 2 * b, cov_g0, linalg.mvdot(cov, g1_x))
    c = linalg.outer(cov_g0, x)
    d1 = fuse(lambda c_, ct, m, xa, xb, s_: c_ + ct + 2 * m - 2 * xa * xb * s_,
              c, linalg.transpose(c), linalg.mmdot(xx, linalg.mmdot(g1, xx)),
              _trail(x, 1), x.reshape(x.shape[:-1] + (1, D)), _trail(linalg.inner(g1_x, x), 2))
    return [d0.reshape(d0.shape[:-1] + tuple(shape)),
            d1.reshape(d1.shape[:-2] + tuple(shape) + tuple(shape))]


# ---------------------------------------------------------------------------
# families: the five VMP formulas per node type
# ---------------------------------------------------------------------------
class _Deferred:
    def __init__(self, make):
        self.make = make


class _LazyList(list):
    """A list whose _Deferred entries are evaluated when first read."""

    def __getitem__(self, i):
        v = list.__getitem__(self, i)
        if isinstance(v, _Deferred):
            v = v.make()
            list.__setitem__(self, i, v)
        return v

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class Terms:
    """A message entry (or a bound term) that is a SUM of products: ``[(coef, [factor, ...]), ...]``.
    The router plate-sums every product with ONE fused launch and adds the (parent-sized) results,
    so e.g. the message of an observed GaussianARD to its precision, sum_n (x m - q / 2 - x^2 / 2),
    is three reductions over the data instead of a plates-sized temporary and its reduction."""

    def __init__(self, terms):
        self.terms = list(terms)


class Family:

    def __init__(self, node):
        self.node = node

    def plates_to_parent(self, index):
        return self.node.plates

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        raise NotImplementedError

    def gradient(self, rg, u, phi):
        """Euclidean gradient from the Riemannian one (expfamily.py:64-70)."""
        raise NotImplementedError("Standard gradient not yet implemented for %s"
                                  % type(self.node).__name__)


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


class MixtureFamily(Family):
    """mixture.py:26-356 over the last parameter plate."""

    def __init__(self, node, base):
        super().__init__(node)
        self.base = base                 # family of the mixed distribution (on node._proto)
        self.K = node.clusters
        self.ndims = [len(d) for d in node.dims]
        # the cluster axis among the plates of the mixed distribution (negative; -1 = last).
        # Internally the formulas always see it as the LAST of those plates: parameter moments
        # are re-viewed with the axis moved there (_cluster_last) and messages moved back
        self.cp = node.cluster_plate

    def _plates_with_cluster(self, k):
        """The node's plates with the cluster axis (of extent k) at its position."""
        p = list(self.node.plates)
        p.insert(len(p) + self.cp + 1, k)
        return tuple(p)

    def _extra(self, index):
        """Number of variable axes the mixed family maps onto plates of parameter `index`."""
        return len(self.plates_to_parent(index)) - len(self.node.plates) - 1

    def plates_to_parent(self, index):
        if index == 0:
            return self.node.plates
        saved = self.base.node.plates
        self.base.node.plates = self._plates_with_cluster(self.K)
        try:
            return self.base.plates_to_parent(index - 1)
        finally:
            self.base.node.plates = saved

    def mask_to_parent(self, index, mask):
        if index == 0:
            return mask
        mask = np.asarray(mask)
        if self.cp == -1:
            mask = mask.reshape(mask.shape + (1,))
        elif mask.ndim >= -self.cp - 1:
            mask = np.expand_dims(mask, mask.ndim + self.cp + 1)
        return self.base.mask_to_parent(index - 1, mask)

    def _cluster_last(self, up):
        """Parameter moments with the cluster axis moved behind the other plates of the mixed
        distribution (stride-only views)."""
        if self.cp == -1:
            return up
        out = [up[0]]
        for j, u in enumerate(up[1:], start=1):
            ex = self._extra(j)
            par = self.node.parents[j]
            full = len(self.node.plates) + 1 + ex
            moved = []
            for i, x in enumerate(u):
                if not isinstance(x, DArray):
                    moved.append(x)
                    continue
                nd = 0 if isinstance(par, Constant) else len(par.dims[i])
                # variable axes of a constant parameter: whatever exceeds the full plate rank
                if isinstance(par, Constant):
                    nd = max(0, x.ndim - full)
                if x.ndim - nd < full:
                    x = x.reshape((1,) * (full - (x.ndim - nd)) + x.shape)
                moved.append(misc.moveaxis(x, self.cp - ex - nd, -1 - ex - nd))
            out.append(moved)
        return out

    def _cluster_back(self, m, index, nd):
        """A message to parameter `index` (cluster axis last of the plates) in the parameter's
        own axis order."""
        if self.cp == -1:
            return m
        ex = self._extra(index)
        full = len(self.node.plates) + 1 + ex + nd

        def back(x):
            x = _arr(x)
            if x.ndim < full:
                x = x.reshape((1,) * (full - x.ndim) + x.shape)
            return misc.moveaxis(x, -1 - ex - nd, self.cp - ex - nd)
        return tuple(back(x) for x in m) if isinstance(m, tuple) else back(m)

    def constant_moments(self, index, value):
        if index == 0:
            # fixed class labels (categorical.py:30-46)
            return [misc.onehot(np.asarray(value).astype(np.int64), self.K)]
        return self.base.constant_moments(index - 1, value)

    def _with_cluster_axis(self, u):
        """u_i (plates + dims_i) -> (plates, 1, dims_i)."""
        out = []
        for ui, nd in zip(u, self.ndims):
            ui = _arr(ui)
            out.append(ui.reshape(ui.shape[:ui.ndim - nd] + (1,) + ui.shape[ui.ndim - nd:]))
        return out

    def phi_from_parents(self, up):
        up = self._cluster_last(up)
        p = up[0][0]
        phik = self.base.phi_from_parents(up[1:])
        out = []
        for ph, nd in zip(phik, self.ndims):
            ph = _arr(ph)
            out.append(misc.sum_multiply(_trail(p, nd), ph, axis=-(nd + 1)))
        return out

    def moments_and_cgf(self, phi):
        return self.base.moments_and_cgf(phi)

    def cgf_from_parents(self, up):
        up = self._cluster_last(up)
        p = up[0][0]
        gk = self.base.cgf_from_parents(up[1:])
        return misc.sum_multiply(p, _arr(gk), axis=-1)

    def fixed_moments_and_f(self, x):
        return self.base.fixed_moments_and_f(x)

    def gradient(self, rg, u, phi):
        return self.base.gradient(rg, u, phi)          # mixture.py:352-356

    def _loglik(self, u, up, uk=None):
        """E[log p(y | cluster k)] - f(y) for every plate and cluster (mixture.py:67-104,
        expfamily.py:45-61); ``up`` with the cluster axis last.  f(y) is left out like in the
        reference (it passes f = 0, mixture.py:92-98): it is the same for every cluster and cancels
        in the normalisation of q(z).  The last answer stands while the arrays it was made from are
        the same objects: the message to the assignments and, one node later, the bound term of
        the observed mixture ask for the same array."""
        deps = [a for a in u] + [a for j in up[1:] for a in j]
        key = tuple(id(a) for a in deps)
        hit = getattr(self, '_ll_cache', None)
        if hit is not None and hit[0] == key and all(isinstance(a, DArray) for a in deps):
            return hit[2]
        if uk is None:
            uk = self._with_cluster_axis(u)
        phik = self.base.phi_from_parents(up[1:])
        parts = [(1.0, _arr(self.base.cgf_from_parents(up[1:])))]
        for ph, ui, nd in zip(phik, uk, self.ndims):
            if nd > 0 and getattr(self.base, 'finite_phi', False):
                # phi_k . u_n as a contraction (plates x clusters, over the variable axes: a
                # matrix-core GEMM) -- not a plates x clusters x D x D product and its sum
                parts.append((1.0, misc.sum_multiply(_arr(ph), ui, axis=tuple(range(-nd, 0)))))
                continue
            t = fuse(lambda a, b: da.where_nonzero(b, a) * b, _arr(ph), ui)
            parts.append((1.0, _sum_last(t, nd)))
        L = _wsum(parts)                         # (one pass over plates x clusters)
        self._ll_cache = (key, deps, L)          # `deps` keeps the keyed arrays alive
        return L

    def observed_bound_terms(self, u, up):
        """cgf_from_parents + f + phi_p . u of a fully observed mixture over its plates
        (expfamily.py:400-480 with mixture.py:53-65): sum_k r_nk (g_k + phi_k . u_n) + f_n -- the
        responsibilities times the array the message to the assignments is made of, instead of
        forming phi_n = sum_k r_nk phi_k (plates x D x D) and contracting it with u_n.  None when
        the mixed family's natural parameters may be infinite (0 * inf needs the guarded form)."""
        if not getattr(self.base, 'finite_phi', False) or isinstance(self.base, MixtureFamily):
            return None
        if os.environ.get('BAYESPY_AMD_MIXTURE_BOUND', '1') == '0':
            return None
        up = self._cluster_last(up)
        p = up[0][0]
        if not isinstance(p, DArray) or not all(isinstance(a, DArray) for a in u):
            return None
        L = self._loglik(u, up)
        if tuple(broadcasted_shape(p.shape, L.shape)[:-1]) != \
                tuple(broadcasted_shape(self.node.plates, p.shape[:-1], L.shape[:-1])):
            return None
        return [(1.0, [misc.sum_multiply(p, L, axis=-1)])]

    def message_to_parent(self, index, u, up):
        up = self._cluster_last(up)
        uk = self._with_cluster_axis(u)
        if index == 0:
            return [self._loglik(u, up, uk)]
        p = up[0][0]
        self.base._terms_ok = getattr(self, '_terms_ok', False) and not isinstance(self.base, MixtureFamily)
        try:
            msgs = self.base.message_to_parent(index - 1, uk, up[1:])
        finally:
            self.base._terms_ok = False
        out = []
        parent = self.node.parents[index]
        # variable axes the mixed family maps onto plates of this parent (the precision of a
        # GaussianARD has the variable's shape among its plates) trail the cluster axis too
        extra = self._extra(index)
        for i, m in enumerate(msgs):
            if m is None:
                out.append(None)
                continue
            nd = len(parent.dims[i])
            # weight by the responsibilities: a lazy product, fused with the plate sum (a nested
            # mixture hands over a product already: one more factor)
            w = _trail(p, nd + extra)
            if isinstance(m, Terms) or _is_lazy(m):
                out.append(Terms([(c, list(self._cluster_back(tuple(fs) + (w,), index, nd)))
                                  for c, fs in m.terms]))
                continue
            inner = tuple(m) if isinstance(m, tuple) else (_arr(m),)
            out.append(self._cluster_back(inner + (w,), index, nd))
        return out


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


class SumMultiplyFamily:
    """dot.py:19-633: einsum over Gaussian moments and its messages to the parents."""
    deterministic = True

    def mask_to_parent(self, index, mask):
        return mask

    def __init__(self, node):
        self.node = node

    def constant_moments(self, index, value):
        """Delta moments [x, x x^T] of a numeric parent over its key axes (dot.py:186-197,
        gaussian.py:74-84)."""
        x = _arr(value)
        nd = len(self.node.in_keys[index])
        if nd == 0:
            return [x, fuse(lambda v: v * v, x)]
        return [x, linalg.outer(x, x, ndim=nd)]

    def _labels(self, plan_plates):
        n = self.node
        npl = len(n.plates)
        plate_labels = ['p%d' % i for i in range(npl)]
        sizes = {lab: s for lab, s in zip(plate_labels, n.plates)}
        for k, s in n.key_sizes.items():
            sizes['k%d' % k] = s
            sizes['K%d' % k] = s
        return plate_labels, sizes

    def _parent_labels(self, i, second):
        n = self.node
        par = n.parents[i]
        pl, _ = self._labels(None)
        lead = pl[len(pl) - len(par.plates):] if len(par.plates) else []
        ks = ['k%d' % k for k in n.in_keys[i]]
        if second:
            ks = ks + ['K%d' % k for k in n.in_keys[i]]
        return list(lead) + ks

    @staticmethod
    def _is_factored(xx):
        return isinstance(xx, FactoredMoment)

    def _second_choices(self, ups, skip=None):
        """The second-moment operands of the parents as a list of alternatives per parent: a dense
        <x x^T> is one alternative; a factored one (Cov + <x><x>^T) is two -- [Cov] and
        [<x> over the first key copy, <x> over the second].  The product of the parents' second
        moments is the sum over one pick per parent."""
        per_parent = []
        for j, u in enumerate(ups):
            if j == skip:
                continue
            xx = u[1]
            l0 = self._parent_labels(j, False)
            l1 = self._parent_labels(j, True)
            nkj = len(self.node.in_keys[j])
            if not self._is_factored(xx) and nkj > 0 and isinstance(xx, DArray) \
                    and not isinstance(xx, FactoredMoment):
                # a small dense second moment without plates of its own (a prior-initialised
                # node: one K x K matrix) factors trivially: Cov = <x x^T> - <x><x>^T
                x0 = _arr(u[0])
                if all(e == 1 for e in xx.shape[:xx.ndim - 2 * nkj]) \
                        and all(e == 1 for e in x0.shape[:x0.ndim - nkj]) and xx.size <= (1 << 16):
                    cov0 = fuse(lambda q, o: q - o, xx, linalg.outer(x0, x0, ndim=nkj))
                    xx = FactoredMoment(cov0, x0, nkj)
            if self._is_factored(xx):
                x, cov = xx.mean, xx.cov
                nk = len(self.node.in_keys[j])
                lK = l0[:len(l0) - nk] + ['K%d' % k for k in self.node.in_keys[j]]
                per_parent.append([
                    ('cov', [(cov, l1[len(l1) - cov.ndim:])]),
                    ('mean', [(x, l0[len(l0) - x.ndim:]), (x, lK[len(lK) - x.ndim:])])])
            else:
                a = _arr(xx)
                per_parent.append([('dense', [(a, l1[len(l1) - a.ndim:])])])
        return per_parent

    @staticmethod
    def _picks(per_parent):
        import itertools
        for combo in itertools.product(*per_parent):
            kinds = [c[0] for c in combo]
            ops = [o for c in combo for o in c[1]]
            yield kinds, ops

    @staticmethod
    def _add_terms(terms):
        acc = terms[0]
        i = 1
        while i < len(terms):
            rest = terms[i:i + 3]
            if len(rest) == 3:
                acc = fuse(lambda a, b, c, d: a + b + c + d, acc, *rest)
            elif len(rest) == 2:
                acc = fuse(lambda a, b, c: a + b + c, acc, *rest)
            else:
                acc = fuse(lambda a, b: a + b, acc, rest[0])
            i += 3
        return acc

    def moments(self, ups):
        n = self.node
        pl, sizes = self._labels(None)
        ops0, labs0 = [], []
        for i, u in enumerate(ups):
            x = _arr(u[0])
            l0 = self._parent_labels(i, False)
            ops0.append(x)
            labs0.append(l0[len(l0) - x.ndim:])
        out0 = pl + ['k%d' % k for k in n.out_keys]
        out1 = out0 + ['K%d' % k for k in n.out_keys]
        per_parent = self._second_choices(ups)
        all_factored = all(len(alts) > 1 for alts in per_parent)
        if all_factored and not n.out_keys and os.environ.get('BAYESPY_AMD_LAZY_SUMS', '1') != '0' \
                and os.environ.get('BAYESPY_AMD_LAZY_DOT', '1') != '0':
            # <f> stays a contraction until somebody needs the array (LazyContract)
            f0 = LazyContract(ops0, labs0, out0, sizes, pl)
        else:
            f0 = misc.contract(ops0, labs0, out0, sizes, compress=pl)
        if not all(len(alts) > 1 for alts in per_parent):
            # some parent carries a dense second moment: the product needs the dense arrays of
            # all of them (a quadratic form per plate pair: D N K^2 flops, the matrix-core GEMM
            # of the dense path is the right tool)
            ops1, labs1 = [], []
            for i, u in enumerate(ups):
                xx = _arr(u[1])
                xx = DArray(xx.t) if isinstance(xx, FactoredMoment) else xx
                l1 = self._parent_labels(i, True)
                ops1.append(xx)
                labs1.append(l1[len(l1) - xx.ndim:])
            return [f0, misc.contract(ops1, labs1, out1, sizes, compress=pl)]
        # every parent factored: <f f^T> = sum over one pick (Cov | <x><x>^T) per parent; the
        # all-means pick is <f><f>^T itself, a pick with means is contracted in two steps --
        # T = (the rest) . <x> over the second key copy (a GEMM), then T . <x> over the first --
        # so that no plates x K x K array is ever formed (the reference's dot.py:355,403)
        terms = []
        nk = len(n.out_keys)
        for kinds, _ in self._picks(per_parent):
            if all(k == 'mean' for k in kinds):
                terms.append(('sq', None))
                continue
            picked = [alts[0 if kd == 'cov' else 1] for alts, kd in zip(per_parent, kinds)]
            first_mean = next((i for i, kd in enumerate(kinds) if kd == 'mean'), None)
            if first_mean is None:
                ops = [o for pk in picked for o in pk[1]]
                terms.append(('t', misc.contract([o[0] for o in ops], [o[1] for o in ops], out1,
                                                 sizes, compress=pl)))
                continue
            (xk, lk), (xK, lK) = picked[first_mean][1]
            rest = [o for i, pk in enumerate(picked) if i != first_mean for o in pk[1]]
            if len(rest) + 1 > 6:
                raise NotImplementedError('SumMultiply over %d factored parents' % len(ups))
            keys_k = [l for l in lk if l.startswith('k')]
            # T keeps: the output labels, this parent's first key copy, every plate label in use
            used = []
            for _, ls in rest + [(xK, lK)]:
                for l in ls:
                    if l not in used:
                        used.append(l)
            # (a key of the second copy that is an OUTPUT key stays; one that is contracted goes)
            t_out = [l for l in pl if l in used] + [l for l in out1 if l not in pl and l in used]
            t_out += [l for l in keys_k if l in used and l not in t_out]
            def two_steps(rest=rest, xK=xK, lK=lK, xk=xk, lk=lk, t_out=t_out):
                T = misc.contract([o[0] for o in rest] + [xK], [o[1] for o in rest] + [lK], t_out,
                                  sizes, compress=pl)
                return misc.contract([T, xk], [t_out, lk], out1, sizes, compress=pl)
            if nk == 0 and os.environ.get('BAYESPY_AMD_LAZY_SUMS', '1') != '0' \
                    and os.environ.get('BAYESPY_AMD_LAZY_QUAD', '1') != '0':
                # x^T C x per plate stays a contraction (dense form: the two steps above): whoever
                # sums it over the plates contracts <x><x>^T first -- sum_n x_n^T C x_n = C : sum_n
                # x_n x_n^T, a sum the sweep has anyway -- and never forms the per-plate rows
                terms.append(('t', LazyContract([o[0] for o in rest] + [xK, xk],
                                                [o[1] for o in rest] + [lK, lk], out1, sizes, pl,
                                                make=two_steps)))
            else:
                terms.append(('t', two_steps()))
        arrs = [t[1] for t in terms if t[0] == 't']
        if any(t[0] == 'sq' for t in terms):
            if nk == 0 and os.environ.get('BAYESPY_AMD_LAZY_SUMS', '1') != '0':
                sq = f0
                shape = broadcasted_shape(f0.shape, *[a.shape for a in arrs])

                def dense(sq=sq, arrs=arrs):
                    if len(arrs) == 1:
                        return fuse(lambda f, a: f * f + a, sq, arrs[0])
                    if len(arrs) == 2:
                        return fuse(lambda f, a, b: f * f + a + b, sq, *arrs)
                    if len(arrs) == 3:
                        return fuse(lambda f, a, b, c: f * f + a + b + c, sq, *arrs)
                    return self._add_terms([fuse(lambda f: f * f, sq)] + arrs)
                f1 = LazySum([(1.0, [sq, sq])] + [(1.0, [a]) for a in arrs], shape, dense)
            elif nk == 0:
                sq = f0
                if len(arrs) == 1:
                    f1 = fuse(lambda f, a: f * f + a, sq, arrs[0])
                elif len(arrs) == 2:
                    f1 = fuse(lambda f, a, b: f * f + a + b, sq, *arrs)
                elif len(arrs) == 3:
                    f1 = fuse(lambda f, a, b, c: f * f + a + b + c, sq, *arrs)
                else:
                    f1 = self._add_terms([fuse(lambda f: f * f, sq)] + arrs)
            else:
                f1 = self._add_terms([linalg.outer(f0, f0, ndim=nk)] + arrs)
        else:
            f1 = self._add_terms(arrs)
        return [f0, f1]

    def message_to_parent(self, index, m_child, ups, mask=None):
        """Messages to parent ``index`` already summed to its plates (dot.py:425-633)."""
        n = self.node
        pl, sizes = self._labels(None)
        par = n.parents[index]
        npl, nparpl = len(pl), len(par.plates)

        def one_term(ops, labs, second, lazy=False):
            present = set()
            for a, ls in zip(ops, labs):
                for ax, lab in enumerate(ls):
                    if a.shape[ax] != 1:
                        present.add(lab)
            # plate axes: kept (parent has them and some operand varies along them),
            # broadcast-compressed (parent has them, no operand varies), summed (parent
            # lacks them, some operand varies) or an integer factor (parent lacks them and
            # every operand is unit there -- utils/misc.py:761-802)
            mult = 1
            lout, final = [], []
            for ax, lab in enumerate(pl):
                pax = ax - (npl - nparpl)
                in_parent = pax >= 0 and par.plates[pax] != 1
                if in_parent:
                    if lab in present:
                        lout.append(lab)
                        final.append(sizes[lab])
                    else:
                        final.append(1)
                elif pax >= 0:
                    final.append(1)
                    if lab not in present:
                        mult *= sizes[lab]
                elif lab not in present:
                    mult *= sizes[lab]
            keys = ['k%d' % k for k in n.in_keys[index]]
            if second:
                keys = keys + ['K%d' % k for k in n.in_keys[index]]
            final = tuple(final) + tuple(sizes[k] for k in keys)
            # plate-free factors (tau of the observed child's message) multiply the result when
            # that is the smaller array, else the smallest operand -- never the (D, N) data
            ones = [a for a in ops if a.size == 1]
            if ones and len(ops) - len(ones) >= 1:
                rest = [(a, ls) for a, ls in zip(ops, labs) if a.size != 1]
                small = min(range(len(rest)), key=lambda q: rest[q][0].size)
                nres = int(np.prod(final))
                if rest[small][0].size < nres:
                    a0 = rest[small][0]
                    for s_ in ones:
                        a0 = fuse(lambda a_, b_: a_ * b_, a0, s_.reshape(()))
                    rest[small] = (a0, rest[small][1])
                    ones = []
                if lazy and not ones and mult == 1 and len(rest) == 2 \
                        and max(a.size for a, _ in rest) >= int(
                            os.environ.get('BAYESPY_AMD_LAZY_DOT_MIN', 1 << 14)):
                    # the message stays a contraction of its two operands (the data and the other
                    # parent's means): the receiving node's update may stream the data itself
                    # (GenericPlan._shared_cov_update); anything else evaluates it on first use
                    outl, sz, nu = [], dict(sizes), 0
                    for ax, lab in enumerate(pl):
                        pax = ax - (npl - nparpl)
                        if pax < 0:
                            continue
                        if par.plates[pax] != 1 and lab in present:
                            outl.append(lab)
                        else:
                            sz['u%d' % nu] = 1
                            outl.append('u%d' % nu)
                            nu += 1
                    return LazyContract([a for a, _ in rest], [ls for _, ls in rest],
                                        outl + keys, sz, ())
                res = misc.contract([a for a, _ in rest], [ls for _, ls in rest], lout + keys,
                                    sizes, scale=float(mult)).reshape(final)
                for s_ in ones:
                    res = fuse(lambda a_, b_: a_ * b_, res, s_.reshape(()))
                return res
            res = misc.contract(ops, labs, lout + keys, sizes, scale=float(mult))
            return res.reshape(final)

        out = []
        for second in (False, True):
            m = m_child[1 if second else 0]
            if m is None:
                out.append(None)
                continue
            m = _arr(m)
            lm = pl + ['k%d' % k for k in n.out_keys]
            if second:
                lm = lm + ['K%d' % k for k in n.out_keys]
            if _is_lazy(m) and len(m.terms) == 1 and m.terms[0][0] == 1.0 \
                    and len(m.terms[0][1]) + len(ups) + (mask is not None) <= 6:
                # a product of arrays (tau * y): its factors join the contraction
                base_ops = list(m.terms[0][1])
                base_labs = [lm[len(lm) - f.ndim:] for f in base_ops]
            else:
                base_ops, base_labs = [m], [lm[len(lm) - m.ndim:]]
            if mask is not None:
                base_ops.append(mask)
                base_labs.append(pl[npl - mask.ndim:])
            if not second:
                ops, labs = list(base_ops), list(base_labs)
                for j, u in enumerate(ups):
                    if j == index:
                        continue
                    a = _arr(u[0])
                    lj = self._parent_labels(j, False)
                    ops.append(a)
                    labs.append(lj[len(lj) - a.ndim:])
                out.append(one_term(ops, labs, False, lazy=getattr(self, '_lazy_first', False)))
                continue
            # second moments of the other parents: dense, or factored (Cov + <x><x>^T) and then
            # expanded term by term -- e.g. the message to W of a PCA model,
            # m (N Cov_X + sum_n <x_n><x_n>^T), without the (N, K, K) array
            per_parent = self._second_choices(ups, skip=index)
            terms = []
            if not all(len(alts) > 1 for alts in per_parent):
                per_parent = []           # a dense second moment among them: the dense product
                terms = None
            for kinds, extra in (self._picks(per_parent) if terms is not None else ()):
                if len(base_ops) + len(extra) > 6:
                    # more operands than one launch takes: fall back to the dense arrays
                    terms = None
                    break
                terms.append(one_term(base_ops + [o[0] for o in extra],
                                      base_labs + [o[1] for o in extra], True))
            if terms is None:
                ops, labs = list(base_ops), list(base_labs)
                for j, u in enumerate(ups):
                    if j == index:
                        continue
                    a = DArray(_arr(u[1]).t)
                    lj = self._parent_labels(j, True)
                    ops.append(a)
                    labs.append(lj[len(lj) - a.ndim:])
                out.append(one_term(ops, labs, True))
            else:
                out.append(self._add_terms(terms) if len(terms) > 1 else terms[0])
        return out


def make_family(node):
    from .extension import registered_family
    from .families_extra import make_extra_family
    # node types registered from outside the package (plans/extension.py: the reference's
    # Distribution contract, writingnodes.rst) come first: a registration may also replace a
    # built-in family
    fam = registered_family(node)
    if fam is not None:
        return fam
    fam = make_extra_family(node)
    if fam is not None:
        return fam
    if isinstance(node, Mixture):
        return MixtureFamily(node, make_family(node._proto))
    if isinstance(node, Gamma):
        return GammaFamily(node)
    if isinstance(node, GaussianGamma):
        return GaussianGammaFamily(node)
    if isinstance(node, GaussianToGaussianGamma):
        return GaussianToGaussianGammaFamily(node)
    if isinstance(node, WrapToGaussianGamma):
        return WrapToGaussianGammaFamily(node)
    if isinstance(node, GaussianARD):
        return GaussianARDFamily(node)
    if isinstance(node, Gaussian):
        return GaussianFamily(node)
    if isinstance(node, Wishart):
        return WishartFamily(node)
    if isinstance(node, Dirichlet):
        return DirichletFamily(node)
    if isinstance(node, Multinomial):
        return MultinomialFamily(node)
    if isinstance(node, Categorical):
        return CategoricalFamily(node)
    if isinstance(node, SumMultiply):
        return SumMultiplyFamily(node)
    if isinstance(node, GaussianMarkovChain):
        return GaussianMarkovChainFamily(node)
    if isinstance(node, MarkovChainToGaussian):
        return ChainToGaussianFamily(node)
    raise NotImplementedError('no device family for node type %s (a node type defined outside the '
                              'package registers its formulas with '
                              'bayespy_amd.inference.register_family, plans/extension.py)'
                              % type(node).__name__)


# ---------------------------------------------------------------------------
# the plan: state + routing
# ---------------------------------------------------------------------------
class _State:
    __slots__ = ('u', 'phi', 'g', 'f', 'observed', 'mask', 'ready', 'u_obs', 'obs_mask',
                 'partial', 'stale')

    def __init__(self):
        self.u = self.phi = None
        self.g = self.f = None
        self.observed = False
        self.mask = None
        self.ready = False
        self.u_obs = None        # fixed moments of the data (partially observed node)
        self.obs_mask = None     # 0/1 device array over the plates: where data were given
        self.partial = False     # observed with an array mask that leaves plates latent
        self.stale = False       # q of the latent plates awaits a refresh (leaf nodes: lazily)


def _operation(method):
    """Run a plan-level operation inside Runtime.operation() (one stream lookup, validity
    checks read back together at the end)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        rt = self.rt
        if rt._op_depth == 0:
            self._graph_note(method.__name__, args)
        prev = misc._CUR_MEMO[0]
        misc._CUR_MEMO[0] = self.__dict__.setdefault('_contract_memo', {})
        try:
            with rt.operation():
                if rt._op_depth == 1:
                    self._seed_sums()
                return method(self, *args, **kwargs)
        finally:
            misc._CUR_MEMO[0] = prev
    wrapped._notes_graph = True
    return wrapped


def _noting(cls):
    """Every public operation of the plan reports to the graph bookkeeping (graph_iter.py): an
    operation outside the recorded sweep decides whether the recorded graph still stands."""
    import functools
    for name, fn in list(vars(cls).items()):
        if name.startswith('_') or not callable(fn) or isinstance(fn, (staticmethod, classmethod,
                                                                         property)) \
                or getattr(fn, '_notes_graph', False):
            continue

        def make(fn):
            @functools.wraps(fn)
            def noted(self, *args, **kwargs):
                if self.rt._op_depth == 0:
                    self._graph_note(fn.__name__, args)
                return fn(self, *args, **kwargs)
            noted._notes_graph = True
            return noted
        setattr(cls, name, make(fn))
    return cls


@_noting
class GenericPlan(GraphIteration):

    @staticmethod
    def describe():
        return 'any graph of Gamma, GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, ' \
               'Mixture, SumMultiply/Dot nodes (generic device message passing)'

    def __init__(self, nodes):
        self.all = []
        seen = set()

        def visit(n):
            if id(n) in seen:
                return
            seen.add(id(n))
            for p in n.parents:
                visit(p)
            self.all.append(n)
            for c, _ in n.children:
                visit(c)
        for n in nodes:
            visit(n)
        self.family = {}
        self.state = {}
        for n in self.all:
            if isinstance(n, Constant):
                continue
            self.family[id(n)] = make_family(n)
            if isinstance(n, Stochastic):
                self.state[id(n)] = _State()
            n._plan = self
        self._const_cache = {}
        self._masks_ready = False
        self._graph_init()

    def nodes(self):
        return [n for n in self.all if not isinstance(n, Constant)]

    def has_state(self):
        """Device state exists (a recompilation would discard it)."""
        return any(st.ready for st in self.state.values())

    def invalidate(self, node):
        st = self.state.get(id(node))
        if st is not None:
            st.ready = False
        self._masks_ready = False

    # -- moments -----------------------------------------------------------------------
    def _ensure(self, node):
        """Materialise the device state of a stochastic node (prior / value / data)."""
        st = self.state[id(node)]
        if st.ready:
            return st
        fam = self.family[id(node)]
        st.ready = True          # guards recursion through parents
        if node.observed:
            data, om = node._data, node._mask
            st.partial = om is not True and not bool(np.all(om))
            if st.partial:
                # Missing entries usually carry NaN / inf placeholders.  The reference never
                # reads them: it writes moments with np.copyto(where=mask)
                # (stochastic.py:223-250).  Here masks multiply, so the placeholders are
                # replaced by a finite value before upload (0 * finite = 0, never NaN).
                data = self._fill_missing(node, fam, data, om)
                st.obs_mask = DArray.from_host(
                    np.ascontiguousarray(np.broadcast_to(np.asarray(om, dtype=bool), node.plates)
                                         .astype(np.float64)))
            u, f = fam.fixed_moments_and_f(data)
            st.u, st.f, st.g = u, f, None
            st.u_obs = u
            st.stale = st.partial
            st.observed = True
            st.phi = None
        else:
            st.observed = False
            up = self._parent_moments(node)
            st.phi = fam.phi_from_parents(up)
            init = node._init
            if init is None:
                st.u, st.g = fam.moments_and_cgf(st.phi)      # initialize_from_prior
            elif init[0] == 'parameters':
                # initialize_from_parameters (expfamily.py:187-190): the given values stand
                # in for the parents
                if len(init[1]) != len(node.parents):
                    raise ValueError('%s has %d parents, %d parameters were given'
                                     % (node.name, len(node.parents), len(init[1])))
                ups = [fam.constant_moments(i, a) for i, a in enumerate(init[1])]
                st.phi = fam.phi_from_parents(ups)
                st.u, st.g = fam.moments_and_cgf(st.phi)
            elif init[0] == 'phi':
                st.phi = [_arr(np.array(x, dtype=np.float64)) if not isinstance(x, DArray) else x
                          for x in init[1]]
                st.u, st.g = fam.moments_and_cgf(st.phi)
            elif init[0] == 'value':
                st.u, _ = fam.fixed_moments_and_f(init[1])
                st.g = np.inf
            else:
                u, _ = fam.moments_and_cgf(st.phi)
                st.u, _ = fam.fixed_moments_and_f(self._sample(node, fam, u))
                st.g = np.inf
            st.f = None
        return st

    def _fill_missing(self, node, fam, data, om):
        """``data`` with a finite placeholder where the observation mask is False."""
        fill = float(getattr(fam, 'missing_fill', 0.0))
        nd = len(node.dims[0])
        m = np.asarray(om, dtype=bool)
        m = m.reshape(m.shape + (1,) * nd)
        torch = self.rt.torch
        if isinstance(data, torch.Tensor):
            mt = torch.from_numpy(np.ascontiguousarray(m)).to(data.device)
            return torch.where(mt, data, torch.full((), fill, dtype=data.dtype, device=data.device))
        a = np.asarray(data)
        return np.where(m, a, np.asarray(fill, dtype=a.dtype if a.dtype.kind == 'f' else np.float64))

    def _refresh_partial(self, node, st):
        """A partially observed node: the plates without data are ordinary latent plates, the
        reference updates them (stochastic.py:276-282 with ``mask = not observed``).  q of
        those plates = prior from the parents + messages of the children; the observed plates
        keep the fixed moments of the data."""
        fam = self.family[id(node)]
        phi = self._optimal_phi(node)
        uq, gq = fam.moments_and_cgf(phi)
        st.u = [fuse(lambda m, a, b: m * a + (1.0 - m) * b, _trail(st.obs_mask, len(node.dims[i])),
                     _arr(st.u_obs[i]), _arr(uq[i])) for i in range(len(uq))]
        st.phi, st.g = phi, gq          # q of the latent plates (bound term, expfamily.py:431-466)
        st.stale = False

    def _sample(self, node, fam, u):
        """A draw from the current q (initialize_from_random, expfamily.py:206-212); set-up
        only, on the host -- RNG streams are not part of the parity contract."""
        if isinstance(node, Categorical):
            p = np.broadcast_to(u[0].numpy(), node.plates + (fam.K,)).reshape(-1, fam.K)
            c = np.cumsum(p, axis=1)
            r = np.random.rand(p.shape[0], 1) * c[:, -1:]
            return (r > c).sum(axis=1).clip(0, fam.K - 1).reshape(node.plates)
        if isinstance(node, (GaussianARD, Gaussian)):
            shape = node.plates + node.dims[0]
            m = np.broadcast_to(u[0].numpy(), shape)
            if node.ndim == 0:
                v = np.broadcast_to(u[1].numpy(), shape) - m * m
                return m + np.sqrt(np.maximum(v, 0)) * np.random.randn(*shape)
            D = int(np.prod(node.dims[0]))
            mm = np.broadcast_to(u[1].numpy(), node.plates + node.dims[1]).reshape(-1, D, D)
            mf = m.reshape(-1, D)
            cov = mm - mf[:, :, None] * mf[:, None, :]
            Lc = np.linalg.cholesky(cov + 1e-12 * np.eye(D))
            z = np.random.randn(mf.shape[0], D)
            return (mf + np.einsum('nij,nj->ni', Lc, z)).reshape(shape)
        if hasattr(fam, 'sample'):
            return fam.sample(self.state[id(node)])
        if isinstance(node, Gamma):
            st = self.state[id(node)]
            a = np.broadcast_to(_arr(st.phi[1]).numpy(), node.plates)
            b = np.broadcast_to(-_arr(st.phi[0]).numpy(), node.plates)
            return np.random.gamma(a, 1.0 / b)
        if isinstance(node, Dirichlet):
            st = self.state[id(node)]
            a = np.broadcast_to(_arr(st.phi[0]).numpy(), node.plates + node.dims[0])
            x = np.random.gamma(a)
            x = x / x.sum(axis=-1, keepdims=True)
            return x[..., 0] if type(node).__name__ == 'Beta' else x     # beta.py:100-105
        raise NotImplementedError('random draws for %s' % type(node).__name__)

    def _moments(self, node):
        if isinstance(node, Stochastic):
            return self._ensure(node).u
        # Deterministic nodes hold no state (deterministic.py:62-64), and the reference
        # recomputes their moments on every request -- e.g. the D x N x K^2 contraction of a
        # PCA model twice per iteration.  Moment arrays are never modified in
        # place here, so "same parent arrays" means "same result": keep the last one.
        fam = self.family[id(node)]
        ups = self._parent_moments(node)
        key = tuple(id(a) for u in ups for a in u)
        cache = self.__dict__.setdefault('_det_cache', {})
        hit = cache.get(id(node))
        if hit is not None and hit[0] == key:
            return hit[2]
        out = fam.moments(ups)
        if os.environ.get('BAYESPY_AMD_DET_CACHE', '1') != '0':
            cache[id(node)] = (key, ups, out)          # `ups` keeps the keyed arrays alive
        return out

    def _parent_moments(self, node, skip=None):
        """Moments of the parents; ``skip``: the index of a parent whose moments the caller does
        not read (the target of a message, by conjugacy) -- they are evaluated only if somebody
        indexes them after all."""
        fam = self.family[id(node)]

        def one(i, p):
            if isinstance(p, Constant):
                key = (id(node), i)
                if key not in self._const_cache:
                    self._const_cache[key] = fam.constant_moments(i, p.value)
                return self._const_cache[key]
            return self._moments(p)
        out = _LazyList()
        for i, p in enumerate(node.parents):
            if i == skip and not isinstance(p, Constant):
                out.append(_Deferred(lambda i=i, p=p: one(i, p)))
            else:
                out.append(one(i, p))
        return out

    # -- masks (node.py:457-526) -----------------------------------------------------------
    def _update_masks(self):
        if self._masks_ready:
            return
        memo = {}

        def mask_of(n):
            if id(n) in memo:
                return memo[id(n)]
            m = np.array(False)
            for c, idx in n.children:
                if isinstance(c, Constant) or id(c) not in self.family:
                    continue
                cm = mask_of(c)
                fam = self.family[id(c)]
                pm = fam.mask_to_parent(idx, cm)
                # "sum" (logical or) over the plates that are unit in this node
                nd = len(n.plates)
                pm = np.asarray(pm)
                while pm.ndim > nd:
                    pm = np.any(pm, axis=0)
                tgt = (1,) * (nd - pm.ndim) + pm.shape
                pm = pm.reshape(tgt)
                axes = tuple(i for i in range(nd) if n.plates[i] == 1 and pm.shape[i] != 1)
                if axes:
                    pm = np.any(pm, axis=axes, keepdims=True)
                if self._is_sharded(c) and not self._is_sharded(n):
                    # the "or" over a plate partitioned over the ranks is global: a replicated node
                    # is an ignored plate only where NO rank has an active child (node.py:457-526)
                    pm = self._any_over_ranks(np.broadcast_to(pm, np.broadcast_shapes(
                        pm.shape, tuple(n.plates))))
                m = np.logical_or(m, pm)
            if isinstance(n, Stochastic) and n.observed:
                om = np.asarray(n._mask, dtype=bool)
                m = np.logical_or(m, om)
            memo[id(n)] = m
            return m
        for n in self.nodes():
            m = mask_of(n)
            if isinstance(n, Stochastic):
                self.state[id(n)].mask = m
            else:
                n._gmask = m
        self._dev_masks = {}
        self._masks_ready = True
        self._mask_epoch = getattr(self, '_mask_epoch', 0) + 1

    def _any_over_ranks(self, mask):
        rt = self.rt
        t = rt.torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float64)).to(rt.device)
        rt.all_reduce_sum_(t)
        return t.cpu().numpy() > 0.0

    def _mask_array(self, node):
        self._update_masks()
        if isinstance(node, Stochastic):
            return self.state[id(node)].mask
        return node._gmask

    def get_mask(self, node):
        return np.array(self._mask_array(node))

    def _mask_factor(self, key, make_host_mask):
        """None when everything is active, else a 0/1 device array.  Cached per `key` until the
        masks change (host masks are N-sized: scanning them every message would dominate)."""
        self._update_masks()
        if key not in self._dev_masks:
            host_mask = np.asarray(make_host_mask())
            if np.all(host_mask):
                ent = (None, True)
            else:
                ent = (DArray.from_host(host_mask.astype(np.float64)), bool(np.any(host_mask)))
            self._dev_masks[key] = ent
        return self._dev_masks[key]

    # -- message routing (node.py:570-688) ------------------------------------------------------
    def _message_to_parent(self, child, index):
        fam = self.family[id(child)]
        parent = child.parents[index]
        # plate multiplier: the part of this node's multiplier the parent does not carry
        # (node.py:589-632)
        r = multiplier_factor(child.plates_multiplier, parent.plates_multiplier)
        if getattr(fam, 'deterministic', False):
            if r != 1.0:
                raise NotImplementedError('plate multipliers through %s are not built'
                                          % type(child).__name__)
            m_child = self._messages_from_children(child)
            ups = self._parent_moments(child)
            mask, _ = self._mask_factor((id(child), 'self'), lambda: self._mask_array(child))
            msgs = fam.message_to_parent(index, m_child, ups, mask)
            if getattr(fam, 'plate_sum', False) and not isinstance(parent, Constant):
                # families that answer with the node's own plates: sum over the plates the
                # parent does not have (node.py:633-655)
                for i, m in enumerate(msgs):
                    if m is None or not isinstance(m, DArray):
                        continue
                    dims = tuple(parent.dims[i])
                    own = tuple(fam.plates_to_parent(index)) \
                        if hasattr(fam, 'plates_to_parent') else tuple(child.plates)
                    msgs[i] = misc.sum_multiply_to_plates(
                        m, to_plates=parent.plates + dims, from_plates=own + dims, ndim=0)
            return msgs
        u = self._moments(child)
        indep = getattr(fam, 'message_independent_of_target', False)
        up = self._parent_moments(child, skip=index if indep else None)
        # A message is a function of the child's moments and of the OTHER parents' moments
        # (conjugacy): while those arrays are the same objects the last answer stands -- e.g. the
        # message of the observed node of a PCA model to F, asked for once by W.update() and once
        # by X.update() of every iteration (two (D, N) passes each time).
        ckey = None
        if indep and os.environ.get('BAYESPY_AMD_DET_CACHE', '1') != '0':
            deps = list(u) + [a for j in range(len(up)) if j != index for a in up[j]]
            if all(isinstance(a, DArray) for a in deps):
                self._update_masks()
                ckey = (tuple(id(a) for a in deps), self._mask_epoch, r)     # (a freed dict's id can come back)
                hit = self.__dict__.setdefault('_msg_cache', {}).get((id(child), index))
                if hit is not None and hit[0] == ckey:
                    return list(hit[2])
        fam._terms_ok = True
        try:
            msgs = fam.message_to_parent(index, u, up)
        finally:
            fam._terms_ok = False
        plates_self = tuple(fam.plates_to_parent(index))
        mask, _ = self._mask_factor(
            (id(child), index),
            lambda: fam.mask_to_parent(index, np.asarray(self._mask_array(child))))
        out = []
        for i, m in enumerate(msgs):
            if m is None:
                out.append(None)
                continue
            nd = len(parent.dims[i])
            to_shape = parent.plates + parent.dims[i]
            if _is_lazy(m) and mask is None and r == 1.0 \
                    and tuple(m.shape) == tuple(plates_self) + tuple(parent.dims[i]) == tuple(to_shape):
                out.append(m)          # nothing to sum: the parent reads the factors (or .t)
                continue
            terms = m.terms if isinstance(m, Terms) or _is_lazy(m) else \
                [(1.0, list(m) if isinstance(m, tuple) else [_arr(m)])]
            parts = []
            for coef, factors in terms:
                factors = list(factors)
                mshape = broadcasted_shape(*[f.shape for f in factors])
                dims = broadcasted_shape(mshape[len(mshape) - nd:], parent.dims[i]) if nd else ()
                from_shape = plates_self + dims
                if mask is not None:
                    factors.append(_trail(mask, nd))
                parts.append((float(coef) * r, self._plate_sum(factors, to_shape, from_shape)))
            out.append(_wsum(parts))
        if ckey is not None:
            self._msg_cache[(id(child), index)] = (ckey, (u, up), list(out))    # keeps the keyed arrays alive
        return out

    @property
    def rt(self):
        from ...device import get_runtime
        return get_runtime()

    def _is_sharded(self, node):
        """The node carries a plate axis partitioned over the ranks: it was declared with
        Node.shard(), or it descends from such a node (a child's plates contain its parents'
        plates, node.py:303-345, so the partition is inherited)."""
        memo = self.__dict__.setdefault('_shard_memo', {})
        key = id(node)
        if key not in memo:
            memo[key] = False          # guards cycles; graphs are DAGs
            memo[key] = (getattr(node, '_shard_axis', None) is not None
                         or any(self._is_sharded(p) for p in node.parents))
        if not memo[key]:
            return False
        rt = self.rt
        rt._refresh_dist()
        # (BAYESPY_AMD_SHARD_WORLD1=1: a world of ONE rank runs the sharded code path too, every
        # collective included -- how the RCCL path is exercised on a one-GPU box)
        return rt.world > 1 or os.environ.get('BAYESPY_AMD_SHARD_WORLD1') == '1'

    def _messages_from_children(self, node):
        total = [None] * len(node.dims)
        partial = [None] * len(node.dims)     # sums over a sharded plate: local parts only
        replicated = not self._is_sharded(node)
        for c, idx in node.children:
            if id(c) not in self.family:
                continue
            m = self._message_to_parent(c, idx)
            acc = partial if (replicated and self._is_sharded(c)) else total
            for i in range(len(total)):
                if m[i] is None:
                    continue
                acc[i] = m[i] if acc[i] is None else fuse(lambda a, b: a + b, acc[i], m[i])
        for i, p in enumerate(partial):
            if p is None:
                continue
            # child -> parent message sum over the sharded plate (node.py:650, dot.py:581):
            # complete it over the ranks (RCCL all-reduce on GPUs)
            p = fuse(lambda x: x + 0.0, p)          # private dense copy: reduced in place
            self.rt.all_reduce_sum_(p.t)
            total[i] = p if total[i] is None else fuse(lambda a, b: a + b, total[i], p)
        return total

    # -- node operations -------------------------------------------------------------------------
    def _phi_parts(self, node, lazy=False):
        """(prior natural parameters from the parents, summed messages of the children);
        ``lazy``: a Dot child may hand its first-moment message over as a contraction."""
        fam = self.family[id(node)]
        up = self._parent_moments(node)
        phi = fam.phi_from_parents(up)
        flagged = []
        if lazy:
            for c, _ in node.children:
                cf = self.family.get(id(c))
                if isinstance(cf, SumMultiplyFamily):
                    cf._lazy_first = True
                    flagged.append(cf)
        try:
            msgs = self._messages_from_children(node)
        finally:
            for cf in flagged:
                cf._lazy_first = False
        return phi, msgs

    def _combine_phi(self, node, phi, msgs):
        a = float(getattr(node, 'annealing', 1.0))
        phi = list(phi)
        for i in range(len(phi)):
            if msgs[i] is not None:
                phi[i] = fuse(lambda p, m: p + m, _arr(phi[i]), msgs[i])
            if a != 1.0:
                # deterministic annealing (expfamily.py:343-350)
                phi[i] = fuse(lambda p, a_=a: a_ * p, _arr(phi[i]))
        return phi

    def _optimal_phi(self, node):
        """Natural parameters of the VB-optimal factor: prior from the parents plus the
        messages of the children (expfamily.py:215-257)."""
        phi, msgs = self._phi_parts(node)
        return self._combine_phi(node, phi, msgs)

    # -- the fused update of a shared-covariance Gaussian node (vmp_gaussian_shared_update) ------
    def _shared_cov_candidate(self, node, fam):
        if os.environ.get('BAYESPY_AMD_SHARED_UPDATE', '1') == '0':
            return False
        if type(fam) not in (GaussianARDFamily, GaussianFamily) or fam.ndim != 1 \
                or getattr(fam, 'mu_gg', False):
            return False
        if float(getattr(node, 'annealing', 1.0)) != 1.0:
            return False
        K = int(fam.shape[0])
        nplates = int(np.prod(node.plates)) if node.plates else 1
        return 1 <= K <= 64 and nplates >= max(_factored_min_plates(), 2) \
            and hasattr(self.rt.lib, 'vmp_gaussian_shared_update')

    @staticmethod
    def _dot_operands(msg, K):
        """(Y tensor, stride along its rows d, stride along the plates n, D, N, B array, its strides)
        of a first-moment Dot message m_nk = sum_d Y[d, n] B[d, k] kept as a contraction, or None."""
        if not isinstance(msg, LazyContract) or len(msg.ops) != 2 or msg._dense is not None:
            return None
        big = [lab for lab in msg.out if msg.sizes[lab] != 1]
        if len(big) != 2 or msg.sizes[big[-1]] != K:
            return None
        nlab, klab = big

        def varying(a, ls):
            return {lab: a.t.stride(ax) for ax, lab in enumerate(ls) if a.shape[ax] != 1}
        v = [varying(a, ls) for a, ls in zip(msg.ops, msg.labs)]
        for iy, ib in ((0, 1), (1, 0)):
            vy, vb = v[iy], v[ib]
            if nlab in vy and klab not in vy and klab in vb and nlab not in vb:
                dl = [lab for lab in vy if lab != nlab]
                if len(dl) != 1 or set(vb) != {dl[0], klab}:
                    continue
                d = dl[0]
                D, N = int(msg.sizes[d]), int(msg.sizes[nlab])
                if D > 256 or (vy[nlab] != 1 and vy[d] != 1):
                    return None
                return (msg.ops[iy], vy[d], vy[nlab], D, N, msg.ops[ib], vb[d], vb[klab])
        return None

    def _shared_cov_update(self, node, st, fam, phi_p, msgs):
        """node.update() of a Gaussian node whose precision carries no plate axis (every plate
        shares Cov = (-2 phi1)^-1) as ONE pass: <x_n> = Cov (phi0_prior + m_n) written once, with
        the plate sums sum <x>, sum <x><x>^T (and sum y <x>^T when the message is the Dot message
        of a data array, which the pass then streams itself instead of reading a formed message)
        made on the fly -- gaussian.py:649-706 behind dot.py:581.  False: not this case."""
        rt = self.rt
        K = int(fam.shape[0])
        p1 = _arr(phi_p[1])
        if msgs[1] is not None:
            if not isinstance(msgs[1], DArray) or isinstance(msgs[1], (LazySum, LazyContract)):
                return False
            p1 = fuse(lambda p, m: p + m, p1, msgs[1])
        if p1.size != K * K or msgs[0] is None:
            return False
        m0, p0 = msgs[0], _arr(phi_p[0])
        if not isinstance(m0, DArray):
            return False
        xshape = tuple(broadcasted_shape(p0.shape, m0.shape))
        if len(xshape) < 1 or xshape[-1] != K:
            return False
        N = int(np.prod(xshape[:-1])) if len(xshape) > 1 else 1
        if N < 2 or int(np.prod(node.plates)) != N:
            return False          # (the means must span the node's plates: no plate multiplier)
        dot = self._dot_operands(m0, K)
        if dot is not None and dot[4] != N:
            dot = None
        if p0.size == K:
            p0v = contiguous(p0.reshape((K,)))
        else:
            # a prior that varies over the plates joins the message rows
            m0 = fuse(lambda p, m: p + m, p0, m0)
            p0v, dot = None, None
        U = linalg.chol(fuse(lambda p: -2.0 * p, p1.reshape((K, K))))
        cov = linalg.chol_inv(U)
        ld = linalg.chol_logdet(U)
        x = DArray.empty(xshape)
        torch = rt.torch
        D = dot[3] if dot is not None else 0
        stats = DArray.empty((K + K * K + D * K,))
        nbytes = int(rt.lib.vmp_gaussian_shared_update_workspace_bytes(D, K))
        ws = self.__dict__.setdefault('_gs_ws', {})
        if ws.get('n', -1) < nbytes:
            ws['t'] = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=rt.device)
            ws['n'] = nbytes
        vp = ctypes.c_void_p
        if dot is not None:
            Yop, y_sd, y_sn, D, _, Bop, b_sd, b_sk = dot
            rt.note_reads([Yop, Bop, p0v, cov])
            rc = rt.lib.vmp_gaussian_shared_update(
                rt.ctx, N, K, D, vp(Yop.t.data_ptr()), y_sd, y_sn, vp(Bop.t.data_ptr()), b_sd, b_sk,
                None, 0, 0, vp(p0v.t.data_ptr()), vp(cov.t.data_ptr()), vp(x.t.data_ptr()), K, 1,
                vp(stats.t.data_ptr()), vp(ws['t'].data_ptr()), nbytes)
            keep = [Yop, Bop, p0v, cov]
        else:
            m2 = contiguous(_arr(m0).broadcast_to(xshape)).reshape((N, K))
            rt.note_reads([m2, p0v, cov])
            rc = rt.lib.vmp_gaussian_shared_update(
                rt.ctx, N, K, 0, None, 0, 0, None, 0, 0, vp(m2.t.data_ptr()), K, 1,
                None if p0v is None else vp(p0v.t.data_ptr()), vp(cov.t.data_ptr()),
                vp(x.t.data_ptr()), K, 1, vp(stats.t.data_ptr()), vp(ws['t'].data_ptr()), nbytes)
            keep = [m2, p0v, cov]
        rt.check(rc)
        del keep          # (launched at once, never queued: stream order keeps the operands valid)
        npl = len(xshape) - 1
        sums = PlateSums(DArray(stats.t[:K]), DArray(stats.t[K:K + K * K].view(K, K)), n=N)
        if dot is not None:
            sums.yx = DArray(stats.t[K + K * K:].view(D, K))
            sums.ydesc = (int(Yop.t.data_ptr()), int(y_sd), int(y_sn), int(D))
            sums.ykeep = Yop
        covs = cov.reshape((1,) * npl + (K, K))
        ldp = ld.reshape((1,) * npl)
        phi1 = p1 if p1.ndim >= 2 and p1.shape[-2:] == (K, K) else p1.reshape((K, K))
        st.phi = [DerivedArray('gauss_phi0', (phi1, x), xshape), phi1]
        st.g = DerivedArray('gauss_g', (phi1, x, ldp), xshape[:-1])
        st.u = [x, FactoredMoment(covs, x, 1, logdet_prec=ldp, sums=sums)]
        self._seed_sums()
        return True

    def _seed_sums(self):
        """Offer the plate sums the fused updates made (PlateSums) to whoever asks for the same
        reductions: entries of the plan's memo under the signature misc._launch_sum_multiply forms
        for them -- sum_n y_n <x_n>^T for the Dot message to the other parent and for sum y <f> of the
        message to the noise precision, sum_n <x_n><x_n>^T for the second-moment message, for
        sum <f>^2 and for the node's own bound term.  (The memo is emptied when a sweep ends; the sums
        are state and are offered again.)"""
        memo = self.__dict__.get('_contract_memo')
        if memo is None:
            return
        for st in self.state.values():
            u = st.u
            if not st.ready or st.observed or not isinstance(u, list) or len(u) != 2:
                continue
            fm = u[1]
            if not isinstance(fm, FactoredMoment) or fm.sums is None:
                continue
            sm, x = fm.sums, fm.mean
            if x.ndim < 2 or not x.t.is_contiguous():
                continue
            K, N = int(x.shape[-1]), sm.n
            if N < misc._MEMO_MIN // max(K, 1) or N * K < misc._MEMO_MIN:
                continue
            xp = int(x.t.data_ptr())
            a, b = (xp, (1, 0, K)), (xp, (0, 1, K))
            memo[((a, b) if a < b else (b, a), ((K, False), (K, False), (N, True)), 1.0)] = \
                (sm.xx, [x])
            memo[(((xp, (1, K)),), ((K, False), (N, True)), 1.0)] = (sm.x, [x])
            if sm.yx is not None:
                yp, y_sd, y_sn, D = sm.ydesc
                a, b = (yp, (y_sd, 0, y_sn)), (xp, (0, 1, K))
                memo[((a, b) if a < b else (b, a), ((D, False), (K, False), (N, True)), 1.0)] = \
                    (sm.yx, [x, sm.ykeep])

    @_operation
    def update(self, node):
        if not isinstance(node, Stochastic):
            return
        st = self._ensure(node)
        if st.observed:
            if st.partial:
                # the latent plates see the Markov blanket as it is NOW, like any other update
                # (VB.update visits an observed leaf first: its q is one iteration behind W, X)
                self._refresh_partial(node, st)
            return
        fam = self.family[id(node)]
        cand = self._shared_cov_candidate(node, fam)
        phi_p, msgs = self._phi_parts(node, lazy=cand)
        if cand and self._shared_cov_update(node, st, fam, phi_p, msgs):
            return
        phi = self._combine_phi(node, phi_p, msgs)
        st.phi = phi
        st.u, st.g = fam.moments_and_cgf(phi)

    @_operation
    def gradient_step(self, nodes, scale=1.0):
        """phi <- phi + scale * (phi_optimal - phi) for all ``nodes`` at once: a step along
        the Riemannian (natural) gradient of the lower bound (vmp.py:432-440 with
        expfamily.py:296-340), the global update of stochastic variational inference."""
        todo = []
        for node in nodes:
            if not isinstance(node, Stochastic):
                continue
            st = self._ensure(node)
            if st.observed:
                continue
            todo.append((node, st, self._optimal_phi(node)))      # all gradients first
        s = float(scale)
        for node, st, opt in todo:
            phi = [fuse(lambda p, q, s_=s: p + s_ * (q - p), _arr(p0), _arr(q0))
                   for p0, q0 in zip(st.phi, opt)]
            st.phi = phi
            st.u, st.g = self.family[id(node)].moments_and_cgf(phi)

    def _lower_bound_device(self, node, ignore_masked=True):
        """The node's lower-bound term (expfamily.py:400-480) as (device scalar | None,
        host factor): no device->host read here."""
        if not isinstance(node, Stochastic):
            return None, 0.0
        st = self._ensure(node)
        fam = self.family[id(node)]
        up = self._parent_moments(node)
        closed = None
        # annealing temperature: the entropy part of the term, i.e. phi and g of q, is
        # multiplied by T (expfamily.py:403-411)
        T = 1.0 / float(getattr(node, 'annealing', 1.0))
        partial = st.observed and st.partial
        if st.observed and not partial and hasattr(fam, 'observed_bound_terms'):
            # a fully observed node: every part of its term is a product of moment arrays --
            # plate-summed product by product, no plates-sized temporaries
            terms = fam.observed_bound_terms(st.u, up)
            if terms is not None:
                if isinstance(fam, MixtureFamily):
                    # (its terms leave f(y) out, like the message they share an array with)
                    terms = list(terms) + [(1.0, [st.f]) if isinstance(st.f, DArray)
                                           else (float(st.f), [])]
                return self._finish_bound(node, terms, ignore_masked)
        phi_p = fam.phi_from_parents(up)
        L = _arr(fam.cgf_from_parents(up))
        fast = self._shared_cov_bound(node, st, fam, phi_p, L, T, ignore_masked) \
            if not st.observed else None
        if fast is not None:
            return fast
        if partial:
            # np.where(observed, f, -T g) and phi_q zeroed on the observed plates
            # (expfamily.py:431-466): the latent plates of the node count like any latent node
            if st.stale:
                self._refresh_partial(node, st)
            fobs = st.f if isinstance(st.f, DArray) else float(st.f)
            L = fuse(lambda a, m, f, g, T_=T: a + m * f - (1.0 - m) * T_ * g, L, st.obs_mask,
                     fobs, _arr(st.g))
        elif st.observed:
            L = fuse(lambda a, b: a + b, L, st.f if isinstance(st.f, DArray) else float(st.f))
        else:
            if not isinstance(st.g, DArray):
                return None, (float(-np.inf) if np.isinf(st.g) else float('nan'))
            closed = getattr(fam, 'q_term', None) if T == 1.0 else None
            if closed is not None:
                # Gaussian factors: -(g_q + phi_q . u_q) in closed form, no K x K contraction
                L = fuse(lambda a, q: a + q, L, closed(st.phi, st.u, st.g))
            else:
                L = fuse(lambda a, g, T_=T: a - T_ * g, L, st.g)
        for i, nd in enumerate(len(d) for d in node.dims):
            if closed is not None and nd > 0:
                # finite Gaussian prior parameters: phi_p . u as one contraction, no temporary
                if i == 1 and isinstance(st.u[i], FactoredMoment):
                    L = fuse(lambda a, b: a + b, L, _inner_second(phi_p[i], st.u[i], nd // 2))
                    continue
                L = fuse(lambda a, b: a + b, L,
                         misc.sum_multiply(_arr(phi_p[i]), _arr(st.u[i]),
                                           axis=tuple(range(-nd, 0))))
                continue
            if partial:
                t = fuse(lambda pp, pq, m, u, T_=T:
                         da.where_nonzero(u, pp - T_ * (1.0 - m) * pq) * u,
                         _arr(phi_p[i]), _arr(st.phi[i]), _trail(st.obs_mask, nd), _arr(st.u[i]))
            elif st.observed or closed is not None:
                t = fuse(lambda pp, u: da.where_nonzero(u, pp) * u, _arr(phi_p[i]), _arr(st.u[i]))
            else:
                t = fuse(lambda pp, pq, u, T_=T: da.where_nonzero(u, pp - T_ * pq) * u,
                         _arr(phi_p[i]), _arr(st.phi[i]), _arr(st.u[i]))
            L = fuse(lambda a, b: a + b, L, _sum_last(t, nd))
        return self._finish_bound(node, [(1.0, [L])], ignore_masked)

    def _shared_cov_bound(self, node, st, fam, phi_p, cgf, T, ignore_masked):
        """Bound term of a latent Gaussian node whose posterior covariance is shared over its plates
        (a ``FactoredMoment``), from plate SUMS instead of per-plate arrays:

            sum_n [ cgf_p + k/2 - log|Lambda|/2 + phi_p0 . <x_n> + phi_p1 : (Cov + <x_n><x_n>^T) ]

        -- the entropy part -(g_q + phi_q . u_q) of a Gaussian is k/2 - log|Lambda|/2 whatever its mean
        (expfamily.py:449-468 evaluates it per plate), and the quadratic part needs the plates only
        through sum_n <x_n> and sum_n <x_n><x_n>^T, which the sweep has formed for the messages
        anyway.  One pass over <x> (two when the second-moment sum is not remembered) instead of
        eleven over plates x K arrays.  Declines (None) whenever a plate mask, annealing, a prior
        that varies over the plates of <x>, or a moment without its log-determinant is involved."""
        if T != 1.0 or getattr(fam, 'q_term', None) is None or len(node.dims) != 2:
            return None
        u0, xx = st.u
        nd = len(node.dims[0])
        if nd < 1 or not isinstance(xx, FactoredMoment) or xx.logdet_prec is None \
                or not isinstance(st.g, DArray):
            return None
        mask, any_active = self._mask_factor((id(node), 'self'), lambda: self._mask_array(node))
        if (mask is not None and ignore_masked) or not any_active:
            return None
        x, cov = _arr(xx.mean), _arr(xx.cov)
        p0, p1 = _arr(phi_p[0]), _arr(phi_p[1])
        npl = len(node.plates)
        xpl = x.shape[:x.ndim - nd]
        xpl = (1,) * (npl - len(xpl)) + tuple(xpl)
        # the prior's parameters must not vary over a plate that <x> spans in full
        for arr, nv in ((p0, nd), (p1, 2 * nd)):
            apl = arr.shape[:arr.ndim - nv]
            apl = (1,) * (npl - len(apl)) + tuple(apl)
            if len(apl) != npl or any(a != 1 and b != 1 for a, b in zip(apl, xpl)):
                return None
        cpl = cov.shape[:cov.ndim - 2 * nd]
        if any(c != 1 for c in cpl):
            return None
        k = float(np.prod(node.dims[0]))
        if p0.size != int(k) or p1.size != int(k) * int(k):
            return None           # a prior that varies over the plates: the general route
        # plate-constant parts: the plate sum multiplies them with the number of plates
        qc = fuse(lambda ld, k_=k: 0.5 * k_ - 0.5 * ld, _arr(xx.logdet_prec))
        tr = misc.sum_multiply(p1, cov, axis=tuple(range(-2 * nd, 0)))
        terms = [(1.0, [cgf]), (1.0, [qc]), (1.0, [tr])]
        # plate sums of <x> and <x><x>^T over the plates <x> spans (multiplier of the plates it
        # lacks included), contracted with the prior's parameters
        D = int(np.prod(node.dims[0]))
        xf = x.reshape(x.shape[:x.ndim - nd] + (D,))
        if xx.sums is not None and nd == 1 and xx.sums.n == int(np.prod(node.plates)):
            # the pass that wrote <x> summed it (and <x><x>^T) over these very plates
            s1, s2 = xx.sums.x, xx.sums.xx
        else:
            s1 = misc.sum_multiply_to_plates(xf, to_plates=(), from_plates=node.plates, ndim=1)
            s2 = misc.sum_multiply_to_plates(_trail(xf, 1), xf.reshape(xf.shape[:-1] + (1, D)),
                                             to_plates=(), from_plates=node.plates, ndim=2)
        p0f, p1f = p0.reshape((-1, D)), p1.reshape((-1, D, D))
        pre = fuse(lambda a, b: a + b, misc.sum_multiply(p0f, s1.reshape((1, D))),
                   misc.sum_multiply(p1f, s2.reshape((1, D, D))))
        return self._finish_bound(node, terms, ignore_masked, presummed=pre)

    def _plate_sum(self, factors, to_plates, from_plates):
        """sum over the plates of prod(factors), remembered while the factor arrays live: the
        same sums feed a node's message to its precision parent and its lower-bound term
        (sum x <m>, sum <m^2>, sum x^2 of an observed Gaussian node), and the ones over constants
        never change.  Plate-free factors multiply the sum afterwards so that they do not key it."""
        import weakref
        factors = list(factors)
        for i, f in enumerate(factors):
            if _is_lazy(f):
                # a sum of products among the factors: one plate sum per product
                rest = factors[:i] + factors[i + 1:]
                return _wsum([(coef, self._plate_sum(rest + list(fs), to_plates, from_plates))
                              for coef, fs in f.terms])
        if any(isinstance(f, LazyContract) for f in factors):
            return self._plate_sum_contract(factors, to_plates, from_plates)
        big = [f for f in factors if f.size > 1]
        small = [f for f in factors if f.size <= 1]
        if not big:
            big, small = list(factors), []
        # factors that do not vary along any summed axis leave the (long) reduction and multiply
        # its (small) result: sum_n r_nk M_k = M_k sum_n r_nk, and the sum is remembered under the
        # varying factors alone (sum_n r_nk serves the messages to the precision, to its degrees of
        # freedom and to the mixing weights of a mixture)
        full = tuple(broadcasted_shape(*[f.shape for f in big]))
        nf, nt = len(full), len(tuple(to_plates))
        to = ((1,) * (nf - nt) + tuple(to_plates)) if nf >= nt else tuple(to_plates)[nt - nf:]
        red = [i for i in range(nf) if full[i] != 1 and to[i] == 1]
        inv = []
        if red and len(big) >= 2:
            def varies(f):
                off = nf - f.ndim
                return any(ax - off >= 0 and f.shape[ax - off] != 1 for ax in red)
            var = [f for f in big if varies(f)]
            if var and len(var) < len(big) \
                    and int(np.prod([full[i] for i in red])) >= int(
                        os.environ.get('BAYESPY_AMD_HOIST_MIN', 1024)):
                inv = [f for f in big if not any(f is v for v in var)]
                big = var
        if len(big) >= 3 and self._pairwise_pays(big, to_plates):
            # three or more arrays under a plate sum: pair by pair when a pair's result is much
            # smaller than its operands (sum_n r_nk y_nd mu_ke = (sum_n r_nk y_nd) mu_ke)
            t = self._plate_sum_contract(big, to_plates, from_plates)
        else:
            # two factors commute exactly, so either order may answer for both; three or more are
            # multiplied left to right and only the same order is the same number
            ids = tuple(id(f) for f in big)
            key = (tuple(sorted(ids)) if len(big) <= 2 else ids, tuple(to_plates),
                   tuple(from_plates))
            cache = self.__dict__.setdefault('_sum_cache', {})
            hit = cache.get(key)
            if hit is not None and sorted(id(r()) for r in hit[0]) == sorted(ids):
                # (a dead reference gives id(None): never among the ids of live factors)
                t = hit[1]
            else:
                t = misc.sum_multiply_to_plates(*big, to_plates=tuple(to_plates),
                                                from_plates=tuple(from_plates), ndim=0)
                for k in [k for k, v in cache.items() if any(r() is None for r in v[0])]:
                    del cache[k]
                cache[key] = ([weakref.ref(f) for f in big], t)
        for f in inv:
            t = fuse(lambda t_, s_: t_ * s_, t, f)
        if inv:
            s_ = t.shape
            while len(s_) > nt and s_[0] == 1:
                s_ = s_[1:]
            t = t.reshape(s_)
        for f in small:
            t = fuse(lambda t_, s_: t_ * s_, t, f.reshape(()))
        return t

    @staticmethod
    def _pairwise_pays(factors, to_plates):
        """Would contracting ``factors`` pair by pair (misc.plan_contraction) keep every
        intermediate result much smaller than the largest factor?  (An elementwise-like product
        -- three (N, K) arrays -- is one fused launch; a pair whose result is (N, K) again is not
        worth a temporary.)"""
        if os.environ.get('BAYESPY_AMD_PAIRWISE_SUMS', '1') == '0' or len(factors) > 6:
            return False
        full = tuple(broadcasted_shape(*[f.shape for f in factors]))
        n, nt = len(full), len(tuple(to_plates))
        to = ((1,) * (n - nt) + tuple(to_plates)) if n >= nt else tuple(to_plates)[nt - n:]
        sizes = {'p%d' % i: full[i] for i in range(n)}
        out_labels = ['p%d' % i for i in range(n) if full[i] != 1 and to[i] != 1]
        varying = [['p%d' % (n - f.ndim + ax) for ax in range(f.ndim) if f.shape[ax] != 1]
                   for f in factors]
        biggest = max(f.size for f in factors)
        if biggest < int(os.environ.get('BAYESPY_AMD_PAIRWISE_MIN', 1 << 16)):
            return False
        steps = misc.plan_contraction(varying, out_labels, sizes)
        if not steps:
            return False
        for _, _, res in steps:
            ext = 1
            for lab in res:
                ext *= sizes[lab]
            if ext * 8 > biggest:
                return False
        return True

    def _plate_sum_contract(self, factors, to_plates, from_plates):
        """_plate_sum of a product that contains contractions (LazyContract): one labelled
        contraction over the plate axes and the contractions' own keys, evaluated pair by pair.
        Same result shape and plate multiplier as misc.sum_multiply_to_plates."""
        import weakref
        from ...utils.shapes import broadcasting_multiplier
        small = [f for f in factors if f.size <= 1 and not isinstance(f, LazyContract)]
        if small:
            # plate-free factors multiply the result (and do not key it)
            t = self._plate_sum_contract([f for f in factors if not any(f is s_ for s_ in small)],
                                         to_plates, from_plates)
            for f in small:
                t = fuse(lambda t_, s_: t_ * s_, t, f.reshape(()))
            return t
        # a fixed order whatever the caller's (arrays before contractions, larger first): the same
        # product asked for by the message and by the bound is the same contraction
        factors = sorted(factors, key=lambda f: (isinstance(f, LazyContract), -f.size))
        ids = tuple(id(f) for f in factors)
        key = (ids, tuple(to_plates), tuple(from_plates), 'contract')
        cache = self.__dict__.setdefault('_sum_cache', {})
        hit = cache.get(key)
        if hit is not None and [id(r()) for r in hit[0]] == list(ids):
            return hit[1]
        full = tuple(broadcasted_shape(*[f.shape for f in factors]))
        n = len(full)
        r = broadcasting_multiplier(tuple(from_plates), full, tuple(to_plates))
        to = (1,) * (n - len(to_plates)) + tuple(to_plates) if n >= len(to_plates) \
            else tuple(to_plates)[len(to_plates) - n:]
        sizes = {'p%d' % i: full[i] for i in range(n)}
        out_labels = ['p%d' % i for i in range(n) if full[i] != 1 and to[i] != 1]
        ops, labs = [], []
        for q, f in enumerate(factors):
            if isinstance(f, LazyContract):
                off = n - len(f.out)
                ren = {lab: 'p%d' % (off + j) for j, lab in enumerate(f.out)}
                for a, ls in zip(f.ops, f.labs):
                    new = []
                    for lab in ls:
                        if lab not in ren:
                            ren[lab] = 'c%d_%s' % (q, lab)
                            sizes[ren[lab]] = f.sizes[lab]
                        new.append(ren[lab])
                    ops.append(a)
                    labs.append(new)
            else:
                ops.append(f)
                labs.append(['p%d' % (n - f.ndim + ax) for ax in range(f.ndim)])
        t = misc.contract_path(ops, labs, out_labels, sizes, scale=float(r))
        keep = tuple(full[i] if ('p%d' % i) in out_labels else 1 for i in range(n))
        while len(keep) > len(to_plates) and keep[0] == 1:
            keep = keep[1:]
        t = t.reshape(keep)
        for k in [k for k, v in cache.items() if any(r_() is None for r_ in v[0])]:
            del cache[k]
        cache[key] = ([weakref.ref(f) for f in factors], t)
        return t

    def _finish_bound(self, node, terms, ignore_masked, presummed=None):
        """sum over the node's plates of sum_k coef_k prod(factors_k), masked, completed over the
        ranks for a sharded node, with the plate multiplier (expfamily.py:470-480)."""
        mask, any_active = self._mask_factor((id(node), 'self'), lambda: self._mask_array(node))
        if not ignore_masked:
            mask, any_active = None, True
        sharded = self._is_sharded(node)
        if not any_active and not sharded:
            return None, 0.0
        parts = []
        for coef, factors in terms:
            factors = list(factors) if factors else [_ones(())]
            if mask is not None:
                factors.append(mask)
            parts.append((float(coef), self._plate_sum(factors, (), node.plates).reshape(())))
        if presummed is not None:
            # a part of the term that is a sum over this rank's plates already
            parts.append((1.0, _arr(presummed).reshape(())))
        tot = _wsum(parts)
        if sharded:
            tot = fuse(lambda x: x + 0.0, tot) if any_active else DArray.zeros(())
            self.rt.all_reduce_sum_(tot.t)
        return tot, float(np.prod(node.plates_multiplier))

    @_operation
    def lower_bound_contribution(self, node, ignore_masked=True):
        tot, factor = self._lower_bound_device(node, ignore_masked)
        return factor if tot is None else tot.item() * factor

    @_operation
    def lower_bound_contributions(self, nodes):
        """Terms of several nodes with ONE device->host read (VB.loglikelihood_lowerbound)."""
        stash, self._g_stash = self._g_stash, None
        if stash is not None and stash[0] == tuple(id(n) for n in nodes):
            return list(stash[1])          # evaluated inside the recorded sweep (graph_iter.py)
        parts = [self._lower_bound_device(n) for n in nodes]
        self.__dict__.get('_contract_memo', {}).clear()       # a sweep ends here
        dev = [t.t.reshape(1) for t, _ in parts if t is not None]
        self.rt.host_access('lower_bound_contributions')      # (flushes the queue of small operations)
        vals = iter(self.rt.torch.cat(dev).cpu().numpy() if dev else ())
        return [f if t is None else float(next(vals)) * f for t, f in parts]

    @_operation
    def get_moments(self, node):
        if isinstance(node, Stochastic):
            st = self._ensure(node)
            if st.observed and st.partial and st.stale:
                self._refresh_partial(node, st)
        return [np.asarray(_arr(m).numpy()) for m in self._moments(node)]

    # -- persistence (stochastic.py:305-355, expfamily.py:507-535) ------------------------------
    def save_state(self, put, nodes, index):
        for node in nodes:
            if not isinstance(node, Stochastic):
                continue
            st = self._ensure(node)
            base = 'nodes/%s/' % node.name
            for i, ui in enumerate(st.u):
                put(base + 'u%d' % i, _arr(ui).numpy())
            if st.phi is not None:
                for i, pi in enumerate(st.phi):
                    put(base + 'phi%d' % i, _arr(pi).numpy())
            put(base + 'f', 0.0 if st.f is None else (_arr(st.f).numpy() if isinstance(st.f, DArray)
                                                     else np.asarray(st.f, dtype=np.float64)))
            put(base + 'g', np.inf if st.g is None else (
                _arr(st.g).numpy() if isinstance(st.g, DArray) else np.asarray(st.g, dtype=np.float64)))
            put(base + 'observed', bool(st.observed))

    def load_state(self, reader, nodes, index):
        for node in nodes:
            if not isinstance(node, Stochastic):
                continue
            base = 'nodes/%s/' % node.name
            if not reader.has(base + 'u0'):
                raise Exception("File does not contain variable %s" % node.name)
            st = self._ensure(node)
            if bool(reader.get(base + 'observed')) != bool(st.observed):
                raise ValueError('node %s: the file and the model disagree on whether it is '
                                 'observed' % node.name)
            st.u = [DArray.from_host(np.array(reader.get(base + 'u%d' % i), dtype=np.float64))
                    for i in range(len(st.u))]
            if not st.observed:
                st.phi = [DArray.from_host(np.array(reader.get(base + 'phi%d' % i),
                                                    dtype=np.float64))
                          for i in range(len(node.dims))]
                g = np.array(reader.get(base + 'g'), dtype=np.float64)
                st.g = float(g) if (g.ndim == 0 and not np.isfinite(g)) else DArray.from_host(g)

    # -- rotations (inference/transformations.py) ------------------------------------------------
    def gamma_posterior_shape(self, node):
        st = self._ensure(node)
        return np.asarray(_arr(st.phi[1]).numpy())

    @_operation
    def rotation_rows(self, node):
        """Per-plate means (N, K) and covariances (N, K, K) of a vector GaussianARD with ONE plate
        axis (the dynamics matrix of a state-space model, whose rows rotate with the state space)."""
        if not isinstance(node, GaussianARD) or node.ndim != 1 or len(node.plates) != 1:
            raise NotImplementedError('row statistics of %s' % node.name)
        st = self._ensure(node)
        N, K = node.plates[0], node.dims[0][-1]
        m = np.broadcast_to(np.asarray(_arr(st.u[0]).numpy()), (N, K)).copy()
        mm = np.broadcast_to(np.asarray(_arr(st.u[1]).numpy()), (N, K, K))
        return dict(mean=m, cov=mm - m[:, :, None] * m[:, None, :])

    def _chain_sums(self, node):
        """K x K sums over sequences and time of the chain's moments (transformations.py:1239-1273)."""
        st = self._ensure(node)
        T, D = node.N, node.D
        nseq = int(np.prod(node.plates)) if node.plates else 1
        u0 = _arr(st.u[0]).broadcast_to(node.plates + (T, D))
        u1 = _arr(st.u[1]).broadcast_to(node.plates + (T, D, D))
        u2 = _arr(st.u[2]).broadcast_to(node.plates + (max(T - 1, 0), D, D))
        npl = len(node.plates)
        lead = tuple(range(npl + 1))            # sequence plates and time

        def total(a, axes):
            return misc.sum_multiply(a, axis=axes) if axes else a

        pick = (slice(None),) * npl
        out = dict(nvec=float(T * nseq),
                   X0=total(DArray(u0.t[pick + (0,)]), tuple(range(npl))),
                   X0X0=total(DArray(u1.t[pick + (0,)]), tuple(range(npl))),
                   XnXn=total(DArray(u1.t[pick + (slice(1, None),)]), lead),
                   XpXp=total(DArray(u1.t[pick + (slice(0, T - 1),)]), lead),
                   XpXn=total(u2, lead))
        if self._is_sharded(node):
            for k in ('X0', 'X0X0', 'XnXn', 'XpXp', 'XpXn'):
                v = fuse(lambda x: x + 0.0, _arr(out[k]))
                self.rt.all_reduce_sum_(v.t)
                out[k] = v
            out['nvec'] = float(T * self.rt.all_reduce_int(nseq))
        return {k: (np.asarray(_arr(v).numpy()) if k != 'nvec' else v) for k, v in out.items()}

    def rotation_statistics(self, node):
        """sum over the plates of <x x^T> (K x K) and the plate count of a vector GaussianARD; for a
        GaussianMarkovChain the sums of its moments over sequences and time."""
        if isinstance(node, GaussianMarkovChain) and type(node) is GaussianMarkovChain:
            return self._chain_sums(node)
        if not isinstance(node, GaussianARD) or node.ndim != 1:
            raise NotImplementedError('rotation of %s' % node.name)
        st = self._ensure(node)
        K = node.dims[0][-1]
        if isinstance(st.u[1], FactoredMoment):
            # (plates sharing the covariance) x Cov + sum over the plates of <x><x>^T
            fm = st.u[1]
            x = fm.mean
            xx = fuse(lambda a, b: a + b,
                      misc.sum_multiply_to_plates(fm.cov, to_plates=(K, K),
                                                  from_plates=node.plates + (K, K), ndim=0),
                      misc.sum_multiply_to_plates(x.reshape(x.shape + (1,)),
                                                  x.reshape(x.shape[:-1] + (1, K)), to_plates=(K, K),
                                                  from_plates=node.plates + (K, K), ndim=0))
        else:
            xx = misc.sum_multiply_to_plates(_arr(st.u[1]), to_plates=(K, K),
                                             from_plates=node.plates + (K, K), ndim=0)
        nplates = float(np.prod(node.plates))
        if self._is_sharded(node):
            xx = fuse(lambda x: x + 0.0, xx)
            self.rt.all_reduce_sum_(xx.t)
            nplates = float(self.rt.all_reduce_int(int(nplates)))
        return dict(XX=np.asarray(xx.numpy()), nplates=nplates)

    @_operation
    def rotate_node(self, node, R, invR, logdetR, Q=None):
        """q(node) <- distribution of R x: phi0 <- R^-T phi0, phi1 <- R^-T phi1 R^-1,
        u0 <- R u0, u1 <- R u1 R^T, g <- g - log|det R|  (gaussian.py:1693-1741); the same with the
        cross moments and T log|det R| for a GaussianMarkovChain (gaussian_markov_chain.py:51-65,
        :167-185).  ``Q``: additionally the (approximate) rotation of the single plate axis of a
        GaussianARD: means exactly, precisions scaled by the inverse squared column sums of Q
        (gaussian.py:1743-1772)."""
        chain = isinstance(node, GaussianMarkovChain) and type(node) is GaussianMarkovChain
        if not chain and (not isinstance(node, GaussianARD) or node.ndim != 1):
            raise NotImplementedError('rotation of %s' % node.name)
        st = self._ensure(node)
        if st.observed:
            raise ValueError('cannot rotate the observed node %s' % node.name)
        Rd = DArray.from_host(np.ascontiguousarray(R))
        Rt = DArray.from_host(np.ascontiguousarray(R.T))
        Ri = DArray.from_host(np.ascontiguousarray(invR))
        Rit = DArray.from_host(np.ascontiguousarray(invR.T))

        def rot2(L_, a, R_):
            return linalg.mmdot(linalg.mmdot(L_, _arr(a)), R_)
        fm = None if chain else st.u[1]
        if Q is None and isinstance(fm, FactoredMoment) and fm.sums is not None \
                and fm.logdet_prec is not None and st.phi is not None \
                and isinstance(st.phi[0], DerivedArray) and isinstance(st.g, DerivedArray):
            # the state of the fused update keeps its form: means and covariance rotate, the plate
            # sums with them (sum <x> -> R sum <x>, sum <x><x>^T -> R . R^T, sum y <x>^T -> . R^T),
            # log|Lambda| -> log|Lambda| - 2 log|det R|; phi0 and g stay functions of those
            u0 = linalg.mvdot(Rd, _arr(st.u[0]))
            phi1 = rot2(Rit, st.phi[1], Ri)
            sm = fm.sums
            sums = PlateSums(linalg.mvdot(Rd, sm.x), rot2(Rd, sm.xx, Rt),
                             None if sm.yx is None else linalg.mmdot(sm.yx, Rt), sm.ydesc, sm.ykeep,
                             sm.n)
            ld = fuse(lambda l: l - 2.0 * float(logdetR), _arr(fm.logdet_prec))
            st.u = [u0, FactoredMoment(rot2(Rd, fm.cov, Rt), u0, node.ndim, logdet_prec=ld,
                                       sums=sums)]
            st.phi = [DerivedArray('gauss_phi0', (phi1, u0), st.phi[0].shape), phi1]
            st.g = DerivedArray('gauss_g', (phi1, u0, ld), st.g.shape)
            self._seed_sums()
            return
        if st.phi is not None:
            phi = [linalg.mvdot(Rit, _arr(st.phi[0]))] + [rot2(Rit, p, Ri) for p in st.phi[1:]]
        else:
            phi = None                       # delta moments (initialize_from_value): no parameters
        st.phi = phi
        if not chain and isinstance(st.u[1], FactoredMoment):
            # the factors rotate separately: Cov <- R Cov R^T, <x> <- R <x>
            u0 = linalg.mvdot(Rd, _arr(st.u[0]))
            st.u = [u0, FactoredMoment(rot2(Rd, st.u[1].cov, Rt), u0, node.ndim)]
        else:
            st.u = [linalg.mvdot(Rd, _arr(st.u[0]))] + [rot2(Rd, u, Rt) for u in st.u[1:]]
        scale = float(node.N) if chain else 1.0
        if isinstance(st.g, DArray):
            st.g = fuse(lambda g: g - scale * float(logdetR), st.g)
        if Q is None:
            return
        if chain or len(node.plates) != 1:
            raise NotImplementedError('plate rotation of %s' % node.name)
        if st.phi is None:
            raise ValueError('%s holds delta moments: its plates cannot be rotated' % node.name)
        sQ = Q.sum(axis=0)
        Qd = DArray.from_host(np.ascontiguousarray(Q))
        u0 = linalg.mmdot(Qd, _arr(st.u[0]))                         # rows mixed: (N, K)
        inv2 = DArray.from_host((1.0 / (sQ * sQ)).reshape(-1, 1, 1))
        phi1 = fuse(lambda p, w: p * w, _arr(st.phi[1]).broadcast_to(node.plates + node.dims[1]),
                    inv2)
        phi0 = fuse(lambda v: -2.0 * v, linalg.mvdot(phi1, u0))
        st.phi = [phi0, phi1]
        st.u, st.g = self.family[id(node)].moments_and_cgf(st.phi)

    # -- natural parameters, gradients, densities (expfamily.py:258-340, :483-542) --------------
    def _latent_state(self, node):
        st = self._ensure(node)
        if st.observed or st.phi is None:
            raise ValueError('node %s is observed: it has no variational parameters' % node.name)
        return st

    def natural_parameters(self, node):
        """phi of q(node) as device arrays (shared with the plan: do not modify)."""
        return [_arr(p) for p in self._latent_state(node).phi]

    def get_parameters(self, node):
        return [np.array(p.numpy()) for p in self.natural_parameters(node)]

    @_operation
    def set_parameters(self, node, x):
        st = self._latent_state(node)
        if len(x) != len(st.phi):
            raise ValueError('%s has %d natural parameters, %d were given'
                             % (node.name, len(st.phi), len(x)))
        phi = []
        for i, xi in enumerate(x):
            xi = xi if isinstance(xi, DArray) else _arr(np.array(xi, dtype=np.float64))
            if not is_shape_subset(xi.shape, node.plates + node.dims[i]):
                raise ValueError('parameter %d of shape %s does not broadcast to %s'
                                 % (i, xi.shape, node.plates + node.dims[i]))
            phi.append(xi)
        u, g = self.family[id(node)].moments_and_cgf(phi)
        self.rt.check_deferred()         # an invalid phi leaves the node as it was
        st.phi, st.u, st.g = phi, u, g

    def log_normalizer(self, node):
        """(g, f) of the node as host values (nan where not defined, expfamily.py:125-126)."""
        st = self._ensure(node)

        def host(v):
            if v is None:
                return np.array(np.nan)
            return np.array(v.numpy()) if isinstance(v, DArray) else np.array(float(v))
        return host(st.g), host(st.f)

    @_operation
    def riemannian_gradient(self, node):
        """annealing * (phi_prior + sum of messages) - phi, with the full shape of the
        parameters (expfamily.py:258-278)."""
        st = self._latent_state(node)
        opt = self._optimal_phi(node)
        out = []
        for i, (q, p) in enumerate(zip(opt, st.phi)):
            d = fuse(lambda a, b: a - b, _arr(q), _arr(p))
            full = node.plates + node.dims[i]
            if d.shape != full:
                d = fuse(lambda a, o: a * o, d, _ones(full))
            out.append(d)
        return out

    @_operation
    def gradient(self, node, rg):
        """Euclidean gradient with respect to phi from the Riemannian one
        (expfamily.py:281-294)."""
        st = self._latent_state(node)
        rg = [r if isinstance(r, DArray) else _arr(np.asarray(r, dtype=np.float64)) for r in rg]
        g = self.family[id(node)].gradient(rg, st.u, st.phi)
        a = float(getattr(node, 'annealing', 1.0))
        if a != 1.0:
            g = [fuse(lambda v, a_=a: v / a_, gi) for gi in g]
        return g

    @_operation
    def logpdf(self, node, X):
        """log q(X) = g + f(X) + sum_i phi_i . u_i(X)   (expfamily.py:483-498)."""
        st = self._latent_state(node)
        u, f = self.family[id(node)].fixed_moments_and_f(X)
        z = fuse(lambda g, f_: g + f_, _arr(st.g), f if isinstance(f, DArray) else float(f))
        for i, nd in enumerate(len(d) for d in node.dims):
            t = fuse(lambda p, v: p * v, _arr(st.phi[i]), _arr(u[i]))
            z = fuse(lambda a, b: a + b, z, _sum_last(t, nd))
        return np.array(z.numpy())

    def random(self, node):
        """A draw from q(node) on the host (set-up / inspection; RNG streams are not part of
        the parity contract)."""
        st = self._latent_state(node)
        return self._sample(node, self.family[id(node)], st.u)
