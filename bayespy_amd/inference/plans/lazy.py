"""
Lazily evaluated device arrays and small helpers of the generic engine.

What the families (``plans/families``) and the router (``plans/generic.py``) pass around besides plain
``DArray``s: second moments kept as factors (``FactoredMoment``, with the plate sums the fused update
made: ``PlateSums``), arrays that are functions of other state arrays (``DerivedArray``), plates-sized
arrays known as sums of products (``LazySum``) or as contractions (``LazyContract``), message entries
that are sums of products (``Terms``); constants uploaded once; weighted sums as one launch.
"""
import os

import numpy as np

from ... import darray as da
from ...darray import DArray, fuse, contiguous
from ...utils import misc, linalg
from ...utils.shapes import broadcasted_shape

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
