"""
Extension point of the generic engine: node types defined OUTSIDE the package.

The reference's boundary for new nodes is a Python class protocol (SURVEY.md 8b;
``doc/source/dev_guide/writingnodes.rst``, "Distributions" / "Stochastic exponential family
nodes"; code ``stochastic.py:16-80``, ``expfamily.py:17-70``): a ``Distribution`` class with

    compute_message_to_parent(parent, index, u_self, *u_parents) -> list   (one entry per moment of the parent)
    compute_phi_from_parents(*u_parents, mask=True)              -> list   (one entry per moment of the node)
    compute_moments_and_cgf(phi, mask=True)                      -> (list u, g)
    compute_cgf_from_parents(*u_parents)                         -> g
    compute_fixed_moments_and_f(x, mask=True)                    -> (list u, f)

plus a node class that fixes plates and dims.  ``register_family(NodeClass, DistributionClass)``
makes such a pair runnable here: the node class derives from ``bayespy_amd.nodes.Stochastic``
(constructor: parents, plates, dims), the distribution class carries the five formulas under the
reference's names, written with ordinary arithmetic and the functions of ``bayespy_amd.darray``
(``log``, ``exp``, ``digamma``, ``gammaln``, ``sqrt``, ``square``, ``maximum``, ``where_nonzero``, ...).
The adapter below calls each formula with symbolic operands and compiles every entry it returns
into ONE fused elementwise kernel (``darray.fuse`` -> ``vmp_ewise``), so a user node costs what a
built-in scalar-valued node costs.  Formulas that need more than elementwise arithmetic (sums over
variable axes, linear algebra) take the arrays themselves: set ``elementwise = False`` on the
distribution class and write them with ``darray.fuse`` / ``utils.misc.sum_multiply`` /
``utils.linalg`` directly.

Moments of NUMERIC parents (the reference takes them from the parent's Moments class,
``node.py:266-300``): ``compute_fixed_parent_moments(index, x) -> list``; default ``[x]``.

A class written against the engine's own family interface (``plans.generic.Family``:
``phi_from_parents(up)``, ``moments_and_cgf(phi)``, ...) registers the same way and is used as it is.
"""
import numbers

import numpy as np

from ... import darray as da
from ...darray import DArray, Expr, fuse

_REGISTRY = []          # (node class, factory(node) -> family), most recent first


def register_family(node_cls, family_cls):
    """Run nodes of type ``node_cls`` on the generic engine with the formulas of ``family_cls``
    (see the module docstring).  Later registrations win; returns ``family_cls`` so that it can be
    used as a class decorator factory: ``register_family(MyNode, MyDistribution)``."""
    if not isinstance(node_cls, type):
        raise TypeError('register_family(node_class, family_class)')
    from .generic import Family
    if isinstance(family_cls, type) and issubclass(family_cls, Family):
        factory = family_cls
    elif all(hasattr(family_cls, m) for m in DistributionAdapter.REQUIRED):
        def factory(node, cls=family_cls):
            return DistributionAdapter(node, cls)
    else:
        missing = [m for m in DistributionAdapter.REQUIRED if not hasattr(family_cls, m)]
        raise TypeError('%s is neither a plans.generic.Family nor a Distribution of the '
                        'reference\'s contract (missing: %s)'
                        % (getattr(family_cls, '__name__', family_cls), ', '.join(missing)))
    _REGISTRY.insert(0, (node_cls, factory))
    return family_cls


def unregister_family(node_cls):
    _REGISTRY[:] = [(c, f) for c, f in _REGISTRY if c is not node_cls]


def registered_family(node):
    for cls, factory in _REGISTRY:
        if isinstance(node, cls):
            return factory(node)
    return None


def _is_array(x):
    return isinstance(x, DArray)


class _Traced:
    """Call ``fn(*args)`` -- nested lists of device arrays and numbers -- with the arrays replaced by
    symbolic operands and compile output entry ``pick(result)`` into one fused kernel."""

    def __init__(self, fn, args):
        self.fn, self.args = fn, args
        self.arrays = []
        self._collect(args)

    def _collect(self, a):
        if _is_array(a):
            if not any(a is b for b in self.arrays):
                self.arrays.append(a)
        elif isinstance(a, (list, tuple)):
            for x in a:
                self._collect(x)

    def _subst(self, a, leaves):
        if _is_array(a):
            return leaves[[i for i, b in enumerate(self.arrays) if b is a][0]]
        if isinstance(a, (list, tuple)):
            return [self._subst(x, leaves) for x in a]
        return a

    def structure(self):
        """The result of the formula on symbolic operands (to learn its structure)."""
        leaves = [Expr('in', val=i) for i in range(len(self.arrays))]
        return self.fn(*self._subst(self.args, leaves))

    def entry(self, pick):
        probe = pick(self.structure())
        if probe is None:
            return None
        if isinstance(probe, numbers.Number) or (isinstance(probe, np.ndarray) and probe.ndim == 0):
            return float(probe)                    # a constant entry (-1, 0.5, ...: legal)
        if not self.arrays:
            raise TypeError('a formula without array operands returned a non-constant')
        return fuse(lambda *leaves: pick(self.fn(*self._subst(self.args, list(leaves)))),
                    *self.arrays)


class DistributionAdapter:
    """A ``plans.generic.Family`` around a Distribution class of the reference's contract."""
    REQUIRED = ('compute_message_to_parent', 'compute_phi_from_parents', 'compute_moments_and_cgf',
                'compute_cgf_from_parents', 'compute_fixed_moments_and_f')

    def __init__(self, node, dist_cls):
        self.node = node
        try:
            self.dist = dist_cls(node)
        except TypeError:
            self.dist = dist_cls()
        self.elementwise = bool(getattr(self.dist, 'elementwise', True))
        for name in ('finite_phi', 'missing_fill', 'message_independent_of_target'):
            if hasattr(self.dist, name):
                setattr(self, name, getattr(self.dist, name))

    # -- structure ---------------------------------------------------------------------------------
    def plates_to_parent(self, index):
        f = getattr(self.dist, 'plates_to_parent', None)
        return tuple(f(index, self.node.plates)) if f else self.node.plates

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        from .generic import _arr
        f = getattr(self.dist, 'compute_fixed_parent_moments', None)
        v = _arr(value)
        if f is None:
            return [v]
        return self._run(lambda x: f(index, x), [v], many=True)

    # -- the five formulas ---------------------------------------------------------------------------
    def _run(self, fn, args, many, with_scalar=False):
        """Evaluate ``fn(*args)``: a list of entries (``many``), optionally paired with a scalar
        field ``(list, g)``; every entry one fused kernel when the class is elementwise."""
        from .generic import _arr
        if not self.elementwise:
            out = fn(*args)
            return out
        tr = _Traced(fn, args)
        probe = tr.structure()
        if with_scalar:
            n = len(probe[0])
            u = [tr.entry(lambda r, i=i: r[0][i]) for i in range(n)]
            g = tr.entry(lambda r: r[1])
            return [(_arr(x) if isinstance(x, float) else x) for x in u], g
        if many:
            return [tr.entry(lambda r, i=i: r[i]) for i in range(len(probe))]
        return tr.entry(lambda r: r)

    def phi_from_parents(self, up):
        from .generic import _arr
        ups = [list(u) for u in up]
        out = self._run(lambda *u: self.dist.compute_phi_from_parents(*u), ups, many=True)
        return [(_arr(x) if isinstance(x, float) else x) for x in out]

    def moments_and_cgf(self, phi):
        from .generic import _arr
        u, g = self._run(lambda p: self.dist.compute_moments_and_cgf(p), [list(phi)], many=True,
                         with_scalar=True)
        return u, (_arr(g) if isinstance(g, float) else g)

    def cgf_from_parents(self, up):
        ups = [list(u) for u in up]
        return self._run(lambda *u: self.dist.compute_cgf_from_parents(*u), ups, many=False)

    def fixed_moments_and_f(self, x):
        from .generic import _arr
        x = _arr(np.asarray(x, dtype=np.float64)) if not isinstance(x, DArray) \
            and not hasattr(x, 'is_cuda') else _arr(x)
        check = getattr(self.dist, 'check_value', None)
        if check is not None:
            check(x)
        return self._run(lambda v: self.dist.compute_fixed_moments_and_f(v), [x], many=True,
                         with_scalar=True)

    def message_to_parent(self, index, u, up):
        parent = self.node.parents[index]
        # the moments of the parent the message goes to are not read by a conjugate message
        # (and may not have been evaluated): handed over only to a class that asks for them
        # (``message_reads_target = True``)
        reads = bool(getattr(self.dist, 'message_reads_target', False))
        rows = [list(up[j]) if (j != index or reads) else []
                for j in range(len(self.node.parents))]
        fn = lambda us, *rest: self.dist.compute_message_to_parent(parent, index, us, *rest)
        return self._run(fn, [list(u)] + rows, many=True)

    def sample(self, st):
        f = getattr(self.dist, 'random', None)
        if f is None:
            raise NotImplementedError('random draws for %s' % type(self.node).__name__)
        from .generic import _arr
        return f(*[np.asarray(_arr(p).numpy()) for p in st.phi], plates=self.node.plates)


__all__ = ['register_family', 'unregister_family', 'registered_family', 'DistributionAdapter', 'da']
