"""
Node-graph core: graph edges, plates, names, the hand-off to compiled plans.

Host-side mirror of the reference's node protocol (``Node`` node.py:223,
``Stochastic`` stochastic.py:83, ``Constant`` constant.py:13): the same
constructor arguments (``plates=``, ``name=``), the same user-visible methods
(``observe``, ``update``, ``get_moments``, ``lower_bound_contribution``,
``initialize_from_value/random/prior``) and attributes (``plates``, ``dims``,
``u``, ``observed``).  Unlike the reference, a node holds NO arrays on the
host: all moments live in HBM inside the *plan* that ``VB`` compiled for the
model block the node belongs to (``bayespy_amd/inference/plans``), and every
method that touches numbers forwards to that plan, which launches HIP kernels
through the C ABI.
"""
import numpy as np

from ..utils.shapes import broadcasted_shape


class Node:
    """Base class of all nodes (reference: node.py:223-301)."""

    _counter = 0

    def __init__(self, *parents, plates=(), dims=(), name=None):
        Node._counter += 1
        self._uid = Node._counter
        self.parents = [ensure_node(p) for p in parents]
        self.children = []
        self.plates = tuple(int(p) for p in plates)
        self.dims = tuple(tuple(d) for d in dims)
        self.name = name if name else '%s_%d' % (type(self).__name__, self._uid)
        self._plan = None
        self._shard_axis = None
        self._plates_multiplier_arg = None
        for i, p in enumerate(self.parents):
            p.children.append((self, i))

    # -- plate multipliers (node.py:294-301, :403-420) -----------------------------------------
    @property
    def plates_multiplier(self):
        """Per-plate-axis factors by which this node's plates stand for more replications than
        they hold (mini-batches of stochastic variational inference): messages to parents
        that lack the multiplier and the lower-bound term are scaled by it (node.py:589-632,
        expfamily.py:470-480).  Default: inherited from the parents."""
        from ..utils.shapes import multiplier_shape
        parents = [p.plates_multiplier for p in self.parents]
        own = self._plates_multiplier_arg
        return multiplier_shape(own, *parents)

    @plates_multiplier.setter
    def plates_multiplier(self, value):
        self._plates_multiplier_arg = None if value is None else tuple(value)
        if self._plan is not None and hasattr(self._plan, 'invalidate'):
            self._plan.invalidate(self)

    # -- multi-GPU: sharded plates ---------------------------------------
    def shard(self, axis=-1):
        """Declare that plate axis ``axis`` of this node is partitioned over the ranks of
        ``torch.distributed`` -- this process holds only its contiguous part, ``plates`` are the
        LOCAL sizes.  Every sum over that axis in a message to a replicated (non-sharded)
        parent, and the node's lower-bound term, is then completed with an all-reduce
        (the reference's plate sums node.py:650 / dot.py:581 /
        expfamily.py:470-480 become local partial sum + RCCL all-reduce).  Mark the top-most
        nodes that carry the sharded plate (e.g. X of a PCA model); descendants (F, Y) inherit
        the partition.  Returns ``self``."""
        if not isinstance(axis, int):
            raise ValueError('Plate axis must be integer')
        if axis >= 0:
            axis -= len(self.plates)
        if axis < -len(self.plates) or axis >= 0:
            raise ValueError('Plate axis out of bounds')
        self._shard_axis = axis
        return self

    # -- plan hand-off ----------------------------------------------------------
    def _require_plan(self):
        if self._plan is None:
            # A model becomes executable when VB(...) compiles it.  Build a
            # throw-away engine so that stand-alone node calls work like in the
            # reference (e.g. ``X.update()`` in a script without VB).
            from ..inference.vb import compile_for_node
            compile_for_node(self)
        return self._plan

    def get_moments(self):
        """Host copies of the node's moments in the reference's shapes
        (stochastic.py:172-175, deterministic.py:62-64)."""
        return self._require_plan().get_moments(self)

    @property
    def u(self):
        return self.get_moments()

    def lower_bound_contribution(self, **kwargs):
        return 0.0

    def get_shape(self, i):
        return self.plates + self.dims[i]

    def __getitem__(self, index):
        """Basic indexing of the plates (node.py:761-763)."""
        from .take import Slice
        return Slice(self, index, name=self.name + '.__getitem__')

    def __repr__(self):
        return '<%s %r plates=%s>' % (type(self).__name__, self.name, self.plates)


class Constant(Node):
    """Fixed numeric parent (reference: constant.py:13-86)."""

    def __init__(self, value, name=None):
        self.value = np.asarray(value, dtype=np.float64)
        super().__init__(plates=self.value.shape, dims=((),), name=name)

    def is_scalar(self):
        return self.value.size == 1

    def scalar(self):
        return float(self.value.reshape(-1)[0])

    def get_moments(self):
        return [self.value]


class GaussianConstant(Constant):
    """A numeric array standing in for a Gaussian-moment parent: the reference wraps it in a
    constant node with DeltaMoments of GaussianMoments (node.py:110-179, gaussian.py:35-84):
    moments [x, x x^T] over the last ``ndim`` axes, the leading axes are plates."""

    def __init__(self, value, ndim, name=None):
        super().__init__(value, name=name)
        shp = self.value.shape
        if ndim > len(shp):
            raise ValueError('Array of shape %s has fewer than %d variable axes' % (shp, ndim))
        tail = tuple(shp[len(shp) - ndim:]) if ndim else ()
        self.plates = tuple(shp[:len(shp) - ndim])
        self.dims = (tail, tail + tail)
        self.ndim = ndim


def ensure_node(x):
    """Numeric arguments become Constant nodes (node.py:360-376)."""
    if isinstance(x, Node):
        return x
    return Constant(x)


class DeviceMask:
    """An observation mask that already lives in HBM (a boolean tensor): kept in place for the
    fused blocks; converts to a host array on demand (``np.asarray``) for everything else."""

    def __init__(self, tensor):
        self.tensor = tensor.to(dtype=tensor.new_empty(0, dtype=bool).dtype)
        self.shape = tuple(tensor.shape)
        self.all_true = bool(self.tensor.all().item())

    def __array__(self, dtype=None, copy=None):
        a = self.tensor.cpu().numpy()
        return a if dtype is None else a.astype(dtype)


class Stochastic(Node):
    """Base of the exponential-family nodes (stochastic.py:83-376,
    expfamily.py:94-542)."""

    def __init__(self, *parents, plates=(), dims=(), name=None):
        super().__init__(*parents, plates=plates, dims=dims, name=name)
        self.observed = False
        self._data = None          # pending observation (host ndarray / device tensor)
        self._mask = True
        self._init = None          # pending initialisation: ('value', x) | ('random',) | None
        self.annealing = 1.0       # deterministic annealing coefficient (expfamily.py:123)

    # -- data / initialisation (expfamily.py:168-212, :369-398) ------------------
    def observe(self, x, mask=True):
        """Fix the node to data.  ``x`` may be a host ndarray or a fp64 tensor
        already resident in HBM (then it is used in place)."""
        self._check_value_shape(x)
        if hasattr(mask, 'is_cuda') and mask.is_cuda:
            mask = DeviceMask(mask)
        if mask is True or (not isinstance(mask, DeviceMask) and np.ndim(mask) == 0
                            and bool(mask)):
            mask = True
        else:
            if not isinstance(mask, DeviceMask):
                mask = np.asarray(mask, dtype=bool)
            try:
                ok = broadcasted_shape(mask.shape, self.plates) == self.plates
            except ValueError:
                ok = False
            if not ok:
                raise ValueError('Mask of shape %s does not broadcast to plates %s'
                                 % (mask.shape, self.plates))
        self._data = x
        self._mask = mask
        self._fully_observed = (mask is True or (mask.all_true if isinstance(mask, DeviceMask)
                                                 else bool(np.all(mask))))
        self.observed = True
        if self._plan is not None:
            self._plan.invalidate(self)

    def initialize_from_value(self, x):
        self._check_value_shape(x)
        self._init = ('value', x)
        if self._plan is not None:
            self._plan.invalidate(self)

    def initialize_from_random(self):
        self._init = ('random',)
        if self._plan is not None:
            self._plan.invalidate(self)

    def initialize_from_prior(self):
        self._init = None
        if self._plan is not None:
            self._plan.invalidate(self)

    def initialize_from_parameters(self, *args):
        """q := the node's own distribution with the given values in place of the parents
        (expfamily.py:187-190), e.g. ``GaussianARD.initialize_from_parameters(mu, alpha)``."""
        self._init = ('parameters', args)
        if self._plan is not None:
            self._plan.invalidate(self)

    def unobserve(self):
        """stochastic.py:284-287."""
        self.observed = False
        self._data = None
        self._mask = True
        if self._plan is not None:
            self._plan.invalidate(self)

    # -- state of q read by users and tests -- host copies ------------------------
    def _plan_call(self, method, *args, **kwargs):
        fn = getattr(self._require_plan(), method, None)
        if fn is None:
            raise NotImplementedError(
                "%s of node %s: the fused %s plan does not expose it; build the engine with "
                "VB(..., engine='generic')" % (method, self.name,
                                               type(self._plan).__name__))
        return fn(self, *args, **kwargs)

    @property
    def phi(self):
        """Natural parameters of q (expfamily.py:215-257)."""
        return self._plan_call('get_parameters')

    @property
    def g(self):
        """Log-normaliser term of q; ``inf`` after initialize_from_value (expfamily.py:125)."""
        return self._plan_call('log_normalizer')[0]

    @property
    def f(self):
        """Fixed term f(x) of an observed node (expfamily.py:126)."""
        return self._plan_call('log_normalizer')[1]

    @property
    def mask(self):
        """Observation mask after propagation from the children (node.py:457-526)."""
        return self._plan_call('get_mask')

    def get_parameters(self):
        return self._plan_call('get_parameters')

    def set_parameters(self, x):
        self._plan_call('set_parameters', x)

    def get_riemannian_gradient(self):
        """annealing * (phi_prior + messages) - phi  (expfamily.py:258-278)."""
        return [np.array(d.numpy()) for d in self._plan_call('riemannian_gradient')]

    def get_gradient(self, rg):
        """Euclidean gradient with respect to phi given the Riemannian gradient
        (expfamily.py:281-294)."""
        return [np.array(d.numpy()) for d in self._plan_call('gradient', rg)]

    def logpdf(self, X, mask=True):
        if mask is not True:
            raise NotImplementedError('Mask not yet implemented')
        return self._plan_call('logpdf', X)

    def pdf(self, X, mask=True):
        return np.exp(self.logpdf(X, mask=mask))

    def random(self):
        return self._plan_call('random')

    def _check_value_shape(self, x):
        shape = tuple(x.shape) if hasattr(x, 'shape') else np.shape(x)
        full = self.plates + self.dims[0]
        try:
            out = broadcasted_shape(shape, full)
        except ValueError:
            out = None
        if out != full:
            raise ValueError('Value of shape %s does not match node %s with plates+dims %s'
                             % (shape, self.name, full))

    # -- inference ------------------------------------------------------------------
    def update(self):
        """Recompute q(node) from the current moments of its Markov blanket
        (stochastic.py:276-282).  Fully observed nodes are skipped like in the reference; the
        plates of a partially observed node that carry no data are updated (``if not
        np.all(self.observed)``)."""
        if self.observed and getattr(self, '_fully_observed', True):
            return
        self._require_plan().update(self)

    def lower_bound_contribution(self, gradient=False, ignore_masked=True):
        """E_q[log p(node | parents) - log q(node)] (expfamily.py:400-480)."""
        if gradient:
            raise NotImplementedError('gradient of the lower bound term')
        if ignore_masked:
            return self._require_plan().lower_bound_contribution(self)
        return self._plan_call('lower_bound_contribution', ignore_masked=False)
