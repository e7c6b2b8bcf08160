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

# the engine in three parts (round 6): lazily evaluated arrays and helpers (lazy.py), the families
# (families/), and this module -- the plan: state + message routing.  The names stay importable from
# here (tests, tools, families_extra, extension).
from .lazy import (DerivedArray,
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
from .families import (Family, GammaFamily, WishartFamily, DirichletFamily, CategoricalFamily,
                       MultinomialFamily, GaussianARDFamily, GaussianFamily, GaussianGammaFamily,
                       GaussianToGaussianGammaFamily, WrapToGaussianGammaFamily, MixtureFamily,
                       GaussianMarkovChainFamily, ChainToGaussianFamily, SumMultiplyFamily,
                       make_family)


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

    def _prior_terms(self, node, fam, up, cgf=False):
        """phi_from_parents (and, ``cgf``, cgf_from_parents) of ``node``.  When every parent is a
        constant these never change: formed once and kept with the constants (a Gamma or Gaussian
        node with fixed hyperparameters paid 3-5 small launches for them in its update and again
        in its lower-bound term, every sweep)."""
        if os.environ.get('BAYESPY_AMD_PRIOR_CACHE', '1') == '0' or not node.parents \
                or not all(isinstance(p, Constant) for p in node.parents):
            return fam.phi_from_parents(up), (fam.cgf_from_parents(up) if cgf else None)
        key = (id(node), 'prior')
        hit = self._const_cache.get(key)
        if hit is None:
            hit = [fam.phi_from_parents(up), None]
            self._const_cache[key] = hit
        if cgf and hit[1] is None:
            hit[1] = _arr(fam.cgf_from_parents(up))
        return hit[0], hit[1]

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
        phi, _ = self._prior_terms(node, fam, up)
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
        phi_p, L = self._prior_terms(node, fam, up, cgf=True)
        L = _arr(L)
        pend = None
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
                # (cgf_p - T g_q joins the formula of the first moment when that is a scalar per plate:
                # same operations in the same order, one launch less)
                pend = (L, st.g)
                L = None
        fold = os.environ.get('BAYESPY_AMD_BOUND_FOLD', '1') != '0'

        def settle():
            nonlocal L, pend
            if pend is not None:
                L = fuse(lambda a, g, T_=T: a - T_ * g, pend[0], pend[1])
                pend = None
        for i, nd in enumerate(len(d) for d in node.dims):
            if closed is not None and nd > 0:
                settle()
                # finite Gaussian prior parameters: phi_p . u as one contraction, no temporary
                if i == 1 and isinstance(st.u[i], FactoredMoment):
                    L = fuse(lambda a, b: a + b, L, _inner_second(phi_p[i], st.u[i], nd // 2))
                    continue
                L = fuse(lambda a, b: a + b, L,
                         misc.sum_multiply(_arr(phi_p[i]), _arr(st.u[i]),
                                           axis=tuple(range(-nd, 0))))
                continue
            if partial:
                settle()
                t = fuse(lambda pp, pq, m, u, T_=T:
                         da.where_nonzero(u, pp - T_ * (1.0 - m) * pq) * u,
                         _arr(phi_p[i]), _arr(st.phi[i]), _trail(st.obs_mask, nd), _arr(st.u[i]))
            elif st.observed or closed is not None:
                settle()
                if nd == 0 and fold:
                    L = fuse(lambda a, pp, u: a + da.where_nonzero(u, pp) * u,
                             L, _arr(phi_p[i]), _arr(st.u[i]))
                    continue
                t = fuse(lambda pp, u: da.where_nonzero(u, pp) * u, _arr(phi_p[i]), _arr(st.u[i]))
            else:
                if nd == 0 and fold:
                    # a scalar moment per plate: its term and the running sum in ONE formula
                    if pend is not None:
                        L = fuse(lambda a, g, pp, pq, u, T_=T:
                                 (a - T_ * g) + da.where_nonzero(u, pp - T_ * pq) * u,
                                 pend[0], pend[1], _arr(phi_p[i]), _arr(st.phi[i]), _arr(st.u[i]))
                        pend = None
                    else:
                        L = fuse(lambda a, pp, pq, u, T_=T: a + da.where_nonzero(u, pp - T_ * pq) * u,
                                 L, _arr(phi_p[i]), _arr(st.phi[i]), _arr(st.u[i]))
                    continue
                settle()
                t = fuse(lambda pp, pq, u, T_=T: da.where_nonzero(u, pp - T_ * pq) * u,
                         _arr(phi_p[i]), _arr(st.phi[i]), _arr(st.u[i]))
            L = fuse(lambda a, b: a + b, L, _sum_last(t, nd))
        settle()
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
