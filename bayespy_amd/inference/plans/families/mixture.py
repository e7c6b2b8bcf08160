"""Mixture over the last parameter plate (mixture.py:26-356)."""
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
