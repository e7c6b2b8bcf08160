"""
Device families of the count / probability nodes and of the small deterministic nodes
(Beta, Complement, Binomial / Bernoulli, Poisson, Add).  Same protocol as the families in
``generic.py`` (the five VMP formulas of stochastic.py:16-80 / the two of
deterministic.py:16-96), every array operation a HIP kernel launch.
"""
import numpy as np

from ... import darray as da
from ...darray import DArray, fuse
from ...nodes.node import Constant
from ...nodes.beta import Beta, Complement
from ...nodes.binomial import Binomial
from ...nodes.poisson import Poisson
from ...nodes.add import Add, ConcatGaussian
from ...nodes.take import Take, Concatenate, Gate, Slice
from ...nodes.categorical_markov_chain import (CategoricalMarkovChain,
                                                CategoricalMarkovChainToCategorical)
from ...utils import misc, linalg
from ...utils import random as drandom
from .generic import Family, DirichletFamily, GaussianMarkovChainFamily, _arr, _trail, _const
from ...nodes.gaussian_markov_chain import (SwitchingGaussianMarkovChain,
                                            VaryingGaussianMarkovChain)


def _flip2(x):
    """Swap the two entries of the last axis (a strided device view cannot be reversed: one
    tiny contraction with the exchange matrix)."""
    x = _arr(x)
    ex = _const(('exchange2',), lambda: np.array([[0.0, 1.0], [1.0, 0.0]]))
    return misc.sum_multiply(_trail(x, 1), ex, axis=-2)


def _pair(p):
    """Probabilities -> [p, 1 - p] on a new last axis (beta.py:33-37)."""
    p = _arr(np.asarray(p, dtype=np.float64)) if not isinstance(p, DArray) else p
    s = _const(('pair_s',), lambda: np.array([1.0, -1.0]))
    o = _const(('pair_o',), lambda: np.array([0.0, 1.0]))
    return fuse(lambda v, s_, o_: v * s_ + o_, _trail(p, 1), s, o)


class BetaFamily(DirichletFamily):
    """beta.py:46-110: Dirichlet formulas, scalar realisations."""

    def fixed_moments_and_f(self, p):
        return super().fixed_moments_and_f(_pair(p))


class ComplementFamily:
    """beta.py:194-214."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node

    def mask_to_parent(self, index, mask):
        return mask

    def moments(self, ups):
        return [_flip2(ups[0][0])]

    def message_to_parent(self, index, m_child, ups, mask=None):
        m = m_child[0]
        if m is None:
            return [None]
        m = _flip2(m)
        if mask is not None:
            m = fuse(lambda a, w: a * w, m, _trail(mask, 1))
        return [m]


class BinomialFamily(Family):
    """binomial.py:54-165 (Bernoulli: one trial, bernoulli.py:33-41)."""

    def __init__(self, node):
        super().__init__(node)
        self.trials = np.asarray(node.trials, dtype=np.float64)
        self.Nd = DArray.from_host(self.trials)

    def constant_moments(self, index, value):
        return [fuse(lambda q: da.log(q), _pair(value))]          # BetaMoments, beta.py:33-37

    def phi_from_parents(self, up):
        lp = _arr(up[0][0])
        return [fuse(lambda a, b: a - b, lp[..., 0], lp[..., 1])]

    def moments_and_cgf(self, phi):
        p = _arr(phi[0])
        u0 = fuse(lambda n, x: n / (1.0 + da.exp(-x)), self.Nd, p)
        # log(1 + e^x) without overflow: max(x, 0) + log(1 + e^-|x|)
        g = fuse(lambda n, x: -n * (da.maximum(x, 0.0)
                                    + da.log(1.0 + da.exp(-da.maximum(x, -x)))), self.Nd, p)
        return [u0], g

    def cgf_from_parents(self, up):
        lp = _arr(up[0][0])
        return fuse(lambda n, b: n * b, self.Nd, lp[..., 1])

    def fixed_moments_and_f(self, x):
        x = _arr(np.asarray(x, dtype=np.float64))
        f = fuse(lambda n, c: da.gammaln(n + 1.0) - da.gammaln(c + 1.0) - da.gammaln(n - c + 1.0),
                 self.Nd, x)
        return [x], f

    def message_to_parent(self, index, u, up):
        # [x, n - x] on a new last axis (binomial.py:80-84)
        s = _const(('pair_s',), lambda: np.array([1.0, -1.0]))
        o = _const(('pair_o',), lambda: np.array([0.0, 1.0]))
        return [fuse(lambda x, n, s_, o_: x * s_ + n * o_, _trail(_arr(u[0]), 1),
                     _trail(self.Nd, 1), s, o)]

    def plates_to_parent(self, index):
        return self.node.plates

    def sample(self, st):
        p = 1.0 / (1.0 + np.exp(-np.broadcast_to(_arr(st.phi[0]).numpy(), self.node.plates)))
        return np.random.binomial(np.broadcast_to(self.trials, self.node.plates).astype(np.int64),
                                  p)


class PoissonFamily(Family):
    """poisson.py:52-120."""

    def constant_moments(self, index, value):
        v = _arr(value)
        return [v, fuse(lambda l: da.log(l), v)]                  # GammaMoments, gamma.py:60-75

    def phi_from_parents(self, up):
        return [fuse(lambda l: 1.0 * l, up[0][1])]

    def moments_and_cgf(self, phi):
        u0 = fuse(lambda p: da.exp(p), phi[0])
        return [u0], fuse(lambda v: -v, u0)

    def cgf_from_parents(self, up):
        return fuse(lambda l: -l, up[0][0])

    def fixed_moments_and_f(self, x):
        x = _arr(np.asarray(x, dtype=np.float64))
        return [x], fuse(lambda c: -da.gammaln(c + 1.0), x)

    def message_to_parent(self, index, u, up):
        return [-1.0, u[0]]

    def sample(self, st):
        return np.random.poisson(np.exp(np.broadcast_to(_arr(st.phi[0]).numpy(),
                                                        self.node.plates)))


class AddFamily:
    """add.py:95-154."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.ndim = node.ndim

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        v = _arr(value)
        return [v, linalg.outer(v, v, ndim=self.ndim)]

    def moments(self, ups):
        u0, u1 = _arr(ups[0][0]), _arr(ups[0][1])
        for up in ups[1:]:
            x = _arr(up[0])
            xy = linalg.outer(u0, x, ndim=self.ndim)     # cross terms with the running sum
            yx = linalg.transpose(xy, ndim=self.ndim)
            u1 = fuse(lambda a, b, c, d: a + b + c + d, u1, _arr(up[1]), xy, yx)
            u0 = fuse(lambda a, b: a + b, u0, x)
        return [u0, u1]

    def message_to_parent(self, index, m_child, ups, mask=None):
        m0, m1 = m_child
        if m1 is None:
            out = [m0, None]
        else:
            others = [_arr(up[0]) for i, up in enumerate(ups) if i != index]
            s = others[0]
            for x in others[1:]:
                s = fuse(lambda a, b: a + b, s, x)
            m1 = _arr(m1)
            t = linalg.mvdot(fuse(lambda q: 2.0 * q, m1), s, ndim=self.ndim)
            out = [t if m0 is None else fuse(lambda a, b: a + b, _arr(m0), t), m1]
        if mask is not None:
            out = [None if m is None else
                   fuse(lambda a, w: a * w, _arr(m), _trail(mask, (1 + i) * self.ndim))
                   for i, m in enumerate(out)]
        return out


class ConcatGaussianFamily:
    """concat_gaussian.py:63-116."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.off = node.offsets

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        v = _arr(value)
        return [v, linalg.outer(v, v)]

    def moments(self, ups):
        xs = [_arr(u[0]) for u in ups]
        x = misc.concatenate(xs, axis=-1)
        rows = []
        for a, ua in enumerate(ups):
            blocks = [_arr(ua[1]) if a == b else linalg.outer(xs[a], xs[b])
                      for b in range(len(ups))]
            rows.append(misc.concatenate(blocks, axis=-1))
        return [x, misc.concatenate(rows, axis=-2)]

    def message_to_parent(self, index, m_child, ups, mask=None):
        r = self.off
        m0, m1 = m_child
        a, b = r[index], r[index + 1]
        out0 = None if m0 is None else _arr(m0)[..., a:b]
        out1 = None
        if m1 is not None:
            m1 = _arr(m1)
            out1 = m1[..., a:b, a:b]
            for j, u in enumerate(ups):
                if j == index:
                    continue
                t = linalg.mvdot(fuse(lambda q: 2.0 * q, m1[..., a:b, r[j]:r[j + 1]]), _arr(u[0]))
                out0 = t if out0 is None else fuse(lambda p, q: p + q, out0, t)
        out = [out0, out1]
        if mask is not None:
            out = [None if m is None else fuse(lambda p, w: p * w, _arr(m), _trail(mask, 1 + i))
                   for i, m in enumerate(out)]
        return out


def _masked(m, mask, nd):
    if mask is None or m is None:
        return m
    return fuse(lambda a, w: a * w, _arr(m), _trail(mask, nd))


class TakeFamily:
    """take.py:72-140: moments are gathered along the plate axis, messages are accumulated
    back (several picks of the same element add up)."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.axis = node.plate_axis
        self.imap = misc.IndexMap(node.indices, node.original_length)
        self.ndims = [len(d) for d in node.dims]

    def plates_to_parent(self, index):
        p = self.node.plates
        end_before = self.axis - self.node.indices.ndim + 1
        start_after = self.axis + 1
        head = p[:len(p) + end_before] if end_before != 0 else p
        tail = p[len(p) + start_after:] if start_after != 0 else ()
        return tuple(head) + (self.node.original_length,) + tuple(tail)

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask, dtype=bool)
        nax = self.node.indices.ndim
        first = self.axis - nax + 1
        if mask.ndim < -first:
            mask = mask.reshape((1,) * (-first - mask.ndim) + mask.shape)
        a = mask.ndim + first
        full = mask.shape[:a] + self.node.indices.shape + mask.shape[a + nax:]
        m = np.broadcast_to(mask, full).reshape(full[:a] + (-1,) + full[a + nax:])
        out = np.zeros(full[:a] + (self.node.original_length,) + full[a + nax:], dtype=np.int64)
        flat = np.where(self.node.indices < 0, self.node.indices + self.node.original_length,
                        self.node.indices).reshape(-1)
        np.add.at(np.moveaxis(out, a, 0), flat, np.moveaxis(m, a, 0).astype(np.int64))
        return out > 0

    def moments(self, ups):
        out = []
        L = self.node.original_length
        for ui, nd in zip(ups[0], self.ndims):
            ui = _arr(ui)
            ax = self.axis - nd
            if ui.ndim < -ax:
                ui = ui.reshape((1,) * (-ax - ui.ndim) + ui.shape)
            if ui.shape[ax] != L:             # broadcast along the taken axis
                sh = list(ui.shape)
                sh[ax] = L
                ui = ui.broadcast_to(tuple(sh))
            out.append(misc.take(ui, self.imap, axis=ax))
        return out

    def message_to_parent(self, index, m_child, ups, mask=None):
        out = []
        for m, nd in zip(m_child, self.ndims):
            if m is None:
                out.append(None)
                continue
            m = _masked(m, mask, nd)
            out.append(misc.put_simple(_arr(m), self.imap, axis=self.axis - nd))
        return out


class ConcatenateFamily:
    """concatenate.py:96-167."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.axis = node.axis
        self.ndims = [len(d) for d in node.dims]

    def plates_to_parent(self, index):
        p = list(self.node.plates)
        p[self.axis] = self.node.lengths[index]
        return tuple(p)

    def _slice(self, x, index, ax):
        """The part of ``x`` that belongs to parent ``index`` along array axis ``ax`` (all of
        it when that axis is missing or broadcast)."""
        nd = np.ndim(x) if not isinstance(x, DArray) else x.ndim
        shape = x.shape
        if nd >= -ax and shape[ax] > 1:
            a, b = int(self.node.offsets[index]), int(self.node.offsets[index + 1])
            sl = [slice(None)] * nd
            sl[ax] = slice(a, b)
            return x[tuple(sl)]
        return x

    def mask_to_parent(self, index, mask):
        return self._slice(np.asarray(mask), index, self.axis)

    def moments(self, ups):
        out = []
        for i, nd in enumerate(self.ndims):
            ax = self.axis - nd
            parts = []
            for up, n in zip(ups, self.node.lengths):
                x = _arr(up[i])
                if x.ndim < -ax:
                    x = x.reshape((1,) * (-ax - x.ndim) + x.shape)
                if x.shape[ax] != n:
                    sh = list(x.shape)
                    sh[ax] = n
                    x = x.broadcast_to(tuple(sh))
                parts.append(x)
            out.append(misc.concatenate(parts, axis=ax))
        return out

    def message_to_parent(self, index, m_child, ups, mask=None):
        out = []
        for m, nd in zip(m_child, self.ndims):
            if m is None:
                out.append(None)
                continue
            m = _arr(_masked(m, mask, nd))
            out.append(self._slice(m, index, self.axis - nd))
        return out


class GateFamily:
    """gate.py:71-205: moments  sum_k <z_k> u_X[k]  over the gated plate axis."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.gp = node.gated_plate
        self.K = node.K
        self.ndims = [len(d) for d in node.dims]

    def constant_moments(self, index, value):
        if index != 0:
            raise NotImplementedError('the gated parent must be a node')
        return [misc.onehot(np.asarray(value).astype(np.int64), self.K)]

    def plates_to_parent(self, index):
        if index == 0:
            return self.node.plates
        p = list(self.node.plates)
        p.insert(len(p) + self.gp + 1, self.K)
        return tuple(p)

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        if index == 0 or mask.ndim < abs(self.gp):
            return mask
        return np.expand_dims(mask, self.gp)

    def _gated_last_plate(self, x, nd):
        """x (plates of X + dims_i) -> the gated axis moved to just before the dims."""
        x = _arr(x)
        ax = self.gp - nd
        if x.ndim < -ax:
            return x.reshape(x.shape[:x.ndim - nd] + (1,) + x.shape[x.ndim - nd:])
        return misc.moveaxis(x, ax, -nd - 1)

    def moments(self, ups):
        z = _arr(ups[0][0])
        out = []
        for x, nd in zip(ups[1], self.ndims):
            out.append(misc.sum_multiply(_trail(z, nd), self._gated_last_plate(x, nd),
                                         axis=-nd - 1))
        return out

    def message_to_parent(self, index, m_child, ups, mask=None):
        z = _arr(ups[0][0])
        if index == 0:
            tot = None
            for m, x, nd in zip(m_child, ups[1], self.ndims):
                if m is None:
                    continue
                c = _arr(_masked(m, mask, nd))
                c = c.reshape(c.shape[:c.ndim - nd] + (1,) + c.shape[c.ndim - nd:])
                t = misc.sum_multiply(c, self._gated_last_plate(x, nd),
                                      axis=tuple(range(-nd, 0))) if nd else \
                    fuse(lambda a, b: a * b, c, self._gated_last_plate(x, nd))
                tot = t if tot is None else fuse(lambda a, b: a + b, tot, t)
            if tot is None:
                return [None]
            if tot.shape[-1] != self.K:           # the class axis must not be broadcast
                tot = fuse(lambda a, o: a * o, tot, _const(('ones', (self.K,)),
                                                           lambda: np.ones(self.K)))
            return [tot]
        out = []
        for m, nd in zip(m_child, self.ndims):
            if m is None:
                out.append(None)
                continue
            c = _arr(_masked(m, mask, nd))
            c = c.reshape(c.shape[:c.ndim - nd] + (1,) + c.shape[c.ndim - nd:])
            mi = fuse(lambda a, b: a * b, _trail(z, nd), c)     # plates + (K,) + dims
            ax = self.gp - nd
            if mi.ndim < -ax:
                mi = mi.reshape((1,) * (-ax - mi.ndim) + mi.shape)
            out.append(misc.moveaxis(mi, -nd - 1, ax))
        return out


class CategoricalMarkovChainFamily(Family):
    """categorical_markov_chain.py:71-205; the moments are one launch of the forward-backward
    kernel for all chains and time instances (utils/random.py:357-422)."""

    def __init__(self, node):
        super().__init__(node)
        self.K = node.categories
        self.N = node.states

    def constant_moments(self, index, value):
        return [fuse(lambda p: da.log(p), _arr(value))]          # DirichletMoments of a value

    def plates_to_parent(self, index):
        if index == 0:
            return self.node.plates
        return self.node.plates + (self.N - 1, self.K)

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        return mask if index == 0 else mask[..., None, None]

    def phi_from_parents(self, up):
        logP = _arr(up[1][0])
        want = (self.N - 1, self.K, self.K)
        if logP.ndim < 3:
            logP = logP.reshape((1,) * (3 - logP.ndim) + logP.shape)
        if logP.shape[-3:] != want:
            logP = logP.broadcast_to(logP.shape[:-3] + want)     # time-invariant transitions
        return [_arr(up[0][0]), logP]

    def moments_and_cgf(self, phi):
        z0, zz, g = drandom.alpha_beta_recursion(phi[0], phi[1])
        return [z0, zz], g

    def cgf_from_parents(self, up):
        return 0.0

    def fixed_moments_and_f(self, x):
        # one-hot first state and one-hot transitions (categorical_markov_chain.py:38-58)
        x = np.asarray(x).astype(np.int64)
        K = self.K
        u0 = misc.onehot(x[..., 0], K)
        pair = x[..., :-1] * K + x[..., 1:]
        us = misc.onehot(pair, K * K)
        return [u0, us.reshape(us.shape[:-1] + (K, K))], 0.0

    def message_to_parent(self, index, u, up):
        return [u[index]]

    def sample(self, st):
        plates = self.node.plates
        p0 = np.broadcast_to(_arr(st.u[0]).numpy(), plates + (self.K,)).reshape(-1, self.K)
        zz = np.broadcast_to(_arr(st.u[1]).numpy(),
                             plates + (self.N - 1, self.K, self.K)).reshape(-1, self.N - 1,
                                                                             self.K, self.K)
        B = p0.shape[0]
        Z = np.zeros((B, self.N), dtype=np.int64)

        def draw(p):
            c = np.cumsum(p, axis=-1)
            r = np.random.rand(p.shape[0], 1) * c[:, -1:]
            return (r > c).sum(axis=1).clip(0, self.K - 1)
        Z[:, 0] = draw(p0)
        rows = np.arange(B)
        for n in range(self.N - 1):
            Z[:, n + 1] = draw(zz[rows, n, Z[:, n]] + 1e-300)     # q(z_{n+1} | z_n)
        return Z.reshape(plates + (self.N,))


class ChainToCategoricalFamily:
    """CategoricalMarkovChainToCategorical (categorical_markov_chain.py:363-438): marginals of
    all time instances with time as the last plate."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node

    def plates_to_parent(self, index):
        return self.node.plates[:-1]

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        return np.any(mask, axis=-1) if mask.ndim >= 1 else mask

    def moments(self, ups):
        z0, zz = _arr(ups[0][0]), _arr(ups[0][1])
        p = misc.sum_multiply(zz, axis=-2)                      # q(z_{n+1}), (..., N-1, K)
        first = z0.reshape(z0.shape[:-1] + (1, z0.shape[-1]))
        return [misc.concatenate([first, p], axis=-2)]

    def message_to_parent(self, index, m_child, ups, mask=None):
        m = m_child[0]
        if m is None:
            return [None, None]
        m = _arr(m)
        if mask is not None:
            m = fuse(lambda a, w: a * w, m, _trail(mask, 1))
        N = self.node.plates[-1]
        if m.ndim < 2 or m.shape[-2] != N:
            # constant over time: give it the time axis
            lead = m.shape[:-1] if m.ndim >= 1 else ()
            lead = lead[:-1] if (m.ndim >= 2 and m.shape[-2] == 1) else lead
            m = m.broadcast_to(lead + (N, self.node.categories)) if m.ndim >= 2 else \
                m.reshape((1, -1)).broadcast_to((N, self.node.categories))
        m0 = m[..., 0, :]
        m1 = m[..., 1:, :]
        m1 = m1.reshape(m1.shape[:-1] + (1, m1.shape[-1]))       # (..., N-1, 1, K)
        return [m0, m1]


class SliceFamily:
    """Slice (node.py:868-1130): moments are strided views of the parent's moments; a message
    goes back into a zero array of the parent's plates, one accumulating put per indexed axis
    (no duplicates: every target element has at most one source)."""
    deterministic = True
    plate_sum = True

    def __init__(self, node):
        self.node = node
        self.parent = node.parents[0]
        self.ndims = [len(d) for d in node.dims]
        self.P = len(self.parent.plates)
        # per parent plate axis: ('int', s) | ('slice', slice, IndexMap or None)
        self.axes = []
        j = 0
        for sl in node.slices:
            if sl is None:
                continue
            L = self.parent.plates[j]
            if isinstance(sl, int):
                self.axes.append(('int', sl, misc.IndexMap([sl], L)))
            else:
                rng = range(sl.start, sl.stop, sl.step)
                full = (len(rng) == L and sl.step == 1)
                self.axes.append(('slice', sl, None if full else misc.IndexMap(list(rng), L)))
            j += 1

    def plates_to_parent(self, index):
        return self.parent.plates

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask, dtype=bool)
        full = np.broadcast_to(mask, self.node.plates)
        out = np.zeros(self.parent.plates, dtype=bool)
        child_idx, parent_idx = [], []
        for sl in self.node.slices:
            if sl is None:
                child_idx.append(0)
            elif isinstance(sl, int):
                parent_idx.append(sl)
            else:
                child_idx.append(slice(None))
                parent_idx.append(sl)
        out[tuple(parent_idx)] = full[tuple(child_idx)]
        return out

    def moments(self, ups):
        out = []
        for x, nd in zip(ups[0], self.ndims):
            x = _arr(x)
            npl = x.ndim - nd
            if npl < self.P:
                x = x.reshape((1,) * (self.P - npl) + x.shape)
            idx = []
            j = 0
            for sl in self.node.slices:
                if sl is None:
                    idx.append(None)
                    continue
                kind, s_, imap = self.axes[j]
                bc = x.shape[j] == 1 and self.parent.plates[j] != 1      # broadcast axis
                if kind == 'int':
                    idx.append(0 if bc else s_)
                elif bc or imap is None:
                    idx.append(slice(None))
                elif s_.step > 0:
                    idx.append(s_)
                else:
                    # a reversed slice is not a strided view of a device tensor: gather it
                    x = misc.take(x, imap, axis=j - x.ndim)
                    idx.append(slice(None))
                j += 1
            out.append(x[tuple(idx) + (Ellipsis,)])
        return out

    def message_to_parent(self, index, m_child, ups, mask=None):
        out = []
        C = len(self.node.plates)
        for m, nd in zip(m_child, self.ndims):
            if m is None:
                out.append(None)
                continue
            m = _arr(_masked(m, mask, nd))
            npl = m.ndim - nd
            if npl < C:
                m = m.reshape((1,) * (C - npl) + m.shape)
            # rearrange the plate axes of the node into the parent's: drop new axes, give
            # integer-indexed parent axes a unit axis
            shape, ci = [], 0
            for sl in self.node.slices:
                if sl is None:
                    ci += 1
                elif isinstance(sl, int):
                    shape.append(1)
                else:
                    shape.append(m.shape[ci])
                    ci += 1
            m = m.reshape(tuple(shape) + m.shape[C:])
            for j in range(self.P - 1, -1, -1):
                kind, s_, imap = self.axes[j]
                if imap is None:
                    continue
                ax = j - self.P - nd
                if kind == 'slice' and m.shape[j] == 1 and imap.n > 1:
                    sh = list(m.shape)
                    sh[j] = imap.n
                    m = m.broadcast_to(tuple(sh))
                m = misc.put_simple(m, imap, axis=ax)
            out.append(m)
        return out


class SwitchingGaussianMarkovChainFamily(GaussianMarkovChainFamily):
    """gaussian_markov_chain.py:1454-1741: the dynamics of transition n are
    sum_k <z_nk> B_k.  The smoother, the fixed moments and the messages to mu / Lambda are those
    of the plain chain; every einsum of the reference is one ``sum_multiply`` launch here."""

    def __init__(self, node):
        super().__init__(node)
        self.K = node.K

    def plates_to_parent(self, index):
        p = self.node.plates
        if index < 2:
            return p
        if index == 2:
            return p + (self.K, self.D)
        if index == 3:
            return p + (self.N - 1,)
        return p + (self.N - 1, self.D)

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        if index < 2:
            return mask
        return mask[..., None] if index == 3 else mask[..., None, None]

    def constant_moments(self, index, value):
        if index == 3:
            return [misc.onehot(np.asarray(value).astype(np.int64), self.K)]
        if index == 4:
            return super().constant_moments(3, value)
        return super().constant_moments(index, value)

    def _v(self, up, i):
        """Innovation precision moment i as (..., D): a unit time axis is dropped."""
        v = _arr(up[4][i])
        par = self.node.parents[4]
        npl = par.value.ndim if isinstance(par, Constant) else len(par.plates)
        if npl >= 2 and v.ndim >= 2:
            v = v[..., 0, :]
        return v

    def _vBB(self, up):
        """sum_d v_d <b_kd b_kd^T>  (..., K, D, D)."""
        D = self.D
        v = self._v(up, 0)
        return misc.sum_multiply(v.reshape(v.shape[:-1] + (1, D, 1, 1)), _arr(up[2][1]), axis=-3)

    def phi_from_parents(self, up):
        D, N = self.D, self.N
        m, Lam = up[0][0], up[1][0]
        Bm, Z = _arr(up[2][0]), _arr(up[3][0])
        v = self._v(up, 0)
        Lm = linalg.mvdot(Lam, m)
        phi0 = fuse(lambda e, q: e * q, self._e0v, _arr(Lm).reshape(_arr(Lm).shape[:-1] + (1, D)))
        # sum_k z_nk sum_d v_d <b_kd b_kd^T> for the N-1 transitions, zero for the last instance
        vBB = self._vBB(up)
        dyn = misc.sum_multiply(Z.reshape(Z.shape + (1, 1)),
                                vBB.reshape(vBB.shape[:-3] + (1,) + vBB.shape[-3:]), axis=-3)
        dyn = misc.concatenate([dyn, DArray.zeros((1, D, D))], axis=-3)
        dv = misc.diag(v.reshape(v.shape[:-1] + (1, D)), ndim=1)            # (..., 1, D, D)
        L = _arr(Lam)
        L = L.reshape(L.shape[:-2] + (1,) + L.shape[-2:])
        phi1 = fuse(lambda a, b, l, d, q: -0.5 * (a * l + b * d + q), self._e0, self._en0, L, dv,
                    dyn)
        # phi2[n, i, j] = sum_k z_nk B_k[j, i] v_j
        Bt = Bm.swapaxes(-1, -2)
        phi2 = misc.sum_multiply(Z.reshape(Z.shape + (1, 1)),
                                 Bt.reshape(Bt.shape[:-3] + (1,) + Bt.shape[-3:]),
                                 v.reshape(v.shape[:-1] + (1, 1, 1, D)), axis=-3)
        return [phi0, phi1, phi2]

    def cgf_from_parents(self, up):
        mm = up[0][1]
        Lam, logdet = up[1]
        s = misc.sum_multiply(self._v(up, 1), axis=-1)
        return fuse(lambda t, ld, ln: -0.5 * t + 0.5 * ld + 0.5 * (self.N - 1) * ln,
                    misc.sum_multiply(Lam, mm, axis=(-1, -2)), logdet, s)

    def message_to_parent(self, index, u, up):
        if index < 2:
            return super().message_to_parent(index, u, up)
        if index == 4:
            raise NotImplementedError('message to the innovation precision of a switching chain')
        D = self.D
        XX, XpXn = _arr(u[1]), _arr(u[2])
        XXp = XX[..., :-1, :, :]                         # <x_{n-1} x_{n-1}^T>, n = 1 .. N-1
        Z = _arr(up[3][0])
        v = self._v(up, 0)
        if index == 2:
            XnXp = XpXn.swapaxes(-1, -2)                 # [i][j] = <x_n[i] x_{n-1}[j]>
            m0 = misc.sum_multiply(XnXp.reshape(XnXp.shape[:-2] + (1, D, D)),
                                   Z.reshape(Z.shape + (1, 1)),
                                   v.reshape(v.shape[:-1] + (1, 1, D, 1)), axis=-4)
            m1 = misc.sum_multiply(XXp.reshape(XXp.shape[:-2] + (1, 1, D, D)),
                                   Z.reshape(Z.shape + (1, 1, 1)),
                                   v.reshape(v.shape[:-1] + (1, 1, D, 1, 1)), axis=-5)
            return [m0, fuse(lambda q: -0.5 * q, m1)]
        # index == 3: expected log-density of every transition under every dynamics matrix
        Bm = _arr(up[2][0])
        t_nn = misc.sum_multiply(misc.get_diag(XX[..., 1:, :, :], ndim=1),
                                 v.reshape(v.shape[:-1] + (1, D)), axis=-1)       # (..., N-1)
        Bt = Bm.swapaxes(-1, -2)                                                   # [k][i][l]
        t_pn = misc.sum_multiply(XpXn.reshape(XpXn.shape[:-2] + (1, D, D)),
                                 v.reshape(v.shape[:-1] + (1, 1, 1, D)),
                                 Bt.reshape(Bt.shape[:-3] + (1,) + Bt.shape[-3:]),
                                 axis=(-1, -2))                                    # (..., N-1, K)
        vBB = self._vBB(up)
        t_pp = misc.sum_multiply(XXp.reshape(XXp.shape[:-2] + (1, D, D)),
                                 vBB.reshape(vBB.shape[:-3] + (1,) + vBB.shape[-3:]),
                                 axis=(-1, -2))                                    # (..., N-1, K)
        slv = misc.sum_multiply(self._v(up, 1), axis=-1)
        c = -0.5 * D * float(np.log(2 * np.pi))
        m0 = fuse(lambda a, b, c_, s: -0.5 * a + b - 0.5 * c_ + 0.5 * s + c,
                  _trail(t_nn, 1), t_pn, t_pp, _trail(_arr(slv), 2))
        return [m0]


class VaryingGaussianMarkovChainFamily(SwitchingGaussianMarkovChainFamily):
    """gaussian_markov_chain.py:930-1198: the dynamics of transition n are sum_k <s_nk> B_k with
    Gaussian weights s_n; B has shape (D, K) per row plate."""

    def plates_to_parent(self, index):
        p = self.node.plates
        if index < 2:
            return p
        if index == 2:
            return p + (self.D,)
        if index == 3:
            return p + (self.N - 1,)
        return p + (self.N - 1, self.D)

    def mask_to_parent(self, index, mask):
        mask = np.asarray(mask)
        if index < 2:
            return mask
        return mask[..., None, None] if index == 4 else mask[..., None]

    def constant_moments(self, index, value):
        if index in (2, 3):
            raise NotImplementedError('the dynamics matrices and their weights must be nodes')
        return super().constant_moments(index, value)

    def _vBB(self, up):
        """sum_d v_d <b_d b_d^T> with b_d = B[d] of shape (D, K):  (..., D_i, K_k, D_j, K_l)."""
        D = self.D
        v = self._v(up, 0)
        return misc.sum_multiply(v.reshape(v.shape[:-1] + (D, 1, 1, 1, 1)), _arr(up[2][1]), axis=-5)

    def phi_from_parents(self, up):
        D, N, K = self.D, self.N, self.K
        m, Lam = up[0][0], up[1][0]
        Bm = _arr(up[2][0])                                  # (..., D_j, D_i, K)
        S, SS = _arr(up[3][0]), _arr(up[3][1])               # (..., N-1, K), (..., N-1, K, K)
        v = self._v(up, 0)
        Lm = linalg.mvdot(Lam, m)
        phi0 = fuse(lambda e, q: e * q, self._e0v, _arr(Lm).reshape(_arr(Lm).shape[:-1] + (1, D)))
        vBB = self._vBB(up)                                  # (..., i, k, j, l)
        # sum_kl vBB[i,k,j,l] <s_nk s_nl>  ->  (..., N-1, i, j)
        dyn = misc.sum_multiply(SS.reshape(SS.shape[:-2] + (1, K, 1, K)),
                                vBB.reshape(vBB.shape[:-4] + (1,) + vBB.shape[-4:]),
                                axis=(-1, -3))
        dyn = misc.concatenate([dyn, DArray.zeros((1, D, D))], axis=-3)
        dv = misc.diag(v.reshape(v.shape[:-1] + (1, D)), ndim=1)
        L = _arr(Lam)
        L = L.reshape(L.shape[:-2] + (1,) + L.shape[-2:])
        phi1 = fuse(lambda a, b, l, d, q: -0.5 * (a * l + b * d + q), self._e0, self._en0, L, dv,
                    dyn)
        # phi2[n, i, j] = sum_k B[j, i, k] s_nk v_j
        Bt = Bm.swapaxes(-2, -3)                             # [i][j][k]
        phi2 = misc.sum_multiply(Bt.reshape(Bt.shape[:-3] + (1,) + Bt.shape[-3:]),
                                 S.reshape(S.shape[:-1] + (1, 1, K)),
                                 v.reshape(v.shape[:-1] + (1, 1, D, 1)), axis=-1)
        return [phi0, phi1, phi2]

    def message_to_parent(self, index, u, up):
        if index < 2:
            return GaussianMarkovChainFamily.message_to_parent(self, index, u, up)
        if index == 4:
            raise NotImplementedError('message to the innovation precision of a varying chain')
        D, K = self.D, self.K
        XX, XpXn = _arr(u[1]), _arr(u[2])
        XXp = XX[..., :-1, :, :]
        S, SS = _arr(up[3][0]), _arr(up[3][1])
        v = self._v(up, 0)
        XnXp = XpXn.swapaxes(-1, -2)                         # [i][j] = <x_n[i] x_{n-1}[j]>
        if index == 2:
            # m0[i, j, k] = sum_n <x_n[i] x_{n-1}[j]> s_nk v_i
            m0 = misc.sum_multiply(XnXp.reshape(XnXp.shape + (1,)),
                                   S.reshape(S.shape[:-1] + (1, 1, K)),
                                   v.reshape(v.shape[:-1] + (1, D, 1, 1)), axis=-4)
            # m1[d, i, k, j, l] = -1/2 v_d sum_n <x x^T>[n, i, j] <s s^T>[n, k, l]
            t = misc.sum_multiply(XXp.reshape(XXp.shape[:-2] + (D, 1, D, 1)),
                                  SS.reshape(SS.shape[:-2] + (1, K, 1, K)), axis=-5)
            m1 = fuse(lambda a, b: -0.5 * a * b,
                      t.reshape(t.shape[:-4] + (1,) + t.shape[-4:]),
                      v.reshape(v.shape[:-1] + (D, 1, 1, 1, 1)))
            return [m0, m1]
        # index == 3
        Bm = _arr(up[2][0])                                  # (..., D_i, D_j, K)
        m0 = misc.sum_multiply(XnXp.reshape(XnXp.shape + (1,)),
                               Bm.reshape(Bm.shape[:-3] + (1,) + Bm.shape[-3:]),
                               v.reshape(v.shape[:-1] + (1, D, 1, 1)), axis=(-2, -3))
        vBB = self._vBB(up)                                  # (..., i, k, j, l)
        m1 = misc.sum_multiply(XXp.reshape(XXp.shape[:-2] + (D, 1, D, 1)),
                               vBB.reshape(vBB.shape[:-4] + (1,) + vBB.shape[-4:]),
                               axis=(-2, -4))
        return [m0, fuse(lambda q: -0.5 * q, m1)]


def make_extra_family(node):
    if isinstance(node, ConcatGaussian):
        return ConcatGaussianFamily(node)
    if isinstance(node, VaryingGaussianMarkovChain):
        return VaryingGaussianMarkovChainFamily(node)
    if isinstance(node, SwitchingGaussianMarkovChain):
        return SwitchingGaussianMarkovChainFamily(node)
    if isinstance(node, Slice):
        return SliceFamily(node)
    if isinstance(node, CategoricalMarkovChain):
        return CategoricalMarkovChainFamily(node)
    if isinstance(node, CategoricalMarkovChainToCategorical):
        return ChainToCategoricalFamily(node)
    if isinstance(node, Take):
        return TakeFamily(node)
    if isinstance(node, Concatenate):
        return ConcatenateFamily(node)
    if isinstance(node, Gate):
        return GateFamily(node)
    if isinstance(node, Beta):
        return BetaFamily(node)
    if isinstance(node, Complement):
        return ComplementFamily(node)
    if isinstance(node, Binomial):
        return BinomialFamily(node)
    if isinstance(node, Poisson):
        return PoissonFamily(node)
    if isinstance(node, Add):
        return AddFamily(node)
    return None
