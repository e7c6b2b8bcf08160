"""SumMultiply / Dot (dot.py:19-633): einsum over Gaussian moments and its messages."""
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
