"""
Contraction / broadcast utilities on device arrays.

Device counterparts of ``bayespy.utils.misc``: ``sum_multiply`` (:851-933),
``sum_product`` (:935-945), ``sum_multiply_to_plates`` (:805-844),
``broadcasting_multiplier`` (:761-802), ``logsumexp`` / ``normalized_exp``
(:1366-1401), ``multidigamma`` (:1146-1151), ``diag`` / ``get_diag``
(:1253, :1207), ``add_trailing_axes`` (:1052).  Same names, argument meaning
and error behaviour; every reduction is the ``vmp_sum_multiply`` HIP kernel.
"""
import ctypes
import functools
import operator
import os

import numpy as np

from ..darray import DArray, asdarray, fuse, contiguous, is_scalar, _strides
from ..darray import digamma as _digamma
from ..device import get_runtime
from .shapes import broadcasted_shape, broadcasting_multiplier, is_shape_subset  # noqa: F401

_ws = {}


def _workspace(rt):
    key = id(rt)
    if key not in _ws:
        nbytes = rt.lib.vmp_sum_multiply_workspace_bytes()
        _ws[key] = rt.empty(nbytes // 8)
    return _ws[key]


def _merge_axes(dims, shape, *stride_lists):
    """Merge the axes `dims` (outer -> inner) into one strided axis for every stride list;
    returns (extent, [stride per list]) or None when some operand is not laid out densely
    across them."""
    if not dims:
        return 1, [0] * len(stride_lists)
    ext = 1
    for d in dims:
        ext *= shape[d]
    out = []
    for st in stride_lists:
        for a, b in zip(dims[:-1], dims[1:]):
            if st[a] != st[b] * shape[b]:
                return None
        out.append(st[dims[-1]])
    return ext, out


def _try_gemm(rt, arrays, shape, reduce_axes, out, scale):
    """Route a dense two-operand contraction to the fp64 MFMA kernel (``vmp_gemm_strided``).
    Returns True when launched."""
    ops = list(arrays)
    # fold operands of "subset" shape into a partner (e.g. a plate mask into its message)
    while len(ops) > 2:
        done = False
        # (the smallest operand first, into the smallest partner that contains its shape: a
        # plate-free factor multiplies a K x K matrix, not the (D, N) data)
        order = sorted(range(len(ops)), key=lambda q: ops[q].size)
        for i in order:
            for j in order:
                if i == j:
                    continue
                try:
                    bs = broadcasted_shape(ops[i].shape, ops[j].shape)
                except ValueError:
                    continue
                full_j = (1,) * (len(bs) - ops[j].ndim) + ops[j].shape
                if bs == full_j:
                    prod = fuse(lambda a, b: a * b, ops[i], ops[j])
                    ops = [o for k, o in enumerate(ops) if k not in (i, j)] + [prod]
                    done = True
                    break
            if done:
                break
        if not done:
            return False
    if len(ops) != 2:
        return False
    A, B = ops
    nd = len(shape)
    sa, sb, so = _strides(A.t, shape), _strides(B.t, shape), _strides(out.t, shape)
    Md, Nd, Kd, Bd = [], [], [], []
    for d in range(nd):
        if shape[d] == 1:
            continue
        va, vb = sa[d] != 0, sb[d] != 0
        if d in reduce_axes:
            if va and vb:
                Kd.append(d)
            elif not va and not vb:
                scale *= shape[d]
            else:
                return False
        elif va and vb:
            Bd.append(d)
        elif va:
            Md.append(d)
        elif vb:
            Nd.append(d)
        else:
            return False
    if len(Bd) > 3:
        return False
    mm = _merge_axes(Md, shape, sa, so)
    nn = _merge_axes(Nd, shape, sb, so)
    kk = _merge_axes(Kd, shape, sa, sb)
    if mm is None or nn is None or kk is None:
        return False
    (M, (a_ms, c_ms)), (N, (b_ns, c_ns)), (K, (a_ks, b_ks)) = mm, nn, kk
    nbatch = 1
    for d in Bd:
        nbatch *= shape[d]
    # (short contractions with a large output are worth it too: the generic kernels pay a
    # full index decode per output element)
    if M < 8 or N < 8 or K < 4 or M * N * K * nbatch < (1 << 20) or nbatch > 65535:
        return False
    if nbatch == 1 and N >= 4096 and M <= 64 and K <= 64 and c_ms == 1:
        # few rows, many columns, output contiguous along the rows: the transposed problem
        # C^T = B^T A^T is the library's row-streaming form (gemm_tall_kernel)
        A, B = B, A
        M, N = N, M
        a_ms, a_ks, b_ks, b_ns = b_ns, b_ks, a_ks, a_ms
        c_ms, c_ns = c_ns, c_ms
    c3 = ctypes.c_int64 * 3
    bshape = c3(*([shape[d] for d in Bd] + [1] * (3 - len(Bd))))
    a_bs = c3(*([sa[d] for d in Bd] + [0] * (3 - len(Bd))))
    b_bs = c3(*([sb[d] for d in Bd] + [0] * (3 - len(Bd))))
    c_bs = c3(*([so[d] for d in Bd] + [0] * (3 - len(Bd))))
    ws = _workspace(rt)
    rt.sync_stream()
    rt.check(rt.lib.vmp_gemm_strided(
        rt.ctx, len(Bd), bshape, M, N, K, ctypes.c_void_p(A.t.data_ptr()), a_bs, a_ms, a_ks,
        ctypes.c_void_p(B.t.data_ptr()), b_bs, b_ks, b_ns, ctypes.c_void_p(out.t.data_ptr()),
        c_bs, c_ms, c_ns, float(scale), ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8))
    return True


def _hoist_invariant(arrays, shape, reduce_axes, out_shape_keep, scale):
    """sum_r a(r, k) b(k) = b(k) sum_r a(r, k): operands that do not vary along any reduced
    axis leave the (long) reduction and multiply its (small) result instead.  None if nothing
    can be hoisted or the reduction is too short to matter."""
    nd = len(shape)
    nred = 1
    for ax in reduce_axes:
        nred *= int(shape[ax])
    if nred < 1024:
        return None

    def varies(a):
        off = nd - a.ndim
        return any(ax - off >= 0 and a.shape[ax - off] != 1 for ax in reduce_axes)

    var = [a for a in arrays if varies(a)]
    inv = [a for a in arrays if not varies(a)]
    if not inv or not var:
        return None
    shape_var = [1] * nd
    for a in var:
        off = nd - a.ndim
        for d in range(a.ndim):
            if a.shape[d] != 1:
                shape_var[off + d] = a.shape[d]
    keep_var = tuple(1 if ax in reduce_axes else shape_var[ax] for ax in range(nd))
    partial = _launch_sum_multiply(var, tuple(shape_var), reduce_axes, keep_var, scale)
    out = fuse(lambda *xs: functools.reduce(operator.mul, xs), partial, *inv)
    if out.shape != tuple(out_shape_keep):
        return out.reshape(tuple(out_shape_keep)) if out.size == int(np.prod(out_shape_keep)) \
            else None
    return out


def _launch_sum_multiply(arrays, shape, reduce_axes, out_shape_keep, scale=1.0):
    """out (shape with size-1 on reduced axes) = scale * sum_{reduce_axes} prod arrays."""
    rt = get_runtime()
    nd = len(shape)
    if nd > 8 or len(arrays) > 6:
        raise NotImplementedError('sum_multiply supports <= 8 axes and <= 6 operands')
    if len(arrays) == 1 and scale == 1.0 and all(shape[ax] == 1 for ax in reduce_axes) \
            and tuple(arrays[0].shape) == tuple(out_shape_keep):
        # nothing to multiply, nothing to sum: the operand IS the result (device arrays are never
        # modified in place; a plate "sum" of a message that already has the parent's plates was
        # a 0.3 ms copy of a (D, N) array at N = 1e6; "sums" over axes of extent one -- scalar terms
        # of the bound on their way through the plate sums -- were a record each)
        return arrays[0]
    memo, sig = _CUR_MEMO[0], None
    if memo is not None and len(reduce_axes) > 0 and any(a.size >= _MEMO_MIN for a in arrays):
        # the same reduction of the same (immutable) arrays, asked for again within one sweep --
        # sum_n y_dn x_nk for the Dot message to W and for sum y <f> of the message to tau;
        # sum_n r_nk y_nd for the messages of a mixture to its means and to its precisions.  The key
        # is the memory the operands occupy and how the index space walks it (unit axes dropped:
        # (N, 1, D, 1) and (N, 1, 1, D) views of one array under the same sum are the same numbers)
        sig = _canonical_signature(arrays, shape, reduce_axes, scale)
        hit = memo.get(sig)
        if hit is not None:
            return hit[0].reshape(tuple(out_shape_keep))
    rt.note_reads(arrays)
    if len(arrays) >= 2 and len(reduce_axes) > 0:
        hoisted = _hoist_invariant(arrays, shape, reduce_axes, out_shape_keep, scale)
        if hoisted is not None:
            return _remember(memo, sig, hoisted, arrays)
    out = DArray.empty(out_shape_keep)
    # a small contraction inside an operation joins the queue of small operations (one interpreter
    # launch for a run of them) instead of being a GEMM launch of its own
    small = rt.queue_collects_sums() and int(np.prod(out_shape_keep)) <= 2048 \
        and int(np.prod(shape)) <= 32768
    if not small and len(arrays) >= 2 and len(reduce_axes) > 0 \
            and _try_gemm(rt, arrays, shape, reduce_axes, out, scale):
        return _remember(memo, sig, out, arrays)
    # coalesce neighbouring axes of the same role (both kept or both reduced) that every
    # operand and the output walk densely: the kernels decode a flat index into axes with
    # 64-bit divisions, so fewer axes is directly fewer instructions per element
    in_str = [_strides(a.t, shape) for a in arrays]
    ostr = _strides(out.t, shape)
    red = set(reduce_axes)
    m_shape, m_red, m_in, m_out = [], [], [[] for _ in arrays], []
    for ax in range(nd):
        if shape[ax] == 1:
            continue
        if m_shape and (m_red[-1] == (ax in red)):
            ext = shape[ax]
            lists = m_in + ([m_out] if ax not in red else [])
            cur = in_str + ([ostr] if ax not in red else [])
            if all(l[-1] == c[ax] * ext for l, c in zip(lists, cur)):
                m_shape[-1] *= ext
                for l, c in zip(lists, cur):
                    l[-1] = c[ax]
                if ax in red:
                    m_out[-1] = 0
                continue
        m_shape.append(shape[ax])
        m_red.append(ax in red)
        for l, c in zip(m_in, in_str):
            l.append(c[ax])
        m_out.append(0 if ax in red else ostr[ax])
    nd = len(m_shape)
    mask = 0
    for i, r in enumerate(m_red):
        if r:
            mask |= 1 << i
    c_shape = (ctypes.c_int64 * max(nd, 1))(*m_shape)
    c_in = (ctypes.c_void_p * len(arrays))(*[a.t.data_ptr() for a in arrays])
    flat = []
    for l in m_in:
        flat += l
    c_str = (ctypes.c_int64 * max(len(flat), 1))(*flat)
    c_ostr = (ctypes.c_int64 * max(nd, 1))(*m_out)
    ws = _workspace(rt)
    rt.sync_stream()
    rt.check(rt.lib.vmp_sum_multiply(
        rt.ctx, nd, c_shape, len(arrays), c_in, c_str, c_ostr, ctypes.c_uint32(mask),
        float(scale), ctypes.c_void_p(out.t.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
        ws.numel() * 8))
    rt.keep_until_flush(arrays, out, kind='sm')
    return _remember(memo, sig, out, arrays)


def _canonical_signature(arrays, shape, reduce_axes, scale):
    nd = len(shape)
    red = set(reduce_axes)
    strides = [_strides(a.t, shape) for a in arrays]
    # kept axes in their order (it is the layout of the result), then the summed ones in an order
    # of their own (extent, strides): sum over (k, e, n) of one pair of arrays = sum over (n, k, d)
    axes = [ax for ax in range(nd) if shape[ax] != 1 and ax not in red] + \
        sorted((ax for ax in range(nd) if shape[ax] != 1 and ax in red),
               key=lambda ax: (int(shape[ax]), tuple(st[ax] for st in strides)))
    ops = tuple((a.t.data_ptr(), tuple(st[ax] for ax in axes)) for a, st in zip(arrays, strides))
    if len(ops) == 2 and ops[1] < ops[0]:
        ops = (ops[1], ops[0])             # two factors commute exactly
    return (ops, tuple((int(shape[ax]), ax in red) for ax in axes), float(scale))


def _remember(memo, sig, out, arrays):
    if sig is None:
        return out

    def held(entry):
        return sum(int(t.size) * 8 for t in [entry[0]] + list(entry[1]))
    # bounded by entries AND by the bytes the entries keep alive (operands + results): a loop of
    # single node.update() calls never reaches the end-of-sweep clear of the plan
    while len(memo) >= 32 or (memo and sum(held(e) for e in memo.values()) > _MEMO_BYTES):
        memo.pop(next(iter(memo)))
    memo[sig] = (out, list(arrays))        # the operands stay alive: their addresses stay theirs
    return out


def sum_multiply(*args, axis=None, sumaxis=True, keepdims=False):
    """
    ``sum(arg[0]*arg[1]*..., axis=...)`` without forming the product
    (reference: utils/misc.py:851-933).  ``axis`` lists the axes to sum
    (``sumaxis=True``; None = all) or to keep (``sumaxis=False``; None = all).
    """
    if len(args) == 0:
        raise ValueError("You must give at least one input array")
    scale = 1.0
    arrays = []
    for a in args:
        if is_scalar(a):
            scale *= float(a)
        else:
            arrays.append(asdarray(a))
    if not arrays:
        return DArray.from_host(np.float64(scale))
    max_dim = max(a.ndim for a in arrays)
    if sumaxis:
        if axis is None:
            keep = []
        else:
            if np.isscalar(axis):
                axis = [axis]
            keep = [i for i in range(max_dim) if i not in axis and (i - max_dim) not in axis]
    else:
        if axis is None:
            keep = list(range(max_dim))
        else:
            if np.isscalar(axis):
                axis = [axis]
            keep = sorted(i if i >= 0 else i + max_dim for i in axis)
    if len(keep) > 0 and (min(keep) < 0 or max(keep) >= max_dim):
        raise ValueError("Axis index out of bounds")
    shape = broadcasted_shape(*[a.shape for a in arrays])
    red = [i for i in range(max_dim) if i not in keep]
    keep_shape = tuple(shape[i] if i in keep else 1 for i in range(max_dim))
    out = _launch_sum_multiply(arrays, shape, red, keep_shape, scale)
    if not keepdims:
        out = out.reshape(tuple(shape[i] for i in keep))
    return out


def sum_product(*args, axes_to_keep=None, axes_to_sum=None, keepdims=False):
    if axes_to_keep is not None:
        return sum_multiply(*args, axis=axes_to_keep, sumaxis=False, keepdims=keepdims)
    return sum_multiply(*args, axis=axes_to_sum, sumaxis=True, keepdims=keepdims)


def sum_multiply_to_plates(*arrays, to_plates=(), from_plates=None, ndim=0):
    """
    Product of the arguments summed to the plates ``to_plates`` (their trailing
    ``ndim`` axes are variable axes and are kept), times the integer plate
    multiplier for axes that are broadcast in every operand
    (reference: utils/misc.py:805-844 -- the plate sum of ``node.py:650``).
    """
    arrays = [asdarray(a) for a in arrays]

    def plates_of(s):
        return tuple(s) if ndim == 0 else tuple(s[:len(s) - ndim])

    plate_shapes = [plates_of(a.shape) for a in arrays]
    product_plates = broadcasted_shape(*plate_shapes)
    if from_plates is None:
        r = 1
    else:
        r = broadcasting_multiplier(tuple(from_plates), product_plates, tuple(to_plates))
    full = broadcasted_shape(*[a.shape for a in arrays])
    npl = len(full) - ndim
    to = (1,) * (npl - len(to_plates)) + tuple(to_plates) if npl >= len(to_plates) else None
    if to is None:
        # the target has more plate axes than the product: nothing to sum on those
        to = tuple(to_plates)[len(to_plates) - npl:]
    red = [i for i in range(npl) if full[i] != 1 and to[i] == 1]
    keep_shape = tuple(1 if i in red else full[i] for i in range(len(full)))
    out = _launch_sum_multiply(arrays, full, red, keep_shape, float(r))
    want = len(to_plates) + ndim
    s = out.shape
    while len(s) > want and s[0] == 1:
        s = s[1:]
    if len(s) > want:
        raise ValueError('cannot squeeze %s to %d axes' % (out.shape, want))
    return out.reshape(s)


def add_trailing_axes(x, n):
    if is_scalar(x):
        return x
    x = asdarray(x)
    return x.reshape(x.shape + (1,) * n)


def add_leading_axes(x, n):
    if is_scalar(x):
        return x
    x = asdarray(x)
    return x.reshape((1,) * n + x.shape)


def logsumexp(X, axis=-1, keepdims=False):
    """Stable log-sum-exp over the LAST axis (utils/misc.py:1366-1385)."""
    if axis not in (-1, asdarray(X).ndim - 1):
        raise NotImplementedError('device logsumexp reduces the last axis')
    _, lse = normalized_exp(X)
    return lse if keepdims else lse.reshape(lse.shape[:-1])


def normalized_exp(phi):
    """(p, logsum_p): exp(phi) normalised over the last axis with the reference's
    second renormalisation (utils/misc.py:1388-1401) -- ``vmp_softmax_moments``."""
    rt = get_runtime()
    phi = contiguous(asdarray(phi))
    K = phi.shape[-1]
    rows = phi.size // K if K else 0
    p = DArray.empty(phi.shape)
    lse = DArray.empty(phi.shape[:-1] + (1,))
    rt.sync_stream()
    rt.note_reads([phi])
    rt.check(rt.lib.vmp_softmax_moments(rt.ctx, rows, K, ctypes.c_void_p(phi.t.data_ptr()),
                                        ctypes.c_void_p(p.t.data_ptr()),
                                        ctypes.c_void_p(lse.t.data_ptr())))
    return p, lse


_HALF_ARANGE = {}


def _half_arange(d):
    """0.5 * arange(d) on the device, uploaded once per runtime."""
    key = (id(get_runtime()), d)
    if key not in _HALF_ARANGE:
        _HALF_ARANGE[key] = DArray.from_host(0.5 * np.arange(d))
    return _HALF_ARANGE[key]


def multidigamma(a, d):
    """sum_{i<d} digamma(a - i/2)  (utils/misc.py:1146-1151)."""
    a = asdarray(a)
    half = _half_arange(int(d))
    terms = fuse(lambda x, h: _digamma(x - h), a.reshape(a.shape + (1,)), half)
    return sum_multiply(terms, axis=-1)


def diag(X, ndim=1):
    """Embed the trailing ``ndim`` axes as the diagonal of ``2*ndim`` axes
    (utils/misc.py:1253-1262)."""
    X = asdarray(X)
    if ndim == 0:
        return X
    sh = X.shape[len(X.shape) - ndim:]
    n = int(np.prod(sh)) if sh else 1
    key = (id(get_runtime()), 'eye', tuple(sh))
    if key not in _HALF_ARANGE:
        _HALF_ARANGE[key] = DArray.from_host(np.eye(n).reshape(sh + sh))     # uploaded once
    return fuse(lambda x, e: x * e, X.reshape(X.shape + (1,) * ndim), _HALF_ARANGE[key])


def get_diag(X, ndim=1):
    """Diagonal of the trailing ``2*ndim`` axes (utils/misc.py:1207-1250) -- a strided VIEW of
    the array (stride sum of the paired axes), no kernel launch."""
    X = asdarray(X)
    if ndim == 0:
        return X
    sh = X.shape[len(X.shape) - ndim:]
    lead = X.shape[:len(X.shape) - 2 * ndim]
    t = X.t
    nl = len(lead)
    strides = list(t.stride())
    dstr = [strides[nl + i] + strides[nl + ndim + i] for i in range(ndim)]
    view = t.as_strided(tuple(lead) + tuple(sh), strides[:nl] + dstr, t.storage_offset())
    return DArray(view)


def onehot(labels, K):
    """One-hot fixed moments of a categorical variable: integer indexing, bit-exact
    (categorical.py:30-46).  ``labels``: host integer array (any shape) -> (..., K)."""
    rt = get_runtime()
    lab = np.ascontiguousarray(np.asarray(labels))
    if not np.issubdtype(lab.dtype, np.integer):
        raise ValueError("Class indices must be integers")
    lab = lab.astype(np.int64)
    n = int(lab.size)
    dl = rt.torch.from_numpy(lab.reshape(-1)).to(rt.device)
    out = DArray.empty(lab.shape + (K,))
    info = rt.torch.zeros(1, dtype=rt.torch.int32, device=rt.device)
    rt.sync_stream()
    rt.check(rt.lib.vmp_onehot_i64(rt.ctx, n, K, ctypes.c_void_p(dl.data_ptr()),
                                   ctypes.c_void_p(out.t.data_ptr()),
                                   ctypes.c_void_p(info.data_ptr())))
    rt.host_access('onehot')
    if int(info.item()) != 0:
        raise ValueError("Class indices out of range [0, %d)" % K)
    return out


def contract(operands, labels, out_labels, sizes, scale=1.0, compress=()):
    """
    Labeled broadcast contraction (one ``vmp_sum_multiply`` launch):

        out[out_labels] = scale * sum_{labels not in out_labels} prod_i operands[i][labels[i]]

    ``labels[i]`` names every axis of ``operands[i]`` (an axis of extent 1 is a
    broadcast axis whatever its label); ``sizes`` maps label -> extent.  This is
    the einsum of ``SumMultiply`` (dot.py:355, :403, :581) on device arrays.  Output labels
    listed in ``compress`` keep extent 1 when no operand varies along them (plate axes of
    broadcast-compressed moments, node.py:311-345).
    """
    ops = [asdarray(a) for a in operands]
    all_labels = list(out_labels)
    for ls in labels:
        for lab in ls:
            if lab not in all_labels:
                all_labels.append(lab)
    nd = len(all_labels)
    # a summed label along which EVERY operand is unit contributes no implicit factor
    # (plate multipliers are explicit, via `scale`)
    varying = set()
    for a, ls in zip(ops, labels):
        for ax, lab in enumerate(ls):
            if a.shape[ax] != 1:
                varying.add(lab)
    shape = tuple(int(sizes[lab]) if ((lab in out_labels and lab not in compress)
                                      or lab in varying) else 1
                  for lab in all_labels)
    views = []
    for a, ls in zip(ops, labels):
        if a.ndim != len(ls):
            raise ValueError('operand with %d axes given %d labels' % (a.ndim, len(ls)))
        # a strided view of `a` laid out along all_labels (extent 1 / stride 0 where absent)
        st, sh = [], []
        for lab in all_labels:
            if lab in ls:
                ax = list(ls).index(lab)
                if a.shape[ax] == 1:
                    st.append(0)
                    sh.append(1)
                else:
                    if a.shape[ax] != sizes[lab]:
                        raise ValueError('axis %r has extent %d, expected %d'
                                         % (lab, a.shape[ax], sizes[lab]))
                    st.append(a.t.stride(ax))
                    sh.append(a.shape[ax])
            else:
                st.append(0)
                sh.append(1)
        views.append(DArray(a.t.as_strided(sh, st, a.t.storage_offset())))
    red = [i for i, lab in enumerate(all_labels) if lab not in out_labels]
    keep_shape = tuple(1 if i in red else shape[i] for i in range(nd))
    out = _launch_sum_multiply(views, shape, red, keep_shape, scale)
    out = out.reshape(tuple(shape[i] for i in range(len(out_labels))))
    return out


# the memo of the plan whose operation is running (plans/generic.py sets and clears it); None: off
_CUR_MEMO = [None]
_MEMO_BYTES = 512 << 20          # operands and results a memo may keep alive
_MEMO_MIN = int(os.environ.get('BAYESPY_AMD_MEMO_MIN', 1 << 16))    # reductions with an operand of at least this many elements are remembered


def plan_contraction(varying, out_labels, sizes):
    """The pairwise order :func:`contract_path` uses, from shapes alone: ``varying[i]`` = the labels
    operand i really varies along.  Returns steps ``(i, j, result_labels)`` over a growing operand
    list (each step appends its result and retires i and j; indices refer to the list as it stands
    when the step runs, results appended at the end) until at most two operands are left.  Greedy:
    smallest result first, ties by the cheaper product, then by position -- deterministic."""
    order = list(out_labels)
    for ls in varying:
        for lab in ls:
            if lab not in order:
                order.append(lab)

    def extent(labs):
        n = 1
        for lab in labs:
            n *= int(sizes[lab])
        return n
    live = [list(v) for v in varying]
    steps = []
    while len(live) > 2:
        best = None
        for i in range(len(live)):
            for j in range(i + 1, len(live)):
                others = set(out_labels)
                for q, v in enumerate(live):
                    if q != i and q != j:
                        others.update(v)
                union = [lab for lab in order if lab in live[i] or lab in live[j]]
                res = [lab for lab in union if lab in others]
                cand = (extent(res), extent(union), i, j, res)
                if best is None or cand[:4] < best[:4]:
                    best = cand
        _, _, i, j, res = best
        steps.append((i, j, res))
        live = [v for q, v in enumerate(live) if q != i and q != j] + [res]
    return steps


def contract_path(operands, labels, out_labels, sizes, scale=1.0):
    """The contraction of :func:`contract` evaluated pair by pair (the reference's einsum runs
    ``optimize=False``: one loop nest over every label, utils/misc.py:906).  Greedy
    (:func:`plan_contraction`): the pair whose result is smallest goes first (ties: the cheaper
    product); a label leaves a pair's result as soon as no other operand and not the output carries
    it.  E.g. sum_dn y_dn w_dk x_nk: (y, x) -> (d, k) is a GEMM over n, then (T, w) -> a number --
    never the (d, n) product."""
    ops = [(asdarray(a), list(ls)) for a, ls in zip(operands, labels)]
    varying = [[lab for ax, lab in enumerate(ls) if a.shape[ax] != 1] for a, ls in ops]
    for i, j, res in plan_contraction(varying, out_labels, sizes):
        t = contract([ops[i][0], ops[j][0]], [ops[i][1], ops[j][1]], res, sizes)
        ops = [o for q, o in enumerate(ops) if q != i and q != j] + [(t, res)]
    return contract([a for a, _ in ops], [ls for _, ls in ops], out_labels, sizes, scale=scale)


# ---------------------------------------------------------------------------
# plate re-indexing (take / put_simple / concatenate, utils/misc.py:549-585 and the
# np.take / np.concatenate call sites take.py:72-94, concatenate.py:130-167)
# ---------------------------------------------------------------------------
class IndexMap:
    """A constant integer index array along one axis of length ``length``, resident on the
    device together with its inverse (CSR: for every target row the source positions, in
    increasing order) so that the accumulation of ``put_simple`` has a fixed order."""

    def __init__(self, indices, length):
        idx = np.asarray(indices)
        if not np.issubdtype(idx.dtype, np.integer):
            raise ValueError("Indices must be integers")
        idx = idx.astype(np.int64)
        if np.any(idx < -length) or np.any(idx >= length):
            raise ValueError("Index out of bounds")
        flat = np.where(idx < 0, idx + length, idx).reshape(-1)
        self.shape = tuple(idx.shape)
        self.length = int(length)
        self.n = int(flat.size)
        ptr = np.zeros(length + 1, dtype=np.int64)
        np.add.at(ptr, flat + 1, 1)
        rt = get_runtime()
        up = lambda a: rt.torch.from_numpy(np.ascontiguousarray(a)).to(rt.device)
        self.idx = up(flat)
        self.ptr = up(np.cumsum(ptr))
        self.perm = up(np.argsort(flat, kind='stable').astype(np.int64))


def _split3(shape, axis, nax=1):
    """(outer, inner) sizes around the ``nax`` axes starting at negative ``axis``."""
    nd = len(shape)
    a = nd + axis
    outer = int(np.prod(shape[:a], dtype=np.int64))
    inner = int(np.prod(shape[a + nax:], dtype=np.int64))
    return a, outer, inner


def take(x, imap, axis=-1):
    """``np.take(x, indices, axis)`` for a negative ``axis``; an index array with several
    axes creates as many axes in the result."""
    x = contiguous(asdarray(x))
    if axis >= 0 or -axis > x.ndim:
        raise ValueError("axis must be a negative index into the array")
    if x.shape[axis] != imap.length:
        raise ValueError("axis has length %d, the index map expects %d"
                         % (x.shape[axis], imap.length))
    a, outer, inner = _split3(x.shape, axis)
    out = DArray.empty(x.shape[:a] + imap.shape + x.shape[a + 1:])
    rt = get_runtime()
    rt.sync_stream()
    rt.note_reads([x])
    rt.check(rt.lib.vmp_take_axis(rt.ctx, outer, imap.length, inner,
                                  ctypes.c_void_p(x.t.data_ptr()), imap.n,
                                  ctypes.c_void_p(imap.idx.data_ptr()),
                                  ctypes.c_void_p(out.t.data_ptr()), imap.n, 0))
    return out


def put_simple(y, imap, axis=-1):
    """Accumulating inverse of :func:`take`: the ``len(imap.shape)`` axes of ``y`` that end at
    negative ``axis`` collapse into one axis of length ``imap.length``; entries with the same
    index are added (utils/misc.py:549-585)."""
    y = asdarray(y)
    nax = len(imap.shape)
    if axis >= 0:
        raise ValueError("Axis index must be negative")
    first = axis - nax + 1                  # negative position of the first index axis
    need = -first
    if y.ndim < need:
        y = y.reshape((1,) * (need - y.ndim) + y.shape)
    a = y.ndim + first
    want = y.shape[:a] + imap.shape + y.shape[a + nax:]
    if y.shape != want:
        y = y.broadcast_to(want)            # broadcast index axes are real terms of the sum
    y = contiguous(y)
    _, outer, inner = _split3(y.shape, first, nax)
    out = DArray.empty(y.shape[:a] + (imap.length,) + y.shape[a + nax:])
    rt = get_runtime()
    rt.sync_stream()
    rt.note_reads([y])
    rt.check(rt.lib.vmp_segment_sum_axis(rt.ctx, outer, imap.n, inner,
                                         ctypes.c_void_p(y.t.data_ptr()), imap.length,
                                         ctypes.c_void_p(imap.ptr.data_ptr()),
                                         ctypes.c_void_p(imap.perm.data_ptr()),
                                         ctypes.c_void_p(out.t.data_ptr())))
    return out


def concatenate(arrays, axis=-1):
    """``np.concatenate`` along a negative ``axis`` with broadcasting of all other axes (the
    explicit broadcast of concatenate.py:140-163)."""
    arrays = [asdarray(a) for a in arrays]
    if axis >= 0:
        raise ValueError("Currently, only negative axis indeces are allowed.")
    nd = max(max(a.ndim for a in arrays), -axis)
    arrays = [a.reshape((1,) * (nd - a.ndim) + a.shape) for a in arrays]
    ax = nd + axis
    others = [tuple(1 if i == ax else s for i, s in enumerate(a.shape)) for a in arrays]
    common = broadcasted_shape(*others)
    lengths = [a.shape[ax] for a in arrays]
    total = int(sum(lengths))
    out = DArray.empty(common[:ax] + (total,) + common[ax + 1:])
    _, outer, inner = _split3(out.shape, axis)
    rt = get_runtime()
    rt.sync_stream()
    off = 0
    for a, n in zip(arrays, lengths):
        src = contiguous(a.broadcast_to(common[:ax] + (n,) + common[ax + 1:]))
        rt.note_reads([src])
        rt.check(rt.lib.vmp_take_axis(rt.ctx, outer, n, inner, ctypes.c_void_p(src.t.data_ptr()),
                                      n, None, ctypes.c_void_p(out.t.data_ptr()), total, off))
        off += n
    return out


def moveaxis(x, src, dst):
    """Stride-only view with one axis moved (utils/misc.py:947-960)."""
    x = asdarray(x)
    return DArray(x.t.movedim(src, dst))
