"""
TEST DOUBLE of the GENERIC part of the C ABI (include/vmp_hip.h: vmp_ewise, vmp_sum_multiply,
vmp_gemm_strided, vmp_spd_batched, vmp_gaussian_moments, vmp_softmax_moments, vmp_onehot_i64,
vmp_take_axis, vmp_segment_sum_axis, vmp_block_banded_solve, vmp_gaussian_shared_update, the queue
entry points as no-ops) on host memory with NumPy.

It lets the host logic of the generic engine (bayespy_amd/inference/plans/generic.py: message
routing, lazily evaluated sums and contractions, contraction planning, the carried plate sums)
run in CPU-only tests against the live-reference golden vectors.  It lives under tests/ and is
never imported by the product; the product path has no CPU fallback.

    from host_generic import install
    install()            # a CPU Runtime whose `lib` is this double becomes the process runtime
"""
import ctypes

import numpy as np
from scipy import special


def _v(x):
    return x.value if hasattr(x, 'value') else x


def _ptr(x):
    p = _v(x)
    return 0 if p is None else int(p)


def _view(ptr, shape, strides, dtype=np.float64):
    """ndarray over raw memory with ELEMENT strides (0 = broadcast)."""
    shape = tuple(int(s) for s in shape)
    item = np.dtype(dtype).itemsize
    if any(s == 0 for s in shape):
        return np.zeros(shape, dtype=dtype)
    span = 1 + sum((s - 1) * abs(int(st)) for s, st in zip(shape, strides))
    buf = (ctypes.c_char * (span * item)).from_address(_ptr(ptr))
    base = np.frombuffer(buf, dtype=dtype)
    return np.lib.stride_tricks.as_strided(base, shape, tuple(int(st) * item for st in strides))


def _dense(ptr, shape, dtype=np.float64):
    shape = tuple(int(s) for s in shape)
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(s, 1)
    return _view(ptr, shape, tuple(reversed(st)), dtype)


(OP_IN, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_LOG, OP_EXP, OP_SQR, OP_SQRT,
 OP_RECIP, OP_DIGAMMA, OP_LGAMMA, OP_MAX, OP_MIN, OP_WHERE_NZ, OP_DUP, OP_SWAP,
 OP_TRIGAMMA) = range(20)


class HostGenericLib:
    """The method set ``Runtime.lib`` needs for the generic engine."""

    def __init__(self):
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # -- housekeeping -------------------------------------------------------------------------------
    def vmp_last_error(self, ctx):
        return b'host double'

    def vmp_tune_set(self, key, value):
        return 0

    def vmp_queue_begin(self, ctx):
        return 0

    vmp_queue_end = vmp_queue_flush = vmp_queue_commit = vmp_queue_begin

    def vmp_ctx_set_stream(self, ctx, s):
        return 0

    def vmp_sum_multiply_workspace_bytes(self):
        return 1024

    # -- formulas -------------------------------------------------------------------------------------
    def vmp_ewise(self, ctx, nd, shape, nin, ins, in_strides, nops, ops, nconsts, consts, out):
        self._count('vmp_ewise')
        shape = [int(shape[i]) for i in range(nd)]
        arrs = [_view(ins[i], shape, [in_strides[i * nd + d] for d in range(nd)])
                for i in range(nin)]
        stack = []
        with np.errstate(all='ignore'):
            for q in range(nops):
                code = int(ops[q])
                op, arg = code & 0xff, code >> 8
                if op == OP_IN:
                    stack.append(arrs[arg])
                elif op == OP_CONST:
                    stack.append(np.float64(consts[arg]))
                elif op == OP_DUP:
                    stack.append(stack[-1])
                elif op == OP_SWAP:
                    stack[-1], stack[-2] = stack[-2], stack[-1]
                elif op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MAX, OP_MIN, OP_WHERE_NZ):
                    b = stack.pop()
                    a = stack.pop()
                    if op == OP_ADD:
                        r = a + b
                    elif op == OP_SUB:
                        r = a - b
                    elif op == OP_MUL:
                        r = a * b
                    elif op == OP_DIV:
                        r = a / b
                    elif op == OP_MAX:
                        r = np.maximum(a, b)
                    elif op == OP_MIN:
                        r = np.minimum(a, b)
                    else:
                        r = np.where(np.asarray(a) != 0, b, 0.0)
                    stack.append(r)
                else:
                    a = stack.pop()
                    if op == OP_NEG:
                        r = -a
                    elif op == OP_LOG:
                        r = np.log(a)
                    elif op == OP_EXP:
                        r = np.exp(a)
                    elif op == OP_SQR:
                        r = a * a
                    elif op == OP_SQRT:
                        r = np.sqrt(a)
                    elif op == OP_RECIP:
                        r = 1.0 / a
                    elif op == OP_DIGAMMA:
                        r = special.digamma(a)
                    elif op == OP_LGAMMA:
                        r = special.gammaln(a)
                    elif op == OP_TRIGAMMA:
                        r = special.polygamma(1, a)
                    else:
                        raise ValueError('opcode %d' % op)
                    stack.append(r)
        o = _dense(out, shape)
        o[...] = np.broadcast_to(stack[-1], tuple(shape))
        return 0

    # -- contractions -----------------------------------------------------------------------------
    def vmp_sum_multiply(self, ctx, nd, shape, nin, ins, in_strides, out_strides, mask, scale, out,
                         ws, ws_bytes):
        self._count('vmp_sum_multiply')
        shape = [int(shape[i]) for i in range(nd)]
        mask = int(_v(mask))
        arrs = [_view(ins[i], shape, [in_strides[i * nd + d] for d in range(nd)])
                for i in range(nin)]
        letters = 'abcdefgh'[:nd]
        keep = ''.join(c for i, c in enumerate(letters) if not (mask >> i) & 1)
        res = np.einsum(','.join([letters] * nin) + '->' + keep, *arrs, optimize=True) \
            if nd else np.prod([a for a in arrs])
        oshape = [s for i, s in enumerate(shape) if not (mask >> i) & 1]
        ostr = [int(out_strides[i]) for i in range(nd) if not (mask >> i) & 1]
        o = _view(out, oshape, ostr)
        o[...] = float(_v(scale)) * res
        return 0

    def vmp_gemm_strided(self, ctx, nb, bshape, M, N, K, A, a_bs, a_ms, a_ks, B, b_bs, b_ks, b_ns, C,
                         c_bs, c_ms, c_ns, scale, ws, ws_bytes):
        self._count('vmp_gemm_strided')
        bsh = [int(bshape[i]) for i in range(3)]
        a = _view(A, bsh + [M, K], [a_bs[0], a_bs[1], a_bs[2], a_ms, a_ks])
        b = _view(B, bsh + [K, N], [b_bs[0], b_bs[1], b_bs[2], b_ks, b_ns])
        c = _view(C, bsh + [M, N], [c_bs[0], c_bs[1], c_bs[2], c_ms, c_ns])
        c[...] = float(_v(scale)) * np.matmul(a, b)
        return 0

    # -- linear algebra -------------------------------------------------------------------------------
    def vmp_spd_batched(self, ctx, n, batch, A, Ainv, logdet, info):
        self._count('vmp_spd_batched')
        n, batch = int(n), int(batch)
        if batch == 0:
            return 0
        a = _dense(A, (batch, n, n))
        flag = _dense(info, (batch,), np.int32)
        flag[...] = 0
        inv = np.empty_like(a)
        ld = np.empty(batch)
        for b in range(batch):
            try:
                L = np.linalg.cholesky(a[b])
                Li = np.linalg.inv(L)
                inv[b] = Li.T @ Li
                ld[b] = 2.0 * np.sum(np.log(np.diag(L)))
            except np.linalg.LinAlgError:
                flag[b] = 1
                inv[b] = np.nan
                ld[b] = np.nan
        if _ptr(Ainv):
            _dense(Ainv, (batch, n, n))[...] = inv
        if _ptr(logdet):
            _dense(logdet, (batch,))[...] = ld
        return 0

    def vmp_gaussian_moments(self, ctx, n, batch, phi0, phi1, u0, u1, g, info):
        self._count('vmp_gaussian_moments')
        n, batch = int(n), int(batch)
        p0, p1 = _dense(phi0, (batch, n)), _dense(phi1, (batch, n, n))
        flag = _dense(info, (batch,), np.int32)
        flag[...] = 0
        U0, U1, G = _dense(u0, (batch, n)), _dense(u1, (batch, n, n)), _dense(g, (batch,))
        for b in range(batch):
            try:
                L = np.linalg.cholesky(-2.0 * p1[b])
            except np.linalg.LinAlgError:
                flag[b] = 1
                continue
            Li = np.linalg.inv(L)
            cov = Li.T @ Li
            x = cov @ p0[b]
            U0[b], U1[b] = x, np.outer(x, x) + cov
            G[b] = -0.5 * x @ p0[b] + np.sum(np.log(np.diag(L)))
        return 0

    def vmp_block_banded_solve(self, ctx, T, K, nm, ny, A, B, y, V, C, x, ldet, info):
        """utils/linalg.py:468-575 restated: forward Cholesky sweep of the block-tridiagonal
        matrix, backward recursion for the blocks of the inverse and the solution."""
        self._count('vmp_block_banded_solve')
        T, K, nm, ny = int(T), int(K), int(nm), int(ny)
        a = _dense(A, (nm, T, K, K))
        b = _dense(B, (nm, max(T - 1, 1), K, K))
        yy = _dense(y, (ny, T, K))
        Vo = _dense(V, (nm, T, K, K))
        Co = _dense(C, (nm, max(T - 1, 0), K, K))
        xo = _dense(x, (ny, T, K))
        ld = _dense(ldet, (nm,))
        flag = _dense(info, (max(nm, 1),), np.int32)
        flag[...] = 0
        per = ny // nm if nm else 0
        for m in range(nm):
            n = T * K
            full = np.zeros((n, n))
            for t in range(T):
                full[t * K:(t + 1) * K, t * K:(t + 1) * K] = a[m, t]
                if t + 1 < T:
                    full[t * K:(t + 1) * K, (t + 1) * K:(t + 2) * K] = b[m, t]
                    full[(t + 1) * K:(t + 2) * K, t * K:(t + 1) * K] = b[m, t].T
            try:
                L = np.linalg.cholesky(full)
            except np.linalg.LinAlgError:
                flag[m] = 1
                continue
            inv = np.linalg.inv(full)
            ld[m] = 2.0 * np.sum(np.log(np.diag(L)))
            for t in range(T):
                Vo[m, t] = inv[t * K:(t + 1) * K, t * K:(t + 1) * K]
                if t + 1 < T:
                    Co[m, t] = inv[t * K:(t + 1) * K, (t + 1) * K:(t + 2) * K]
            rows = range(m * per, (m + 1) * per) if nm > 1 else range(ny)
            for r in rows:
                xo[r] = (inv @ yy[r].reshape(-1)).reshape(T, K)
        return 0

    def vmp_gaussian_shared_update_workspace_bytes(self, D, K):
        return 64

    def vmp_gaussian_shared_update(self, ctx, N, K, D, Y, y_sd, y_sn, B, b_sd, b_sk, m0, m0_sn,
                                   m0_sk, p0, cov, x, x_sn, x_sk, stats, ws, ws_bytes):
        """include/vmp_hip.h: x_n = Cov (p0 + m_n), m_n = B^T y_n (or the given rows), and the plate
        sums [sum x; sum x x^T; sum y x^T]."""
        self._count('vmp_gaussian_shared_update')
        N, K, D = int(N), int(K), int(D)
        c = _dense(cov, (K, K))
        if _ptr(Y):
            yv = _view(Y, (D, N), (y_sd, y_sn))
            bv = _view(B, (D, K), (b_sd, b_sk))
            m = yv.T @ bv
        else:
            yv = None
            m = _view(m0, (N, K), (m0_sn, m0_sk))
        if _ptr(p0):
            m = m + _dense(p0, (K,))[None, :]
        xs = m @ c.T
        _view(x, (N, K), (x_sn, x_sk))[...] = xs
        st = _dense(stats, (K + K * K + (D * K if yv is not None else 0),))
        st[:K] = xs.sum(axis=0)
        st[K:K + K * K] = (xs.T @ xs).reshape(-1)
        if yv is not None:
            st[K + K * K:] = (yv @ xs).reshape(-1)
        return 0

    # -- categorical ----------------------------------------------------------------------------------
    def vmp_softmax_moments(self, ctx, rows, K, phi, p, lse):
        self._count('vmp_softmax_moments')
        rows, K = int(rows), int(K)
        ph = _dense(phi, (rows, K))
        with np.errstate(all='ignore'):
            mx = np.max(ph, axis=1, keepdims=True)
            mx = np.where(np.isfinite(mx), mx, 0.0)
            e = np.exp(ph - mx)
            s = e.sum(axis=1, keepdims=True)
            l = np.log(s) + mx
            q = np.exp(ph - l)
            q = q / q.sum(axis=1, keepdims=True)
        _dense(p, (rows, K))[...] = q
        _dense(lse, (rows,))[...] = l[:, 0]
        return 0

    def vmp_onehot_i64(self, ctx, n, K, labels, out, info):
        self._count('vmp_onehot_i64')
        n, K = int(n), int(K)
        lab = _dense(labels, (n,), np.int64)
        flag = _dense(info, (1,), np.int32)
        o = _dense(out, (n, K))
        o[...] = 0.0
        bad = (lab < 0) | (lab >= K)
        flag[0] = int(bad.any())
        ok = ~bad
        o[np.nonzero(ok)[0], lab[ok]] = 1.0
        return 0

    # -- plate re-indexing ------------------------------------------------------------------------
    def vmp_take_axis(self, ctx, outer, src_len, inner, src, n, idx, dst, dst_len, dst_off):
        self._count('vmp_take_axis')
        outer, src_len, inner, n = int(outer), int(src_len), int(inner), int(n)
        dst_len, dst_off = int(dst_len), int(dst_off)
        s = _dense(src, (outer, src_len, inner))
        d = _dense(dst, (outer, dst_len, inner))
        ix = _dense(idx, (n,), np.int64) if _ptr(idx) else np.arange(n)
        d[:, dst_off:dst_off + n, :] = s[:, ix, :]
        return 0

    def vmp_segment_sum_axis(self, ctx, outer, src_len, inner, src, out_len, ptr, perm, dst):
        self._count('vmp_segment_sum_axis')
        outer, src_len, inner, out_len = int(outer), int(src_len), int(inner), int(out_len)
        s = _dense(src, (outer, src_len, inner))
        d = _dense(dst, (outer, out_len, inner))
        pt = _dense(ptr, (out_len + 1,), np.int64)
        pm = _dense(perm, (src_len,), np.int64)
        for l in range(out_len):
            acc = np.zeros((outer, inner))
            for t in range(pt[l], pt[l + 1]):
                acc = acc + s[:, pm[t], :]
            d[:, l, :] = acc
        return 0


def install():
    """Make a CPU runtime with the NumPy double the process runtime; returns it."""
    from bayespy_amd import device
    rt = device.Runtime(device='cpu')
    rt.lib = HostGenericLib()
    device.set_runtime(rt)
    return rt


def uninstall():
    from bayespy_amd import device
    device.set_runtime(None)
