"""
GPU parity of the generic plate-broadcast kernels (bayespy_amd.utils.misc /
linalg / darray.fuse) against NumPy brute force on seeded inputs and against
known answers produced by the reference's own utility functions
(tests/golden/utils_known_answers.npz, made by oracle/make_golden.py).

Tolerance: fp64, rtol 1e-12 for sums/products, 1e-10 for inverses; one-hot is
bit-exact.
"""
import os

import numpy as np
import pytest
from scipy import special

pytestmark = pytest.mark.gpu


def _brute(arrs, axis, sumaxis, keepdims):
    y = 1
    for a in arrs:
        y = y * a
    nd = np.ndim(y)
    if sumaxis:
        if axis is None:
            return np.sum(y, keepdims=keepdims)
        return np.sum(y, axis=tuple(axis) if not np.isscalar(axis) else axis, keepdims=keepdims)
    if axis is None:
        return y
    keep = sorted(a if a >= 0 else a + nd for a in ([axis] if np.isscalar(axis) else axis))
    red = tuple(i for i in range(nd) if i not in keep)
    return np.sum(y, axis=red, keepdims=keepdims) if red else y


SM_CASES = [
    (((),), dict()),
    (((), (), ()), dict()),
    (((3,),), dict(axis=())),
    (((3, 1, 5), (4, 1), (5,), ()), dict(axis=(), keepdims=True)),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(1,))),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(-2,))),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(1, -2))),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(1,), keepdims=True)),
    (((3, 1, 5, 6), (4, 1, 6), (4, 1, 1), ()), dict(axis=(1, -2), keepdims=True)),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(0,), sumaxis=False)),
    (((3, 1), (1, 4), (3, 4)), dict(axis=(-1,), sumaxis=False, keepdims=True)),
    (((3, 1, 5, 6), (4, 1, 6), (4, 1, 1), ()), dict(axis=(1, -1), sumaxis=False)),
    (((2, 3, 4, 5, 2, 3),), dict(axis=(0, 2, 4))),
    (((7, 1, 1), (1, 5000, 1), (7, 5000, 3)), dict(axis=(1,))),           # long reduction
    (((100000, 3), (100000, 1)), dict(axis=(0,))),                          # plate sum
    (((3, 40000, 2, 2), (1, 40000, 1, 1)), dict(axis=(1,), keepdims=True)),
    (((64, 1, 32), (1, 1000, 32)), dict(axis=(-1,))),                       # many outputs
    # few outputs, long reduction over the leading plates, kept axes not dense in every
    # operand (messages of the chain to its dynamics): fat-thread kernel, 16 / 64 accumulators
    (((300, 250, 4, 4), (1, 4, 1)), dict(axis=(0, 1))),
    (((300, 250, 1, 4, 4), (1, 4, 1, 1)), dict(axis=(0, 1))),
    (((90000, 3, 5),), dict(axis=(0,))),
    (((70000, 2, 3), (70000, 1, 1), (1, 2, 1)), dict(axis=(0,), keepdims=True)),
    # many outputs each with a long contiguous reduction (per-sequence sums)
    (((3000, 700, 2, 2),), dict(axis=(1, 2, 3))),
    (((3000, 1100), (1, 1100)), dict(axis=(1,))),
    # many outputs, short dense reductions: lane-group kernel (8 / 16 / 32 / 64 lanes)
    (((5000, 16), (5000, 16)), dict(axis=(1,))),
    (((5000, 9),), dict(axis=(1,))),
    (((4100, 3, 40), (1, 3, 40)), dict(axis=(2,))),
    (((4500, 70), (4500, 1), (1, 70)), dict(axis=(1,), keepdims=True)),
    (((300, 20, 6, 6), (300, 20, 6, 6)), dict(axis=(2, 3))),
    # dense two-axis forms (16-byte lanes over the flat index): column sums with per-plate /
    # per-column / scalar factors, merged trailing axes, widest and narrowest inner extents
    (((100000, 16), (100000, 1)), dict(axis=(0,))),
    (((50001, 4, 4), (50001, 1, 1), (1, 4, 4)), dict(axis=(0,))),
    (((70000, 8), (70000, 8), ()), dict(axis=(0,), keepdims=True)),
    (((4097, 512),), dict(axis=(0,))),
    (((9000, 2), (9000, 1)), dict(axis=(0,))),
    (((3, 40000, 2, 2), (1, 40000, 1, 1)), dict(axis=(0, 1))),
    # ... and row sums
    (((4500, 64), (4500, 1), (1, 64)), dict(axis=(1,), keepdims=True)),
    (((6001, 2, 2),), dict(axis=(1, 2))),
    (((4099, 128), (4099, 128), ()), dict(axis=(1,))),
    (((5000, 4, 4), (5000, 4, 4)), dict(axis=(-1, -2))),
]


@pytest.mark.parametrize('shapes,kw', SM_CASES)
def test_sum_multiply_matches_bruteforce(shapes, kw):
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(len(shapes) * 7 + sum(len(s) for s in shapes))
    arrs = [rs.normal(size=s) if s else float(rs.normal()) for s in shapes]
    got = misc.sum_multiply(*arrs, **kw).numpy()
    ref = _brute(arrs, kw.get('axis'), kw.get('sumaxis', True), kw.get('keepdims', False))
    assert got.shape == np.shape(ref)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * max(1.0, np.max(np.abs(ref))))
    # transposed (non-dense) trailing axes of the first operand take the strided paths
    if np.ndim(arrs[0]) >= 2 and np.shape(arrs[0])[-1] == np.shape(arrs[0])[-2]:
        from bayespy_amd.darray import DArray
        a0 = DArray.from_host(np.ascontiguousarray(np.swapaxes(arrs[0], -1, -2))).swapaxes(-1, -2)
        got = misc.sum_multiply(a0, *arrs[1:], **kw).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-12,
                                   atol=1e-12 * max(1.0, np.max(np.abs(ref))))


def test_sum_multiply_dense_forms_fall_back_on_unaligned_operands():
    """The 16-byte forms need aligned bases; an operand that starts at an odd element takes the
    strided kernels and gives the same sums."""
    import torch
    from bayespy_amd.darray import DArray
    from bayespy_amd.device import get_runtime
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(5)
    N = 20000
    flat = rs.normal(size=N * 16 + 1)
    w = rs.normal(size=(N, 1))
    t = torch.from_numpy(flat).to(get_runtime().device)
    a = DArray(t[1:].view(N, 16))
    ref = flat[1:].reshape(N, 16)
    np.testing.assert_allclose(misc.sum_multiply(a, w, axis=(0,)).numpy(), (ref * w).sum(0),
                               rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(misc.sum_multiply(a, a, axis=(1,)).numpy(), (ref * ref).sum(1),
                               rtol=1e-12, atol=1e-10)


def test_sum_multiply_errors():
    from bayespy_amd.utils import misc
    with pytest.raises(ValueError):
        misc.sum_multiply()
    with pytest.raises(ValueError):
        misc.sum_multiply(np.ones((3, 4)), axis=(5,), sumaxis=False)
    with pytest.raises(ValueError):
        misc.sum_multiply(np.ones((3, 4)), np.ones((5, 4)))


def test_sum_multiply_to_plates_known_answers(golden_dir):
    from bayespy_amd.utils import misc
    g = np.load(os.path.join(golden_dir, 'utils_known_answers.npz'))
    for i in range(int(g['smtp_n'])):
        nin, ndim = [int(v) for v in g['smtp%d_meta' % i]]
        arrs = [g['smtp%d_in%d' % (i, j)] for j in range(nin)]
        to = tuple(int(v) for v in g['smtp%d_to' % i])
        frm = tuple(int(v) for v in g['smtp%d_from' % i])
        got = misc.sum_multiply_to_plates(*arrs, to_plates=to, from_plates=frm, ndim=ndim).numpy()
        ref = g['smtp%d_out' % i]
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_softmax_and_special_known_answers(golden_dir):
    from bayespy_amd.utils import misc
    g = np.load(os.path.join(golden_dir, 'utils_known_answers.npz'))
    p, lse = misc.normalized_exp(g['lse_in'])
    np.testing.assert_allclose(lse.numpy(), g['nexp_lse'], rtol=1e-13)
    ref_p = g['nexp_p']
    ok = np.isfinite(ref_p)
    np.testing.assert_allclose(p.numpy()[ok], ref_p[ok], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(misc.logsumexp(g['lse_in']).numpy(), g['lse_out'], rtol=1e-13)
    np.testing.assert_allclose(misc.multidigamma(g['mdg_in'], 5).numpy(), g['mdg_out5'],
                               rtol=1e-13)


def test_large_softmax_rows():
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(3)
    x = rs.normal(size=(20011, 64)) * 30
    p, lse = misc.normalized_exp(x)
    m = x.max(axis=1, keepdims=True)
    ref_l = np.log(np.exp(x - m).sum(axis=1, keepdims=True)) + m
    ref_p = np.exp(x - ref_l)
    ref_p /= ref_p.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(lse.numpy(), ref_l, rtol=1e-13)
    np.testing.assert_allclose(p.numpy(), ref_p, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize('K', [1, 2, 3, 5, 16, 17, 64, 100, 255, 256, 257, 300])
def test_softmax_row_widths_and_impossible_states(K):
    """Every lanes-per-row instance (K <= 256: registers, one pass) and the wide-row kernel;
    -inf entries give exact zeros, an all -inf row follows misc.py:1375-1378 (max -> 0)."""
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(K)
    x = rs.normal(size=(1003, K)) * 20
    x[rs.rand(1003, K) < 0.1] = -np.inf
    x[7] = -np.inf
    x[11, :] = 0.0
    p, lse = misc.normalized_exp(x)
    with np.errstate(all='ignore'):
        m = x.max(axis=1, keepdims=True)
        m[~np.isfinite(m)] = 0.0
        ref_l = np.log(np.exp(x - m).sum(axis=1, keepdims=True)) + m
        ref_p = np.exp(x - ref_l)
        ref_p /= ref_p.sum(axis=1, keepdims=True)
    got_p, got_l = p.numpy(), lse.numpy()
    ok = np.isfinite(ref_l[:, 0])
    np.testing.assert_allclose(got_l[ok], ref_l[ok], rtol=1e-13)
    np.testing.assert_allclose(got_p[ok], ref_p[ok], rtol=1e-12, atol=1e-300)
    assert np.all(got_p[ok][~np.isfinite(x[ok])] == 0.0)
    assert np.all(np.isnan(got_p[~ok])) and np.all(got_l[~ok] == -np.inf)
    # an odd row stride (K odd) or an odd base take the 8-byte instance: the same values
    if K % 2 == 0 and K > 1:
        import torch
        from bayespy_amd.darray import DArray
        from bayespy_amd.device import get_runtime
        flat = np.concatenate([[0.0], x.ravel()])
        t = torch.from_numpy(flat).to(get_runtime().device)
        p2, lse2 = misc.normalized_exp(DArray(t[1:].view(1003, K)))
        np.testing.assert_allclose(p2.numpy()[ok], got_p[ok], rtol=1e-14, atol=1e-300)
        np.testing.assert_allclose(lse2.numpy()[ok], got_l[ok], rtol=1e-14)


def test_batched_spd_known_answers_and_random(golden_dir):
    from bayespy_amd.utils import linalg
    from bayespy_amd import _lib
    g = np.load(os.path.join(golden_dir, 'utils_known_answers.npz'))
    for n in (1, 3, 8, 20):
        U = linalg.chol(g['chol%d_C' % n])
        np.testing.assert_allclose(linalg.chol_inv(U).numpy(), g['chol%d_inv' % n], rtol=1e-9,
                                   atol=1e-12)
        np.testing.assert_allclose(linalg.chol_logdet(U).numpy(), g['chol%d_logdet' % n],
                                   rtol=1e-12)
        np.testing.assert_allclose(linalg.chol_solve(U, g['chol%d_b' % n]).numpy(),
                                   g['chol%d_solve' % n], rtol=1e-9, atol=1e-12)
    rs = np.random.RandomState(5)
    for n, batch in ((2, 1000), (5, 333), (9, 64), (16, 100), (32, 50), (64, 7),
                     (9, 5003), (13, 2049), (16, 4099), (17, 1500), (24, 1031), (32, 1027)):
        A = rs.normal(size=(batch, n, n))
        C = A @ A.transpose(0, 2, 1) + n * np.eye(n)
        U = linalg.chol(C)
        np.testing.assert_allclose(linalg.chol_inv(U).numpy(), np.linalg.inv(C), rtol=1e-8,
                                   atol=1e-11)
        np.testing.assert_allclose(linalg.chol_logdet(U).numpy(), np.linalg.slogdet(C)[1],
                                   rtol=1e-11)
    bad = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(_lib.NotPositiveDefiniteError):
        linalg.chol(bad)
    # one indefinite matrix inside a large batch (row-per-lane kernels)
    for n in (12, 20):
        C = np.tile(np.eye(n) * 2.0, (3000, 1, 1))
        C[1234, 3, 3] = -1.0
        with pytest.raises(_lib.NotPositiveDefiniteError):
            linalg.chol(C)
    with pytest.raises(NotImplementedError):
        linalg.chol(np.eye(65))


def test_linalg_products():
    from bayespy_amd.utils import linalg
    rs = np.random.RandomState(9)
    A, B = rs.normal(size=(4, 1, 3, 5)), rs.normal(size=(6, 5, 2))
    np.testing.assert_allclose(linalg.mmdot(A, B).numpy(), A @ B, rtol=1e-12)
    b = rs.normal(size=(6, 5))
    np.testing.assert_allclose(linalg.mvdot(A, b).numpy(), np.einsum('...ij,...j->...i', A, b),
                               rtol=1e-12)
    x, y = rs.normal(size=(7, 3)), rs.normal(size=(1, 3))
    np.testing.assert_allclose(linalg.outer(x, y).numpy(), x[..., :, None] * y[..., None, :],
                               rtol=1e-15)
    np.testing.assert_allclose(linalg.inner(x, y).numpy(), np.sum(x * y, axis=-1), rtol=1e-13)
    X = rs.normal(size=(2, 3, 4))
    np.testing.assert_array_equal(linalg.transpose(X).numpy(), np.swapaxes(X, -1, -2))


def test_fused_elementwise_formulas():
    from bayespy_amd import darray as da
    rs = np.random.RandomState(11)
    a = rs.gamma(2.0, size=(50, 1, 7)) + 0.1
    b = rs.normal(size=(3, 7))
    A, B = da.DArray.from_host(a), da.DArray.from_host(b)
    # GaussianARD scalar moments (gaussian.py:675-678): u0, u1, g
    phi0, phi1 = b, -a
    P0, P1 = da.DArray.from_host(phi0), da.DArray.from_host(phi1)
    u0 = da.fuse(lambda p0, p1: -p0 / (2 * p1), P0, P1)
    np.testing.assert_allclose(u0.numpy(), -phi0 / (2 * phi1), rtol=1e-15)
    g = da.fuse(lambda u, p0, p1: -0.5 * u * p0 + 0.5 * da.log(-2 * p1), u0, P0, P1)
    np.testing.assert_allclose(g.numpy(), -0.5 * (-phi0 / (2 * phi1)) * phi0
                               + 0.5 * np.log(-2 * phi1), rtol=1e-14)
    # Gamma moments (gamma.py:142-148)
    r = da.fuse(lambda x, y: da.digamma(x) - da.log(y * y + 1.0), A, B)
    np.testing.assert_allclose(r.numpy(), special.digamma(a) - np.log(b * b + 1), rtol=1e-13,
                               atol=1e-14)
    r = da.fuse(lambda x: x * da.log(x) - da.gammaln(x), A)
    np.testing.assert_allclose(r.numpy(), a * np.log(a) - special.gammaln(a), rtol=1e-12,
                               atol=1e-13)
    # 0 * -inf guard (expfamily.py:463-464)
    u = np.array([0.0, 1.0, 0.0, 2.0])
    v = np.array([-np.inf, 3.0, np.inf, -1.0])
    r = da.fuse(lambda x, y: da.where_nonzero(x, y) * x, u, v)
    np.testing.assert_array_equal(r.numpy(), np.array([0.0, 3.0, 0.0, -2.0]))
    # operators, scalars, 0-d, broadcasting
    np.testing.assert_allclose((A * B + 2.0 - B / A).numpy(), a * b + 2 - b / a, rtol=1e-14)
    np.testing.assert_allclose((-(3.0 - A)).numpy(), -(3 - a), rtol=1e-15)
    s = da.DArray.from_host(np.float64(2.5))
    np.testing.assert_allclose((s * s).numpy(), 6.25)
    np.testing.assert_allclose(da.fuse(lambda x: da.maximum(x, 0.5) + da.minimum(x, 0.5) +
                                       da.sqrt(x) + da.exp(-x) + da.recip(x) + x ** 2, A).numpy(),
                               np.maximum(a, 0.5) + np.minimum(a, 0.5) + np.sqrt(a) + np.exp(-a)
                               + 1 / a + a * a, rtol=1e-14)
    # a view is an operand like any other
    V = A[:, 0, ::2]
    np.testing.assert_allclose((V + 1.0).numpy(), a[:, 0, ::2] + 1, rtol=1e-15)
    with pytest.raises(ValueError):
        da.fuse(lambda x, y: x + y, np.ones((3, 4)), np.ones((5, 4)))


def test_onehot_is_bit_exact_and_validates():
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(2)
    lab = rs.randint(0, 64, size=(1234, 3))
    got = misc.onehot(lab, 64).numpy()
    ref = np.zeros((1234, 3, 64))
    ref[np.arange(1234)[:, None], np.arange(3)[None, :], lab] = 1
    assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        misc.onehot(np.array([0, 64]), 64)
    with pytest.raises(ValueError):
        misc.onehot(np.array([-1, 3]), 64)
    with pytest.raises(ValueError):
        misc.onehot(np.array([0.5]), 4)


def test_diag_helpers():
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(4)
    x = rs.normal(size=(5, 3))
    d = misc.diag(x).numpy()
    ref = np.zeros((5, 3, 3))
    for i in range(5):
        ref[i] = np.diag(x[i])
    np.testing.assert_array_equal(d, ref)
    M = rs.normal(size=(5, 3, 3))
    np.testing.assert_allclose(misc.get_diag(M).numpy(), np.einsum('...ii->...i', M), rtol=1e-15)


GEMM_CASES = [
    # (shapes, reduce axes) -- all large enough to take the MFMA contraction path
    (((64, 5000, 1, 1), (1, 5000, 16, 16)), (1,)),            # masked-PCA message to W
    (((64, 5000, 1, 1), (64, 1, 16, 16)), (0,)),              # ... to X
    (((70, 1, 9, 9), (1, 3000, 9, 9)), (2, 3)),               # SumMultiply second moment (E2)
    (((3000, 17, 1, 1), (3000, 1, 6, 6)), (0,)),              # mixture weighted statistics
    (((4, 300, 20, 8, 1), (4, 1, 1, 8, 33)), (3,)),           # batched matmul, broadcast batch
    (((129, 1, 257), (1, 65, 257)), (2,)),                    # ragged M, N, K
    (((64, 5000, 1, 1), (1, 5000, 16, 16), (64, 5000, 1, 1)), (1,)),   # mask folded in
    (((2, 3, 40, 50, 1), (2, 1, 1, 50, 60)), (3,)),           # two batch axes
]


@pytest.mark.parametrize('shapes,red', GEMM_CASES)
def test_dense_contractions_on_matrix_cores(shapes, red):
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(17)
    arrs = [rs.normal(size=s) for s in shapes]
    nd = max(len(s) for s in shapes)
    axis = tuple(a - nd for a in red) if red else None
    got = misc.sum_multiply(*arrs, axis=tuple(red), keepdims=True).numpy()
    y = 1
    for a in arrs:
        y = y * a
    ref = np.sum(y, axis=tuple(red), keepdims=True)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-10 * np.abs(ref).max())


def test_gemm_path_is_taken_and_deterministic(monkeypatch):
    from bayespy_amd.utils import misc
    calls = []
    orig = misc._try_gemm

    def spy(*a, **k):
        r = orig(*a, **k)
        calls.append(r)
        return r
    monkeypatch.setattr(misc, '_try_gemm', spy)
    rs = np.random.RandomState(3)
    a, b = rs.normal(size=(64, 20000, 1, 1)), rs.normal(size=(1, 20000, 16, 16))
    r1 = misc.sum_multiply(a, b, axis=(1,)).numpy()
    r2 = misc.sum_multiply(a, b, axis=(1,)).numpy()
    assert calls == [True, True] and np.array_equal(r1, r2)
    # small or one-sided contractions stay on the generic kernels
    misc.sum_multiply(rs.normal(size=(3, 4)), rs.normal(size=(4,)), axis=(1,))
    assert calls[-1] is False


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (16, 16, 100000, True, False),     # sum_n x_n x_n^T: both operands N-contiguous rows
    (64, 16, 100003, False, False),    # sum_n y_dn x_nk: A K-contiguous, ragged K
    (16, 64, 5000, True, True),
    (20, 30, 4096 + 7, False, True),   # partial 16 x 16 blocks on both sides
    (9, 48, 70001, True, False),
    (33, 16, 4096, False, False)])
def test_skinny_gemm_matches_numpy(M, N, K, ta, tb):
    """gemm_skinny_kernel (csrc/vmp_gemm.hip): long contractions with at most four 16 x 16 output
    blocks, K slices per wavefront, operands straight from HBM in every stride pattern the engine
    produces (views of (K, M) / (M, K) arrays); fixed-order combination => run-to-run identical."""
    from bayespy_amd.utils import misc
    from bayespy_amd.darray import DArray
    rs = np.random.RandomState(M * 131 + N * 7 + K)
    a = rs.normal(size=(K, M) if ta else (M, K))
    b = rs.normal(size=(N, K) if tb else (K, N))
    A = DArray.from_host(a)
    B = DArray.from_host(b)
    A = A.swapaxes(0, 1) if ta else A                  # (M, K) view
    B = B.swapaxes(0, 1) if tb else B                  # (K, N) view
    A3 = DArray(A.t.unsqueeze(1))                      # (M, 1, K) view
    B3 = DArray(B.t.transpose(0, 1).unsqueeze(0))      # (1, N, K) view
    r1 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    r2 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    ref = (a.T if ta else a) @ (b.T if tb else b)
    assert r1.shape == (M, N) and np.array_equal(r1, r2)
    np.testing.assert_allclose(r1, ref, rtol=1e-12, atol=1e-12 * np.sqrt(K))


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (100000, 16, 16, False, False),    # a K x K matrix applied to every plate (mvdot)
    (50001, 16, 64, True, False),      # the Dot message to the plate-side parent: A M-contiguous
    (4099, 64, 8, False, True),
    (20000, 33, 20, True, True),
    (8192, 9, 5, False, False),
    (30000, 24, 50, False, False)])
def test_tall_gemm_matches_numpy(M, N, K, ta, tb):
    """gemm_tall_kernel (csrc/vmp_gemm.hip): many rows, at most 64 columns, K <= 64 -- B in
    registers, A streamed by blocks of 16 rows, every stride pattern, ragged edges."""
    from bayespy_amd.utils import misc
    from bayespy_amd.darray import DArray
    rs = np.random.RandomState(M + N * 7 + K)
    a = rs.normal(size=(K, M) if ta else (M, K))
    b = rs.normal(size=(N, K) if tb else (K, N))
    A = DArray.from_host(a)
    B = DArray.from_host(b)
    A = A.swapaxes(0, 1) if ta else A
    B = B.swapaxes(0, 1) if tb else B
    A3 = DArray(A.t.unsqueeze(1))
    B3 = DArray(B.t.transpose(0, 1).unsqueeze(0))
    r1 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    ref = (a.T if ta else a) @ (b.T if tb else b)
    assert r1.shape == (M, N)
    np.testing.assert_allclose(r1, ref, rtol=1e-12, atol=1e-12)
    # the same product asked for the other way round (few rows, many columns; rows contiguous in
    # the output): routed to the same kernel as the transposed problem
    r2 = misc.sum_multiply(B3, A3, axis=(2,)).numpy()
    np.testing.assert_allclose(r2, ref, rtol=1e-12, atol=1e-12)
    if K == N:
        from bayespy_amd.utils import linalg
        Mx = rs.normal(size=(K, K))
        got = linalg.mvdot(Mx, A).numpy()                      # (M, K): Mx applied to every row
        np.testing.assert_allclose(got, (a.T if ta else a) @ Mx.T, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('shape_a,shape_b,axes', [
    ((1, 1, 16, 16), (1, 1, 16, 16), (0, 1, 2, 3)),     # tr(A B^T): one output, 256 products
    ((16, 16), (1, 1, 16, 16), (0, 1, 2, 3)),
    ((64, 1, 16), (1, 1, 16), (0,)),                    # 16 outputs over 64 products
    ((7, 100), (7, 100), (1,)),
    ((3, 5, 40), (1, 5, 40), (1, 2))])
def test_short_reductions_with_few_outputs(shape_a, shape_b, axes):
    """A handful of outputs over 64 .. 1023 products run on a workgroup per output
    (sum_multiply_block_kernel), not on one thread per output."""
    from bayespy_amd.utils import misc
    rs = np.random.RandomState(sum(shape_a) + len(axes))
    a, b = rs.normal(size=shape_a), rs.normal(size=shape_b)
    got = misc.sum_multiply(a, b, axis=axes, keepdims=True).numpy()
    full = a * b
    ref = np.sum(full, axis=tuple(ax + full.ndim - max(a.ndim, b.ndim) for ax in axes), keepdims=True)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('n,batch', [(9, 3000), (16, 4099), (12, 1025), (17, 1100), (32, 1029),
                                     (24, 2000)])
def test_fused_gaussian_moments(n, batch):
    """vmp_gaussian_moments (compute_moments_and_cgf of a Gaussian with a covariance per
    plate, gaussian.py:680-706) against NumPy: u0 = Cov phi0, u1 = u0 u0^T + Cov,
    g = -1/2 u0.phi0 + 1/2 log|-2 phi1|."""
    from bayespy_amd.utils import linalg
    from bayespy_amd import _lib
    rs = np.random.RandomState(n * 7 + batch)
    A = rs.normal(size=(batch, n, n))
    Lam = A @ A.transpose(0, 2, 1) + n * np.eye(n)
    phi1 = -0.5 * Lam
    phi0 = rs.normal(size=(batch, n))
    out = linalg.gaussian_moments(phi0, phi1)
    assert out is not None
    u0, u1, g = [o.numpy() for o in out]
    cov = np.linalg.inv(Lam)
    r0 = np.einsum('bij,bj->bi', cov, phi0)
    np.testing.assert_allclose(u0, r0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(u1, cov + r0[:, :, None] * r0[:, None, :], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g, -0.5 * np.sum(r0 * phi0, axis=1) + 0.5 * np.linalg.slogdet(Lam)[1],
                               rtol=1e-10)
    # sizes / layouts outside the fused kernel are declined, indefinite input is reported
    assert linalg.gaussian_moments(phi0[:10], phi1[:10]) is None
    assert linalg.gaussian_moments(np.zeros((2000, 4)), np.tile(-np.eye(4), (2000, 1, 1))) is None
    bad = phi1.copy()
    bad[batch // 2] = np.eye(n)
    with pytest.raises(_lib.NotPositiveDefiniteError):
        linalg.gaussian_moments(phi0, bad)


def test_cabi_of_the_index_and_chain_kernels_directly():
    """vmp_take_axis / vmp_segment_sum_axis / vmp_alpha_beta_recursion through the raw C ABI:
    results on plain device buffers, and the status codes for bad arguments."""
    import torch
    from bayespy_amd import _lib
    from bayespy_amd.device import get_runtime, ptr
    rt = get_runtime()
    lib = rt.lib
    dev = rt.device
    rt.sync_stream()
    src = torch.arange(2 * 5 * 3, dtype=torch.float64, device=dev)          # (outer 2, axis 5, inner 3)
    idx = torch.tensor([4, 0, 0, 2], dtype=torch.int64, device=dev)
    dst = torch.full((2, 6, 3), -1.0, dtype=torch.float64, device=dev)
    rt.check(lib.vmp_take_axis(rt.ctx, 2, 5, 3, ptr(src), 4, ptr(idx), ptr(dst), 6, 1))
    ref = src.reshape(2, 5, 3)[:, idx]
    assert torch.equal(dst[:, 1:5], ref) and torch.all(dst[:, 0] == -1) and torch.all(dst[:, 5] == -1)
    rt.check(lib.vmp_take_axis(rt.ctx, 2, 5, 3, ptr(src), 5, None, ptr(dst), 6, 0))   # block copy
    assert torch.equal(dst[:, :5], src.reshape(2, 5, 3))
    assert lib.vmp_take_axis(rt.ctx, 2, 5, 3, ptr(src), 4, ptr(idx), ptr(dst), 6, 3) == _lib.VMP_ERR_INVALID
    assert lib.vmp_take_axis(rt.ctx, 2, 5, 3, None, 4, ptr(idx), ptr(dst), 6, 0) == _lib.VMP_ERR_INVALID
    # segment sum: rows {1,2} -> target 0, row 0 -> target 2, target 1 empty
    ptr_ = torch.tensor([0, 2, 2, 3], dtype=torch.int64, device=dev)
    perm = torch.tensor([1, 2, 0], dtype=torch.int64, device=dev)
    y = torch.arange(2 * 3 * 2, dtype=torch.float64, device=dev)             # (2, 3, 2)
    out = torch.empty(2, 3, 2, dtype=torch.float64, device=dev)
    rt.check(lib.vmp_segment_sum_axis(rt.ctx, 2, 3, 2, ptr(y), 3, ptr(ptr_), ptr(perm), ptr(out)))
    yy = y.reshape(2, 3, 2)
    assert torch.equal(out[:, 0], yy[:, 1] + yy[:, 2]) and torch.all(out[:, 1] == 0)
    assert torch.equal(out[:, 2], yy[:, 0])
    assert lib.vmp_segment_sum_axis(rt.ctx, 2, 3, 2, ptr(y), 3, None, ptr(perm), ptr(out)) == _lib.VMP_ERR_INVALID
    # forward-backward recursion: two states, one transition -- by hand
    logp0 = torch.log(torch.tensor([[0.25, 0.75]], dtype=torch.float64, device=dev))
    P = torch.tensor([[[0.9, 0.1], [0.5, 0.5]]], dtype=torch.float64, device=dev)
    z0 = torch.empty(1, 2, dtype=torch.float64, device=dev)
    zz = torch.empty(1, 1, 2, 2, dtype=torch.float64, device=dev)
    g = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(16, dtype=torch.float64, device=dev)
    rt.check(lib.vmp_alpha_beta_recursion(rt.ctx, 1, 2, 1, ptr(logp0), 2, ptr(torch.log(P)), 4, 4,
                                          ptr(z0), ptr(zz), ptr(g), ptr(ws), ws.numel() * 8))
    joint = torch.tensor([[0.25 * 0.9, 0.25 * 0.1], [0.75 * 0.5, 0.75 * 0.5]], dtype=torch.float64)
    np.testing.assert_allclose(zz.cpu().numpy()[0, 0], joint.numpy(), rtol=1e-14)
    np.testing.assert_allclose(z0.cpu().numpy()[0], [0.25, 0.75], rtol=1e-14)
    np.testing.assert_allclose(g.cpu().numpy(), [0.0], atol=1e-15)
    rc = lib.vmp_alpha_beta_recursion(rt.ctx, 1, 2, 1, ptr(logp0), 2, ptr(P), 4, 4, ptr(z0), ptr(zz),
                                      ptr(g), ptr(ws), 8)
    assert rc == _lib.VMP_ERR_INVALID                    # workspace too small
    with pytest.raises(ValueError, match='workspace'):
        rt.check(rc)
    big = torch.zeros(65 * 65, dtype=torch.float64, device=dev)
    rc = lib.vmp_alpha_beta_recursion(rt.ctx, 1, 65, 1, ptr(big), 65, ptr(big), 0, 0, ptr(big),
                                      ptr(big), ptr(big), ptr(big), big.numel() * 8)
    assert rc == _lib.VMP_ERR_UNSUPPORTED                # more than 64 states


@pytest.mark.parametrize('spec,out', [
    ('dn,dk,nk', ''),            # sum y <f>: (y, x) -> (d, k) first, never the (d, n) product
    ('dk,nk,dl,nl', ''),         # sum <f>^2 = (W^T W) : (X^T X)
    ('dn,dk,nk', 'd'),           # the same sums kept per row (a precision with plates (D, 1))
    ('dn,dn,dk,nk', ''),         # with a plate mask
    ('nk,kl,nl', 'n'),           # a chain of three with a kept plate
    ('ab,bc,cd,de', 'ae')])      # a matrix chain
def test_contract_path_matches_einsum(spec, out):
    """misc.contract_path: the labelled contraction of misc.contract evaluated pair by pair
    (smallest intermediate first) against np.einsum, deterministic."""
    from bayespy_amd.utils import misc
    sizes = dict(d=24, n=30000, k=6, l=6, a=40, b=50, c=30, e=20)
    rs = np.random.RandomState(len(spec))
    terms = spec.split(',')
    arrs = [rs.normal(size=tuple(sizes[c] for c in t)) for t in terms]
    ref = np.einsum(spec + '->' + out, *arrs)
    got = misc.contract_path(arrs, [list(t) for t in terms], list(out), sizes)
    got2 = misc.contract_path(arrs, [list(t) for t in terms], list(out), sizes)
    assert got.shape == ref.shape and np.array_equal(got.numpy(), got2.numpy())
    np.testing.assert_allclose(got.numpy(), ref, rtol=1e-11, atol=1e-9 * np.abs(ref).max())


def test_plate_sums_over_lazy_dot_products():
    """GenericPlan._plate_sum with LazyContract factors (the first moment of a Dot node kept as a
    contraction): sum y <f>, sum <f>^2, with a plate-free factor and with a kept row axis, against
    the dense arrays; the dense form itself (.t) equals W X^T."""
    import bayespy_amd.inference.plans.generic as G
    from bayespy_amd.darray import DArray
    D, N, K = 12, 50000, 5
    rs = np.random.RandomState(2)
    w, x, y = rs.normal(size=(D, 1, K)), rs.normal(size=(1, N, K)), rs.normal(size=(D, N))
    W, X, Y = DArray.from_host(w), DArray.from_host(x), DArray.from_host(y)
    sizes = {'p0': D, 'p1': N, 'k0': K}
    f = G.LazyContract([W, X], [['p0', 'p1', 'k0'], ['p0', 'p1', 'k0']], ['p0', 'p1'], sizes,
                       ['p0', 'p1'])
    fd = w[:, 0, :] @ x[0].T
    assert f.shape == (D, N)
    plan = object.__new__(G.GenericPlan)
    a = DArray.from_host(np.asarray(0.7))
    cases = [([Y, f], (), fd * y), ([f, f], (), fd * fd), ([a, f, Y], (), 0.7 * fd * y),
             ([Y, f], (D, 1), fd * y), ([f, f], (1, N), fd * fd)]
    for factors, to, dense in cases:
        got = plan._plate_sum(factors, to, (D, N)).numpy()
        axes = tuple(i for i in range(2) if len(to) == 0 or to[i] == 1)
        ref = dense.sum(axis=axes, keepdims=len(to) > 0)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-9 * np.abs(ref).max())
    np.testing.assert_allclose(f.numpy(), fd, rtol=1e-13, atol=1e-13)


def test_cabi_queue_and_graph_of_small_operations():
    """The raw C ABI a binding would use for a sweep: small formulas / plate sums recorded by the
    queue (vmp_queue_begin / _end) and the whole call sequence recorded into a HIP graph
    (vmp_graph_begin / _end / _launch) on a stream of the context's own; replays with new
    contents of the input arrays give what NumPy gives, and queued == unqueued bit for bit."""
    import ctypes
    import torch
    from bayespy_amd import _lib
    from bayespy_amd.darray import OP_IN, OP_MUL, OP_ADD, OP_CONST
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    stream = torch.cuda.Stream(dev)
    ctx = ctypes.c_void_p()
    assert lib.vmp_ctx_create(0, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(ctx)) == 0
    # tune keys are per process: the Python runtime keeps sums and inverses out of the queue unless
    # asked (BAYESPY_AMD_SMALL_QUEUE=all); this test is about the library's own default
    assert lib.vmp_tune_set(b'small_queue_sm', 1) == 0
    try:
        K = 16
        a = torch.randn(K, K, dtype=torch.float64, device=dev)
        b = torch.randn(K, dtype=torch.float64, device=dev)
        t1 = torch.empty(K, K, dtype=torch.float64, device=dev)      # a * b + 2
        t2 = torch.empty(K, dtype=torch.float64, device=dev)         # sum_j t1[i, j] * b[j]
        t3 = torch.empty((), dtype=torch.float64, device=dev)        # sum_i t2[i]
        spd = torch.empty(K, K, dtype=torch.float64, device=dev)     # a a^T + K I
        sinv = torch.empty(K, K, dtype=torch.float64, device=dev)
        sld = torch.empty(1, dtype=torch.float64, device=dev)
        sinfo = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(1 << 20, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        vp = lambda t: ctypes.c_void_p(t.data_ptr())

        def sweep():
            shape = (ctypes.c_int64 * 2)(K, K)
            ins = (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
            strides = (ctypes.c_int64 * 4)(K, 1, 0, 1)
            ops = (ctypes.c_int32 * 5)(OP_IN | (0 << 8), OP_IN | (1 << 8), OP_MUL, OP_CONST | (0 << 8), OP_ADD)
            consts = (ctypes.c_double * 1)(2.0)
            assert lib.vmp_ewise(ctx, 2, shape, 2, ins, strides, 5, ops, 1, consts, vp(t1)) == 0
            ins2 = (ctypes.c_void_p * 2)(t1.data_ptr(), b.data_ptr())
            ostr = (ctypes.c_int64 * 2)(1, 0)
            assert lib.vmp_sum_multiply(ctx, 2, shape, 2, ins2, strides, ostr, ctypes.c_uint32(2), 1.0,
                                        vp(t2), vp(ws), ws.numel() * 8) == 0
            shape1 = (ctypes.c_int64 * 1)(K)
            ins3 = (ctypes.c_void_p * 1)(t2.data_ptr())
            s3 = (ctypes.c_int64 * 1)(1)
            o3 = (ctypes.c_int64 * 1)(0)
            assert lib.vmp_sum_multiply(ctx, 1, shape1, 1, ins3, s3, o3, ctypes.c_uint32(1), 1.0,
                                        vp(t3), vp(ws), ws.numel() * 8) == 0
            # spd = a a^T + K I (a contraction over the second axis of both), then its inverse
            shape3 = (ctypes.c_int64 * 3)(K, K, K)
            ins4 = (ctypes.c_void_p * 2)(a.data_ptr(), a.data_ptr())
            str4 = (ctypes.c_int64 * 6)(K, 0, 1, 0, K, 1)
            ostr4 = (ctypes.c_int64 * 3)(K, 1, 0)
            assert lib.vmp_sum_multiply(ctx, 3, shape3, 2, ins4, str4, ostr4, ctypes.c_uint32(4), 1.0,
                                        vp(spd), vp(ws), ws.numel() * 8) == 0
            ins5 = (ctypes.c_void_p * 1)(spd.data_ptr())
            str5 = (ctypes.c_int64 * 2)(K, 1)
            ops5 = (ctypes.c_int32 * 1)(OP_IN | (0 << 8))
            del ins5, str5, ops5
            assert lib.vmp_spd_batched(ctx, K, 1, vp(spd), vp(sinv), vp(sld), vp(sinfo)) == 0

        def expect():
            A, Bv = a.cpu().numpy(), b.cpu().numpy()
            T1 = A * Bv + 2.0
            return T1, T1 @ Bv, (T1 @ Bv).sum()

        def check_spd():
            S = a.cpu().numpy() @ a.cpu().numpy().T
            np.testing.assert_allclose(spd.cpu().numpy(), S, rtol=1e-12, atol=1e-12)
            S = spd.cpu().numpy()
            if np.linalg.eigvalsh(S).min() > 1e-6:
                np.testing.assert_allclose(sinv.cpu().numpy() @ S, np.eye(K), atol=1e-6)
                np.testing.assert_allclose(float(sld.cpu()), np.linalg.slogdet(S)[1], rtol=1e-9)
                assert int(sinfo.cpu()) == 0

        # unqueued, then queued: the five operations are ONE launch of the interpreter
        sweep()
        assert lib.vmp_ctx_sync(ctx) == 0
        ref = (t1.cpu().numpy().copy(), t2.cpu().numpy().copy(), float(t3.cpu()),
               sinv.cpu().numpy().copy(), float(sld.cpu()))
        np.testing.assert_allclose(ref[0], expect()[0], rtol=1e-14)
        np.testing.assert_allclose(ref[1], expect()[1], rtol=1e-12)
        check_spd()
        l0, n0 = ctypes.c_int64(), ctypes.c_int64()
        assert lib.vmp_queue_begin(ctx) == 0
        for t in (t1, t2, t3, spd, sinv, sld):
            t.zero_()
        torch.cuda.synchronize()
        sweep()
        assert lib.vmp_queue_end(ctx) == 0 and lib.vmp_ctx_sync(ctx) == 0
        assert lib.vmp_queue_stats(ctx, ctypes.byref(l0), ctypes.byref(n0)) == 0
        assert l0.value == 1 and n0.value == 5
        # formulas and inverses: the arithmetic of the stand-alone kernels, bit for bit; sums: another
        # order of the additions
        assert np.array_equal(t1.cpu().numpy(), ref[0])
        np.testing.assert_allclose(t2.cpu().numpy(), ref[1], rtol=1e-14)
        np.testing.assert_allclose(float(t3.cpu()), ref[2], rtol=1e-14)
        np.testing.assert_allclose(sinv.cpu().numpy(), ref[3], rtol=1e-10, atol=1e-13)
        check_spd()
        # "small_queue_sm" = 0: sums and inverses launch on their own, the formula waits for them
        assert lib.vmp_tune_set(b'small_queue_sm', 0) == 0
        try:
            assert lib.vmp_queue_begin(ctx) == 0
            sweep()
            assert lib.vmp_queue_end(ctx) == 0 and lib.vmp_ctx_sync(ctx) == 0
            l1, n1 = ctypes.c_int64(), ctypes.c_int64()
            assert lib.vmp_queue_stats(ctx, ctypes.byref(l1), ctypes.byref(n1)) == 0
            assert l1.value - l0.value == 1 and n1.value - n0.value == 1
            assert np.array_equal(t2.cpu().numpy(), ref[1]) and np.array_equal(sinv.cpu().numpy(), ref[3])
        finally:
            assert lib.vmp_tune_set(b'small_queue_sm', 1) == 0     # (for the graph part below)
        # the sweep as a graph (the queue open inside: its flush is recorded too), replayed on new data
        g = ctypes.c_void_p()
        assert lib.vmp_queue_begin(ctx) == 0
        assert lib.vmp_graph_begin(ctx) == 0
        sweep()
        assert lib.vmp_graph_end(ctx, ctypes.byref(g)) == 0 and g.value
        assert lib.vmp_queue_end(ctx) == 0
        for rep in range(3):
            with torch.cuda.stream(stream):
                a.normal_()
                b.normal_()
            assert lib.vmp_graph_launch(ctx, g) == 0 and lib.vmp_ctx_sync(ctx) == 0
            e1, e2, e3 = expect()
            np.testing.assert_allclose(t1.cpu().numpy(), e1, rtol=1e-14)
            np.testing.assert_allclose(t2.cpu().numpy(), e2, rtol=1e-12)
            np.testing.assert_allclose(float(t3.cpu()), e3, rtol=1e-12)
            check_spd()
        assert lib.vmp_graph_destroy(ctx, g) == 0
    finally:
        from bayespy_amd.device import get_runtime
        lib.vmp_tune_set(b'small_queue_sm', int(get_runtime()._tune_sm))
        lib.vmp_ctx_destroy(ctx)


@pytest.mark.parametrize('N,D,K,ylay', [
    (4099, 64, 16, 'n'), (4099, 64, 16, 'd'), (777, 20, 5, 'n'), (777, 20, 5, 'd'),
    (33, 7, 3, 'n'), (5000, 128, 32, 'n'), (2500, 130, 40, 'd'), (1500, 256, 64, 'n'),
    (9001, 33, 17, 'n'), (64, 0, 16, 'rows'), (3001, 0, 5, 'rows'), (2000, 0, 33, 'rows'),
    (700, 0, 64, 'rows')])
def test_gaussian_shared_update_through_the_c_abi(N, D, K, ylay):
    """vmp_gaussian_shared_update (include/vmp_hip.h) through raw ctypes against NumPy: <x_n> =
    Cov (p0 + B^T y_n) -- or the given message rows -- and the plate sums sum <x>, sum <x><x>^T,
    sum y <x>^T; both memory orders of Y, ragged D / K / N, every tile instance up to D = 256,
    K = 64; run twice: identical bits (fixed-order combination of the partial sums)."""
    import ctypes
    import torch
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    rs = np.random.RandomState(N + 3 * D + K)
    dev = rt.device
    A = rs.normal(size=(K, K))
    cov = np.linalg.inv(A @ A.T + K * np.eye(K))
    p0 = rs.normal(size=K)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    nbytes = int(rt.lib.vmp_gaussian_shared_update_workspace_bytes(D, K))
    assert nbytes > 0
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    x = torch.full((N, K), np.nan, dtype=torch.float64, device=dev)
    nst = K + K * K + D * K
    outs = []
    p0t, covt = up(p0), up(cov)          # (device arrays stay referenced while the call runs)
    for rep in range(2):
        stats = torch.full((nst,), np.nan, dtype=torch.float64, device=dev)
        rt.sync_stream()
        if ylay == 'rows':
            m = rs.normal(size=(N, K)) if rep == 0 else m
            mt = up(m)
            rc = rt.lib.vmp_gaussian_shared_update(
                rt.ctx, N, K, 0, None, 0, 0, None, 0, 0, vp(mt), K, 1, vp(p0t), vp(covt),
                vp(x), K, 1, vp(stats), vp(ws), nbytes)
            ref = (m + p0) @ cov.T
        else:
            if rep == 0:
                y = rs.normal(size=(D, N))
                B = rs.normal(size=(D, K))
            Bt = up(B)
            if ylay == 'n':
                yt = up(y)
                y_sd, y_sn = N, 1
            else:
                yt = up(y.T)
                y_sd, y_sn = 1, D
            rc = rt.lib.vmp_gaussian_shared_update(
                rt.ctx, N, K, D, vp(yt), y_sd, y_sn, vp(Bt), K, 1, None, 0, 0, vp(p0t),
                vp(covt), vp(x), K, 1, vp(stats), vp(ws), nbytes)
            ref = (y.T @ B + p0) @ cov.T
        rt.check(rc)
        torch.cuda.synchronize()
        outs.append((x.cpu().numpy().copy(), stats.cpu().numpy().copy()))
    xs, st = outs[0]
    tol = dict(rtol=1e-11, atol=1e-11 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(xs, ref, **tol)
    big = dict(rtol=1e-10, atol=1e-10 * N)
    np.testing.assert_allclose(st[:K], ref.sum(axis=0), **big)
    np.testing.assert_allclose(st[K:K + K * K].reshape(K, K), ref.T @ ref, **big)
    if ylay != 'rows':
        np.testing.assert_allclose(st[K + K * K:].reshape(D, K), y @ ref, **big)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (5000, 32, 256, False, True),      # mixture log-likelihood: y y^T (N, D^2) against Lambda_k (K, D^2)
    (256, 32, 30000, True, False),     # weighted statistics: sum_n r_nk y y^T, split over the plates
    (130, 17, 300, False, False),      # ragged in every direction
    (4097, 24, 100, True, True),
    (65, 32, 64 + 16, False, False)])
def test_narrow_output_gemm_matches_numpy(M, N, K, ta, tb):
    """gemm_kernel<128, 32> (csrc/vmp_gemm.hip): outputs of at most 32 columns on a 128 x 32 tile
    (the 64 x 64 tile ran half empty for the 32 components of a mixture), every stride pattern,
    split-K included; identical to the 64 x 64 tile's result up to the order of the K slices
    (tune key gemm_narrow_tile = 0) and run-to-run identical."""
    from bayespy_amd.utils import misc
    from bayespy_amd.darray import DArray
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    rs = np.random.RandomState(M + N * 11 + K)
    a = rs.normal(size=(K, M) if ta else (M, K))
    b = rs.normal(size=(N, K) if tb else (K, N))
    A = DArray.from_host(a)
    B = DArray.from_host(b)
    A = A.swapaxes(0, 1) if ta else A
    B = B.swapaxes(0, 1) if tb else B
    A3 = DArray(A.t.unsqueeze(1))
    B3 = DArray(B.t.transpose(0, 1).unsqueeze(0))
    ref = (a.T if ta else a) @ (b.T if tb else b)
    r1 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    r2 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    assert r1.shape == (M, N) and np.array_equal(r1, r2)
    np.testing.assert_allclose(r1, ref, rtol=1e-12, atol=1e-12 * np.sqrt(K))
    try:
        rt.lib.vmp_tune_set(b'gemm_narrow_tile', 0)
        r0 = misc.sum_multiply(A3, B3, axis=(2,)).numpy()
    finally:
        rt.lib.vmp_tune_set(b'gemm_narrow_tile', 1)
    np.testing.assert_allclose(r1, r0, rtol=1e-13, atol=1e-13 * np.sqrt(K))


def test_queue_places_small_arrays_in_lds_without_changing_results():
    """The interpreter of small operations keeps the small arrays of a launch in LDS (DESIGN.md
    section 5.7).  A chain that exercises the placement rules -- strided VIEWS into an earlier record's
    result (diagonal, transposed, a row), results too large or too many for the arena (read back from
    memory behind a fence), an inverse between formulas, an operand from outside used twice, more
    records than one staging chunk -- gives, formula by formula, the bits of the same chain with every
    operation launched on its own, and sums / inverses to rounding."""
    import torch
    from bayespy_amd.utils import misc, linalg
    from bayespy_amd.darray import DArray, fuse
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    rs = np.random.RandomState(11)
    K = 16
    A = DArray.from_host(rs.normal(size=(K, K)))
    v = DArray.from_host(rs.normal(size=(K,)))
    BIG = [DArray.from_host(rs.normal(size=(2048,))) for _ in range(9)]      # 9 x 2048 > the arena
    G = rs.normal(size=(K, K))
    S = DArray.from_host(G @ G.T + K * np.eye(K))

    def chain():
        out = []
        a = fuse(lambda x, y: x * y + 2.0, A, v)                  # (K, K), broadcast operand
        d = misc.get_diag(a)                                      # strided view of a record's result
        b = fuse(lambda x, y: x + 3.0 * y, d, v)
        at = a.swapaxes(0, 1)                                     # transposed view
        c = fuse(lambda x, y: x - y, a, at)
        row = a[3]                                                # a row
        e = fuse(lambda x, y: x * y, row, b)
        bigs = [fuse(lambda x: x * 1.5 + 1.0, g) for g in BIG]    # the later ones find no room
        f = fuse(lambda x, y: x + y, bigs[0], bigs[8])            # one from the arena, one from memory
        s1 = misc.sum_multiply(a, c, axis=(-1, -2))               # scalar
        s2 = misc.sum_multiply(a, v, axis=(-1,))                  # (K,)
        inv = linalg.chol_inv(linalg.chol(S))                     # a queued 16 x 16 inverse
        h = fuse(lambda x, y: x * y, inv, a)                      # reads an inverse's result
        s3 = misc.sum_multiply(h, axis=(-1, -2))
        t = e
        for _ in range(24):                                       # beyond one staging chunk
            t = fuse(lambda x, y: 0.5 * x + y, t, v)
        out = [a, b, c, e, f, s1, s2, inv, h, s3, t]
        return out

    def run(queue):
        rt.lib.vmp_tune_set(b'small_queue', queue)
        try:
            if queue:
                with rt.operation():
                    res = chain()
            else:
                res = chain()
            rt.flush_small()
            return [np.asarray(r.numpy()) for r in res]
        finally:
            rt.lib.vmp_tune_set(b'small_queue', 1)

    s0 = rt.queue_stats()
    q = run(1)
    s1 = rt.queue_stats()
    assert s1['operations'] - s0['operations'] >= 40 and s1['launches'] - s0['launches'] <= 6
    u = run(0)
    names = ['a', 'b', 'c', 'e', 'f', 's1', 's2', 'inv', 'h', 's3', 't']
    for n_, x, y in zip(names, q, u):
        if n_ in ('s1', 's2', 's3', 'inv', 'h'):
            np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-12, err_msg=n_)
        else:
            assert np.array_equal(x, y), n_
