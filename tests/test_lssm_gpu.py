"""
GPU: the fused linear state-space model block (BASELINE.json config 5; vmp_lssm_* through
LSSMPlan) against the chunk-free NumPy oracle (oracle/lssm.py, pinned on the live reference) on
seeded inputs incl. tiny / ragged sizes; the live-reference traces themselves are in
tests/test_chain_gpu.py::test_lssm_matches_reference[fused].
Bars: bound rtol 1e-9 per iteration and per node term; moments rtol 1e-7.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(M, B, T, D, seed):
    rs = np.random.RandomState(seed)
    a_true = 0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]
    x = np.zeros((B, T, D))
    x[:, 0] = rs.normal(size=(B, D))
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + rs.normal(size=(B, D))
    c_true = rs.normal(size=(M, D))
    y = np.einsum('md,btd->mbt', c_true, x) + 0.3 * rs.normal(size=(M, B, T))
    return y, rs.normal(size=(B, T, D)), rs.normal(size=(M, D))


def _build(y, x0, c0, gamma_nu, shard=False):
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain
    from bayespy_amd.inference import VB
    M, B, T = y.shape
    D = x0.shape[-1]
    alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    A.initialize_from_value(np.identity(D))
    nu = Gamma(1e-3, 1e-3, plates=(D,), name='nu') if gamma_nu else np.ones(D)
    X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, nu, n=T, plates=(B,), name='X')
    if shard:
        X.shard(-1)
    X.initialize_from_value(x0)
    gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
    gamma.initialize_from_value(1e-2 * np.ones(D))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
    C.initialize_from_value(c0.reshape(M, 1, 1, D))
    tau = Gamma(1e-5, 1e-5, name='tau')
    tau.initialize_from_value(1e2)
    F = SumMultiply('i,i', C, X, name='F')
    Y = GaussianARD(F, tau, name='Y')
    Y.observe(y)
    nodes = [Y, F, C, gamma, X, A, alpha, tau] + ([nu] if gamma_nu else [])
    Q = VB(*nodes)
    Q.ignore_bound_checks = True
    return Q


@pytest.mark.parametrize('M,B,T,D', [(1, 1, 1, 1), (3, 5, 2, 2), (8, 70, 33, 4), (16, 300, 20, 3),
                                     (5, 1000, 50, 8), (8, 257, 1000, 4), (2, 64, 3, 5),
                                     # wide observations: the sweeps on the projected data
                                     # tau C^T y, a separate y <x>^T pass (M > 8, > 16 at D <= 4)
                                     (30, 300, 40, 3), (9, 70, 33, 8), (17, 513, 9, 4),
                                     (64, 40, 100, 8), (33, 1, 20, 1), (20, 260, 1000, 4),
                                     # 8 < D <= 16: the big-state path (round 6)
                                     (5, 300, 40, 9), (12, 70, 33, 12), (8, 513, 120, 16),
                                     (40, 257, 9, 13), (3, 1, 1, 16), (16, 1000, 50, 16)])
@pytest.mark.parametrize('gamma_nu', [False, True])
def test_fused_lssm_vs_oracle(M, B, T, D, gamma_nu):
    from oracle.lssm import LSSMOracle
    y, x0, c0 = _data(M, B, T, D, seed=M + B + T + D)
    Q = _build(y, x0, c0, gamma_nu)
    assert type(Q.plans[0]).__name__ == 'LSSMPlan'
    iters = 3
    Q.update(repeat=iters, verbose=False)
    o = LSSMOracle(y, x0, c0, nu_prior=(1e-3, 1e-3) if gamma_nu else None)
    o.iterate(iters)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-9)
    for nm in ('Y', 'C', 'A', 'X', 'gamma', 'alpha', 'tau') + (('nu',) if gamma_nu else ()):
        np.testing.assert_allclose(Q.l[Q[nm]][:iters], [t[nm] for t in o.L_terms], rtol=1e-8,
                                   atol=1e-7, err_msg=nm)
    np.testing.assert_allclose(Q['A'].u[0], o.Am, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q['A'].u[1], o.AA, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q['C'].u[0].reshape(M, D), o.Cm, rtol=1e-7, atol=1e-10)
    xu = Q['X'].u
    # the means come out of two T-step recursions: absolute accuracy relative to their scale
    sx = float(np.abs(o.X).max())
    np.testing.assert_allclose(xu[0], o.X, rtol=1e-7, atol=1e-8 * sx)
    np.testing.assert_allclose(xu[1], o.V[None] + o.X[:, :, :, None] * o.X[:, :, None, :],
                               rtol=1e-7, atol=1e-8 * sx * sx)
    if T > 1:
        np.testing.assert_allclose(xu[2], o.Cn[None] + o.X[:, :-1, :, None] * o.X[:, 1:, None, :],
                                   rtol=1e-7, atol=1e-8 * sx * sx)
    np.testing.assert_allclose(np.array(Q['tau'].u, dtype=np.float64).ravel(),
                               [o.tau, o.logtau], rtol=1e-9)


def test_fused_lssm_config_scale_direct_oracle_parity():
    """the dimensions of BASELINE config 5 (T=1e3, M=8, D=4) with B = 2e4 + 13
    sequences -- 313 wavefronts x 1000 steps, a ragged last wavefront -- against the chunk-free
    oracle on the same data (generated on the device), two iterations: the bound, every node's
    term, <A>, <C>, <tau> and a strided sample of the chain means."""
    import torch
    from bayespy_amd.device import get_runtime
    from oracle.lssm import LSSMOracle
    dev = get_runtime().device
    M, B, T, D = 8, 20_013, 1000, 4
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    rs = np.random.RandomState(11)
    a_true = torch.from_numpy(0.9 * np.linalg.qr(rs.normal(size=(D, D)))[0]).to(dev)
    c_true = torch.from_numpy(rs.normal(size=(M, D))).to(dev)
    x = torch.empty(B, T, D, device=dev, dtype=torch.float64)
    x[:, 0] = torch.randn(B, D, generator=g, device=dev, dtype=torch.float64)
    for t in range(1, T):
        x[:, t] = x[:, t - 1] @ a_true.T + torch.randn(B, D, generator=g, device=dev,
                                                       dtype=torch.float64)
    y = torch.einsum('md,btd->mbt', c_true, x)
    y += 0.3 * torch.randn(M, B, T, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(B, T, D, generator=g, device=dev, dtype=torch.float64)
    c0 = rs.normal(size=(M, D))
    del x
    Q = _build(y, x0, c0, False)
    assert type(Q.plans[0]).__name__ == 'LSSMPlan'
    iters = 2
    Q.update(repeat=iters, verbose=False)
    o = LSSMOracle(y.cpu().numpy(), x0.cpu().numpy(), c0)
    del y, x0
    o.iterate(iters)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-9)
    for nm in ('Y', 'C', 'A', 'X', 'gamma', 'alpha', 'tau'):
        np.testing.assert_allclose(Q.l[Q[nm]][:iters], [t[nm] for t in o.L_terms], rtol=1e-8,
                                   atol=1e-7 + 2e-14 * M * B * T, err_msg=nm)
    np.testing.assert_allclose(Q['A'].u[0], o.Am, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q['C'].u[0].reshape(M, D), o.Cm, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(np.array(Q['tau'].u, dtype=np.float64).ravel(),
                               [o.tau, o.logtau], rtol=1e-9)
    xm = Q['X'].u[0]
    sx = float(np.abs(o.X).max())
    sel = np.r_[0:B:211, B - 14:B]
    np.testing.assert_allclose(xm[sel], o.X[sel], rtol=1e-7, atol=1e-8 * sx)


def _lssm_edge_cases():
    rs = np.random.RandomState(3)

    def mk(M, B, T, D, scale=1.0):
        return rs.normal(size=(M, B, T)) * scale, rs.normal(size=(B, T, D)), rs.normal(size=(M, D))
    return [('one_step', mk(2, 3, 1, 2)), ('all_ones', mk(1, 1, 2, 1)),
            ('zero_data', mk(3, 4, 6, 2, 0.0)), ('tiny_scale', mk(3, 4, 6, 2, 1e-6)),
            ('huge_scale', mk(3, 4, 6, 2, 1e6)), ('more_states_than_observed', mk(2, 6, 12, 5)),
            ('white_noise_long', mk(4, 40, 700, 4))]


@pytest.mark.parametrize('gamma_nu', [False, True])
@pytest.mark.parametrize('case', _lssm_edge_cases(), ids=lambda c: c[0])
def test_fused_lssm_edge_regimes_vs_oracle(case, gamma_nu):
    """Degenerate inputs of the state-space block: a single time step, all-zero data, data without
    any dynamics, extreme scales, D > M.  The oracle agrees with the live reference to <= 1e-15 on
    these inputs (1e-4 at the 1e6 scale, where the reference's own phi . u sums cancel)."""
    from oracle.lssm import LSSMOracle
    _, (y, x0, c0) = case
    M, B, T = y.shape
    Q = _build(y, x0, c0, gamma_nu)
    assert type(Q.plans[0]).__name__ == 'LSSMPlan'
    iters = 4
    Q.update(repeat=iters, verbose=False)
    o = LSSMOracle(y, x0, c0, nu_prior=(1e-3, 1e-3) if gamma_nu else None)
    o.iterate(iters)
    assert np.all(np.isfinite(Q.L[:iters]))
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-8)
    sx = max(float(np.abs(o.X).max()), 1e-300)
    np.testing.assert_allclose(Q['X'].u[0], o.X, rtol=1e-6, atol=1e-8 * sx)
    np.testing.assert_allclose(Q['A'].u[0], o.Am, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.array(Q['tau'].u, dtype=np.float64).ravel(), [o.tau, o.logtau],
                               rtol=1e-8)


def test_stationary_stretch_of_the_covariance_recursion_is_filled_in():
    """Long chains: once the filter's Riccati map has converged to working precision (8 ulp) the
    remaining interior steps reuse one (S^-1, J) and one (V, C).  In the first iteration of this
    model both recursions converge after a few hundred steps: the shortcut is taken (diagnostics)
    and the results still match the oracle, which computes every step."""
    from oracle.lssm import LSSMOracle
    M, B, T, D = 2, 70, 600, 2
    y, x0, c0 = _data(M, B, T, D, seed=5)
    Q = _build(y, x0, c0, False)
    Q.update(repeat=1, verbose=False)
    fwd, bwd = Q.plans[0].cov_stationary_from()
    assert 0 < fwd < T - 2 and 0 < bwd < T - 2
    Q.update(repeat=2, verbose=False)
    o = LSSMOracle(y, x0, c0, nu_prior=None)
    o.iterate(3)
    np.testing.assert_allclose(Q.L[:3], np.array(o.L), rtol=1e-9)
    np.testing.assert_allclose(Q['X'].u[0], o.X, rtol=1e-7, atol=1e-8 * float(np.abs(o.X).max()))


def test_fused_lssm_device_inputs_and_checkpoint(tmp_path):
    import torch
    from bayespy_amd.device import get_runtime
    y, x0, c0 = _data(8, 300, 40, 4, seed=9)
    d = get_runtime().device
    Q = _build(torch.from_numpy(y).to(d), torch.from_numpy(x0).to(d), c0, False)
    Qh = _build(y, x0, c0, False)
    Q.update(repeat=2, verbose=False)
    Qh.update(repeat=2, verbose=False)
    np.testing.assert_array_equal(Q.L[:2], Qh.L[:2])
    fn = str(tmp_path / 'lssm.ckpt')
    Q.save(filename=fn)
    Q.update(repeat=2, verbose=False)
    Q2 = _build(y, x0, c0, False)
    Q2.load(filename=fn)
    Q2.update(repeat=2, verbose=False)
    np.testing.assert_array_equal(Q2.L[:4], Q.L[:4])
    np.testing.assert_array_equal(Q2['X'].u[0], Q['X'].u[0])


def test_segmented_covariance_recursion_is_bit_identical():
    """vmp_lssm_x_update runs the covariance recursion in time segments beside the forward sweep
    (round 3); the arithmetic is that of the single launch: same bound, same moments, bit for bit,
    also when the stationary stretch is found inside a segment."""
    from bayespy_amd.device import get_runtime
    lib = get_runtime().lib
    for (M, B, T, D, seed) in [(8, 300, 1000, 4, 1), (2, 70, 600, 2, 5), (5, 200, 640, 8, 3),
                               (6, 120, 640, 12, 7), (3, 64, 520, 16, 9)]:
        y, x0, c0 = _data(M, B, T, D, seed=seed)
        res = []
        try:
            for seg in (1, 0):
                lib.vmp_tune_set(b'lssm_segments', seg)
                Q = _build(y, x0, c0, False)
                Q.update(repeat=3, verbose=False)
                res.append((Q.L[:3].copy(), Q['X'].u[0].copy(), Q['A'].u[0].copy()))
        finally:
            lib.vmp_tune_set(b'lssm_segments', 1)
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_array_equal(res[0][2], res[1][2])


def test_checkpoint_form_of_the_sweeps_agrees_with_the_two_array_form():
    """D <= 4, M <= 8: the forward sweep keeps z every 4th step only and the backward sweep forms
    the steps of a block again from the checkpoint (no (T, D, B) array of z is written or read).
    Same recursions in the same order; the compiler contracts a few products of the plate sums
    differently in the two kernels, so the results agree to rounding (bound 1e-13 relative after
    the first iteration, 1e-11 after three; <x> bit for bit after the first iteration) -- incl. T not a multiple of the block,
    sequences that do not fill a workgroup, and the segmented covariance recursion beside it."""
    from bayespy_amd.device import get_runtime
    lib = get_runtime().lib
    for (M, B, T, D, seed) in [(8, 300, 1000, 4, 1), (2, 70, 601, 2, 5), (5, 1, 11, 3, 3),
                               (7, 513, 515, 1, 4), (8, 1000, 8, 4, 6), (3, 40, 9, 4, 7)]:
        y, x0, c0 = _data(M, B, T, D, seed=seed)
        res = []
        try:
            for ck in (1, 0):
                lib.vmp_tune_set(b'lssm_checkpoint', ck)
                Q = _build(y, x0, c0, False)
                Q.update(repeat=1, verbose=False)
                x1 = Q['X'].u[0].copy()
                Q.update(repeat=2, verbose=False)
                res.append((Q.L[:3].copy(), x1, Q['X'].u[0].copy(), Q['A'].u[0].copy(),
                            Q['C'].u[0].copy()))
        finally:
            lib.vmp_tune_set(b'lssm_checkpoint', 1)
        np.testing.assert_allclose(res[0][0][:1], res[1][0][:1], rtol=1e-13)
        np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-11)
        np.testing.assert_array_equal(res[0][1], res[1][1])
        # (three iterations on: the rounding differences of the sums have passed through two more
        # 1000-step recursions)
        for a, b in zip(res[0][2:], res[1][2:]):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8 * np.abs(b).max())


@pytest.mark.parametrize('D', [4, 8, 12, 16])
def test_state_rotation_of_the_means_through_the_c_abi(D):
    """vmp_lssm_rotate_x: <x_bt> <- R <x_bt> on the time-major array (transformations.py:1167-1176);
    D <= 8 with R in scalar registers, 8 < D <= 16 with R in LDS (round 6)."""
    import ctypes
    import torch
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    rs = np.random.RandomState(D)
    T, B, BL = 7, 300, 320
    z = rs.normal(size=(T, D, BL))
    R = rs.normal(size=(D, D))
    zt = torch.from_numpy(z.copy()).to(rt.device)
    Rt = torch.from_numpy(R).to(rt.device)
    rt.sync_stream()
    rt.check(rt.lib.vmp_lssm_rotate_x(rt.ctx, D, T, B, BL, ctypes.c_void_p(Rt.data_ptr()),
                                      ctypes.c_void_p(zt.data_ptr())))
    got = zt.cpu().numpy()
    ref = z.copy()
    ref[:, :, :B] = np.einsum('ij,tjb->tib', R, z[:, :, :B])
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize('M,B,T,D', [(6, 301, 45, 12), (8, 96, 130, 16), (3, 33, 17, 9)])
def test_matrix_core_sweeps_agree_with_the_thread_per_sequence_form(M, B, T, D):
    """8 < D <= 16: the sweeps on v_mfma_f64_16x16x4_f64 (state of 16 sequences = one accumulator,
    default) against the same sweeps as one thread per sequence (tune key lssm_big_mfma = 0): same
    bound trace and means to rounding; both against the oracle in test_fused_lssm_vs_oracle."""
    from bayespy_amd.device import get_runtime
    y, x0, c0 = _data(M, B, T, D, seed=5 * D + M)
    out = []
    rt = get_runtime()
    try:
        for mf in (1, 0):
            rt.lib.vmp_tune_set(b'lssm_big_mfma', mf)
            Q = _build(y, x0, c0, True)
            assert type(Q.plans[0]).__name__ == 'LSSMPlan'
            Q.update(repeat=3, verbose=False)
            out.append((np.array(Q.L[:3]), np.asarray(Q['X'].u[0])))
    finally:
        rt.lib.vmp_tune_set(b'lssm_big_mfma', 1)
    # (the two forms also sum the plates in different orders: matrix-core tiles / thread blocks)
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-10)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-8, atol=1e-8 * np.abs(out[1][1]).max())


def _cov_recursion_numpy(Dg0, Dgm, DgT, E, T):
    """The block recursion of vmp_lssm_cov, every step computed (oracle/lssm.py: block LDL^T of the
    block-tridiagonal precision; utils/linalg.py:468-575 in the reference)."""
    D = E.shape[0]
    Sinv = np.zeros((T, D, D))
    J = np.zeros((T - 1, D, D))
    S = Dg0.copy()
    ld = 0.0
    for t in range(T):
        ld += np.linalg.slogdet(S)[1]
        Sinv[t] = np.linalg.inv(S)
        if t < T - 1:
            J[t] = Sinv[t] @ E
            S = (Dgm if t + 1 < T - 1 else DgT) - E.T @ J[t]
    V = Sinv[T - 1].copy()
    sv, sc = V.copy(), np.zeros((D, D))
    for t in range(T - 2, -1, -1):
        C = -J[t] @ V
        V = Sinv[t] - C @ J[t].T
        sv += V
        sc += C
    return Sinv, J, sv, V, sc, ld


@pytest.mark.parametrize('form', [1, 0])
@pytest.mark.parametrize('D,T', [(12, 700), (16, 500), (9, 400), (13, 90)])
def test_big_state_covariance_recursion_fills_in_its_stationary_stretch(D, T, form):
    """8 < D <= 16, through the C ABI, both forms of the covariance recursion (tune key lssm_cov_mfma:
    1 = one wavefront on the matrix cores, block sweeps with 4 x 4 pivots; 0 = 256 threads, scalar
    Gauss-Jordan through LDS): each applies the rule of the D <= 8 kernel (iterates within 8 ulp: the
    interior steps behind are filled in).  On a
    strongly contracting map the shortcut is taken (diagnostics); S^-1, J, the sums of V and C and
    log|Phi| agree with the recursion that computes every step (tune key lssm_cov_shortcut = 0) and
    with a NumPy restatement."""
    import torch
    from bayespy_amd.device import get_runtime
    rt = get_runtime()
    rng = np.random.RandomState(D)
    G = rng.randn(D, D)
    E = -0.35 * np.linalg.qr(G)[0] + 0.02 * rng.randn(D, D)
    W = rng.randn(D, D) * 0.1
    Dgm = 3.0 * np.eye(D) + W @ W.T
    Dg0 = Dgm + np.eye(D)
    DgT = Dgm - 0.5 * np.eye(D)
    dev = rt.device
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    ins = [up(Dg0), up(Dgm), up(DgT), up(E)]
    ref = _cov_recursion_numpy(Dg0, Dgm, DgT, E, T)
    out = []
    try:
        rt.lib.vmp_tune_set(b'lssm_cov_mfma', form)
        for sc in (8, 0):
            rt.lib.vmp_tune_set(b'lssm_cov_shortcut', sc)
            Sinv = torch.zeros(T, D, D, dtype=torch.float64, device=dev)
            J = torch.zeros(T, D, D, dtype=torch.float64, device=dev)
            sums = torch.zeros(5 * D * D + 16, dtype=torch.float64, device=dev)
            rt.check(rt.lib.vmp_lssm_cov(rt.ctx, T, D, vp(ins[0]), vp(ins[1]), vp(ins[2]), vp(ins[3]),
                                         vp(Sinv), vp(J), vp(sums)))
            rt.sync_stream()
            sm = sums.cpu().numpy()
            fwd, bwd = int(sm[5 * D * D + 2]), int(sm[5 * D * D + 3])
            if sc:
                assert 0 < fwd < T - 2 and 0 < bwd < T - 2
            else:
                assert fwd < 0 and bwd < 0
            assert sm[5 * D * D + 1] == 0.0
            out.append((Sinv.cpu().numpy(), J.cpu().numpy()[:T - 1], sm[:D * D].reshape(D, D),
                        sm[D * D:2 * D * D].reshape(D, D), sm[3 * D * D:4 * D * D].reshape(D, D),
                        sm[5 * D * D]))
    finally:
        rt.lib.vmp_tune_set(b'lssm_cov_shortcut', 8)
        rt.lib.vmp_tune_set(b'lssm_cov_mfma', 1)
    for o in out:
        for got, want in zip(o, ref):
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize('M,B,T,D', [(8, 301, 45, 8), (5, 96, 130, 7), (20, 70, 33, 8), (8, 1000, 20, 12),
                                     (15, 333, 40, 16), (3, 97, 25, 9)])
def test_split_form_of_the_plate_sums_agrees_with_the_other_forms(M, B, T, D):
    """D >= 7 (default): sweeps that carry the state only + the plate sums as a matrix-core pass, one
    wavefront per (32 sequences, time chunk).  Against (a) the workgroup form of that pass
    (lssm_stats_form = 0) and, for D <= 8, (b) the sums carried in the backward sweep's registers
    (lssm_split_from = 9): same bound trace and moments to rounding."""
    from bayespy_amd.device import get_runtime
    y, x0, c0 = _data(M, B, T, D, seed=7 * D + M)
    rt = get_runtime()
    variants = [(7, 1), (7, 0)] + ([(9, 1)] if D <= 8 else [(7, 2)])      # (2: sums outside the backward sweep)
    out = []
    try:
        for frm, form in variants:
            rt.lib.vmp_tune_set(b'lssm_split_from', frm)
            rt.lib.vmp_tune_set(b'lssm_stats_form', 1 if form == 2 else form)
            rt.lib.vmp_tune_set(b'lssm_fuse_stats', 0 if form != 1 else 1)    # (1: fused whatever the size)
            Q = _build(y, x0, c0, True)
            assert type(Q.plans[0]).__name__ == 'LSSMPlan'
            Q.update(repeat=3, verbose=False)
            out.append((np.array(Q.L[:3]), np.asarray(Q['X'].u[0]), np.asarray(Q['C'].u[0])))
    finally:
        rt.lib.vmp_tune_set(b'lssm_split_from', 7)
        rt.lib.vmp_tune_set(b'lssm_stats_form', 1)
        rt.lib.vmp_tune_set(b'lssm_fuse_stats', 2)
    for o in out[1:]:
        np.testing.assert_allclose(out[0][0], o[0], rtol=1e-10)
        np.testing.assert_allclose(out[0][1], o[1], rtol=1e-8, atol=1e-8 * np.abs(o[1]).max())
        np.testing.assert_allclose(out[0][2], o[2], rtol=1e-8, atol=1e-8 * np.abs(o[2]).max())
