"""
GPU: PCA with missing values (demos/pca.py:80-82) against the live-reference traces of
tests/golden/masked_pca.npz (oracle/make_golden.py masked_pca_case; the model script of
tests/models.py runs unchanged on both sides).  The data carry NaN at the missing entries.

Bars: lower bound rtol 1e-9 (north star: 1e-5); posterior moments rtol 1e-7.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ELBO_RTOL = 1e-9


def _inputs(golden_dir):
    g = np.load(os.path.join(golden_dir, 'masked_pca.npz'))
    return g, {k[3:]: g[k] for k in g.files if k.startswith('in_')}


def _check(res, g, tags, mom_rtol=1e-7):
    for tag in tags:
        np.testing.assert_allclose(res[tag + '_L'], g[tag + '_L'], rtol=ELBO_RTOL, err_msg=tag)
        for nm in ('Y', 'W', 'X', 'tau', 'alpha'):
            key = '%s_L_%s' % (tag, nm)
            np.testing.assert_allclose(res[key], g[key], rtol=1e-8, atol=1e-7, err_msg=key)
        for key in ('W_u0', 'W_u1', 'X_u0', 'X_u1_first', 'tau_u', 'alpha_u0', 'alpha_u1',
                    'Y_u0', 'Y_u1'):
            k = '%s_%s' % (tag, key)
            if k in g.files:
                assert np.all(np.isfinite(res[k])), k
                np.testing.assert_allclose(res[k], g[k], rtol=mom_rtol, atol=1e-9, err_msg=k)


def test_generic_engine_masked_pca_with_nan_placeholders(golden_dir):
    """NaN at the masked entries never reaches a message or the bound, and the latent plates
    of partially observed nodes are updated like in the reference (stochastic.py:223-282)."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_masked_pca_cases
    g, inp = _inputs(golden_dir)
    res = run_masked_pca_cases(nodes, VB, inp, only=('m0', 'm1', 'm2', 'po'), engine='generic')
    _check(res, g, ('m0', 'm1', 'm2'))
    np.testing.assert_allclose(res['po_L'], g['po_L'], rtol=ELBO_RTOL)
    for key in ('po_z_u0', 'po_z_u1', 'po_mu_u', 'po_tau_u'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-8, err_msg=key)


def test_fused_block_matches_reference(golden_dir):
    """The fused missing-data block (vmp_mpca_*) on the live-reference traces: four sizes incl.
    D=128, K=32, ragged K=17, a plate without observations, NaN at the missing entries."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.inference.plans.masked_pca import MaskedPCAPlan
    from models import build_masked_pca, run_masked_pca_cases
    g, inp = _inputs(golden_dir)
    Q = build_masked_pca(nodes, VB, inp['m1_y'], inp['m1_mask'], inp['m1_x0'])
    assert isinstance(Q.plans[0], MaskedPCAPlan)
    res = run_masked_pca_cases(nodes, VB, inp, only=('m0', 'm1', 'm2', 'm3'))
    # q of Y's latent entries: the fused block evaluates it when read (current W, X), the
    # reference as of Y's last update -- compared separately below
    for k in list(res):
        if k.endswith('_Y_u0') or k.endswith('_Y_u1'):
            del res[k]
    g2 = {k: g[k] for k in g.files if not (k.endswith('_Y_u0') or k.endswith('_Y_u1'))}

    class G(dict):
        files = list(g2)
    _check(res, G(g2), ('m0', 'm1', 'm2', 'm3'))


@pytest.mark.parametrize('N,D,K,keep', [(1, 1, 1, 1.0), (33, 5, 2, 0.7), (100, 20, 16, 0.5),
                                        (77, 33, 17, 0.8), (1000, 128, 32, 0.9),
                                        (5000, 64, 16, 0.9), (4099, 100, 9, 0.3)])
def test_fused_block_vs_oracle_ragged_sizes(N, D, K, keep):
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_masked_pca
    from oracle.masked_pca import MaskedPCAOracle
    rs = np.random.RandomState(N + D + K)
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < keep
    y = np.where(mask, y, np.nan)
    x0 = rs.normal(size=(N, K))
    Q = build_masked_pca(nodes, VB, y, mask, x0)
    assert type(Q.plans[0]).__name__ == 'MaskedPCAPlan'
    iters = 3
    Q.update(repeat=iters, verbose=False)
    o = MaskedPCAOracle(y, mask, x0)
    o.iterate(iters)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=ELBO_RTOL)
    for nm in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[nm]][:iters], [t[nm] for t in o.L_terms], rtol=1e-8,
                                   atol=1e-7, err_msg=nm)
    W, X = Q['W'], Q['X']
    np.testing.assert_allclose(W.u[0][:, 0], o.W, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(W.u[1][:, 0], o.WW, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(X.u[0][0], o.X, rtol=1e-7, atol=1e-10)
    # second moments of X: re-derived per plate on request
    xx = Q.plans[0].x_second_moments(0, min(N, 50))
    tau_x = float(Q.plans[0].state[Q.plans[0].layout.off_scal + 6].item())
    WWf = o.WW.reshape(D, K * K)
    lam = np.eye(K)[None] + tau_x * (mask[:, :min(N, 50)].T.astype(float) @ WWf).reshape(-1, K, K)
    cov = np.linalg.inv(lam)
    np.testing.assert_allclose(xx, cov + o.X[:min(N, 50), :, None] * o.X[:min(N, 50), None, :],
                               rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_plates_and_dimensions_without_any_observation_match_reference(golden_dir, engine):
    """Erasure patterns at the edges (tests/golden/masked_pca_erasures.npz, live reference):
    plates with no observed dimension (their q(x_n) falls back to the prior), dimensions observed on
    no plate -- ignored plates of W: updated to diag(1/<alpha>), but left out of the message to
    alpha and of W's bound term (node.py:457-526, :624-650) -- and a dimension seen once."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_erasure_cases, build_masked_pca
    g = np.load(os.path.join(golden_dir, 'masked_pca_erasures.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    kw = {} if engine == 'fused' else {'engine': 'generic'}
    Q = build_masked_pca(nodes, VB, inp['e0_y'], inp['e0_mask'], inp['e0_x0'], **kw)
    assert type(Q.plans[0]).__name__ == ('MaskedPCAPlan' if engine == 'fused' else 'GenericPlan')
    tags = ('e0', 'e1', 'e2') if engine == 'fused' else ('e0', 'e1')
    res = run_erasure_cases(nodes, VB, inp, only=tags, **kw)
    if engine == 'fused':
        # q of Y's latent entries is evaluated on request there (test_fused_block_predictive_...)
        res = {k: v for k, v in res.items() if not (k.endswith('_Y_u0') or k.endswith('_Y_u1'))}
    keep = [k for k in g.files if engine != 'fused' or not (k.endswith('_Y_u0') or k.endswith('_Y_u1'))]

    class G(dict):
        files = keep
    _check(res, G({k: g[k] for k in keep}), tags)
    for tag in tags:
        mask = inp[tag + '_mask']
        N = mask.shape[1]
        empty = ~mask.any(axis=0)
        assert empty.sum() >= 3
        assert np.all(res[tag + '_X_u0'][0][empty] == 0.0)       # no message reaches them
        assert np.all(res[tag + '_W_u0'][2, 0] == 0.0)


def test_fused_block_chunking_and_device_inputs():
    """Several chunks per pass == one chunk (up to summation order); data and mask may already be
    device tensors (NaN at the missing entries)."""
    import torch
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime
    from models import build_masked_pca
    rs = np.random.RandomState(3)
    D, N, K = 24, 1000, 6
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < 0.85
    y = np.where(mask, y, np.nan)
    x0 = rs.normal(size=(N, K))
    Ls = []
    for chunk, dev in ((1 << 20, False), (96, False), (160, True)):
        os.environ['BAYESPY_AMD_MPCA_CHUNK'] = str(chunk)
        try:
            if dev:
                d = get_runtime().device
                Q = build_masked_pca(nodes, VB, torch.from_numpy(y).to(d),
                                     torch.from_numpy(mask).to(d), x0)
            else:
                Q = build_masked_pca(nodes, VB, y, mask, x0)
            Q.update(repeat=3, verbose=False)
        finally:
            del os.environ['BAYESPY_AMD_MPCA_CHUNK']
        Ls.append((Q.L[:3].copy(), Q['W'].u[0].copy(), Q['X'].u[0].copy()))
    for L, w, x in Ls[1:]:
        np.testing.assert_allclose(L, Ls[0][0], rtol=1e-12)
        np.testing.assert_allclose(w, Ls[0][1], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(x, Ls[0][2], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize('masked', [True, False])
def test_moments_of_the_deterministic_product_are_read_out_on_request(masked):
    """F.u = [<w.x>, <(w.x)^2>] (dot.py:316-415) is never formed by the fused blocks' updates, but a
    script may read it (predictions at the missing entries): same values as the generic engine."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_masked_pca
    rs = np.random.RandomState(8)
    D, N, K = 12, 300, 4
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = (rs.rand(D, N) < 0.8) if masked else True
    x0 = rs.normal(size=(N, K))
    res = []
    for engine in (None, 'generic'):
        Q = build_masked_pca(nodes, VB, y, mask, x0, engine=engine)
        Q.update(repeat=3, verbose=False)
        res.append((type(Q.plans[0]).__name__, [np.asarray(u) for u in Q['F'].u]))
    assert res[0][0] == ('MaskedPCAPlan' if masked else 'PCAPlan') and res[1][0] == 'GenericPlan'
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_allclose(a, np.broadcast_to(b, a.shape), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('N,D,K', [(777, 40, 32), (333, 17, 20), (64, 128, 32), (203, 9, 7),
                                   (150, 12, 16), (1, 3, 2), (1030, 33, 25)])
def test_plate_stage_variants_agree(N, D, K):
    """The per-plate stage has four forms (vmp_tune_set): four plates per wavefront on the 4x4x4
    matrix instruction (default), two / one plate per wavefront on the 16x16x4 sweep, and the
    vector-ALU Gauss-Jordan with a plate per 16 lanes (16 < K <= 32); the M_d GEMM has two (4x4x4
    row-split default, 16x16x4 column-split).  Same inputs -> the same bound, moments and
    rotation statistic to round-off."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime
    from models import build_masked_pca
    rs = np.random.RandomState(N + K)
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < 0.8
    x0 = rs.normal(size=(N, K))
    lib = get_runtime().lib
    res = []
    defaults = {'mpca_blk4': 1, 'mpca_sweep_nm': 2, 'mpca_rows': 0, 'mpca_stats3': 1}
    try:
        for knobs in ({}, {'mpca_blk4': 0}, {'mpca_blk4': 0, 'mpca_sweep_nm': 1},
                      {'mpca_blk4': 0, 'mpca_rows': 1}, {'mpca_stats3': 0}):
            for k, v in knobs.items():
                lib.vmp_tune_set(k.encode(), v)
            Q = build_masked_pca(nodes, VB, y, mask, x0)
            Q.update(repeat=3, verbose=False)
            st = Q.plans[0].rotation_statistics(Q['X'])
            res.append((Q.L[:3].copy(), Q['W'].u[0].copy(), Q['X'].u[0].copy(),
                        np.array(st['XX'])))
            for k in knobs:
                lib.vmp_tune_set(k.encode(), defaults[k])
    finally:
        for k, v in defaults.items():
            lib.vmp_tune_set(k.encode(), v)
    for L, w, x, xx in res[1:]:
        np.testing.assert_allclose(L, res[0][0], rtol=1e-11)
        np.testing.assert_allclose(w, res[0][1], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(x, res[0][2], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(xx, res[0][3], rtol=1e-9, atol=1e-9)


def test_fused_block_predictive_moments_of_missing_entries(golden_dir):
    """Y.u at the missing entries after an explicit Y.update(): <f>, <f^2> + 1/<tau> from the
    current W, X, tau -- the reference's value when Y is updated last."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_masked_pca
    from oracle.masked_pca import MaskedPCAOracle
    g, inp = _inputs(golden_dir)
    y, mask, x0 = inp['m1_y'], inp['m1_mask'], inp['m1_x0']
    Q = build_masked_pca(nodes, VB, y, mask, x0)
    Q.update(repeat=3, verbose=False)
    Q['Y'].update()
    u0, u1 = Q['Y'].u
    o = MaskedPCAOracle(y, mask, x0)
    o.iterate(3)
    f, tau = o.predictive_Y()
    np.testing.assert_allclose(u0[~mask], f[~mask], rtol=1e-7, atol=1e-10)
    np.testing.assert_array_equal(u0[mask], y[mask])
    assert np.all(np.isfinite(u1)) and np.all(u1[~mask] > f[~mask] ** 2)


def test_fused_block_direct_oracle_parity_n2e5():
    """D=128, K=32 (the headline dims), N=2e5, 10 % missing: three iterations against the
    chunked oracle on the same data."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_masked_pca
    from oracle.masked_pca import MaskedPCAOracle
    rs = np.random.RandomState(42)
    D, N, K = 128, 200_000, 32
    y = rs.normal(size=(D, K)) @ rs.normal(size=(K, N)) + 0.1 * rs.normal(size=(D, N))
    mask = rs.rand(D, N) < 0.9
    x0 = rs.normal(size=(N, K))
    Q = build_masked_pca(nodes, VB, y, mask, x0)
    Q.update(repeat=3, verbose=False)
    o = MaskedPCAOracle(y, mask, x0, chunk=1 << 13)
    o.iterate(3)
    np.testing.assert_allclose(Q.L[:3], np.array(o.L), rtol=ELBO_RTOL)
    np.testing.assert_allclose(Q['W'].u[0][:, 0], o.W, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q.plans[0].Xm[:N:97, :K].cpu().numpy(), o.X[::97], rtol=1e-7,
                               atol=1e-10)


def test_fused_block_config_scale_default_chunks_direct_oracle_parity():
    """D=128, K=32, N = 2.5e6 + 77 (ragged), 10 % missing, the DEFAULT chunk of 2^20
    plates -- two full chunk boundaries and a ragged tail are crossed with the default launch
    parameters -- against the chunked oracle on the same data (generated on the device, NaN
    at the missing entries), two iterations: bound, per-node terms, <w>, a strided sample of <x>."""
    import torch
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.device import get_runtime
    from models import build_masked_pca
    from oracle.masked_pca import MaskedPCAOracle
    dev = get_runtime().device
    D, N, K = 128, 2_500_077, 32
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    y = w @ torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y += 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    mask = torch.rand(D, N, generator=g, device=dev) >= 0.1
    y[~mask] = float('nan')
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    Q = build_masked_pca(nodes, VB, y, mask, x0)
    plan = Q.plans[0]
    assert type(plan).__name__ == 'MaskedPCAPlan'
    assert plan.chunk == 1 << 20 and N > 2 * plan.chunk       # default chunking, > 2 boundaries
    iters = 2
    Q.update(repeat=iters, verbose=False)
    yh, mh, xh = y.cpu().numpy(), mask.cpu().numpy(), x0.cpu().numpy()
    del y, mask, x0
    o = MaskedPCAOracle(yh, mh, xh, chunk=1 << 15)
    o.iterate(iters)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=ELBO_RTOL)
    # (the Gamma terms are differences of O(a log b) ~ 1e9-sized numbers at a = N D / 2: their
    # own rounding is ~1e-16 of THAT size on both sides)
    for nm in ('Y', 'W', 'X', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[nm]][:iters], [t[nm] for t in o.L_terms], rtol=1e-8,
                                   atol=1e-7 + 2e-14 * N * D, err_msg=nm)
    np.testing.assert_allclose(Q['W'].u[0][:, 0], o.W, rtol=1e-7, atol=1e-10)
    # plates on both sides of each chunk boundary and the ragged tail
    idx = np.r_[0:N:9973, (1 << 20) - 2:(1 << 20) + 2, (2 << 20) - 2:(2 << 20) + 2, N - 3:N]
    xm = plan.Xm[torch.from_numpy(idx).to(dev), :K].cpu().numpy()
    np.testing.assert_allclose(xm, o.X[idx], rtol=1e-7, atol=1e-10)


def test_fused_block_rotation_matches_reference(golden_dir):
    """RotationOptimizer / RotateGaussianARD as the VB callback on a model with missing values
    (demos/pca.py:80-94, the demo's default use) on the fused missing-data block."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB, transformations
    from models import build_pca, run_rotation_sequence, check_rotation_results
    g = np.load(os.path.join(golden_dir, 'rotations.npz'))
    y, x0 = g['rotm_y'], g['rotm_x0']
    K = x0.shape[1]
    Q = build_pca(nodes, VB, y, x0, K)
    Q['Y'].observe(y, mask=g['rotm_mask'])
    assert type(Q.plans[0]).__name__ == 'MaskedPCAPlan'
    res = run_rotation_sequence(Q, K, transformations)
    check_rotation_results(res, g, 'rotm')
