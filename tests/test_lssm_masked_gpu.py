"""GPU: the masked state-space block (csrc/vmp_lssmm.hip, plans/lssm_masked.py) against the
live-reference traces of tests/golden/lssm_masked.npz, against oracle/lssm.py at larger sizes, and
against the host build of the same device code (state vector field by field)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from test_lssm_masked_host import build, check_against_golden, golden_of, CASES, WIDE_CASES  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag,B,gamma_nu', CASES + WIDE_CASES)
def test_fused_block_matches_reference(tag, B, gamma_nu):
    """demos/lssm.py's own mask shape (md: (M, T), a fully missing stretch), one mask per sequence
    (mb), a shared mask (ms), erasures (me: a row / a sequence / the end points never observed),
    a single step (m1); 5 ... 8 states (w8 / w6 / w5 / w7: two rows of the blocks per lane); NaN
    placeholders at the masked entries."""
    g = golden_of(tag)
    Q, track = build(g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu,
                     host=False)
    n = len(g[tag + '_L'])
    Q.update(repeat=n, verbose=False)
    check_against_golden(Q, track, g, tag, n)
    np.testing.assert_allclose(Q.l[Q['Y']][:n], g[tag + '_Y_L'], rtol=1e-8, atol=1e-7)


@pytest.mark.parametrize('tag,B,gamma_nu', [('md', None, False), ('mb', 5, True)])
def test_generic_engine_matches_the_same_traces(tag, B, gamma_nu):
    """The same models on the generic message-passing engine (engine='generic')."""
    from bayespy_amd.inference import VB
    g = np.load(os.path.join(GOLDEN, 'lssm_masked.npz'))
    Q, track = build(g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0'], B, gamma_nu,
                     host=False)
    Qg = VB(*Q.model, engine='generic')
    Qg.ignore_bound_checks = True
    n = len(g[tag + '_L'])
    Qg.update(repeat=n, verbose=False)
    np.testing.assert_allclose(Qg.L[:n], g[tag + '_L'], rtol=1e-8)


@pytest.mark.parametrize('M,B,T,D,nu', [(3, 1, 7, 1, False), (5, 3, 9, 2, True), (2, 70, 5, 3, False),
                                        (7, 130, 12, 4, True), (64, 2, 4, 2, False),
                                        (6, 1, 200, 3, False), (8, 1000, 50, 4, False),
                                        (5, 37, 9, 6, True), (32, 3, 5, 8, True),
                                        (9, 70, 7, 5, False), (12, 130, 6, 7, True),
                                        (20, 5, 8, 4, False), (3, 300, 40, 8, False)])
def test_fused_block_vs_oracle(M, B, T, D, nu):
    from oracle.lssm import MaskedLSSMOracle
    rs = np.random.RandomState(M + B + T + D)
    y = rs.normal(size=(M, B, T))
    mask = rs.rand(M, B, T) < 0.6
    mask[0, 0, 0] = True
    if B > 2:
        mask[:, 1] = False                         # a sequence without data
    x0 = rs.normal(size=(B, T, D))
    c0 = rs.normal(size=(M, 1, 1, D))
    Q, track = build(np.where(mask, y, np.nan), mask, x0, c0, B, nu, host=False)
    Q.update(repeat=3, verbose=False)
    o = MaskedLSSMOracle(y, mask, x0, c0.reshape(M, D), nu_prior=(1e-3, 1e-3) if nu else None)
    o.iterate(3)
    np.testing.assert_allclose(Q.L[:3], o.L, rtol=1e-9)
    for nm in ('Y', 'X', 'A', 'C', 'tau', 'alpha', 'gamma') + (('nu',) if nu else ()):
        np.testing.assert_allclose(Q.l[Q[nm]][:3], [t[nm] for t in o.L_terms], rtol=1e-8, atol=1e-6,
                                   err_msg=nm)
    np.testing.assert_allclose(track['X'].u[0], o.X, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(track['X'].u[1], o.P, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(track['C'].u[0].reshape(M, D), o.Cm, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('M,B,T,D,lanes', [(6, 75, 20, 3, 4), (6, 75, 20, 3, 1), (8, 33, 12, 4, 4),
                                           (12, 40, 10, 4, 4), (7, 21, 9, 6, 4), (5, 18, 8, 8, 4)])
def test_device_kernels_equal_their_host_build(M, B, T, D, lanes):
    """The packed state vector and the plate arrays after two iterations: libvmp_hip.so against the
    g++ build of the same header (tests/host_build.py) -- pins the CPU double of this block to the
    kernels (only the order of the plate sums, the contraction of products into fused
    multiply-adds outside the sweeps and the device's reciprocal differ).  Four lanes per sequence
    (statistics carried by the backward sweep for M <= 8, D <= 4; a separate pass otherwise) and one
    thread per sequence."""
    from bayespy_amd import _lib
    rs = np.random.RandomState(3)
    y = rs.normal(size=(M, B, T))
    mask = rs.rand(M, B, T) < 0.5
    x0, c0 = rs.normal(size=(B, T, D)), rs.normal(size=(M, 1, 1, D))
    _lib.load().vmp_tune_set(b'lssmm_lanes', lanes)
    try:
        Qd, _ = build(y, mask, x0, c0, B, True, host=False)
        Qh, _ = build(y, mask, x0, c0, B, True, host=True)
        for Q in (Qd, Qh):
            Q.update(repeat=2, verbose=False)
    finally:
        _lib.load().vmp_tune_set(b'lssmm_lanes', 0)
    pd, ph = Qd.plans[0], Qh.plans[0]
    L = pd.layout
    sd, sh = pd.state.cpu().numpy(), ph.state.numpy()
    for name in ('off_tau', 'off_gamma', 'off_alpha', 'off_nu', 'off_Cm', 'off_CovC', 'off_ldC',
                 'off_SCC', 'off_Am', 'off_AA', 'off_ldA', 'off_tab', 'off_setup', 'off_raw',
                 'off_L'):
        lo = int(getattr(L, name))
        nxt = min([int(getattr(L, f)) for f, _ in L._fields_
                   if f.startswith('off_') and int(getattr(L, f)) > lo] + [int(L.total)])
        np.testing.assert_allclose(sd[lo:nxt], sh[lo:nxt], rtol=1e-10, atol=1e-11, err_msg=name)
    np.testing.assert_allclose(pd.x_means(), ph.x_means(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pd.x_second_moments(), ph.x_second_moments(), rtol=1e-10, atol=1e-12)
    assert np.array_equal(pd.Mw.cpu().numpy(), ph.Mw.numpy())
    assert np.array_equal(pd.Yt.cpu().numpy(), ph.Yt.numpy())


def test_device_resident_inputs_reobserve_and_checkpoint(tmp_path):
    import torch
    from oracle.lssm import MaskedLSSMOracle
    rs = np.random.RandomState(11)
    M, B, T, D = 4, 9, 15, 2
    y = rs.normal(size=(M, B, T))
    mask = rs.rand(M, B, T) < 0.7
    x0, c0 = rs.normal(size=(B, T, D)), rs.normal(size=(M, 1, 1, D))
    dev = torch.device('cuda', 0)
    Q, track = build(torch.from_numpy(y).to(dev), torch.from_numpy(mask).to(dev), x0, c0, B, False,
                     host=False)
    Q.update(repeat=2, verbose=False)
    o = MaskedLSSMOracle(y, mask, x0, c0.reshape(M, D))
    o.iterate(2)
    np.testing.assert_allclose(Q.L[:2], o.L, rtol=1e-10)
    fn = str(tmp_path / 'ck.npz')
    Q.save(filename=fn)
    Q.update(repeat=2, verbose=False)
    L_after = np.array(Q.L[2:4])
    Q2, _ = build(y, mask, x0, c0, B, False, host=False)
    Q2.load(filename=fn)
    Q2.update(repeat=2, verbose=False)
    np.testing.assert_array_equal(np.array(Q2.L[Q2.iter - 2:Q2.iter]), L_after)


def test_config_scale_masked_properties():
    """B = 1e4 sequences x T = 1e3 steps (the masked leg of bench.py), M = 8, D = 4, 30 % missing and
    a stretch without data: the first 64 sequences against the oracle run on them alone is not
    possible (C, A, tau couple the sequences), so: the bound increases monotonically, is finite,
    and the sums of the block equal independent torch reductions of its own plate arrays."""
    import torch
    from tools import workloads
    Q, info = workloads.build_lssm_masked(B=10_000, T=1000, M=8, D=4)
    Q.update(repeat=4, verbose=False)
    L = np.array(Q.L[:4])
    assert np.all(np.isfinite(L)) and np.all(np.diff(L) > -1e-6 * np.abs(L[:-1]))
    p = Q.plans[0]
    st = p.state.cpu().numpy()
    o = p._raw_offsets()
    raw = st[int(p.layout.off_raw):]
    NS, D, M, T, BL = p.NS, p.D, p.M, p.T, p.BL
    P = p.Pm.view(T, NS, BL)[:, :, :p.B]
    Z = p.Z.view(T, D, BL)[:, :, :p.B]
    np.testing.assert_allclose(raw[o['sumP']:o['sumP'] + NS], P.sum(dim=(0, 2)).cpu().numpy(),
                               rtol=1e-10)
    Yt = p.Yt.view(T, M, BL)[:, :, :p.B]
    syx = torch.einsum('tmb,tdb->md', Yt, Z).cpu().numpy()
    np.testing.assert_allclose(raw[o['Syx']:o['Syx'] + M * D].reshape(M, D), syx, rtol=1e-9,
                               atol=1e-6)
    bits = ((p.Mw.view(T, BL)[:, :p.B].unsqueeze(1) >> torch.arange(M, device=P.device)
             .view(1, M, 1)) & 1).to(torch.float64)
    xx = torch.einsum('tmb,tsb->ms', bits, P).cpu().numpy()
    np.testing.assert_allclose(raw[o['XX']:o['XX'] + M * NS].reshape(M, NS), xx, rtol=1e-9,
                               atol=1e-6)


def test_rotations_on_the_masked_block_match_reference():
    """demos/lssm.py as it ships: array mask + rotation after every iteration, on the device."""
    from test_lssm_masked_host import run_masked_rotation_case
    run_masked_rotation_case(host=False)
