"""GPU: the CPU kernel doubles of tests/fake_kernels.py (what the world-2 gloo tests of SURVEY.md
8(e) run on) pinned to the kernels they stand for: every fused plan runs two iterations once on
its double and once on libvmp_hip.so from the same inputs, and the PACKED STATE VECTORS are
compared field by field (offsets from vmp_*_get_layout) together with the plate arrays.  A double
that drifts from the device layout or arithmetic fails here."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu


def fields(L):
    offs = sorted((int(getattr(L, f)), f) for f, _ in L._fields_ if f.startswith('off_'))
    out = []
    for i, (lo, name) in enumerate(offs):
        hi = offs[i + 1][0] if i + 1 < len(offs) else int(L.total)
        out.append((name, lo, hi))
    return out


def compare_states(L, dev, host, skip=(), rtol=1e-9, atol=1e-9):
    bad = []
    for name, lo, hi in fields(L):
        if name in skip or hi <= lo:
            continue
        a, b = dev[lo:hi], host[lo:hi]
        ok = np.isfinite(a) & np.isfinite(b)
        if not np.array_equal(np.isfinite(a), np.isfinite(b)) or \
                not np.allclose(a[ok], b[ok], rtol=rtol, atol=atol):
            with np.errstate(invalid='ignore'):
                bad.append((name, float(np.nanmax(np.abs(a - b)))))
    assert not bad, 'state fields differ between the library and its CPU double: %s' % bad


def test_pca_double_equals_the_kernels():
    import bayespy_amd.nodes as nodes
    from bayespy_amd.device import Runtime
    from bayespy_amd.inference import VB
    from fake_kernels import CPURuntimeKernels
    from models import build_pca
    g = np.load(os.path.join(GOLDEN, 'pca_n777_d20_k5.npz'))
    for stats in ('gram', 'stream'):
        Qd = build_pca(nodes, VB, g['y'], g['x0'], 5)
        Qh = build_pca(nodes, VB, g['y'], g['x0'], 5)
        Qd.plans[0].stats = Qh.plans[0].stats = stats
        rt = Runtime(device='cpu')
        Qh.plans[0]._rt, Qh.plans[0]._kernels = rt, CPURuntimeKernels(rt)
        for Q in (Qd, Qh):
            Q.update(repeat=2, verbose=False)
        pd, ph = Qd.plans[0], Qh.plans[0]
        pd.finish()
        # off_A: the double keeps no copy of <tau> Cov_X <W>^T in the stream form; off_scal holds
        # log-determinants + a status word the double leaves at zero
        compare_states(pd.layout, pd.state.cpu().numpy(), ph.state.numpy(),
                       skip=('off_scal',) + (('off_A', 'off_G') if stats == 'stream' else ()))
        np.testing.assert_allclose(pd.Xd[:5, :777].cpu().numpy(), ph.Xd[:5, :777].numpy(),
                                   rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Qd.L[:2], Qh.L[:2], rtol=1e-11)


def test_gmm_double_equals_the_kernels():
    from test_gmm_plan_host import _build
    from bayespy_amd.inference.plans.gmm import GMMKernels
    from bayespy_amd.device import get_runtime
    g = np.load(os.path.join(GOLDEN, 'gmm_n3000_d8_k16.npz'))
    Qh = _build(g['y'], g['lab0'], 16)
    Qd = _build(g['y'], g['lab0'], 16)
    rt = get_runtime()
    Qd.plans[0]._rt, Qd.plans[0]._kernels = rt, GMMKernels(rt)
    for Q in (Qd, Qh):
        Q.update(repeat=2, verbose=False)
    pd, ph = Qd.plans[0], Qh.plans[0]
    compare_states(pd.layout, pd.state.cpu().numpy(), ph.state.numpy(), skip=('off_scal', 'off_C'),
                   rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(pd.Rd.cpu().numpy().reshape(3000, -1)[:, :16], ph.Rd.numpy().reshape(3000, -1)[:, :16], rtol=1e-8,
                               atol=1e-12)
    np.testing.assert_allclose(Qd.L[:2], Qh.L[:2], rtol=1e-11)


def test_lssm_double_equals_the_kernels():
    from test_lssm_plan_host import _build
    from bayespy_amd.inference.plans.lssm import LSSMKernels
    from bayespy_amd.device import get_runtime
    g = np.load(os.path.join(GOLDEN, 'lssm.npz'))
    for tag, B, nu in (('lssmB', 6, True), ('lssm1', None, False)):
        Qh, _ = _build(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], B, nu)
        Qd, _ = _build(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], B, nu)
        rt = get_runtime()
        Qd.plans[0]._rt, Qd.plans[0]._kernels = rt, LSSMKernels(rt)
        for Q in (Qd, Qh):
            Q.update(repeat=2, verbose=False)
        pd, ph = Qd.plans[0], Qh.plans[0]
        # off_covsums: four D x D sums, then the state of the device recursion between its time
        # segments (S_t; csrc/vmp_lssm.hip), then log|Phi|
        DD = pd.D * pd.D
        L = pd.layout
        sd, sh = pd.state.cpu().numpy(), ph.state.numpy()
        compare_states(L, sd, sh, skip=('off_covsums', 'off_scal'))
        lo = int(L.off_covsums)
        np.testing.assert_allclose(sd[lo:lo + 4 * DD], sh[lo:lo + 4 * DD], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sd[lo + 5 * DD], sh[lo + 5 * DD], rtol=1e-11)
        np.testing.assert_allclose(pd.x_means(), ph.x_means(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Qd.L[:2], Qh.L[:2], rtol=1e-11)


def test_masked_pca_double_equals_the_kernels():
    from test_masked_plan_host import _build, _inputs as _load
    from bayespy_amd.inference.plans.masked_pca import MaskedHIPKernels as MaskedKernels
    from bayespy_amd.device import get_runtime
    g, inp = _load('masked_pca.npz')
    y, mask, x0 = inp['m1_y'], inp['m1_mask'], inp['m1_x0']
    Qh = _build(y, mask, x0)
    Qd = _build(y, mask, x0)
    rt = get_runtime()
    Qd.plans[0]._rt, Qd.plans[0]._kernels = rt, MaskedKernels(rt)
    for Q in (Qd, Qh):
        Q.update(repeat=2, verbose=False)
    pd, ph = Qd.plans[0], Qh.plans[0]
    # panels: the B operands of the device GEMMs in fragment order (the double has no use for them)
    compare_states(pd.layout, pd.state.cpu().numpy(), ph.state.numpy(),
                   skip=('off_panel', 'off_panel_x', 'off_scal'), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(Qd.L[:2], Qh.L[:2], rtol=1e-10)
