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
