"""
GPU: node-level update sequences out of constructor order (VB.update(*nodes), vmp.py:132-172) --
one node updated twice in a row, the bound evaluated in between -- on the PCA, missing-data PCA,
mixture and state-space models, against the live-reference traces of tests/golden/order_probes.npz
(oracle/make_golden.py order_probes_case; tests/models.py run_order_probes runs unchanged on both
sides).  Whatever a fused block queues or caches, every update sees the latest moments of its
Markov blanket.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('engine', ['fused', 'fused-stream', 'generic'])
def test_update_sequences_match_reference(golden_dir, engine, monkeypatch):
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_order_probes
    g = np.load(os.path.join(golden_dir, 'order_probes.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    kw = {'engine': 'generic'} if engine == 'generic' else {}
    if engine == 'fused-stream':
        monkeypatch.setenv('BAYESPY_AMD_PCA_STATS', 'stream')
    seen = []
    if engine != 'generic':
        class Spy(VB):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                seen.append(type(self.plans[0]).__name__)
        res = run_order_probes(nodes, Spy, inp, **kw)
        assert seen == ['PCAPlan', 'MaskedPCAPlan', 'GMMPlan', 'LSSMPlan']
    else:
        res = run_order_probes(nodes, VB, inp, **kw)
    for tag in ('pca', 'mpca', 'gmm', 'lssm'):
        np.testing.assert_allclose(res[tag + '_L'], g[tag + '_L'], rtol=1e-9, err_msg=tag)
    for key in ('pca_W_u0', 'pca_X_u0', 'mpca_W_u0', 'mpca_X_u0', 'gmm_z_u0', 'gmm_mu_u0',
                'lssm_X_u0', 'lssm_A_u0'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)


@pytest.mark.parametrize('engine', ['fused', 'fused-stream', 'generic'])
def test_reobserving_data_keeps_the_posteriors(golden_dir, engine, monkeypatch):
    """Y.observe(new data) between updates changes Y only (stochastic.py:223-273): live-reference
    trace tests/golden/reobserve.npz; the fused PCA block recomputes the data statistics and the
    messages of the new data with the current <x>, no restart and no warning."""
    import warnings
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_reobserve_case
    g = np.load(os.path.join(golden_dir, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    if engine == 'fused-stream':
        monkeypatch.setenv('BAYESPY_AMD_PCA_STATS', 'stream')
    kw = {'engine': 'generic'} if engine == 'generic' else {}
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        res = run_reobserve_case(nodes, VB, inp, **kw)
    np.testing.assert_allclose(res['L'], g['L'], rtol=1e-9)
    np.testing.assert_allclose(res['L_mid'], g['L_mid'], rtol=1e-9)
    np.testing.assert_allclose(res['L_w'], g['L_w'], rtol=1e-9)
    np.testing.assert_allclose(res['W_u0'], g['W_u0'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(res['X_u0'], g['X_u0'], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_reobserving_data_of_the_state_space_model(golden_dir, engine):
    """The same for the fused state-space block: new observations, q(X) and every other posterior kept
    (tests/golden/reobserve.npz, lssm_* entries)."""
    import warnings
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_reobserve_lssm_case
    g = np.load(os.path.join(golden_dir, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    seen = []

    class Spy(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            seen.append(type(self.plans[0]).__name__)
    kw = {'engine': 'generic'} if engine == 'generic' else {}
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        res = run_reobserve_lssm_case(nodes, Spy, inp, **kw)
    assert seen == (['LSSMPlan'] if engine == 'fused' else ['GenericPlan'])
    np.testing.assert_allclose(res['lssm_L'], g['lssm_L'], rtol=1e-9)
    np.testing.assert_allclose(res['lssm_L_mid'], g['lssm_L_mid'], rtol=1e-9)
    np.testing.assert_allclose(res['lssm_L_c'], g['lssm_L_c'], rtol=1e-9)
    for key in ('lssm_X_u0', 'lssm_C_u0', 'lssm_A_u0'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)


@pytest.mark.parametrize('engine', ['fused', 'fused-stream', 'generic'])
def test_reinitialising_a_node_keeps_the_other_posteriors(golden_dir, engine, monkeypatch):
    """initialize_from_value on X, later on W, between updates (expfamily.py:193-204): live-reference
    trace tests/golden/reobserve.npz (ri_*)."""
    import warnings
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_reinitialise_case
    g = np.load(os.path.join(golden_dir, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    if engine == 'fused-stream':
        monkeypatch.setenv('BAYESPY_AMD_PCA_STATS', 'stream')
    kw = {'engine': 'generic'} if engine == 'generic' else {}
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        res = run_reinitialise_case(nodes, VB, inp, **kw)
    np.testing.assert_allclose(res['ri_steps'], g['ri_steps'], rtol=1e-9)
    np.testing.assert_allclose(res['ri_L'], g['ri_L'], rtol=1e-9)
    for key in ('ri_W_u0', 'ri_X_u0', 'ri_alpha_u0'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)
