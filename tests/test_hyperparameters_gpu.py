"""
GPU: the four fused model families with every constant they read from their nodes' parents away from
the demos' defaults -- Gamma priors (0.5, 2.0) / (3.0, 0.1), a latent prior precision of 2.5, a
non-uniform Dirichlet, prior precision of the means, Wishart degrees and a non-diagonal scale, prior mean
and non-diagonal prior precision of the first state, innovation precisions away from one -- against the
live-reference traces of tests/golden/hyper_probes.npz (tests/models.py run_hyper_probes runs unchanged on
both sides; oracle/make_golden.py hyper_probes_case).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('engine', ['fused', 'generic'])
def test_non_default_hyperparameters_match_reference(golden_dir, engine):
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import run_hyper_probes
    g = np.load(os.path.join(golden_dir, 'hyper_probes.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}
    seen = []

    class Spy(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            seen.append(type(self.plans[0]).__name__)
    kw = {'engine': 'generic'} if engine == 'generic' else {}
    res = run_hyper_probes(nodes, Spy, inp, **kw)
    if engine == 'fused':
        assert seen == ['PCAPlan', 'MaskedPCAPlan', 'GMMPlan', 'LSSMPlan', 'LSSMPlan']
    else:
        assert set(seen) == {'GenericPlan'}
    for tag in ('pca', 'mpca', 'gmm', 'lssm', 'lssmnu'):
        np.testing.assert_allclose(res[tag + '_L'], g[tag + '_L'], rtol=1e-9, err_msg=tag)
    for key in g.files:
        if key.startswith('in_') or key.endswith('_L'):
            continue
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)
