"""CPU: the host logic of the four fused plans TOGETHER on their CPU kernel doubles
(tests/fake_kernels.py attach_cpu), driven by the model scripts the GPU tests and the golden
generator share (tests/models.py), against live-reference traces:

* node-level update sequences out of constructor order with the bound in between
  (tests/golden/order_probes.npz) -- whatever a plan queues or caches, every update sees the
  latest moments of its Markov blanket;
* every constant a plan reads from its nodes' parents away from the demos' defaults
  (tests/golden/hyper_probes.npz): Gamma priors, a latent prior precision, a non-uniform
  Dirichlet, Wishart degrees and a non-diagonal scale, prior mean / precision of the first
  state, innovation precisions.
The doubles restate the kernels' arithmetic in NumPy; what is under test here is the plans."""
import os

import numpy as np
import pytest

import bayespy_amd.nodes as nodes
from bayespy_amd.inference import VB

from fake_kernels import attach_cpu


def _cpu_vb(seen, stats=None):
    class CPUVB(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            seen.append(type(self.plans[0]).__name__)
            attach_cpu(self, stats)
    return CPUVB


def _inputs(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    return g, {k[3:]: g[k] for k in g.files if k.startswith('in_')}


@pytest.mark.parametrize('stats', ['gram', 'stream'])
def test_update_sequences_match_reference(golden_dir, stats):
    from models import run_order_probes
    g, inp = _inputs(golden_dir, 'order_probes.npz')
    seen = []
    res = run_order_probes(nodes, _cpu_vb(seen, stats), inp)
    assert seen == ['PCAPlan', 'MaskedPCAPlan', 'GMMPlan', 'LSSMPlan']
    for tag in ('pca', 'mpca', 'gmm', 'lssm'):
        np.testing.assert_allclose(res[tag + '_L'], g[tag + '_L'], rtol=1e-9, err_msg=tag)
    for key in ('pca_W_u0', 'pca_X_u0', 'mpca_W_u0', 'mpca_X_u0', 'gmm_z_u0', 'gmm_mu_u0',
                'lssm_X_u0', 'lssm_A_u0'):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)


def test_non_default_hyperparameters_match_reference(golden_dir):
    from models import run_hyper_probes
    g, inp = _inputs(golden_dir, 'hyper_probes.npz')
    seen = []
    res = run_hyper_probes(nodes, _cpu_vb(seen), inp)
    assert seen == ['PCAPlan', 'MaskedPCAPlan', 'GMMPlan', 'LSSMPlan', 'LSSMPlan']
    for tag in ('pca', 'mpca', 'gmm', 'lssm', 'lssmnu'):
        np.testing.assert_allclose(res[tag + '_L'], g[tag + '_L'], rtol=1e-9, err_msg=tag)
    for key in g.files:
        if key.startswith('in_') or key.endswith('_L'):
            continue
        np.testing.assert_allclose(res[key], g[key], rtol=1e-7, atol=1e-9, err_msg=key)
