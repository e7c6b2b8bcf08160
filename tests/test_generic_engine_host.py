"""
CPU: the HOST LOGIC of the generic engine -- message routing, lazily evaluated sums and
contractions, contraction planning, the fused shared-covariance update with its carried plate sums,
the extension hook -- on the NumPy double of the generic entry points (tests/host_generic.py),
against the same live-reference goldens the GPU tests use.  No kernel is exercised here (the GPU
tests do that); the size thresholds of the planning code are lowered so that the small golden models
take the paths large models take.
"""
import importlib

import numpy as np
import pytest

import host_generic


@pytest.fixture
def host_engine(monkeypatch):
    for k in ('BAYESPY_AMD_LAZY_DOT_MIN', 'BAYESPY_AMD_PAIRWISE_MIN', 'BAYESPY_AMD_HOIST_MIN'):
        monkeypatch.setenv(k, '1')
    from bayespy_amd.utils import misc
    monkeypatch.setattr(misc, '_MEMO_MIN', 1)
    rt = host_generic.install()
    yield rt
    host_generic.uninstall()


def _gpu_module(name):
    return importlib.import_module(name)


@pytest.mark.parametrize('test', [
    'test_quickstart_known_answer_on_device', 'test_masked_pca_matches_reference',
    'test_vector_gaussian_ard_matches_reference', 'test_gaussian_wishart_matches_reference',
    'test_pca_block_through_generic_engine', 'test_mixture_of_gaussian_ard_matches_reference',
    'test_count_probability_and_add_nodes_match_reference',
    'test_gaussian_gamma_nodes_match_reference', 'test_hierarchical_wishart_matches_reference'])
def test_generic_engine_cases_on_the_host_double(host_engine, golden_dir, test):
    getattr(_gpu_module('test_generic_engine_gpu'), test)(golden_dir)
    assert host_engine.lib.calls.get('vmp_ewise', 0) > 0


@pytest.mark.parametrize('name', ['gmm_n400_d3_k4', 'gmm_n3000_d8_k16'])
def test_mixture_on_the_host_double(host_engine, golden_dir, name):
    _gpu_module('test_generic_engine_gpu').test_gaussian_mixture_matches_reference(
        golden_dir, name, 'generic')


@pytest.mark.parametrize('name,K', [('pca_n500_d6_k3', 3), ('pca_n777_d20_k5', 5)])
def test_pca_takes_the_fused_update_and_its_carried_sums(host_engine, golden_dir, name, K):
    """The PCA model on the generic engine: X and W are updated by vmp_gaussian_shared_update (X with
    the data array of the Dot message streamed by the pass itself), and after the first sweep no
    contraction over the plates is launched any more -- the messages to W and tau and the bound
    terms are served by the plate sums the pass made (GenericPlan._seed_sums)."""
    import os
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from bayespy_amd.utils import misc
    from models import build_pca
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    Q = build_pca(nodes, VB, g['y'], g['x0'], K, engine='generic')
    Q.update(repeat=1, verbose=False)
    lib = host_engine.lib
    assert lib.calls.get('vmp_gaussian_shared_update', 0) == 2          # W (rows), X (data streamed)
    N = g['y'].shape[1]
    big = []
    orig = misc._launch_sum_multiply

    def spy(arrays, shape, red, keep, scale=1.0):
        memo = misc._CUR_MEMO[0]
        hit = memo is not None and len(red) > 0 and \
            misc._canonical_signature(arrays, shape, red, scale) in memo
        if not hit and any(N in a.shape for a in arrays) and len(red) > 0:
            big.append([tuple(a.shape) for a in arrays])
        return orig(arrays, shape, red, keep, scale)
    misc._launch_sum_multiply = spy
    try:
        Q.update(repeat=int(g['n_iter']) - 1, verbose=False)
    finally:
        misc._launch_sum_multiply = orig
    assert big == [], big
    np.testing.assert_allclose(Q.L[:Q.iter], g['L'], rtol=1e-9)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:Q.iter], g['L_' + k], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(Q['X'].u[0], g['X_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['W'].u[1], g['W_u1'], rtol=1e-8, atol=1e-10)
    # the natural parameters and the log-normaliser are formed on demand from (phi1, <x>)
    np.testing.assert_allclose(Q['W'].phi[0], g['W_phi0'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Q['X'].g, g['X_g'], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(Q['W'].g, g['W_g'], rtol=1e-8, atol=1e-9)


def test_rotation_keeps_the_form_of_the_fused_state(host_engine, golden_dir):
    _gpu_module('test_generic_engine_gpu').test_rotation_parameter_expansion_matches_reference(
        golden_dir, 'generic')


def test_extension_hook_on_the_host_double(host_engine, golden_dir):
    _gpu_module('test_extension_gpu').test_user_defined_node_through_the_extension_hook(golden_dir)
