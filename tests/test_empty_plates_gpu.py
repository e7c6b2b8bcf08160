"""
GPU: an EMPTY observation plate (N = 0 / no sequences) -- what a rank of a sharded run holds when the
plate has fewer elements than ranks.  The reference runs such models (NumPy handles the zero-size
arrays): its PCA bound at N = 0, D = 4, K = 2 is -8.0890447 (live reference, and oracle/pca.py);
the fused blocks must neither crash nor launch on null pointers.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pca_blocks_with_an_empty_plate():
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_pca, build_masked_pca
    from oracle.pca import PCAOracle
    from oracle.masked_pca import MaskedPCAOracle
    y, x0 = np.zeros((4, 0)), np.zeros((0, 2))
    Q = build_pca(nodes, VB, y, x0, 2)
    assert type(Q.plans[0]).__name__ == 'PCAPlan'
    Q.update(repeat=2, verbose=False)
    o = PCAOracle(y, x0)
    o.iterate(2)
    np.testing.assert_allclose(Q.L[:2], o.L, rtol=1e-12)
    np.testing.assert_allclose(Q.L[0], -8.089044696714964, rtol=1e-12)
    assert np.asarray(Q['X'].u[0]).shape[-2:] == (0, 2)
    assert np.all(np.asarray(Q['W'].u[0]) == 0.0)
    m = np.zeros((4, 0), dtype=bool)
    Q = build_masked_pca(nodes, VB, y, m, x0)
    assert type(Q.plans[0]).__name__ == 'MaskedPCAPlan'
    Q.update(repeat=2, verbose=False)
    o = MaskedPCAOracle(y, m, x0)
    o.iterate(2)
    np.testing.assert_allclose(Q.L[:2], o.L, rtol=0, atol=1e-12)


def test_mixture_block_with_an_empty_plate():
    from test_gmm_gpu import _build
    from oracle.gmm import GMMOracle
    y, lab0 = np.zeros((0, 2)), np.zeros(0, dtype=np.int64)
    Q = _build(y, lab0, 3)
    assert type(Q.plans[0]).__name__ == 'GMMPlan'
    Q.update(repeat=2, verbose=False)
    o = GMMOracle(y, lab0, 3)
    o.iterate(2)
    np.testing.assert_allclose(Q.L[:2], o.L, rtol=0, atol=1e-10)
    assert np.asarray(Q['z'].u[0]).shape == (0, 3)


def test_state_space_block_without_sequences():
    from test_lssm_gpu import _build
    from oracle.lssm import LSSMOracle
    y, x0, c0 = np.zeros((2, 0, 5)), np.zeros((0, 5, 2)), np.ones((2, 2))
    Q = _build(y, x0, c0, False)
    assert type(Q.plans[0]).__name__ == 'LSSMPlan'
    Q.update(repeat=2, verbose=False)
    o = LSSMOracle(y, x0, c0)
    o.iterate(2)
    np.testing.assert_allclose(Q.L[:2], o.L, rtol=1e-9)
