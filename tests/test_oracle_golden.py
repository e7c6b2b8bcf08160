"""CPU: the NumPy oracle against the golden vectors made from the live reference
(oracle/make_golden.py) -- pins the oracle."""
import os

import numpy as np
import pytest

from oracle.pca import PCAOracle, make_pca_data

PCA_CASES = ['pca_n500_d6_k3', 'pca_n777_d20_k5', 'pca_n2048_d128_k32', 'pca_n4000_d64_k16']


@pytest.mark.parametrize('name', PCA_CASES)
@pytest.mark.parametrize('chunk', [257, 1 << 16])
def test_pca_oracle_matches_reference(golden_dir, name, chunk):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    o = PCAOracle(g['y'], g['x0'], a0=float(g['a0']), b0=float(g['b0']), chunk=chunk)
    o.iterate(int(g['n_iter']))
    # ELBO trace: the north-star tolerance is 1e-5 relative; the oracle is ~1e-14
    np.testing.assert_allclose(np.array(o.L), g['L'], rtol=1e-11)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        ref = g['L_' + k]
        got = np.array([t[k] for t in o.L_terms])
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-7)
    m = o.moments()
    np.testing.assert_allclose(m['W'], g['W_u0'][:, 0, :], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m['X'], g['X_u0'][0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m['tau'][0], g['tau_u0'], rtol=1e-11)
    np.testing.assert_allclose(m['tau'][1], g['tau_u1'], rtol=1e-11)
    np.testing.assert_allclose(m['alpha'][0], g['alpha_u0'], rtol=1e-10)
    np.testing.assert_allclose(m['alpha'][1], g['alpha_u1'], rtol=1e-10)
    # second moments: u1 = <x><x>^T + Cov
    x3 = m['X'][:3]
    np.testing.assert_allclose(x3[:, :, None] * x3[:, None, :] + m['CX'], g['X_u1_first'],
                               rtol=1e-9, atol=1e-11)
    w = m['W']
    np.testing.assert_allclose(w[:, :, None] * w[:, None, :] + m['CW'], g['W_u1'][:, 0],
                               rtol=1e-9, atol=1e-11)


def test_quickstart_known_answer(golden_dir):
    """The reference's own doctest vector (doc/source/user_guide/quickstart.rst:111-118)."""
    g = np.load(os.path.join(golden_dir, 'quickstart_n10.npz'))
    printed = ['%e' % v for v in g['L']]
    assert printed == ['-6.020956e+01', '-5.820527e+01', '-5.820290e+01', '-5.820288e+01']


def test_synthetic_data_is_deterministic():
    y1, x1 = make_pca_data(300, 8, 3, seed=5, chunk=128)
    y2, x2 = make_pca_data(300, 8, 3, seed=5, chunk=128)
    assert np.array_equal(y1, y2) and np.array_equal(x1, x2)
    assert y1.shape == (8, 300) and x1.shape == (300, 3)


def test_hmm_oracle_matches_reference_recursions(golden_dir):
    """oracle/hmm.py against random.alpha_beta_recursion of the live reference (six shapes,
    incl. an impossible state and 300 chains)."""
    from oracle import hmm
    f = np.load(os.path.join(golden_dir, 'markov_chains.npz'))
    for tag in ('ab_a', 'ab_b', 'ab_c', 'ab_d', 'ab_e', 'ab_f'):
        z0, zz, g = hmm.alpha_beta_recursion(f[tag + '_logp0'], f[tag + '_logP'])
        np.testing.assert_allclose(g, f[tag + '_g'], rtol=1e-12, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(z0, f[tag + '_z0'], rtol=1e-10, atol=1e-15, err_msg=tag)
        np.testing.assert_allclose(zz, f[tag + '_zz'], rtol=1e-10, atol=1e-15, err_msg=tag)
