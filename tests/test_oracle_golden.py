"""CPU: the NumPy oracle against the golden vectors made from the live reference
(oracle/make_golden.py) -- pins the oracle."""
import os

import numpy as np
import pytest

from oracle.pca import PCAOracle, make_pca_data

PCA_CASES = ['pca_n500_d6_k3', 'pca_n777_d20_k5', 'pca_n2048_d128_k32', 'pca_n4000_d64_k16']


def test_pca_oracle_with_prior_mean_matches_reference(golden_dir):
    """A constant non-zero prior mean of W (oracle/make_golden.py:pca_mean_case, case m3)."""
    g = np.load(os.path.join(golden_dir, 'pca_prior_mean.npz'))
    D = g['m3_y'].shape[0]
    o = PCAOracle(g['m3_y'], g['m3_x0'], mu=g['m3_mu'].reshape(D, -1), chunk=97)
    o.iterate(len(g['m3_L']))
    np.testing.assert_allclose(np.array(o.L), g['m3_L'], rtol=1e-11)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(np.array([t[k] for t in o.L_terms]), g['m3_L_' + k], rtol=1e-9,
                                   atol=1e-7)
    m = o.moments()
    np.testing.assert_allclose(m['W'], g['m3_W_u0'][:, 0, :], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m['alpha'][0], g['m3_alpha_u0'], rtol=1e-10)


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


def test_pca_oracle_matches_seeded_reference_run(golden_dir):
    """D=128, K=32 at N=1e5: the chunked oracle against the largest live-reference run."""
    from models import make_seeded_pca
    g = np.load(os.path.join(golden_dir, 'pca_seeded_n100000_d128_k32.npz'))
    N, D, K, n = int(g['N']), int(g['D']), int(g['K']), int(g['n_iter'])
    y, x0 = make_seeded_pca(int(g['seed']), N, D, K)
    o = PCAOracle(y, x0, chunk=1 << 14)
    o.iterate(n)
    np.testing.assert_allclose(np.array(o.L), g['L'], rtol=1e-11)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose([t[k] for t in o.L_terms], g['L_' + k], rtol=1e-9, atol=1e-6)
    m = o.moments()
    np.testing.assert_allclose(m['W'], g['W_u0'][:, 0, :], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m['X'][::97], g['X_u0_strided'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m['CX'], g['X_cov'], rtol=1e-7, atol=1e-14)


@pytest.mark.parametrize('tag', ['m0', 'm1', 'm2', 'm3', 'e0', 'e1', 'e2'])
@pytest.mark.parametrize('chunk', [7, 1 << 12])
def test_masked_pca_oracle_matches_reference(golden_dir, tag, chunk):
    """oracle/masked_pca.py (chunked sufficient-statistics form) against the live-reference
    traces with NaN placeholders at the missing entries (e*: plates and dimensions without any
    observation, tests/golden/masked_pca_erasures.npz)."""
    from oracle.masked_pca import MaskedPCAOracle
    g = np.load(os.path.join(golden_dir, 'masked_pca.npz' if tag[0] == 'm'
                             else 'masked_pca_erasures.npz'))
    y, mask, x0 = g['in_%s_y' % tag], g['in_%s_mask' % tag], g['in_%s_x0' % tag]
    L = g[tag + '_L']
    o = MaskedPCAOracle(y, mask, x0, chunk=chunk)
    o.iterate(len(L) - 1)
    # VB.update visits Y first (vmp.py:154-160): its latent entries see W, X of the previous
    # iteration
    f_prev, _ = o.predictive_Y()
    o.iterate(1)
    np.testing.assert_allclose(np.array(o.L), L, rtol=1e-11)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose([t[k] for t in o.L_terms], g['%s_L_%s' % (tag, k)], rtol=1e-9,
                                   atol=1e-7, err_msg=k)
    np.testing.assert_allclose(o.W, g[tag + '_W_u0'][:, 0, :], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.WW, g[tag + '_W_u1'][:, 0], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.X, g[tag + '_X_u0'][0], rtol=1e-8, atol=1e-11)
    if tag + '_Y_u0' in g.files:
        miss = ~mask
        np.testing.assert_allclose(f_prev[miss], g[tag + '_Y_u0'][miss], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('tag,nu_prior', [('lssmBc', None), ('lssmB', (1e-3, 1e-3)),
                                          # 8, 12, 16 states (lssm_wide_states.npz; round 6)
                                          ('w8', (1e-3, 1e-3)), ('w12', (1e-3, 1e-3)), ('w16', None)])
def test_lssm_oracle_matches_reference(golden_dir, tag, nu_prior):
    """oracle/lssm.py (plate sums + ONE shared covariance recursion) against the live-reference
    traces of the batched linear state-space model (fixed and Gamma innovation precision)."""
    from oracle.lssm import LSSMOracle
    g = np.load(os.path.join(golden_dir, 'lssm_wide_states.npz' if tag.startswith('w') else 'lssm.npz'))
    o = LSSMOracle(g[tag + '_y'], g[tag + '_x0'], g[tag + '_c0'], nu_prior=nu_prior)
    n = len(g[tag + '_L'])
    o.iterate(n)
    np.testing.assert_allclose(np.array(o.L), g[tag + '_L'], rtol=1e-11)
    for nm in ('X', 'A', 'C', 'tau', 'alpha', 'gamma') + (('nu',) if nu_prior else ()):
        np.testing.assert_allclose([t[nm] for t in o.L_terms], g['%s_%s_L' % (tag, nm)],
                                   rtol=1e-9, atol=1e-8, err_msg=nm)
    # (12 / 16 states: the smoother's conditioning costs two more digits on the entries near zero)
    wide = tag.startswith('w')
    np.testing.assert_allclose(o.X, g[tag + '_X_u0'], rtol=1e-8 if wide else 1e-9,
                               atol=1e-9 if wide else 1e-11)
    at = 1e-10 if wide else 1e-12
    np.testing.assert_allclose(o.Am, g[tag + '_A_u0'], rtol=1e-9, atol=at)
    np.testing.assert_allclose(o.AA, g[tag + '_A_u1'], rtol=1e-9, atol=at)
    np.testing.assert_allclose(o.Cm, g[tag + '_C_u0'].reshape(o.Cm.shape), rtol=1e-9, atol=at)
    xx = o.V[None] + o.X[:, :, :, None] * o.X[:, :, None, :]
    np.testing.assert_allclose(xx, g[tag + '_X_u1'], rtol=1e-8, atol=1e-9 if wide else 1e-11)
    xpxn = o.Cn[None] + o.X[:, :-1, :, None] * o.X[:, 1:, None, :]
    np.testing.assert_allclose(xpxn, g[tag + '_X_u2'], rtol=1e-8, atol=1e-9 if wide else 1e-11)


@pytest.mark.parametrize('name', ['gmm_n400_d3_k4', 'gmm_n3000_d8_k16'])
@pytest.mark.parametrize('chunk', [251, 1 << 15])
def test_gmm_oracle_matches_reference(golden_dir, name, chunk):
    """oracle/gmm.py directly against the live-reference traces of demos/mog.py's model (bound and
    every node's term per iteration, responsibilities, cluster moments) -- VERDICT r03: until now
    this oracle was pinned only through the kernel doubles and the GPU tests."""
    from oracle.gmm import GMMOracle
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    K = g['mu_u0'].shape[0]
    o = GMMOracle(g['y'], g['lab0'], K, chunk=chunk)
    o.iterate(int(g['n_iter']))
    np.testing.assert_allclose(np.array(o.L), g['L'], rtol=1e-11)
    for k in ('Y', 'mu', 'Lambda', 'z', 'alpha'):
        np.testing.assert_allclose([t[k] for t in o.L_terms], g['L_' + k], rtol=1e-9, atol=1e-7,
                                   err_msg=k)
    np.testing.assert_allclose(o.r, g['z_u0'], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(o.mu, g['mu_u0'], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o._mumu(), g['mu_u1'], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.Lam, g['Lambda_u0'], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(o.logdetLam, g['Lambda_u1'], rtol=1e-9)
    np.testing.assert_allclose(o.logpi, g['alpha_u0'], rtol=1e-9)
    np.testing.assert_allclose(o.alpha, g['alpha_phi0'], rtol=1e-10)


MASKED_LSSM = [('md', None), ('mb', (1e-3, 1e-3)), ('ms', None), ('me', (1e-3, 1e-3)), ('m1', None),
               # 5 ... 8 states (lssm_masked_wide.npz)
               ('w8', None), ('w6', (1e-3, 1e-3)), ('w5', None), ('w7', (1e-3, 1e-3))]


def load_masked_lssm(golden_dir, tag):
    """(y (M,B,T), mask (M,B,T) as stored -- possibly broadcastable --, x0 (B,T,D), c0 (M,D), g)."""
    g = np.load(os.path.join(golden_dir, 'lssm_masked_wide.npz' if tag.startswith('w')
                             else 'lssm_masked.npz'))
    y, mask, x0, c0 = g[tag + '_y'], g[tag + '_mask'], g[tag + '_x0'], g[tag + '_c0']
    if y.ndim == 2:                       # the demo's shape: no sequence plate
        y, mask, x0 = y[:, None, :], mask[:, None, :], x0[None]
    return y, mask, x0, c0.reshape(y.shape[0], -1), g


@pytest.mark.parametrize('tag,nu_prior', MASKED_LSSM)
def test_masked_lssm_oracle_matches_reference(golden_dir, tag, nu_prior):
    """oracle/lssm.py:MaskedLSSMOracle (one covariance recursion PER sequence, one posterior per
    row of C, ignored plates) against live-reference traces of demos/lssm.py's model observed
    through array masks: the demo's own (M, T) mask with a fully missing stretch (md), one mask
    per sequence (mb), a mask shared by the sequences (ms), rows / sequences / time steps without
    any observation (me), a single time step (m1).  mb / me come from the reference with its
    chol_solve broadcast defect repaired (oracle/make_golden.py:lssm_masked_cases)."""
    from oracle.lssm import MaskedLSSMOracle
    y, mask, x0, c0, g = load_masked_lssm(golden_dir, tag)
    o = MaskedLSSMOracle(y, mask, x0, c0, nu_prior=nu_prior)
    n = len(g[tag + '_L'])
    o.iterate(n)
    np.testing.assert_allclose(np.array(o.L), g[tag + '_L'], rtol=1e-11)
    for nm in ('Y', 'X', 'A', 'C', 'tau', 'alpha', 'gamma') + (('nu',) if nu_prior else ()):
        np.testing.assert_allclose([t[nm] for t in o.L_terms], g['%s_%s_L' % (tag, nm)],
                                   rtol=1e-9, atol=1e-8, err_msg=nm)
    sh = o.X.shape
    np.testing.assert_allclose(o.X, g[tag + '_X_u0'].reshape(sh), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(o.P, g[tag + '_X_u1'].reshape(sh + sh[-1:]), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(o.Cm, g[tag + '_C_u0'].reshape(o.Cm.shape), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(o._ccm(), g[tag + '_C_u1'].reshape(o._ccm().shape), rtol=1e-8,
                               atol=1e-8)
    np.testing.assert_allclose(o.Am, g[tag + '_A_u0'], rtol=1e-8, atol=1e-11)
    if tag + '_L_defect' in g.files:
        # the unrepaired reference gives every sequence the last one's S^-1 E: a different trace
        assert np.max(np.abs(g[tag + '_L_defect'] - g[tag + '_L'])) > 1e-2
