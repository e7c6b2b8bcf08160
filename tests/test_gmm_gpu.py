"""
GPU parity of the fused Gaussian-mixture block (vmp_gmm_* through the plan) against
the NumPy oracle (oracle/gmm.py, pinned to the live reference) on seeded inputs incl.
ragged / tiny sizes, and size-independent properties at the BASELINE.json config-3
size N=1e7, D=8, K=64.  Tolerances: ELBO rtol 1e-9, responsibilities atol 1e-12,
statistics rtol 1e-10; one-hot initial responsibilities bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(y, lab0, K, engine=None, shard=False):
    from bayespy_amd.nodes import (GaussianARD, Gaussian, Wishart, Dirichlet, Categorical,
                                   Mixture)
    from bayespy_amd.inference import VB
    N, D = y.shape
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
    z = Categorical(alpha, plates=(N,), name='z')
    if shard:
        z.shard(-1)
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    if lab0 is not None:
        z.initialize_from_value(lab0)
    Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha, engine=engine)
    Q.ignore_bound_checks = True
    return Q


@pytest.mark.parametrize('N,D,K', [(1, 1, 1), (5, 2, 3), (17, 3, 4), (100, 4, 16), (1000, 5, 17),
                                   (4099, 8, 64), (3000, 7, 33), (20000, 8, 32), (777, 1, 5),
                                   # 9 <= D <= 16: every feature-tile count (F2P = 64 ... 160) and
                                   # every cluster-tile count, incl. the pair-split form (K > 32)
                                   (500, 9, 5), (1000, 10, 40), (999, 11, 7), (3000, 12, 33),
                                   (1500, 13, 16), (2500, 14, 20), (700, 15, 48), (4100, 16, 32),
                                   (5000, 16, 64), (33, 16, 3), (1, 9, 64),
                                   # 17 <= D <= 32 (vmp_gmm_wide.hip): F2P = 192 ... 576, one / two /
                                   # four cluster tiles sharing a 16-point tile
                                   (500, 17, 5), (1000, 20, 40), (700, 24, 64), (1500, 32, 64),
                                   (2000, 32, 16), (300, 27, 33), (64, 31, 20), (1, 17, 64),
                                   (3000, 29, 7)])
def test_fused_gmm_vs_oracle(N, D, K):
    from oracle.gmm import GMMOracle, make_gmm_data
    y, lab0 = make_gmm_data(N, D, K, seed=N + D + K)
    Q = _build(y, lab0, K)
    assert type(Q.plans[0]).__name__ == 'GMMPlan'
    o = GMMOracle(y, lab0, K)
    # the one-hot initial responsibilities are integer indexing: bit-exact
    oh = np.zeros((N, K))
    oh[np.arange(N), lab0] = 1
    assert np.array_equal(Q['z'].u[0], oh)
    R, S1, S2 = Q.plans[0].statistics()
    np.testing.assert_allclose(R, o.R, rtol=0, atol=0)
    np.testing.assert_allclose(S1, o.S1, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(S2, o.S2, rtol=1e-12, atol=1e-10)
    iters = 3
    Q.update(repeat=iters, verbose=False)
    o.iterate(iters)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-9)
    np.testing.assert_allclose(Q['z'].u[0], o.r, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(Q['mu'].u[0], o.mu, rtol=1e-7, atol=1e-10)
    # (elements of an inverse: absolute accuracy follows the size of the matrix)
    np.testing.assert_allclose(Q['Lambda'].u[0], o.Lam, rtol=1e-7,
                               atol=1e-10 + 1e-11 * np.abs(o.Lam).max())
    np.testing.assert_allclose(Q['Lambda'].u[1], o.logdetLam, rtol=1e-9)
    np.testing.assert_allclose(Q['alpha'].u[0], o.logpi, rtol=1e-9)
    R, S1, S2 = Q.plans[0].statistics()
    np.testing.assert_allclose(R, o.R, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(S2, o.S2, rtol=1e-9, atol=1e-9)


def _edge_cases():
    rs = np.random.RandomState(0)
    out = []
    y = np.concatenate([rs.normal(size=(30, 2)), rs.normal(size=(30, 2)) + 8, [[1e3, -1e3]]])
    out.append(('outlier_and_empty_clusters', y, rs.randint(3, size=61), 8))
    out.append(('fewer_points_than_clusters', rs.normal(size=(5, 3)), np.arange(5), 9))
    y = np.repeat(rs.normal(size=(1, 2)), 40, axis=0)
    out.append(('identical_points', y, rs.randint(4, size=40), 4))
    out.append(('tiny_scale', rs.normal(size=(200, 2)) * 1e-6, rs.randint(3, size=200), 3))
    out.append(('huge_scale', rs.normal(size=(200, 2)) * 1e6, rs.randint(3, size=200), 3))
    return out


@pytest.mark.parametrize('case', _edge_cases(), ids=lambda c: c[0])
def test_fused_gmm_edge_regimes_vs_oracle(case):
    """Clusters that start or become empty, a far outlier (responsibilities underflow to exact
    zeros), more clusters than points, coincident points, extreme data scales.  The oracle agrees
    with the live reference to <= 1e-9 in the bound on exactly these inputs (checked when they were
    added; oracle/gmm.py is pinned by tests/test_oracle_golden.py)."""
    from oracle.gmm import GMMOracle
    _, y, lab0, K = case
    Q = _build(y, lab0, K)
    assert type(Q.plans[0]).__name__ == 'GMMPlan'
    o = GMMOracle(y, lab0, K)
    iters = 6
    Q.update(repeat=iters, verbose=False)
    o.iterate(iters)
    assert np.all(np.isfinite(Q.L[:iters]))
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-8)
    r = Q['z'].u[0]
    np.testing.assert_allclose(r.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(r, o.r, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(Q['mu'].u[0], o.mu, rtol=1e-6, atol=1e-9 * (1 + np.abs(y).max()))
    np.testing.assert_allclose(Q['Lambda'].u[1], o.logdetLam, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('D', [8, 16, 32])
def test_fused_gmm_prior_initialisation_and_determinism(D):
    from oracle.gmm import make_gmm_data
    y, lab0 = make_gmm_data(5000, D, 64, seed=1)
    Q1 = _build(y, lab0, 64)
    Q2 = _build(y, lab0, 64)
    Q1.update(repeat=3, verbose=False)
    Q2.update(repeat=3, verbose=False)
    assert np.array_equal(Q1.L[:3], Q2.L[:3])
    assert np.array_equal(Q1['z'].u[0], Q2['z'].u[0])
    # q(z) initialised from its prior: uniform responsibilities, rows sum to one
    Q3 = _build(y, None, 64)
    r = Q3['z'].u[0]
    np.testing.assert_allclose(r, 1.0 / 64, rtol=1e-12)
    Q3.update(repeat=2, verbose=False)
    assert np.all(np.isfinite(Q3.L[:2]))


def test_fused_gmm_rejects_bad_labels_and_sizes():
    from oracle.gmm import make_gmm_data
    y, lab0 = make_gmm_data(50, 3, 4, seed=2)
    bad = lab0.copy()
    bad[3] = 4
    Q = _build(y, bad, 4)
    with pytest.raises(ValueError):
        Q.update(repeat=1, verbose=False)
    # D beyond the fused block -> generic engine takes the model
    y9, l9 = make_gmm_data(80, 33, 3, seed=3)
    Q9 = _build(y9, l9, 3)
    assert type(Q9.plans[0]).__name__ == 'GenericPlan'
    Q9.update(repeat=2, verbose=False)
    from oracle.gmm import GMMOracle
    o = GMMOracle(y9, l9, 3)
    o.iterate(2)
    np.testing.assert_allclose(Q9.L[:2], np.array(o.L), rtol=1e-9)


@pytest.mark.parametrize('N,D,K', [(10_000_000, 8, 64), (2_000_000, 16, 64), (2_000_000, 12, 32),
                                   (1_000_000, 32, 64), (1_000_000, 24, 32)])
def test_config3_size_properties(N, D, K):
    """N=1e7, D=8, K=64 (BASELINE.json config 3): beyond what the reference can hold
    ((N,K,D,D) temporaries = 328 GB); parity through size-independent properties.  The same
    properties at D = 16 / 12 (the wider instances of the pass, pair-split and not) and at
    D = 32 / 24 (coefficients streamed from L2)."""
    import torch
    dev = torch.device('cuda')
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    lab0 = torch.randint(0, K, (N,), generator=g, device=dev)
    Q = _build(y, lab0.cpu().numpy(), K)
    plan = Q.plans[0]
    Q.update(repeat=3, verbose=False)
    L = Q.L[:3]
    assert np.all(np.isfinite(L)) and np.all(np.diff(L) > -1e-6 * np.abs(L[:-1]))
    # (1) responsibilities: rows sum to one, non-negative
    r = plan.Rd
    rs = r.sum(dim=1)
    assert float((rs - 1).abs().max()) < 1e-12 and float(r.min()) >= 0.0
    # (2) statistics == independent fp64 GEMMs over the written responsibilities
    R, S1, S2 = plan.statistics()
    np.testing.assert_allclose(R, r.sum(dim=0).cpu().numpy(), rtol=1e-11)
    np.testing.assert_allclose(S1, (r.T @ y).cpu().numpy(), rtol=1e-10, atol=1e-6)
    yy = (y[:, :, None] * y[:, None, :]).reshape(N, D * D)
    np.testing.assert_allclose(S2.reshape(K, D * D), (r.T @ yy).cpu().numpy(), rtol=1e-10,
                               atol=1e-5)
    del yy
    # (3) the written r equals the reference recipe on a sample of rows (re-derived on the
    #     host from the device's own mu / Lambda / log pi moments of the previous half-step)
    from oracle.gmm import GMMOracle
    o = GMMOracle.__new__(GMMOracle)
    Lyt = plan.layout
    o.D, o.K = D, K
    st = plan.state.cpu().numpy()
    o.mu = st[Lyt.off_mu:Lyt.off_mu + K * D].reshape(K, D)
    o.Cmu = st[Lyt.off_Cmu:Lyt.off_Cmu + K * D * D].reshape(K, D, D)
    o.Lam = st[Lyt.off_Lam:Lyt.off_Lam + K * D * D].reshape(K, D, D)
    o.logdetLam = st[Lyt.off_logdetLam:Lyt.off_logdetLam + K]
    # <log pi> used by the last z.update() is the one BEFORE the last alpha.update(): redo
    # the half step on a fresh plan state instead: run z.update() again and compare
    Q['z'].update()
    st2 = plan.state.cpu().numpy()
    logpi = st2[Lyt.off_alpha + Lyt.KP:Lyt.off_alpha + Lyt.KP + K]
    c, b = o._coefficients()
    idx = torch.randint(0, N, (2048,), generator=g, device=dev)
    ys = y[idx].cpu().numpy()
    phi = (logpi + c)[None, :] + ys @ b.T - 0.5 * np.einsum('ni,kij,nj->nk', ys, o.Lam, ys)
    m = phi.max(axis=1, keepdims=True)
    p = np.exp(phi - (np.log(np.exp(phi - m).sum(axis=1, keepdims=True)) + m))
    p /= p.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(plan.Rd[idx].cpu().numpy(), p, rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize('N,D,K', [(2_000_000, 8, 64), (500_000, 16, 64), (200_000, 32, 64)])
def test_config3_dims_direct_oracle_parity(N, D, K):
    """D=8, K=64 (BASELINE config 3 dims) at N=2e6, DIRECT parity: the chunked NumPy oracle on
    the same data and initial labels, two iterations (the reference itself builds (N,K,D,D)
    temporaries and stops at N~3e5); bound rel <= 1e-9 per iteration and node term,
    responsibilities of a strided sample atol 1e-12, cluster moments rtol 1e-7.  The same at
    D = 16 (pair-split instance) and D = 32 (coefficients streamed from L2)."""
    import torch
    from oracle.gmm import GMMOracle
    dev = torch.device('cuda')
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    lab0 = torch.randint(0, K, (N,), generator=g, device=dev).cpu().numpy()
    Q = _build(y, lab0, K)
    iters = 2
    Q.update(repeat=iters, verbose=False)
    o = GMMOracle(y.cpu().numpy(), lab0, K, chunk=1 << 16)
    o.iterate(iters, keep_r=True)
    np.testing.assert_allclose(Q.L[:iters], np.array(o.L), rtol=1e-9)
    for nm in ('Y', 'z', 'alpha', 'mu', 'Lambda'):
        np.testing.assert_allclose(Q.l[Q[nm]][:iters], [t[nm] for t in o.L_terms], rtol=1e-9,
                                   atol=1e-4, err_msg=nm)
    sel = torch.arange(0, N, 499, device=dev)
    np.testing.assert_allclose(Q.plans[0].Rd.index_select(0, sel).cpu().numpy(), o.r[::499],
                               rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(Q['mu'].u[0], o.mu, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q['Lambda'].u[0], o.Lam, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(Q['alpha'].u[0], o.logpi, rtol=1e-9)
