"""CPU: pure host logic that needs no device -- plate multipliers (node.py:294-301,
:589-632), sharding declarations, Multinomial validation (multinomial.py:66-82, :141-152),
rotation-block argument checks (transformations.py:376-440), checkpoint container."""
import numpy as np
import pytest

import bayespy_amd.nodes as nodes
from bayespy_amd.utils.shapes import multiplier_shape, multiplier_factor, broadcasting_multiplier


def test_multiplier_algebra():
    assert multiplier_shape(None) == ()
    assert multiplier_shape((10.0,), ()) == (10.0,)
    assert multiplier_shape(None, (4,), (1, 4)) == (1, 4)
    assert multiplier_shape((2, 1), (3,)) == (2, 3)
    with pytest.raises(ValueError):
        multiplier_shape((2,), (3,))
    # the factor a child applies to a message: its multiplier where the parent has none
    assert multiplier_factor((10.0,), ()) == 10.0
    assert multiplier_factor((10.0,), (10.0,)) == 1.0
    assert multiplier_factor((2, 5), (1, 5)) == 2.0
    assert multiplier_factor((), ()) == 1.0
    assert broadcasting_multiplier((3, 4), (1, 4), (4,)) == 3


def test_nodes_inherit_multipliers_and_partitions():
    alpha = nodes.Dirichlet(np.ones(3), name='alpha')
    z = nodes.Categorical(alpha, plates=(5,), plates_multiplier=(12.0,), name='z')
    mu = nodes.GaussianARD(0, 1, shape=(2,), plates=(3,), name='mu')
    y = nodes.Mixture(z, nodes.Gaussian, mu, np.identity(2), name='y')
    assert alpha.plates_multiplier == () and mu.plates_multiplier == ()
    assert z.plates_multiplier == (12.0,) and y.plates_multiplier == (12.0,)
    z.plates_multiplier = (3.0,)
    assert y.plates_multiplier == (3.0,)
    x = nodes.GaussianARD(0, 1, shape=(2,), plates=(1, 7), name='x')
    assert x.shard(-1) is x and x._shard_axis == -1
    assert x.shard(1)._shard_axis == -1
    with pytest.raises(ValueError):
        x.shard(2)
    with pytest.raises(ValueError):
        x.shard(1.5)


def test_multinomial_validation():
    p = nodes.Dirichlet(np.array([1.0, 2.0, 3.0]))
    m = nodes.Multinomial(np.array([4, 2, 7]), p)
    assert m.plates == (3,) and m.dims == ((3,),)
    assert nodes.Multinomial(5, p, plates=(2, 6)).plates == (2, 6)
    with pytest.raises(ValueError, match='integer'):
        nodes.Multinomial(2.5, p)
    with pytest.raises(ValueError, match='non-negative'):
        nodes.Multinomial(-1, p)
    with pytest.raises(ValueError, match='sum to the number of trials'):
        m.observe(np.array([[4, 0, 0], [1, 1, 1], [7, 0, 0]]))
    with pytest.raises(ValueError, match='non-negative'):
        m.observe(np.array([[5, -1, 0], [2, 0, 0], [7, 0, 0]]))
    with pytest.raises(ValueError):
        m.observe(np.array([[4, 0], [2, 0], [7, 0]]))


def test_rotation_block_argument_checks():
    from bayespy_amd.inference import transformations as T
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(4,))
    W = nodes.GaussianARD(0, alpha, shape=(4,), plates=(6, 1))
    X = nodes.GaussianARD(0, 1, shape=(4,), plates=(1, 9))
    assert T.RotateGaussianARD(W, alpha).nodes() == [W, alpha]
    assert T.RotateGaussianARD(X).nodes() == [X]
    with pytest.raises(ValueError, match='Too many'):
        T.RotateGaussianARD(W, alpha, alpha)
    with pytest.raises(ValueError):
        T.RotateGaussianARD(W, nodes.Gamma(1e-2, 1e-2, plates=(4,)))     # not W's precision
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(nodes.GaussianARD(1.0, 1, shape=(4,), plates=(3,)))   # non-zero mean
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(nodes.GaussianARD(0, 1, plates=(3,)))                # scalar node
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(X).setup(plate_axis=0)        # two plate axes: no plate rotation


def test_state_space_rotation_gradients_by_finite_differences():
    """RotateGaussianMarkovChain + RotateGaussianARD with a plate rotation (transformations.py:693-1093,
    :1296-1450): the analytic gradient of the bound w.r.t. R (through x -> R x, the columns of A by
    R^-T and its rows by Q = R) against central differences, on statistics served by a plan double."""
    from scipy import optimize
    from bayespy_amd.inference import transformations as T
    rs = np.random.RandomState(12)
    D, N = 3, 40

    def spd(n=D):
        a = rs.normal(size=(n, n + 2))
        return a @ a.T

    class Plan:
        def rotation_statistics(self, node):
            return dict(nvec=float(N), X0=rs_x0, X0X0=s00, XnXn=snn, XpXn=spn, XpXp=spp)

        def rotation_rows(self, node):
            return dict(mean=am, cov=cova)

        def gamma_posterior_shape(self, node):
            return np.full(D, 1e-5 + 0.5 * D)

    rs_x0, s00, snn, spp = rs.normal(size=D), spd(), N * spd(), N * spd()
    spn = 0.5 * N * rs.normal(size=(D, D))
    am = rs.normal(size=(D, D))
    cova = np.stack([0.1 * spd() for _ in range(D)])
    alpha = nodes.Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
    A = nodes.GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
    X = nodes.GaussianMarkovChain(np.zeros(D) + 0.3, 1e-1 * spd(), A, np.ones(D), n=N, name='X')
    for n in (alpha, A, X):
        n._plan = Plan()
    for with_alpha in (True, False):
        rotA = T.RotateGaussianARD(A, alpha, axis=0) if with_alpha else None
        if rotA is None:
            A2 = nodes.GaussianARD(0, 2.0, shape=(D,), plates=(D,), name='A2')
            X2 = nodes.GaussianMarkovChain(np.zeros(D), np.identity(D), A2, np.ones(D), n=N, name='X2')
            A2._plan = X2._plan = Plan()
            rotX = T.RotateGaussianMarkovChain(X2, T.RotateGaussianARD(A2, axis=0))
        else:
            rotX = T.RotateGaussianMarkovChain(X, rotA)
        rotX.setup()
        R0 = np.identity(D) + 0.2 * rs.normal(size=(D, D))
        f = lambda r: rotX.bound(r.reshape(D, D))[0]        # noqa: E731
        g = rotX.bound(R0)[1]
        gn = optimize.approx_fprime(R0.ravel(), f, 1e-6).reshape(D, D)
        np.testing.assert_allclose(g, gn, rtol=2e-5, atol=2e-5 * np.abs(gn).max())
        # the terms reported per node add up to the bound
        terms = rotX.get_bound_terms(R0)
        np.testing.assert_allclose(sum(terms.values()), rotX.bound(R0)[0], rtol=1e-12)
    with pytest.raises(ValueError):
        T.RotateGaussianARD(A, alpha).bound(np.identity(D), Q=np.identity(D))    # Q without plate_axis


def test_checkpoint_container_round_trip(tmp_path):
    from bayespy_amd.inference.checkpoint import Writer, Reader
    fn = str(tmp_path / 'c.bin')
    w = Writer(fn)
    w.put('nodes/a/u0', np.arange(6.0).reshape(2, 3))
    w.put('iter', 7)
    w.put('converged', False)
    w.close()
    r = Reader(fn)
    assert r.has('nodes/a/u0') and not r.has('nodes/b/u0')
    np.testing.assert_array_equal(r.get('nodes/a/u0'), np.arange(6.0).reshape(2, 3))
    assert int(r.get('iter')) == 7 and not bool(r.get('converged'))
    with pytest.raises(KeyError):
        r.get('missing')
    r.close()


def test_plate_rules_of_the_plate_moving_nodes():
    """Plates and argument checks of Take / Concatenate / Gate / Slice / Choose follow the
    reference (take.py:41-66, :126-131; concatenate.py:27-78; gate.py:33-80,
    nodes/tests/test_gate.py:35-80; node.py:868-1015) -- construction only, no device."""
    import bayespy_amd.nodes as N
    a = N.Gamma(np.ones(3), np.ones(3))
    assert N.Take(a, [1, 1, 2, 2, 1, 0]).plates == (6,)
    assert N.Take(a, [[0, 2], [1, 1]]).plates == (2, 2)
    X = N.GaussianARD(0, 1, shape=(2,), plates=(3, 4))
    t = N.Take(X, [[0, 2], [1, 1]], plate_axis=-2)
    assert t.plates == (2, 2, 4) and t.dims == X.dims
    assert N.Concatenate(N.GaussianARD(0, 1, plates=(3,)), N.GaussianARD(0, 1, plates=(2,))).plates == (5,)
    c = N.Concatenate(N.GaussianARD(0, 1, shape=(2,), plates=(2, 1)),
                      N.GaussianARD(0, 1, shape=(2,), plates=(1, 1)), axis=-2)
    assert c.plates == (3, 1)
    # test_gate.py:35-80
    Z = N.Categorical(np.ones(3) / 3)
    assert N.Gate(Z, N.GaussianARD(0, 1, shape=(), plates=(3,))).plates == ()
    G = N.Gate(Z, N.GaussianARD(0, 1, shape=(2,), plates=(3,)))
    assert G.plates == () and G.dims == ((2,), (2, 2))
    Z4 = N.Categorical(np.ones(3) / 3, plates=(4,))
    assert N.Gate(Z4, N.GaussianARD(0, 1, shape=(2,), plates=(3,))).plates == (4,)
    assert N.Gate(Z, N.GaussianARD(0, 1, shape=(2,), plates=(4, 3))).plates == (4,)
    Z5 = N.Categorical(np.ones(3) / 3, plates=(5,))
    assert N.Gate(Z5, N.GaussianARD(0, 1, shape=(2,), plates=(4, 1, 3))).plates == (4, 5)
    assert N.Gate(Z, N.GaussianARD(0, 1, shape=(), plates=(3, 4)), gated_plate=-2).plates == (4,)
    with pytest.raises(ValueError, match='negative'):
        N.Gate(Z, N.GaussianARD(0, 1, plates=(3,)), gated_plate=0)
    # slicing
    X = N.GaussianARD(0, 1, plates=(4, 5, 6))
    assert X[1:3].plates == (2, 5, 6)
    assert X[..., 0].plates == (4, 5)
    assert X[:, None, ::2].plates == (4, 1, 3, 6)
    assert X[-1, ..., 2:].plates == (5, 4)
    assert X[2].plates == (5, 6) and X[None].plates == (1, 4, 5, 6)
    assert X[::-1, 1].plates == (4, 6)
    assert N.Choose([0, 1, 1], N.GaussianARD(0, 1), N.GaussianARD(1, 1)).plates == (3,)


def test_markov_chain_constructors():
    """categorical_markov_chain.py:283-330 and gaussian_markov_chain.py:1866-1985 / :1331-1452:
    chain lengths, plates and the parent checks."""
    import bayespy_amd.nodes as N
    K = 3
    Z = N.CategoricalMarkovChain(np.ones(K) / K, np.ones((K, K)) / K, states=7)
    assert Z.plates == () and Z.dims == ((K,), (6, K, K))
    A = N.Dirichlet(np.ones((9, K, K)))                # time-varying transitions: plates (9, K)
    Z = N.CategoricalMarkovChain(N.Dirichlet(np.ones(K), plates=(4,)), A)
    assert Z.plates == (4,) and Z.states == 10
    assert Z.as_categorical().plates == (4, 10) and Z.as_categorical() is Z.as_categorical()
    with pytest.raises(ValueError, match='inconsistent'):
        N.CategoricalMarkovChain(np.ones(K) / K, A, states=5)
    with pytest.raises(ValueError, match='not square'):
        N.CategoricalMarkovChain(np.ones(K) / K, N.Dirichlet(np.ones((9, 2, K))))
    D = 2
    B = N.GaussianARD(0, 1, shape=(D,), plates=(K, D))
    Zc = N.CategoricalMarkovChain(np.ones(K) / K, np.ones((K, K)) / K, states=5)
    X = N.SwitchingGaussianMarkovChain(np.zeros(D), np.identity(D), B, Zc, np.ones(D))
    assert (X.N, X.D, X.K) == (6, D, K) and X.plates == ()
    assert X.as_gaussian().plates == (6,)
    with pytest.raises(ValueError, match='N-1'):
        N.SwitchingGaussianMarkovChain(np.zeros(D), np.identity(D), B, Zc, np.ones(D), n=9)
    with pytest.raises(ValueError, match='Fourth parent'):
        N.SwitchingGaussianMarkovChain(np.zeros(D), np.identity(D), B,
                                       N.Categorical(np.ones(2) / 2, plates=(5,)), np.ones(D))
    Bv = N.GaussianARD(0, 1, shape=(D, K), plates=(D,))
    S = N.GaussianMarkovChain(np.zeros(K), np.identity(K), np.identity(K), np.ones(K), n=6)
    Xv = N.VaryingGaussianMarkovChain(np.zeros(D), np.identity(D), Bv, S.as_gaussian()[1:],
                                      np.ones(D))
    assert (Xv.N, Xv.D, Xv.K) == (6, D, K)
    with pytest.raises(ValueError, match='Third parent'):
        N.VaryingGaussianMarkovChain(np.zeros(D), np.identity(D), B, S.as_gaussian()[1:],
                                     np.ones(D))


def test_count_node_constructors():
    """binomial.py:66-76, :214-240; beta.py:134-160; poisson.py:122-150; add.py:52-85."""
    import bayespy_amd.nodes as N
    p = N.Beta([[1.0, 2.0], [0.5, 0.5], [3.0, 1.0]])
    assert p.plates == (3,) and p.dims == ((2,),)
    assert N.Binomial([10, 7, 12], p, plates=(4, 3)).plates == (4, 3)
    assert N.Bernoulli(p.complement(), plates=(6, 3)).plates == (6, 3)
    assert N.Bernoulli(0.3).plates == () and N.Binomial(5, [0.2, 0.4]).plates == (2,)
    assert N.Poisson(N.Gamma(np.ones(3), np.ones(3)), plates=(20, 3)).plates == (20, 3)
    with pytest.raises(ValueError, match=r'\[0, 1\]'):
        N.Bernoulli(1.5)
    with pytest.raises(ValueError, match='beta-like'):
        N.Bernoulli(N.Dirichlet(np.ones(3)))
    a = N.GaussianARD(0, 1, shape=(2,), plates=(5, 1))
    b = N.GaussianARD(0, 1, shape=(2,), plates=(1, 3))
    s = N.Add(a, b, np.zeros(2))
    assert s.plates == (5, 3) and s.dims == ((2,), (2, 2))
    # like the reference, GaussianARD is scalar-valued unless ndim / shape says otherwise: the
    # variable axis of a Gaussian mean becomes a plate (gaussian.py:1617-1640)
    Y = N.GaussianARD(s, 1.0)
    assert Y.plates == (5, 3, 2) and Y.shape == ()
    Y = N.GaussianARD(s, 1.0, ndim=1)
    assert Y.plates == (5, 3) and Y.shape == (2,)
    with pytest.raises(ValueError, match='inconsistent'):
        N.GaussianARD(s, 1.0, ndim=2, shape=(2,))


def test_random_data_generation_helpers():
    """bayespy.utils.random set-up helpers (host, NumPy): shapes, ranges and basic statistics."""
    from bayespy_amd.utils import random as r
    np.random.seed(3)
    m = r.mask(200, 50, p=0.8)
    assert m.shape == (200, 50) and m.dtype == bool and abs(m.mean() - 0.8) < 0.02
    C = r.covariance(4, size=(3,))
    assert C.shape == (3, 4, 4) and np.all(np.linalg.eigvalsh(C) > 0)
    np.testing.assert_allclose(C, np.swapaxes(C, -1, -2), rtol=1e-10)
    Q = r.orth(5)
    np.testing.assert_allclose(Q @ Q.T, np.identity(5), atol=1e-12)
    np.testing.assert_allclose(np.linalg.svd(r.svd(np.array([3.0, 2.0, 1.0])))[1], [3, 2, 1], rtol=1e-10)
    np.testing.assert_allclose(np.diag(r.correlation(4)), 1.0, rtol=1e-12)
    z = r.categorical([0.1, 0.2, 0.7], size=5000)
    assert z.shape == (5000,) and set(np.unique(z)) <= {0, 1, 2}
    assert abs(np.mean(z == 2) - 0.7) < 0.03
    z = r.categorical(np.array([[1.0, 0.0], [0.0, 1.0]]))
    assert list(z) == [0, 1]
    with pytest.raises(ValueError, match='negative'):
        r.categorical([-0.1, 1.1])
    x = r.multinomial([5, 9], [[0.5, 0.5, 0.0], [0.1, 0.1, 0.8]])
    assert x.shape == (2, 3) and list(x.sum(-1)) == [5, 9] and x[0, 2] == 0
    b = r.bernoulli(np.array([0.0, 1.0, 1.0]))
    assert list(b) == [False, True, True]
    d = r.dirichlet(np.ones(4), size=(6,))
    assert d.shape == (6, 4)
    np.testing.assert_allclose(d.sum(-1), 1.0, rtol=1e-12)
    assert r.gamma(2.0, 3.0, size=(4,)).shape == (4,)
    np.testing.assert_allclose(r.logodds_to_probability([0.0, np.log(3.0)]), [0.5, 0.75])
    lat, lon = r.sphere(10)
    assert np.all(np.abs(lat) <= 90) and np.all(np.abs(lon) <= 180)
    with pytest.raises(ValueError, match='greater than'):
        r.covariance(4, nu=3)


def test_user_guide_plate_doctests():
    """The plate examples of doc/source/user_guide/modelconstruct.rst:205-220, :315-328,
    :350-393, :500-511 -- construction only."""
    from bayespy_amd.nodes import (Gaussian, GaussianARD, Gamma, Wishart, Categorical, Mixture, Dot)
    mu0 = Gaussian(np.zeros(3), np.identity(3))
    Lam0 = Wishart(3, np.identity(3))
    y = Gaussian(mu0, Lam0, plates=(10, 30))
    assert y[0].plates == (30,)
    assert y[:, ::2].plates == (10, 15)
    assert y[:5, 10:20:5].plates == (5, 2)
    mu = [[0, 0], [1, 1], [2, 2]]
    Lambda = [[[1.0, 0.0], [0.0, 1.0]], [[1.0, 0.9], [0.9, 1.0]], [[1.0, -0.3], [-0.3, 1.0]]]
    X = Gaussian(mu, Lambda)
    assert X.plates == (3,)
    mu = Gaussian([[0], [0], [0]], [[[1]], [[1]], [[1]]])
    Lambda = Wishart(1, [[[1]], [[1]], [[1]]])
    Z = Categorical([1 / 3, 1 / 3, 1 / 3], plates=(100,))
    X = Mixture(Z, Gaussian, mu, Lambda)
    assert mu.plates == (3,) and Lambda.plates == (3,) and Z.plates == (100,)
    assert X.plates == (100,)
    tau = Gamma(1, 1, plates=(5, 4, 3))
    X = GaussianARD(0, tau, shape=(4, 3))
    assert tau.plates == (5, 4, 3) and X.plates == (5,)
    D = 3
    X = GaussianARD(0, 1, shape=(D,), plates=(1, 100))
    alpha = Gamma(1e-3, 1e-3, plates=(D,))
    C = GaussianARD(0, alpha, shape=(D,), plates=(10, 1))
    assert Dot(C, X).plates == (10, 100)


def test_bench_extra_records_are_compact():
    """The default bench line must stay far below the ~8 KB tail the driver keeps: every leg of
    "extra" is the one-screen record of tools/workloads.py:compact."""
    import json
    from tools import workloads
    full = {
        'metric': 'VB iterations/sec, GMM N=10000000 D=8 K=64', 'value': 342.5, 'steps': 200,
        'ms_per_step': 2.92, 'step_ms': {'ms_mean': 2.92, 'ms_median': 2.913, 'ms_max': 3.16,
                                         'argmax_step': 0, 'timed_s': 0.584},
        'config': {'workload': 'x' * 300}, 'peak_mem_GB': 12.3456789, 'wall_s': 2.123,
        'roofline': {'kernel': 'gmm_pass_kernel', 'bound': 'mfma', 'achieved': 66.9, 'peak': 78.6,
                     'frac': 0.8512345, 'traffic': 5.767e9, 'alg_bytes_per_launch': 5.76e9,
                     'avg_launch_ms': 2.7951234, 'issued_mfma_TFLOPs': 44.01234, 'note': 'y' * 400},
        'cpu_baseline': {'value': 0.0335123, 'cores': 128, 'kind': 'port',
                         'elbo_rel_err_hip_vs_oracle': 8.1234e-15, 'sample': 'z' * 300}}
    c = workloads.compact(full, 'gmm')
    assert c['leg'] == 'gmm' and c['ms_per_step'] == 2.92 and c['ms_max'] == 3.16
    assert c['frac'] == 0.851 and c['kernel_ms'] == 2.795 and c['parity_on'] == 'sample'
    assert abs(c['traffic_over_alg'] - 1.0) < 2e-3
    # the usual values are left out (all 128 cores; no single slow step): nine legs, one 5 KB line
    assert 'cpu_cores' not in c and 'argmax_step' not in c and 'wall_s' not in c
    slow = dict(full, step_ms=dict(full['step_ms'], ms_max=49.0, argmax_step=7),
                cpu_baseline=dict(full['cpu_baseline'], cores=1))
    cs = workloads.compact(slow, 'gmm')
    assert cs['argmax_step'] == 7 and cs['cpu_cores'] == 1
    assert len(json.dumps(c)) < 420
    err = workloads.compact({'error': 'RuntimeError: ' + 'q' * 500, 'wall_s': 1.0}, 'lssm')
    assert err['leg'] == 'lssm' and len(json.dumps(err)) < 260
    assert len(json.dumps([c] * 9)) < 3400


def test_contraction_plans_avoid_plates_sized_intermediates():
    """misc.plan_contraction (the pairwise order of misc.contract_path, from shapes alone): the
    plate sums of a PCA model never pair two operands into a (D, N) product, and the plan is a
    function of the shapes only (deterministic)."""
    from bayespy_amd.utils.misc import plan_contraction
    sizes = dict(d=64, n=1_000_000, k=16, l=16)
    # sum_dn y_dn w_dk x_nk: (y, x) -> (d, k) over n first, then with w
    steps = plan_contraction([['d', 'n'], ['d', 'k'], ['n', 'k']], [], sizes)
    assert steps == [(0, 2, ['d', 'k'])]
    # sum <f>^2 = (W^T W) : (X^T X): the two small Gram matrices, never (d, n)
    steps = plan_contraction([['d', 'k'], ['n', 'k'], ['d', 'l'], ['n', 'l']], [], sizes)
    assert steps == [(0, 2, ['k', 'l']), (0, 1, ['k', 'l'])]
    # with a plate mask the mask joins the data first (one (d, n) product is unavoidable)
    steps = plan_contraction([['d', 'n'], ['d', 'n'], ['d', 'k'], ['n', 'k']], [], sizes)
    assert steps[0] == (0, 1, ['d', 'n']) and steps[1][2] == ['d', 'k']
    # a kept plate stays in the results that carry it
    steps = plan_contraction([['d', 'n'], ['d', 'k'], ['n', 'k']], ['d'], sizes)
    assert steps == [(0, 2, ['d', 'k'])]
    # a matrix chain: smallest intermediate first
    steps = plan_contraction([['a', 'b'], ['b', 'c'], ['c', 'e']], ['a', 'e'],
                             dict(a=1000, b=10, c=1000, e=10))
    assert steps == [(1, 2, ['e', 'b'])]          # (output labels lead the label order)
    for _ in range(3):
        assert plan_contraction([['d', 'k'], ['n', 'k'], ['d', 'l'], ['n', 'l']], [], sizes) == \
            [(0, 2, ['k', 'l']), (0, 1, ['k', 'l'])]
