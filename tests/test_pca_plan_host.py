"""CPU: host logic of the PCA plan (pattern match, update order, lazy state,
lower-bound cache) with the kernel test double, against the live-reference
golden vectors."""
import os

import numpy as np
import pytest
import torch

import bayespy_amd.nodes as nodes
from bayespy_amd.device import Runtime
from bayespy_amd.inference import VB
from bayespy_amd.inference.plans.pca import PCAPlan

from fake_kernels import CPURuntimeKernels
from models import build_pca


def _attach_cpu(Q, stats='gram'):
    rt = Runtime(device='cpu')
    for p in Q.plans:
        p._rt = rt
        p._kernels = CPURuntimeKernels(rt)
        p.stats = stats
    return Q


@pytest.mark.parametrize('stats', ['gram', 'stream'])
@pytest.mark.parametrize('name', ['pca_n500_d6_k3', 'pca_n777_d20_k5'])
def test_plan_reproduces_reference_trace(golden_dir, name, stats):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    K = g['x0'].shape[1]
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], K), stats)
    Q.update(repeat=int(g['n_iter']), verbose=False)
    np.testing.assert_allclose(Q.L[:Q.iter], g['L'], rtol=1e-10)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:Q.iter], g['L_' + k], rtol=1e-8, atol=1e-7)
    assert np.all(Q.l[Q['F']][:Q.iter] == 0)
    W, X, tau, alpha = Q['W'], Q['X'], Q['tau'], Q['alpha']
    np.testing.assert_allclose(W.u[0], g['W_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(W.u[1], g['W_u1'], rtol=1e-8, atol=1e-10)
    assert W.u[0].shape == g['W_u0'].shape and X.u[0].shape == g['X_u0'].shape
    np.testing.assert_allclose(X.u[0], g['X_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(X.u[1][0, :3], g['X_u1_first'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(tau.u[0], g['tau_u0'], rtol=1e-10)
    np.testing.assert_allclose(alpha.u[1], g['alpha_u1'], rtol=1e-9)


def test_update_order_and_one_pass_per_iteration(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(repeat=2, verbose=False)
    calls = Q.plans[0].kernels.calls
    assert calls[:2] == ['gram', 'stats_from_x']
    per_iter = ['update_w', 'prepare_x', 'xpass_tiled', 'update_tau', 'update_alpha']
    # the constant data is re-laid-out tile-major ONCE (before the first pass); the latent pass
    # stays in flight across iterations and is joined once per VB.update()
    assert calls[2:] == (['update_w', 'prepare_x', 'tile_y', 'xpass_tiled', 'update_tau',
                          'update_alpha'] + per_iter + ['xjoin'])
    # explicit node order, as VB.update(*nodes) allows (vmp.py:139-141)
    Q.update(Q['X'], Q['W'], repeat=1, verbose=False)
    assert Q.plans[0].kernels.calls[-4:] == ['prepare_x', 'xpass_tiled', 'update_w', 'xjoin']
    # layout='rows': the pass over the row-major array
    Q3 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q3.plans[0].plate_layout = 'rows'
    Q3.update(repeat=1, verbose=False)
    assert 'xpass' in Q3.plans[0].kernels.calls and 'tile_y' not in Q3.plans[0].kernels.calls
    np.testing.assert_array_equal(Q3.L[:1], Q.L[:1])
    # streaming-statistics form: one fused pass per iteration instead
    Q2 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3), 'stream')
    Q2.update(repeat=1, verbose=False)
    assert Q2.plans[0].kernels.calls == ['stats_from_x', 'update_w', 'prepare_x', 'pass',
                                         'update_tau', 'update_alpha']


def test_rotation_matches_reference(golden_dir):
    """RotationOptimizer + RotateGaussianARD on the fused block (K x K optimisation on the
    host, state rotated through the plan) against the live-reference trace."""
    from bayespy_amd.inference import transformations
    from models import run_rotation_sequence, check_rotation_results
    g = np.load(os.path.join(golden_dir, 'rotations.npz'))
    K = g['rot_x0'].shape[1]
    Q = _attach_cpu(build_pca(nodes, VB, g['rot_y'], g['rot_x0'], K))
    res = run_rotation_sequence(Q, K, transformations)
    check_rotation_results(res, g, 'rot')
    with pytest.raises(ValueError):
        transformations.RotateGaussianARD(Q['X'], subset=[0, 0])
    with pytest.raises(ValueError):
        transformations.RotateGaussianARD(Q['W'], Q['alpha'], axis=1)


def test_rotation_of_a_subset_of_components(golden_dir):
    """``subset=`` (transformations.py:425-455, :639-690).  The reference's own path raises
    AttributeError in setup() unless a plate rotation is requested as well (``self.X`` is only
    set there, :619/:645), so there is no trace to pin: the properties of the transformation are
    checked instead -- the bound does not decrease, <W><X>^T is unchanged, components outside
    the subset are untouched, the full-set rotation is the special case."""
    import warnings
    from bayespy_amd.inference import transformations
    g = np.load(os.path.join(golden_dir, 'rotations.npz'))
    K = g['rot_x0'].shape[1]
    sub = [0, 2]

    def rotated(subset):
        Q = _attach_cpu(build_pca(nodes, VB, g['rot_y'], g['rot_x0'], K))
        Q.update(repeat=2, verbose=False)
        before = (Q.compute_lowerbound(), np.array(Q['W'].u[0]), np.array(Q['X'].u[0]))
        R = transformations.RotationOptimizer(
            transformations.RotateGaussianARD(Q['W'], Q['alpha'], subset=subset),
            transformations.RotateGaussianARD(Q['X'], subset=subset),
            K if subset is None else len(subset))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            R.rotate()
        return before, (Q.compute_lowerbound(), np.array(Q['W'].u[0]), np.array(Q['X'].u[0]))

    (L0, W0, X0), (L1, W1, X1) = rotated(sub)
    assert L1 > L0
    rest = [k for k in range(K) if k not in sub]
    np.testing.assert_array_equal(W1[..., rest], W0[..., rest])
    np.testing.assert_array_equal(X1[..., rest], X0[..., rest])
    assert np.abs(W1[..., sub] - W0[..., sub]).max() > 1e-3
    recon = lambda w, x: np.einsum('dik,ink->dn', w, x)       # noqa: E731
    np.testing.assert_allclose(recon(W1, X1), recon(W0, X0), rtol=1e-9, atol=1e-10)
    # all components as a subset = the plain rotation
    (_, _, _), (La, Wa, Xa) = rotated(list(range(K)))
    (_, _, _), (Lb, Wb, Xb) = rotated(None)
    np.testing.assert_allclose(La, Lb, rtol=1e-12)
    np.testing.assert_allclose(Wa, Wb, rtol=1e-9, atol=1e-12)


def test_reference_pca_doctest_known_answer(golden_dir):
    """doc/source/examples/pca.rst:114-118: first and converged lower bound of the
    reference's PCA example (ARD + rotation callback, run to convergence)."""
    from bayespy_amd.inference import transformations
    from models import run_pca_doctest, check_pca_doctest
    g = np.load(os.path.join(golden_dir, 'pca_doctest.npz'))
    Q, nd = run_pca_doctest(nodes, VB, transformations, g, attach=_attach_cpu)
    assert isinstance(Q.plans[0], PCAPlan)
    check_pca_doctest(Q, nd, g)


def test_checkpoint_round_trip(golden_dir, tmp_path):
    """VB.save / VB.load (vmp.py:237-356): a restored model continues bit-for-bit."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(repeat=2, verbose=False)
    fn = str(tmp_path / 'ckpt.bin')
    Q.save(filename=fn)
    Q.update(repeat=3, verbose=False)
    Q2 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q2.load(filename=fn)
    assert Q2.iter == 2 and np.array_equal(Q2.L[:2], Q.L[:2])
    Q2.update(repeat=3, verbose=False)
    assert np.array_equal(Q2.L[:5], Q.L[:5])
    np.testing.assert_array_equal(Q2['W'].u[0], Q['W'].u[0])
    np.testing.assert_array_equal(Q2['X'].u[0], Q['X'].u[0])
    with pytest.raises(Exception, match='Filename'):
        Q.save()
    # autosave every second iteration writes the same container
    fn2 = str(tmp_path / 'auto.bin')
    Q3 = build_pca(nodes, VB, g['y'], g['x0'], 3, autosave_filename=fn2, autosave_iterations=2)
    _attach_cpu(Q3)
    Q3.update(repeat=2, verbose=False)
    assert os.path.exists(fn2)
    # set_autosave / user_data (vmp.py:121-126, :286-305); the save happens in the end-of-iteration
    # step, i.e. also on the iteration that converges
    fn3 = str(tmp_path / 'auto3.bin')
    Q4 = build_pca(nodes, VB, g['y'], g['x0'], 3,
                   user_data={'seed': 7, 'note': np.arange(3.0), 'cfg': {'a': [1, 'x'], 'b': None}})
    _attach_cpu(Q4)
    Q4.set_autosave(fn3, iterations=1, nodes=[Q4['W'], Q4['tau'], Q4['alpha'], Q4['X'], Q4['Y']])
    Q4.ignore_bound_checks = False
    Q4.update(repeat=50, tol=1e-2, verbose=False)
    assert Q4.converged and os.path.exists(fn3)
    ud = VB.load_user_data(fn3)
    assert int(ud['seed']) == 7 and np.array_equal(ud['note'], np.arange(3.0))
    assert ud['cfg'] == {'a': [1, 'x'], 'b': None}          # non-array values travel as JSON
    Q5 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q5.load(filename=fn3)
    assert Q5.iter == Q4.iter and Q5.converged
    # without a file name the autosave goes to a temporary file
    Q6 = build_pca(nodes, VB, g['y'], g['x0'], 3, autosave_iterations=1)
    _attach_cpu(Q6)
    Q6.update(repeat=1, verbose=False)
    assert os.path.exists(Q6.autosave_filename)


def test_second_vb_over_the_same_nodes_continues(golden_dir):
    """The reference keeps q in the nodes: VB(...) over already-updated nodes continues from their
    posteriors.  Here the plans own the state and are kept when they cover what is asked for."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(repeat=4, verbose=False)
    Qa = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Qa.update(repeat=2, verbose=False)
    plan = Qa.plans[0]
    Qb = VB(*Qa.model)
    assert Qb.plans[0] is plan and Qb.iter == 0
    Qb.update(repeat=2, verbose=False)
    assert np.array_equal(Qb.L[:2], Q.L[2:4])
    # another engine cannot take the state over: a fresh plan, and a warning that says so
    with pytest.warns(UserWarning, match='already hold posterior state'):
        try:
            VB(*Qa.model, engine='generic')
        except Exception:       # noqa: BLE001 -- no device in the CPU suite
            pass


def test_lower_bound_cache_and_observed_skip(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q['Y'].update()           # observed -> no-op (stochastic.py:277)
    assert Q.plans[0].kernels.calls == []
    Q.update(repeat=1, verbose=False)
    a = Q.compute_lowerbound()
    b = Q.compute_lowerbound()
    assert a == b == Q.L[0]


def test_plan_selection_and_loud_failures():
    from bayespy_amd.inference.plans.generic import GenericPlan
    from bayespy_amd.nodes.node import Stochastic
    K, D, N = 3, 4, 10
    # config 1 (quickstart model): no fused block -> generic device message passing
    mu = nodes.GaussianARD(0, 1e-6, name='mu')
    tau = nodes.Gamma(1e-6, 1e-6, name='tau')
    y = nodes.GaussianARD(mu, tau, plates=(N,), name='y')
    Q = VB(y, mu, tau)
    assert isinstance(Q.plans[0], GenericPlan) and len(Q.plans) == 1
    # a fully observed PCA block is fused ...
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,))
    W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1))
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, N))
    Y = nodes.GaussianARD(nodes.Dot(W, X), nodes.Gamma(1e-2, 1e-2))
    Q2 = VB(Y, W, X)
    assert isinstance(Q2.plans[0], PCAPlan)
    # ... and moves to the fused missing-data block as soon as data are missing (array mask)
    from bayespy_amd.inference.plans.masked_pca import MaskedPCAPlan
    Y.observe(np.zeros((D, N)), mask=np.ones((D, N), dtype=bool))
    assert isinstance(Q2.plans[0], MaskedPCAPlan)
    Y.observe(np.zeros((D, N)))                        # the mask is gone again
    assert isinstance(Q2.plans[0], PCAPlan)
    Y.observe(np.zeros((D, N)), mask=np.ones((D, N), dtype=bool))
    assert VB(Y, W, X, engine='generic') is not None

    # the fused block starts tau / alpha from their priors and updates every role: a fixed
    # (observed) tau, an initialised alpha or initialize_from_parameters keep per-node state like
    # the reference -> generic engine, before and after VB(...) (never silently ignored)
    def fresh():
        al = nodes.Gamma(1e-2, 1e-2, plates=(K,))
        Wn = nodes.GaussianARD(0, al, shape=(K,), plates=(D, 1))
        Xn = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, N))
        ta = nodes.Gamma(1e-2, 1e-2)
        Yn = nodes.GaussianARD(nodes.Dot(Wn, Xn), ta)
        Yn.observe(np.zeros((D, N)))
        return Yn, Wn, Xn, ta, al
    tweaks = [lambda Yn, Wn, Xn, ta, al: ta.observe(2.0),
              lambda Yn, Wn, Xn, ta, al: al.initialize_from_value(np.ones(K)),
              lambda Yn, Wn, Xn, ta, al: Wn.initialize_from_parameters(np.zeros(K), np.ones(K)),
              lambda Yn, Wn, Xn, ta, al: Wn.observe(np.ones((D, 1, K)))]
    for tw in tweaks:
        ns = fresh()
        tw(*ns)
        assert isinstance(VB(*ns).plans[0], GenericPlan)
        ns = fresh()
        Qf = VB(*ns)
        assert isinstance(Qf.plans[0], PCAPlan)
        tw(*ns)
        assert isinstance(Qf.plans[0], GenericPlan)
    with pytest.raises(ValueError):
        Y.observe(np.zeros((D + 1, N)))
    with pytest.raises(ValueError):
        Y.observe(np.zeros((D, N)), mask=np.ones((D, N + 1), dtype=bool))

    # node types without device formulas fail loudly -- there is no CPU fallback
    class Exotic(Stochastic):
        pass
    with pytest.raises(NotImplementedError, match='no device family'):
        VB(Exotic(plates=(3,), dims=((),)))


def test_plates_and_shapes_follow_reference_rules():
    K, D, N = 5, 7, 11
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,))
    W = nodes.GaussianARD(0, alpha, shape=(K,), plates=(D, 1))
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, N))
    F = nodes.SumMultiply('i,i', W, X)
    Y = nodes.GaussianARD(F, nodes.Gamma(1e-2, 1e-2))
    assert W.plates == (D, 1) and W.dims == ((K,), (K, K))
    assert F.plates == (D, N) and F.dims == ((), ())
    assert Y.plates == (D, N) and Y.dims == ((), ())
    assert PCAPlan.match([Y, F, W, X]) is not None
    F2 = nodes.SumMultiply(W, [0], X, [0])
    assert F2.plates == (D, N) and F2.out_keys == []
    # W and X now feed two blocks: no longer a private PCA block
    assert PCAPlan.match([Y, F, W, X]) is None
    with pytest.raises(ValueError):
        nodes.SumMultiply('i,i', W, nodes.GaussianARD(0, 1, shape=(K + 1,), plates=(1, N)))


def test_node_state_access_on_the_fused_block(golden_dir):
    """node.phi / node.mask of the fused PCA block (SURVEY.md 8b "node state read by users");
    what the block does not carry fails loudly and names the generic engine."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(repeat=2, verbose=False)
    W, X, tau, alpha = Q['W'], Q['X'], Q['tau'], Q['alpha']
    for nd in (W, X):
        phi = nd.phi
        assert phi[0].shape == nd.plates + (3,)
        cov = np.linalg.inv(-2 * phi[1][0, 0])
        np.testing.assert_allclose(phi[0] @ cov, nd.u[0], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(nd.u[1][0, 0] - np.outer(nd.u[0][0, 0], nd.u[0][0, 0]),
                                   cov, rtol=1e-8, atol=1e-12)
    for nd in (tau, alpha):
        phi = nd.phi                     # [-b, a]: <x> = a / b  (gamma.py:116-148)
        np.testing.assert_allclose(phi[1] / -phi[0], nd.u[0], rtol=1e-12)
    assert bool(W.mask) is True
    with pytest.raises(NotImplementedError, match="engine='generic'"):
        Q.set_annealing(0.5)
    with pytest.raises(NotImplementedError, match="engine='generic'"):
        Q.optimize(W, tau, collapsed=[X, alpha], maxiter=1, verbose=False)
    with pytest.raises(NotImplementedError, match="engine='generic'"):
        W.get_riemannian_gradient()
    Q.set_annealing(1.0)                 # the standard updates are always allowed


def test_bound_of_point_mass_states_and_warning_on_a_restart(golden_dir):
    """initialize_from_value leaves a point mass: the bound is -inf until that node is updated
    (expfamily.py:193-212, :433-447).  A random re-initialisation after updates restarts the fused block from the
    nodes' initial values -- unlike the reference, hence a loud warning (re-observing Y keeps the
    posteriors, and so does initialize_from_value: test_reobserving_data_keeps_the_posteriors,
    test_reinitialising_a_node_keeps_the_other_posteriors)."""
    import warnings
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    assert Q.compute_lowerbound() == -np.inf
    assert Q['X'].lower_bound_contribution() == -np.inf
    assert np.isfinite(Q['W'].lower_bound_contribution())
    Q.update(Q['W'], repeat=1, verbose=False)
    assert Q.compute_lowerbound() == -np.inf
    Q.update(Q['X'], repeat=1, verbose=False)
    assert np.isfinite(Q.compute_lowerbound())
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        Q2 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
        Q2['Y'].observe(g['y'])               # nothing learned yet: silent
    with pytest.warns(RuntimeWarning, match='restarts from the initial state'):
        Q['X'].initialize_from_random()      # a draw from the CURRENT q is not kept by the block: restart


def test_logging_switch_and_per_node_traces(golden_dir, caplog):
    import logging
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.use_logging(True)
    with caplog.at_level(logging.INFO):
        Q.update(repeat=2)
    assert any('Iteration 1: loglike=' in r.getMessage() for r in caplog.records)
    Q.use_logging(False)
    traces = Q.get_iteration_by_nodes()
    assert traces is Q.l and set(traces) >= {Q['W'], Q['X'], Q['tau'], Q['alpha'], Q['Y']}
    np.testing.assert_allclose(sum(traces[n][:2] for n in traces), Q.L[:2], rtol=1e-12)


def test_checkpoint_keeps_point_mass_states(golden_dir, tmp_path):
    """A model saved while X is still the point mass of initialize_from_value reports -inf after
    load as well (the reference stores g = inf with the node, vmp.py:237-285)."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(Q['W'], repeat=1, verbose=False)
    assert Q.compute_lowerbound() == -np.inf
    fn = str(tmp_path / 'delta.bin')
    Q.save(filename=fn)
    Q2 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q2.load(filename=fn)
    assert Q2.compute_lowerbound() == -np.inf
    Q2.update(Q2['X'], repeat=1, verbose=False)
    Q.update(Q['X'], repeat=1, verbose=False)
    assert np.isfinite(Q2.compute_lowerbound())
    assert Q2.compute_lowerbound() == Q.compute_lowerbound()


@pytest.mark.parametrize('stats', ['gram', 'stream'])
def test_reobserving_data_keeps_the_posteriors(golden_dir, stats):
    """Y.observe(new data) after updates changes Y only (stochastic.py:223-273): the fused PCA block
    keeps q(W), q(X), q(tau), q(alpha) and recomputes the messages of the new data with the current
    <x> -- live-reference trace tests/golden/reobserve.npz; no restart, no warning."""
    import warnings
    from models import run_reobserve_case
    g = np.load(os.path.join(golden_dir, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}

    class CPUVB(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            _attach_cpu(self, stats)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        res = run_reobserve_case(nodes, CPUVB, inp)
    np.testing.assert_allclose(res['L'], g['L'], rtol=1e-10)
    np.testing.assert_allclose(res['L_mid'], g['L_mid'], rtol=1e-10)
    np.testing.assert_allclose(res['L_w'], g['L_w'], rtol=1e-10)
    np.testing.assert_allclose(res['W_u0'], g['W_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(res['X_u0'], g['X_u0'], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('stats', ['gram', 'stream'])
def test_reinitialising_a_node_keeps_the_other_posteriors(golden_dir, stats):
    """initialize_from_value on X, later on W, between updates (expfamily.py:193-204): live-reference
    trace tests/golden/reobserve.npz (ri_*); the re-initialised node is a point mass (bound -inf)
    until its next update, nothing else restarts."""
    import warnings
    from models import run_reinitialise_case
    g = np.load(os.path.join(golden_dir, 'reobserve.npz'))
    inp = {k[3:]: g[k] for k in g.files if k.startswith('in_')}

    class CPUVB(VB):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            _attach_cpu(self, stats)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        res = run_reinitialise_case(nodes, CPUVB, inp)
    np.testing.assert_allclose(res['ri_steps'], g['ri_steps'], rtol=1e-10)
    np.testing.assert_allclose(res['ri_L'], g['ri_L'], rtol=1e-10)
    np.testing.assert_allclose(res['ri_W_u0'], g['ri_W_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(res['ri_X_u0'], g['ri_X_u0'], rtol=1e-8, atol=1e-10)


def test_near_misses_of_the_fused_blocks_are_reported():
    """VERDICT r02 #6a: a model that resembles a fused block but misses its matcher runs on the
    generic engine WITH a warning that says which condition failed; engine='generic' is silent."""
    import warnings
    from bayespy_amd.inference.plans.generic import GenericPlan
    D, N, K = 4, 6, 2

    def build(mu_w=0.0, extra_child=False):
        al = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
        Wn = nodes.GaussianARD(mu_w, al, shape=(K,), plates=(D, 1), name='W')
        Xn = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
        ta = nodes.Gamma(1e-2, 1e-2, name='tau')
        Yn = nodes.GaussianARD(nodes.Dot(Wn, Xn), ta, name='Y')
        Yn.observe(np.zeros((D, N)))
        ns = [Yn, Wn, Xn, ta, al]
        if extra_child:
            Z = nodes.GaussianARD(nodes.Dot(Wn, Xn), 1.0, name='Y2')
            Z.observe(np.ones((D, N)))
            ns.append(Z)
        return ns

    def hyper_mean():
        ns = build(mu_w=nodes.GaussianARD(0, 1, shape=(K,), name='m'))
        return ns + [ns[1].parents[0]]

    with pytest.warns(UserWarning, match='prior mean of W is not a constant'):
        Q = VB(*hyper_mean())
    assert isinstance(Q.plans[0], GenericPlan)
    with pytest.warns(UserWarning, match='has other children as well'):
        Q = VB(*build(extra_child=True))
    assert isinstance(Q.plans[0], GenericPlan)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        Q = VB(*hyper_mean(), engine='generic')              # asked for: no warning
        assert isinstance(Q.plans[0], GenericPlan)
        Q = VB(*build())                                     # the fused block: no warning
        assert not isinstance(Q.plans[0], GenericPlan)
        Q = VB(*build(mu_w=1.0))                             # a constant prior mean as well
        assert not isinstance(Q.plans[0], GenericPlan)


def test_tile_major_x_is_invisible_behind_the_row_major_view(golden_dir):
    """Opt-in BAYESPY_AMD_PCA_XTILES: after the first tile-major pass <x> lives tile-major and
    ``PCAPlan.Xd`` forms the row-major copy on demand.  Same trace, moments, rotation and
    checkpoint contents as with the row-major array."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))
    out = []
    for tiles in (False, True):
        Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
        plan = Q.plans[0]
        plan.kernels.x_tiles = tiles
        Q.update(repeat=2, verbose=False)
        assert (plan._Xt is not None) == tiles and plan._x_form == ('tiled' if tiles else 'rows')
        x1 = Q['X'].u[0].copy()
        if tiles:
            assert plan.kernels.calls.count('tile_x') == 1      # formed once, kept until the next pass
            Q['X'].u[0]
            assert plan.kernels.calls.count('tile_x') == 1
        # a rotation changes the row-major array in place: it becomes the current <x>
        R = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 2.0]])
        plan.rotate_node(Q["X"], R, np.linalg.inv(R), np.log(2.0))
        x2 = Q['X'].u[0].copy()
        np.testing.assert_allclose(x2[0], x1[0] @ R.T, rtol=1e-13)
        Q.update(repeat=2, verbose=False)
        out.append((Q.L[:4].copy(), x1, x2, Q['X'].u[0].copy(), Q['W'].u[0].copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)



def build_pca_with_mean(g, tag):
    """The models of oracle/make_golden.py:pca_mean_case."""
    y, x0, mu = g[tag + '_y'], g[tag + '_x0'], g[tag + '_mu']
    D, N = y.shape
    K = x0.shape[1]
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
    W = nodes.GaussianARD(mu, alpha, shape=(K,), plates=(D, 1), name='W')
    X = nodes.GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
    F = nodes.SumMultiply('i,i', W, X, name='F')
    tau = nodes.Gamma(1e-2, 1e-2, name='tau')
    Y = nodes.GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None, :, :])
    if tag == 'mk':
        W.initialize_from_value(g[tag + '_w0'][:, None, :])
    Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha)
    Q.ignore_bound_checks = True
    order = ('W', 'X', 'tau', 'alpha') if tag == 'm3' else ('X', 'W', 'tau', 'alpha')
    return Q, [Q[k] for k in order]


def check_pca_with_mean(Q, order, g, tag):
    n = len(g[tag + '_L'])
    Q.update(*order, repeat=n, verbose=False)
    np.testing.assert_allclose(Q.L[:n], g[tag + '_L'], rtol=1e-10)
    for k in ('Y', 'X', 'W', 'tau', 'alpha'):
        np.testing.assert_allclose(Q.l[Q[k]][:n], g[tag + '_L_' + k], rtol=1e-8, atol=1e-7,
                                   err_msg=k)
    np.testing.assert_allclose(Q['W'].u[0], g[tag + '_W_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['W'].u[1], g[tag + '_W_u1'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['X'].u[0], g[tag + '_X_u0'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Q['tau'].u[0], g[tag + '_tau_u0'], rtol=1e-10)
    np.testing.assert_allclose(Q['alpha'].u[0], g[tag + '_alpha_u0'], rtol=1e-9)
    np.testing.assert_allclose(Q['alpha'].phi[0], g[tag + '_alpha_phi0'], rtol=1e-9)


@pytest.mark.parametrize('stats', ['gram', 'stream'])
@pytest.mark.parametrize('tag', ['m3', 'mk', 'ms'])
def test_constant_prior_mean_of_w_matches_reference(golden_dir, tag, stats):
    """GaussianARD(mu, alpha) for W with a constant mu != 0 stays on the fused block: an array of
    shape (D, 1, K), of shape (K,) with W started from a value, a scalar with W started from its
    prior (live-reference traces, oracle/make_golden.py:pca_mean_case)."""
    g = np.load(os.path.join(golden_dir, 'pca_prior_mean.npz'))
    Q, order = build_pca_with_mean(g, tag)
    assert isinstance(Q.plans[0], PCAPlan) and Q.plans[0].mu0 is not None
    _attach_cpu(Q, stats)
    check_pca_with_mean(Q, order, g, tag)
    from bayespy_amd.inference import transformations
    with pytest.raises(NotImplementedError):
        transformations.RotateGaussianARD(Q['W'], Q['alpha']).setup()


def test_prior_mean_with_missing_values_goes_to_the_generic_engine(golden_dir):
    """The missing-data block keeps requiring a zero prior mean."""
    from bayespy_amd.inference.plans.masked_pca import MaskedPCAPlan
    g = np.load(os.path.join(golden_dir, 'pca_prior_mean.npz'))
    y, mu = g['ms_y'], g['ms_mu']
    D, N = y.shape
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(2,), name='alpha')
    W = nodes.GaussianARD(mu, alpha, shape=(2,), plates=(D, 1), name='W')
    X = nodes.GaussianARD(0, 1, shape=(2,), plates=(1, N), name='X')
    F = nodes.SumMultiply('i,i', W, X, name='F')
    tau = nodes.Gamma(1e-2, 1e-2, name='tau')
    Y = nodes.GaussianARD(F, tau, name='Y')
    Y.observe(y, mask=np.random.RandomState(0).rand(D, N) < 0.8)
    assert MaskedPCAPlan.match([Y, F, W, X, tau, alpha]) is None
    assert PCAPlan.match([Y, F, W, X, tau, alpha]) is None


def test_checkpoint_of_the_older_state_layout_loads(golden_dir):
    """The packed state grew at its END in round 5 (prior mean of W and its sums): the state of a
    file written before that is the prefix of today's and loads into a zero-mean model; a state of
    any other length is refused with a message instead of a shape error from the copy."""
    g = np.load(os.path.join(golden_dir, 'pca_n500_d6_k3.npz'))

    class Reader:
        def __init__(self, items):
            self.items = items

        def has(self, k):
            return k in self.items

        def get(self, k):
            return self.items[k]

    Q = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q.update(repeat=2, verbose=False)
    items = {}
    # (copies: on the CPU double the saved views alias the live arrays)
    Q.plans[0].save_state(lambda k, v: items.__setitem__(k, np.array(v)), [], 0)
    old_len = int(Q.plans[0].layout.off_mu)              # where the state ended before round 5
    assert old_len < items['plans/0/state'].size
    items['plans/0/state'] = items['plans/0/state'][:old_len]
    Q.update(repeat=2, verbose=False)
    Q2 = _attach_cpu(build_pca(nodes, VB, g['y'], g['x0'], 3))
    Q2.update(repeat=1, verbose=False)                    # (device state exists; the load overwrites it)
    Q2.plans[0].load_state(Reader(items), [], 0)
    Q2.iter = 2
    Q2.update(repeat=2, verbose=False)
    np.testing.assert_array_equal(np.array(Q2.L[2:4]), np.array(Q.L[2:4]))
    items['plans/0/state'] = items['plans/0/state'][:-3]
    with pytest.raises(ValueError, match='state values'):
        Q2.plans[0].load_state(Reader(items), [], 0)
