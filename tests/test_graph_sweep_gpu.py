"""
GPU: sweeps of the generic engine replayed from a recorded HIP graph (plans/graph_iter.py) against
the same sweeps issued launch by launch (``BAYESPY_AMD_GRAPH=0``).  A replay is the device work of
the eager sweep, kernel for kernel, so everything must be BIT-identical: bound traces, moments, and
the behaviour around operations that interleave with the sweeps -- read-only ones (the graph stays),
mutating ones (the graph is dropped and recorded again), and models whose sweep needs the host
(the recording is abandoned, the eager path continues from the restored state).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pca(N=20000, D=24, K=6, seed=5):
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    from models import build_pca, make_seeded_pca
    y, x0 = make_seeded_pca(seed, N, D, K)
    return build_pca(nodes, VB, y, x0, K, engine='generic')


def _script(Q):
    """Sweeps with other operations in between; returns everything a caller could observe."""
    out = []
    W, X, tau = Q['W'], Q['X'], Q['tau']
    Q.update(repeat=6, verbose=False)
    out.append(W.get_moments()[0])                       # read-only between sweeps
    out.append(np.array(Q.compute_lowerbound()))
    Q.update(repeat=3, verbose=False)
    X.update()                                           # a single update: not the recorded sweep
    Q.update(repeat=5, verbose=False)
    Q.update(W, X, repeat=4, verbose=False)              # another sweep (a subset of the nodes)
    Q.update(repeat=4, verbose=False)
    for n in (W, X, tau, Q['alpha']):
        out.extend(n.get_moments())
    out.append(Q.L[:Q.iter].copy())
    return out, W._plan.graph_info()


def test_graph_replay_is_bit_identical_to_eager_sweeps(monkeypatch):
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    eager, info0 = _script(_pca())
    assert not info0['recorded']
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '1')
    graph, info1 = _script(_pca())
    assert info1['recorded'] and info1['replays'] >= 2 and info1['disabled'] is None, info1
    assert len(eager) == len(graph)
    for a, b in zip(eager, graph):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_new_data_drops_the_graph_and_rotation_callbacks_keep_it(monkeypatch):
    """observe() between sweeps changes what a recording has baked in (dropped, recorded again); a
    rotation callback after every sweep (the reference's idiom, demos/pca.py:63-68) only replaces
    state arrays: the sweep is recorded with the callback in the loop, every replay first copies the
    rotated state into the graph's inputs, and the bound is evaluated on the rotated state."""
    import warnings
    from bayespy_amd.inference.transformations import RotateGaussianARD, RotationOptimizer

    def run():
        Q = _pca(N=5000)
        Y, W, X = Q['Y'], Q['W'], Q['X']
        Q.update(repeat=5, verbose=False)
        rs = np.random.RandomState(1)
        Y.observe(np.asarray(Y.get_moments()[0]) + 0.01 * rs.normal(size=(24, 5000)))
        Q.update(repeat=5, verbose=False)
        rot = RotationOptimizer(RotateGaussianARD(W, Q['alpha']), RotateGaussianARD(X), 6)
        Q.callback = rot.rotate
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Q.update(repeat=6, verbose=False)
        return [Q.L[:Q.iter].copy()] + list(W.get_moments()) + list(X.get_moments()), \
            W._plan.graph_info()
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    eager, _ = run()
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '1')
    graph, info = run()
    assert info['recorded'] and info['replays'] >= 3, info       # replays WITH the callback
    for a, b in zip(eager, graph):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_recording_that_needs_the_host_falls_back(monkeypatch):
    """A sweep that reads device data on the host cannot be recorded: the attempt is abandoned
    before HIP sees the read, the state is restored, results equal the eager run's."""
    import bayespy_amd.inference.plans.generic as G
    orig = G.GammaFamily.moments_and_cgf

    def peeking(self, phi):
        out = orig(self, phi)
        G._arr(out[0][0]).numpy()              # a host read inside the sweep
        return out

    def run():
        Q = _pca(N=4000)
        Q.update(repeat=8, verbose=False)
        return [Q.L[:Q.iter].copy()] + list(Q['X'].get_moments()), Q['W']._plan.graph_info()
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    eager, _ = run()
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '1')
    monkeypatch.setattr(G.GammaFamily, 'moments_and_cgf', peeking)
    graph, info = run()
    assert not info['recorded'] and 'needs the host' in info['disabled'], info
    for a, b in zip(eager, graph):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize('case', ['count', 'plate', 'markov'])
def test_other_model_families_agree_with_eager(case, golden_dir, monkeypatch):
    """Mixtures, categorical / count nodes, chains: whatever part of their sweeps is recorded or
    not, the results equal the eager ones bit for bit."""
    import bayespy_amd.nodes as nodes
    from bayespy_amd.inference import VB
    import models
    fn, gold = {'count': (models.run_count_node_cases, 'count_nodes.npz'),
                'plate': (models.run_plate_node_cases, 'plate_nodes.npz'),
                'markov': (models.run_markov_chain_cases, 'markov_chains.npz')}[case]
    f = np.load(os.path.join(golden_dir, gold))
    g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    eager = fn(nodes, VB, g)
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '1')
    graph = fn(nodes, VB, g)
    assert sorted(eager) == sorted(graph)

    def same(a, b):
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return np.array_equal(np.asarray(a), np.asarray(b))
    for k in eager:
        assert same(eager[k], graph[k]), k


@pytest.mark.parametrize('mode', ['formulas_only', 'graph_without_queue', 'eager'])
def test_queue_of_small_operations_agrees_with_its_opt_outs(mode, monkeypatch):
    """The queue of small operations: by default formulas, small plate sums and
    K x K inverses of a sweep are records of a few interpreter launches, in eager sweeps and -- as
    nodes of the graph -- in recorded ones.  Against the opt-outs: formulas only (tune small_queue_sm
    = 0: sums by the stand-alone kernels, another order of the additions), recorded sweeps with one
    node per operation (BAYESPY_AMD_GRAPH_QUEUE=0), eager sweeps (BAYESPY_AMD_GRAPH=0: the same
    queued arithmetic, so bit for bit)."""
    from bayespy_amd.device import get_runtime

    def run():
        Q = _pca(N=5000, D=40, K=12)             # K > 8: the inverses are queueable
        Q.update(repeat=8, verbose=False)
        return Q.L[:8].copy(), Q['W'].get_moments()[0], Q['W']._plan.graph_info()

    rt = get_runtime()
    s0 = rt.queue_stats()
    L0, W0, info0 = run()
    s1 = rt.queue_stats()
    assert info0['recorded']
    # operations were in fact queued: far fewer interpreter launches than records
    assert s1['operations'] - s0['operations'] > 3 * (s1['launches'] - s0['launches']) > 0
    if mode == 'formulas_only':
        rt.set_tune('small_queue_sm', 0)
    elif mode == 'eager':
        monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    else:
        monkeypatch.setenv('BAYESPY_AMD_GRAPH_QUEUE', '0')
    try:
        L1, W1, info1 = run()
    finally:
        rt.set_tune('small_queue_sm', 1)
    assert info1['recorded'] == (mode != 'eager')
    if mode == 'eager':
        assert np.array_equal(L1, L0) and np.array_equal(np.asarray(W1), np.asarray(W0))
    np.testing.assert_allclose(L1, L0, rtol=1e-12)
    np.testing.assert_allclose(W1, W0, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('knob', ['BAYESPY_AMD_PRIOR_CACHE', 'BAYESPY_AMD_BOUND_FOLD'])
def test_record_reductions_do_not_change_a_bit(knob, monkeypatch):
    """Round 6 removed records from a sweep without touching its arithmetic: the prior terms of a
    node whose parents are constants are formed once instead of twice per sweep, and a scalar moment's
    bound term joins the running sum in one formula (the interpreter executes the same operations in
    the same order).  With either switched off the traces and moments are the same bits."""
    def run():
        Q = _pca(N=6000, D=20, K=5, seed=9)
        Q.update(repeat=7, verbose=False)
        return [Q.L[:7].copy()] + [np.asarray(m) for n in ('W', 'tau', 'alpha') for m in Q[n].get_moments()]
    on = run()
    monkeypatch.setenv(knob, '0')
    off = run()
    for a, b in zip(on, off):
        assert np.array_equal(a, b)
