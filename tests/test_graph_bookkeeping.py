"""
CPU: the bookkeeping that decides when a sweep of the generic engine may be replayed from its
recorded graph (plans/graph_iter.py) -- without a device: recording and replay are stubbed, the
decisions are the real code.  A graph is recorded after two eager sweeps that followed the expected
pattern, kept across read-only operations, dropped by anything else; its bound terms answer the next
``lower_bound_contributions`` of the same nodes once.
"""
import types

import numpy as np
import pytest

from bayespy_amd.inference.plans.graph_iter import GraphIteration, READ_ONLY


class _Node:
    def __init__(self, name):
        self.name = name
        self.annealing = 1.0
        self.plates_multiplier = (1,)
        self.observed = False


class _Plan(GraphIteration):
    """The decisions of GraphIteration over stubbed device work."""

    def __init__(self, nodes, world=1, device='cuda', library_comm=False):
        self.all = list(nodes)
        self.state = {id(n): object() for n in nodes}
        self._graph_init()
        self.recorded, self.replayed = 0, 0
        self.fail_recording = False
        self.rt = types.SimpleNamespace(device=types.SimpleNamespace(type=device), world=world,
                                        _refresh_dist=lambda: None,
                                        _ensure_comm=lambda: library_comm)

    def _update_masks(self):
        pass

    def _graph_record(self, upd, bound, key):
        if self.fail_recording:
            self._g_disabled = 'needs the host: test'
            return None
        self.recorded += 1
        return types.SimpleNamespace(key=key, replays=0, copy_bytes=0)

    def _graph_replay(self, rec):
        self.replayed += 1
        rec.replays += 1
        return [float(self.replayed)] * 2

    # what VB does in an eager sweep, seen through the operation wrapper
    def eager_sweep(self, upd):
        for n in upd:
            self._graph_note('update', (n,))
        self._graph_note('lower_bound_contributions', ())


def _sweeps(plan, upd, bound, n):
    done = []
    for _ in range(n):
        ok = plan.graph_iteration(upd, bound)
        if not ok:
            plan.eager_sweep(upd)
        done.append(ok)
    return done


def test_two_eager_sweeps_then_recorded_and_replayed(monkeypatch):
    monkeypatch.delenv('BAYESPY_AMD_GRAPH', raising=False)
    a, b = _Node('a'), _Node('b')
    p = _Plan([a, b])
    assert _sweeps(p, [a, b], [a, b], 6) == [False, False, True, True, True, True]
    assert p.recorded == 1 and p.replayed == 4
    assert p.graph_info()['recorded'] and p.graph_info()['replays'] == 4
    # the bound terms of the last replay wait for the next lower_bound_contributions of these nodes
    assert p._g_stash == ((id(a), id(b)), [4.0, 4.0])


def test_read_only_operations_keep_the_graph_and_others_drop_it(monkeypatch):
    monkeypatch.delenv('BAYESPY_AMD_GRAPH', raising=False)
    a, b = _Node('a'), _Node('b')
    p = _Plan([a, b])
    _sweeps(p, [a, b], [a, b], 4)
    assert p._g_rec is not None and p.recorded == 1
    for name in sorted(READ_ONLY - {'lower_bound_contributions', 'graph_iteration'}):
        p._graph_note(name, (a,))
    assert p._g_rec is not None and _sweeps(p, [a, b], [a, b], 1) == [True]
    # operations that only replace state arrays keep the graph (the replay copies the present
    # state into its inputs) but the bound terms of the last replay no longer describe the state
    for name in ('update', 'rotate_node', 'set_parameters'):
        assert _sweeps(p, [a, b], [a, b], 1) == [True] and p._g_stash is not None
        p._graph_note(name, (a,))
        assert p._g_rec is not None and p._g_stash is None
    assert _sweeps(p, [a, b], [a, b], 2) == [True, True] and p.recorded == 1
    # a state of another structure ends the recording's life at the next replay
    p._graph_replay = lambda rec: None
    assert _sweeps(p, [a, b], [a, b], 1) == [False] and p._g_rec is None
    del p._graph_replay
    assert _sweeps(p, [a, b], [a, b], 3) == [False, False, True] and p.recorded == 2
    # anything not known to keep the state's structure drops the graph
    p._graph_note('invalidate', (a,))
    assert p._g_rec is None
    assert _sweeps(p, [a, b], [a, b], 4) == [False, False, True, True]
    # another node list is another sweep
    assert _sweeps(p, [a], [a, b], 4) == [False, False, True, True]
    assert p.recorded == 4
    # a rotation callback between the updates and the bound does not keep a sweep from being recorded
    q = _Plan([a, b])
    for _ in range(2):
        assert not q.graph_iteration([a, b], [a, b])
        for n in (a, b):
            q._graph_note('update', (n,))
        q._graph_note('rotate_node', (a,))
        q._graph_note('update', (b,))            # the rotation's update of a hyperparameter
        q._graph_note('lower_bound_contributions', ())
    assert q.graph_iteration([a, b], [a, b]) and q.recorded == 1


def test_host_side_settings_are_part_of_the_key(monkeypatch):
    monkeypatch.delenv('BAYESPY_AMD_GRAPH', raising=False)
    a, b = _Node('a'), _Node('b')
    p = _Plan([a, b])
    _sweeps(p, [a, b], [a, b], 4)
    a.annealing = 0.5                      # deterministic annealing changed between sweeps
    assert _sweeps(p, [a, b], [a, b], 4) == [False, False, True, True]
    b.plates_multiplier = (10,)            # stochastic VI
    assert _sweeps(p, [a, b], [a, b], 1) == [False]
    p._mask_epoch += 1                     # masks recomputed
    assert _sweeps(p, [a, b], [a, b], 3) == [False, False, True]


def test_graphs_are_declined_where_they_cannot_work(monkeypatch):
    a = _Node('a')
    monkeypatch.setenv('BAYESPY_AMD_GRAPH', '0')
    p = _Plan([a])
    assert _sweeps(p, [a], [a], 5) == [False] * 5 and p.recorded == 0
    monkeypatch.delenv('BAYESPY_AMD_GRAPH')
    # sharded: the sweep holds all-reduces -- recorded only when they are the library's own RCCL
    # launches (torch.distributed's gloo collectives cannot be recorded)
    assert _sweeps(_Plan([a], world=2), [a], [a], 5) == [False] * 5
    assert _sweeps(_Plan([a], world=2, library_comm=True), [a], [a], 5) == [False] * 2 + [True] * 3
    assert _sweeps(_Plan([a], device='cpu'), [a], [a], 5) == [False] * 5
    p = _Plan([a])
    p.fail_recording = True
    assert _sweeps(p, [a], [a], 6) == [False] * 6
    assert 'needs the host' in p.graph_info()['disabled']


def test_every_public_operation_of_the_plan_reports():
    """GenericPlan: each public method goes through the bookkeeping (a new method that changes the
    state cannot silently bypass it), and only the listed ones count as read-only."""
    from bayespy_amd.inference.plans.generic import GenericPlan
    for name, fn in vars(GenericPlan).items():
        if name.startswith('_') or not callable(fn) or isinstance(fn, (staticmethod, classmethod)):
            continue
        assert getattr(fn, '_notes_graph', False), name
    mutators = {'update', 'invalidate', 'load_state', 'rotate_node', 'set_parameters',
                'gradient_step'}
    assert not (mutators & READ_ONLY)
    assert mutators <= set(vars(GenericPlan))


@pytest.fixture
def cpu_runtime():
    """DArray wrappers over CPU tensors (no kernel is launched by the code under test here)."""
    from bayespy_amd import device
    prev = device._runtime
    device.set_runtime(device.Runtime(device='cpu'))
    yield
    device.set_runtime(prev)


def test_state_snapshots_and_rewrapping(cpu_runtime):
    """The leaves of a plan's state (what a replay copies into the graph's inputs), the structure
    signature that decides whether a recording still fits, and the re-wrapping after a replay:
    fresh wrapper objects over the SAME tensors, shared wrappers stay shared, lazily evaluated dense
    forms are left behind."""
    import bayespy_amd.inference.plans.generic as G
    from bayespy_amd.darray import DArray
    mean = DArray.from_host(np.arange(12.0).reshape(1, 4, 3))
    cov = DArray.from_host(np.eye(3).reshape(1, 1, 3, 3))
    fm = G.FactoredMoment(cov, mean, 1)
    assert fm.shape == (1, 4, 3, 3) and fm.size == 36 and fm._dense is None
    state = [mean, fm]
    leaves, sig = [], []
    GraphIteration._leaves(state, leaves, sig, ('n', 'u'))
    assert [t.data_ptr() for t in leaves] == [mean.t.data_ptr(), cov.t.data_ptr(), mean.t.data_ptr()]
    kinds = [s[1] for s in sig]
    assert kinds == ['list', 'tensor', 'factored', 'tensor', 'tensor']
    # host-side values are part of the structure, by value
    leaves2, sig2 = [], []
    GraphIteration._leaves([np.inf, None, 2.5], leaves2, sig2, ('n', 'g'))
    assert leaves2 == [] and [s[2] for s in sig2[1:]] == [float('inf'), None, 2.5]
    with pytest.raises(Exception):
        GraphIteration._leaves(object(), [], [], ())
    # re-wrapping: new objects, same tensors, sharing kept, dense form dropped
    fm._dense = object()
    memo = {}
    new = GraphIteration._rewrap(state, memo)
    assert new is not state and new[0] is not mean and new[1] is not fm
    assert new[0].t is mean.t and new[1].cov.t is cov.t
    assert new[1].mean is new[0] and new[1]._dense is None
    assert GraphIteration._rewrap((1.0, None), {}) == (1.0, None)
    # the signature does not depend on whether somebody evaluated the dense form
    leaves3, sig3 = [], []
    GraphIteration._leaves(new, leaves3, sig3, ('n', 'u'))
    assert sig3 == sig


def test_lazy_arrays_know_their_shape_without_being_evaluated(cpu_runtime):
    """LazySum / LazyContract: shape, size and the lazy flag come from the factors; nothing is
    evaluated by asking (consumers that plate-sum them never form the array)."""
    import bayespy_amd.inference.plans.generic as G
    from bayespy_amd.darray import DArray
    W = DArray.from_host(np.ones((5, 1, 3)))
    X = DArray.from_host(np.ones((1, 7, 3)))
    sizes = {'p0': 5, 'p1': 7, 'k0': 3}
    f = G.LazyContract([W, X], [['p0', 'p1', 'k0']] * 2, ['p0', 'p1'], sizes, ['p0', 'p1'])
    assert f.shape == (5, 7) and f.size == 35 and f.ndim == 2 and f._dense is None
    # a compressed plate no operand varies along stays broadcast
    g = G.LazyContract([W], [['p0', 'p1', 'k0']], ['p0', 'p1'], sizes, ['p0', 'p1'])
    assert g.shape == (5, 1)
    made = []
    s = G.LazySum([(1.0, [f, f]), (1.0, [DArray.from_host(np.ones((1, 7)))])], (5, 7),
                  lambda: made.append(1))
    assert s.shape == (5, 7) and s.size == 35 and G._is_lazy(s) and made == []
    assert not G._is_lazy(W) and not G._is_lazy(f)


def test_vb_offers_a_sweep_to_one_generic_plan_only():
    """VB._graph_sweep: the whole model must belong to ONE plan that knows graph_iteration; fully
    observed nodes are not part of the sweep; anything else is visited node by node."""
    from bayespy_amd.inference.vb import VB

    class Plan:
        def __init__(self):
            self.calls = []

        def graph_iteration(self, upd, bound):
            self.calls.append(([n.name for n in upd], [n.name for n in bound]))
            return True

    def node(name, plan, observed=False):
        n = types.SimpleNamespace(name=name, _plan=plan, observed=observed, _fully_observed=True)
        n.update = lambda: None
        return n
    p, q = Plan(), Plan()
    Q = object.__new__(VB)
    y, w, x = node('y', p, observed=True), node('w', p), node('x', p)
    Q.model = [y, w, x]
    assert VB._graph_sweep(Q, ['y', 'w', 'x']) is True
    assert p.calls == [(['w', 'x'], ['y', 'w', 'x'])]
    # a second plan in the model, a plan without graphs, a node outside any plan
    Q.model = [y, w, node('z', q)]
    assert VB._graph_sweep(Q, ['w']) is False
    Q.model = [node('a', object())]
    assert VB._graph_sweep(Q, ['a']) is False
    Q.model = [node('b', None)]
    assert VB._graph_sweep(Q, ['b']) is False


def test_dead_inputs_of_a_recorded_sweep(monkeypatch):
    """Which leaves of the state snapshot the replay's copy-back may leave out: the natural
    parameters and the log-normaliser of the nodes the sweep updates -- not their moments, and
    nothing of a node the sweep only reads (graph_iter.py:_dead_inputs)."""
    from bayespy_amd.inference.plans.graph_iter import GraphIteration

    class N:
        pass
    a, b = N(), N()
    sig = [((id(a), 'u'), 'list', 2), ((id(a), 'u', 0), 'tensor', (4,)), ((id(a), 'u', 1), 'factored', 1),
           ((id(a), 'u', 1, 'cov'), 'tensor', (2, 2)), ((id(a), 'u', 1, 'mean'), 'tensor', (4,)),
           ((id(a), 'phi'), 'list', 2), ((id(a), 'phi', 0), 'tensor', (4,)), ((id(a), 'phi', 1), 'tensor', (2, 2)),
           ((id(a), 'g'), 'tensor', (4,)), ((id(a), 'f'), 'host', None),
           (id(a), 'flags', False, False, True, False),
           ((id(b), 'u', 0), 'tensor', (3,)), ((id(b), 'phi', 0), 'tensor', (3,)), ((id(b), 'g'), 'tensor', ()),
           (id(b), 'flags', False, False, True, False)]
    leaves = list(range(9))
    dead = GraphIteration._dead_inputs(sig, leaves, [a])
    assert dead == [False, False, False, True, True, True, False, False, False]
    assert GraphIteration._dead_inputs(sig, leaves, [a, b]) == [False, False, False, True, True, True,
                                                               False, True, True]
    # a snapshot whose tags do not line up with its leaves, or the opt-out: everything is copied
    assert GraphIteration._dead_inputs(sig, leaves[:-1], [a]) == [False] * 8
    monkeypatch.setenv('BAYESPY_AMD_GRAPH_COPY_ALL', '1')
    assert GraphIteration._dead_inputs(sig, leaves, [a, b]) == [False] * 9
