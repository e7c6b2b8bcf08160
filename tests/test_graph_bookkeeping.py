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

    def __init__(self, nodes, world=1, device='cuda'):
        self.all = list(nodes)
        self.state = {id(n): object() for n in nodes}
        self._graph_init()
        self.recorded, self.replayed = 0, 0
        self.fail_recording = False
        self.rt = types.SimpleNamespace(device=types.SimpleNamespace(type=device), world=world,
                                        _refresh_dist=lambda: None)

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
    assert _sweeps(_Plan([a], world=2), [a], [a], 5) == [False] * 5      # sharded: all-reduces
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
