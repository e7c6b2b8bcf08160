"""
Point-mass initial states in the fused blocks.

``initialize_from_value`` / ``initialize_from_random`` give a node the DELTA moments of a value with
``g = inf`` (expfamily.py:193-212): until the node's first update its term of the lower bound is
-inf (expfamily.py:433-447: ``z = -g``), and so is the total.  The fused blocks keep only moments,
so they track which of their roles are still point masses.
"""
import warnings

import numpy as np


def delta_roles(roles):
    return {k for k, n in roles.items()
            if getattr(n, '_init', None) is not None and n._init[0] in ('value', 'random')}


def updated(delta, roles, node):
    for k, n in roles.items():
        if n is node:
            delta.discard(k)


def bound_terms(terms, delta):
    if not delta:
        return terms
    out = dict(terms)
    for k in delta:
        if k in out:
            out[k] = -np.inf
    out['total'] = -np.inf
    return out


def save(put, base, delta):
    put(base + 'delta_roles', np.array([ord(c) for c in ','.join(sorted(delta))], dtype=np.uint8))


def load(reader, base):
    if not reader.has(base + 'delta_roles'):
        return set()
    txt = ''.join(chr(int(c)) for c in np.asarray(reader.get(base + 'delta_roles')).ravel())
    return set(t for t in txt.split(',') if t)


def warn_state_discarded(plan, node):
    """observe() / initialize_from_*() on a node of a fused block that has already been updated:
    the block rebuilds ALL of its device state from the nodes' initial values, whereas the
    reference changes the touched node only and keeps the other posteriors (stochastic.py:223-273).
    Loud, because results differ from there on; the generic engine keeps per-node state."""
    if getattr(plan, '_ready', False) and getattr(plan, '_version', 0) > 1:
        warnings.warn("%s: %s was observed / initialised after the block had been updated; the fused "
                      "block restarts from the initial state of every node (the reference would "
                      "keep the other posteriors) -- use VB(..., engine='generic') to continue from "
                      "the current state" % (type(plan).__name__, node.name), RuntimeWarning,
                      stacklevel=4)
