"""
Plate / shape algebra (host-side metadata only; no array arithmetic).

Mirrors the semantics of the reference helpers
``bayespy/utils/misc.py``: ``broadcasted_shape`` (:995), ``is_shape_subset``
(:1028), ``add_trailing_axes`` (:1052), ``axes_to_collapse`` (:1088) and
``broadcasting_multiplier`` (:761-802).  Plates are the reference's leading,
right-aligned broadcastable axes: a unit or missing axis means "same for all".
"""
from functools import reduce


def broadcasted_shape(*shapes):
    """Right-aligned NumPy broadcasting of shapes; ValueError on mismatch
    (reference: utils/misc.py:995-1025)."""
    n = max((len(s) for s in shapes), default=0)
    out = [1] * n
    for s in shapes:
        s = tuple(int(x) for x in s)
        for i, d in enumerate(reversed(s)):
            j = n - 1 - i
            if out[j] == 1:
                out[j] = d
            elif d != 1 and d != out[j]:
                raise ValueError("Shapes %s do not broadcast" % (shapes,))
    return tuple(out)


def is_shape_subset(sub, full):
    """True if an array of shape ``sub`` broadcasts to ``full`` without
    enlarging it (utils/misc.py:1028-1049)."""
    if len(sub) > len(full):
        return False
    for a, b in zip(reversed(tuple(sub)), reversed(tuple(full))):
        if a != 1 and a != b:
            return False
    return True


def prod(shape):
    return reduce(lambda a, b: a * int(b), shape, 1)


def multiplier_shape(own, *parents):
    """Plate multiplier of a node: its own tuple (may hold non-integers, e.g. N / N_batch)
    broadcast against those of its parents (node.py:294-301 with _total_plates :336-360)."""
    out = []
    seqs = [tuple(q) for q in ((own,) if own is not None else ()) + tuple(parents)]
    n = max([len(q) for q in seqs] or [0])
    for j in range(1, n + 1):
        vals = {q[-j] for q in seqs if len(q) >= j and q[-j] != 1}
        if len(vals) > 1:
            raise ValueError('The plate multipliers do not broadcast: %s' % (seqs,))
        out.append(vals.pop() if vals else 1)
    return tuple(reversed(out))


def multiplier_factor(mult, *args):
    """Product of the entries of ``mult`` along axes where every multiplier in ``args`` is
    unit or missing (broadcasting_multiplier of node.py:604,625 on multiplier tuples)."""
    r = 1.0
    for j in range(1, len(mult) + 1):
        if all(len(a) < j or a[-j] == 1 for a in args):
            r *= float(mult[-j])
    return r


def broadcasting_multiplier(plates, *args):
    """
    Integer factor by which a sum over broadcast-compressed arrays of shapes
    ``args`` under-counts the sum over the full ``plates``
    (utils/misc.py:761-802): product of plate sizes over axes where EVERY
    argument has a unit/missing axis.
    """
    plates = tuple(plates)
    r = 1
    for j in range(1, len(plates) + 1):
        if all(len(a) < j or a[-j] == 1 for a in args):
            r *= int(plates[-j])
    return r
