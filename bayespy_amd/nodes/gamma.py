"""
Gamma node (reference: bayespy/inference/vmp/nodes/gamma.py:214-335).

Moments u = [<x>, <log x>], natural parameters phi = [-b, a]
(gamma.py:116-148).  The arithmetic runs in the HIP kernels of the plan that
owns the node (e.g. ``pca_update_tau_kernel`` in csrc/vmp_pca.hip).
"""
from .node import Stochastic
from ..utils.shapes import broadcasted_shape


class Gamma(Stochastic):
    """``Gamma(a, b, plates=(), name=...)`` -- shape a, rate b."""
    _parent_count = 2

    def __init__(self, a, b, plates=None, name=None, plates_multiplier=None):
        super().__init__(a, b, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        pa, pb = self.parents[0].plates, self.parents[1].plates
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, pa, pb)
        if plates is not None and self.plates != given:
            raise ValueError('Plates %s of the parents do not broadcast to plates %s'
                             % ((pa, pb), given))


class Exponential(Gamma):
    """The reference declines this node and points to ``Gamma(1, l)`` (exponential.py:60-67)."""

    def __init__(self, l, **kwargs):
        raise NotImplementedError("Not yet implemented. Use Gamma(1, lambda)")
