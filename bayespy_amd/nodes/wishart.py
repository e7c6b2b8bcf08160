"""
Wishart node (reference: bayespy/inference/vmp/nodes/wishart.py:228-305).

``Wishart(n, V, plates=...)``: degrees of freedom ``n`` and INVERSE scale matrix
``V`` (wishart.py:126-128).  Moments u = [<Lambda>, <log|Lambda|>]
(wishart.py:45-60), natural parameters phi = [-V/2, n/2] (:153-163).
"""

from .node import Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class Wishart(Stochastic):
    _parent_count = 2

    def __init__(self, n, V, plates=None, name=None, plates_multiplier=None):
        super().__init__(n, V, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        n_node, V_node = self.parents
        if not isinstance(n_node, Constant) or not isinstance(V_node, Constant):
            raise NotImplementedError('Wishart parents must be numeric constants')
        Vs = V_node.value.shape
        if len(Vs) < 2 or Vs[-1] != Vs[-2]:
            raise ValueError('V must be a (..., D, D) array')
        D = Vs[-1]
        self.dims = ((D, D), ())
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, n_node.value.shape, Vs[:-2])
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))
