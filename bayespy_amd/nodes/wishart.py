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
        if not isinstance(n_node, Constant):
            raise NotImplementedError('the degrees of freedom of a Wishart node must be numeric '
                                      '(the reference needs WishartPriorMoments there: wishart.py:96-115)')
        if isinstance(V_node, Constant):
            Vs = V_node.value.shape
            if len(Vs) < 2 or Vs[-1] != Vs[-2]:
                raise ValueError('V must be a (..., D, D) array')
            D, v_plates = Vs[-1], Vs[:-2]
        elif isinstance(V_node, Wishart):
            # a hierarchical prior: the inverse scale matrix is itself Wishart; the message to it
            # is [-<Lambda>/2, n/2] (wishart.py:142-150).  Runs on the generic engine.
            D, v_plates = V_node.dims[0][0], V_node.plates
        else:
            raise NotImplementedError('the inverse scale matrix of a Wishart node must be numeric '
                                      'or a Wishart node (WishartMoments, wishart.py:45-60)')
        self.dims = ((D, D), ())
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, n_node.value.shape, v_plates)
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))
