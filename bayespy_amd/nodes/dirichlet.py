"""
Dirichlet node (reference: bayespy/inference/vmp/nodes/dirichlet.py:333-399).

``Dirichlet(alpha, plates=...)`` with concentration vector ``alpha`` (last axis =
categories).  Moments u = [<log p>] (dirichlet.py:25-60); phi = [alpha] (:113-127).
"""
from .node import Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class Dirichlet(Stochastic):
    _parent_count = 1

    def __init__(self, alpha, plates=None, name=None, plates_multiplier=None):
        super().__init__(alpha, plates=(), dims=((),), name=name)
        self._plates_multiplier_arg = plates_multiplier
        a = self.parents[0]
        if not isinstance(a, Constant):
            raise NotImplementedError('Dirichlet concentration must be a numeric constant')
        if a.value.ndim < 1:
            raise ValueError('Concentration must be at least a vector')
        K = a.value.shape[-1]
        self.dims = ((K,),)
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, a.value.shape[:-1])
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))
