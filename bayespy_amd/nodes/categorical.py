"""
Categorical node (reference: bayespy/inference/vmp/nodes/categorical.py:127-201,
multinomial.py:62-231 with one trial).

``Categorical(p, plates=...)``: ``p`` is a Dirichlet node or a probability array.
Moments u = [one-hot probabilities] (categorical.py:25-71); observing /
initialising takes integer class labels.
"""
import numpy as np

from .node import Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class Categorical(Stochastic):
    _parent_count = 1

    def __init__(self, p, plates=None, name=None, plates_multiplier=None):
        super().__init__(p, plates=(), dims=((),), name=name)
        self._plates_multiplier_arg = plates_multiplier
        par = self.parents[0]
        if isinstance(par, Constant):
            if par.value.ndim < 1:
                raise ValueError('Probabilities must be at least a vector')
            K = par.value.shape[-1]
            pplates = par.value.shape[:-1]
        else:
            K = par.dims[0][0]
            pplates = par.plates
        self.categories = K
        self.dims = ((K,),)
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, pplates)
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))

    def _check_value_shape(self, x):
        shape = tuple(np.shape(x))
        if broadcasted_shape(shape, self.plates) != self.plates:
            raise ValueError('Labels of shape %s do not match plates %s' % (shape, self.plates))
