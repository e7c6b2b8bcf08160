"""
Poisson node (reference: bayespy/inference/vmp/nodes/poisson.py:25-177).

``Poisson(l)`` with rate ``l`` a gamma-like node or a positive number / array.  The moment is
the expected count, phi = [<log l>] (poisson.py:66-84); observing takes non-negative integers.
"""
import numpy as np

from .node import Stochastic, Constant
from .gamma import Gamma
from ..utils.shapes import broadcasted_shape


class Poisson(Stochastic):
    _parent_count = 1

    def __init__(self, l, plates=None, name=None, plates_multiplier=None):
        super().__init__(l, plates=(), dims=((),), name=name)
        self._plates_multiplier_arg = plates_multiplier
        par = self.parents[0]
        if isinstance(par, Constant):
            pplates = par.value.shape
        elif isinstance(par, Gamma):
            pplates = par.plates
        else:
            raise ValueError('The rate must be a gamma-like node or a number, not %s'
                             % type(par).__name__)
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, pplates)
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))

    def _check_value_shape(self, x):
        x = np.asarray(x)
        try:
            ok = broadcasted_shape(x.shape, self.plates) == self.plates
        except ValueError:
            ok = False
        if not ok:
            raise ValueError('Counts of shape %s do not match plates %s' % (x.shape, self.plates))
        if np.any(x != np.round(x)):
            raise ValueError("Values must be integers")
        if np.any(x < 0):
            raise ValueError("Values must be positive")
