"""
Multinomial node (reference: bayespy/inference/vmp/nodes/multinomial.py:62-319).

``Multinomial(n, p, plates=...)``: ``n`` trials (a non-negative integer or an integer array
over the plates), ``p`` a Dirichlet node or a probability array.  Moments u = [expected
counts] = n * softmax(phi); observing takes integer count vectors that sum to ``n``.
"""
import numpy as np

from .node import Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class Multinomial(Stochastic):

    def __init__(self, n, p, plates=None, name=None, plates_multiplier=None):
        trials = np.asarray(n)
        if not np.issubdtype(trials.dtype, np.integer):
            if np.any(trials != np.round(trials)):
                raise ValueError("Number of trials must be integer")
            trials = trials.astype(np.int64)
        if np.any(trials < 0):
            raise ValueError("Number of trials must be non-negative")
        super().__init__(p, plates=(), dims=((),), name=name)
        self._plates_multiplier_arg = plates_multiplier
        self.trials = trials
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
        self.plates = broadcasted_shape(given, pplates, trials.shape)
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))

    def _check_value_shape(self, x):
        x = np.asarray(x)
        shape = tuple(x.shape)
        if len(shape) < 1 or shape[-1] != self.categories or \
                broadcasted_shape(shape[:-1], self.plates) != self.plates:
            raise ValueError('Counts of shape %s do not match plates %s and %d categories'
                             % (shape, self.plates, self.categories))
        if np.any(x != np.round(x)):
            raise ValueError("Counts must be integers")
        if np.any(x < 0):
            raise ValueError("Counts must be non-negative")
        if np.any(np.sum(x, axis=-1) != self.trials):
            raise ValueError("Counts must sum to the number of trials")
