"""
Binomial and Bernoulli nodes (reference: bayespy/inference/vmp/nodes/binomial.py:54-246,
bernoulli.py:20-100).

``Binomial(n, p)``: number of successes in ``n`` trials (a non-negative integer or integer
array over the plates); ``p`` is a beta-like node or a probability (array).  The moment is the
expected count, phi = [<log p> - <log(1-p)>] (binomial.py:90-105).  ``Bernoulli(p)`` is the
one-trial case.
"""
import numpy as np

from .node import Stochastic, Constant
from .beta import Beta, Complement
from ..utils.shapes import broadcasted_shape


class Binomial(Stochastic):

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
            if np.any(par.value < 0) or np.any(par.value > 1):
                raise ValueError("Probabilities must be in [0, 1]")
            pplates = par.value.shape
        elif isinstance(par, (Beta, Complement)):
            pplates = par.plates
        else:
            raise ValueError('The probability must be a beta-like node or a number, not %s'
                             % type(par).__name__)
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, pplates, trials.shape)
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
            raise ValueError("Counts must be integer")
        if np.any(x < 0) or np.any(x > self.trials):
            raise ValueError("Invalid count")


class Bernoulli(Binomial):
    _parent_count = 1

    def __init__(self, p, plates=None, name=None, plates_multiplier=None):
        super().__init__(1, p, plates=plates, name=name, plates_multiplier=plates_multiplier)
