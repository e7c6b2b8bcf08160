"""
Categorical Markov chain (reference:
bayespy/inference/vmp/nodes/categorical_markov_chain.py:207-438).

``CategoricalMarkovChain(pi, A, states=N)``: a chain of ``N`` instances over K states with
first-state probabilities ``pi`` (a Dirichlet node or a (..., K) array) and transition
probabilities ``A`` (a Dirichlet node with plates (K,), (..., 1, K) or (..., N-1, K), or a
probability array of shape (K, K) / (..., 1|N-1, K, K)).  Moments: u = [q(z_0),
q(z_n, z_{n+1})], dims ((K,), (N-1, K, K)); they come from the forward-backward kernel
(``vmp_alpha_beta_recursion``).  With ``Mixture`` as the emission distribution this is a
hidden Markov model: a chain handed to ``Mixture`` / ``Gate`` is seen through
``as_categorical()``, the time axis becoming the last plate
(``CategoricalMarkovChainToCategorical``, :363-438).
"""
import numpy as np

from .node import Node, Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class CategoricalMarkovChain(Stochastic):
    _parent_count = 2

    def __init__(self, pi, A, states=None, plates=None, name=None, plates_multiplier=None):
        super().__init__(pi, A, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        p0, P = self.parents

        def split(par, what):
            if isinstance(par, Constant):
                if par.value.ndim < 1:
                    raise ValueError('%s must be at least a vector' % what)
                return par.value.shape[:-1], par.value.shape[-1]
            if len(par.dims) != 1 or len(par.dims[0]) != 1:
                raise ValueError('%s must be a Dirichlet-like node' % what)
            return tuple(par.plates), par.dims[0][0]
        p0_plates, D = split(p0, 'Initial state probabilities')
        P_plates, DP = split(P, 'State transition probabilities')
        if len(P_plates) < 2:
            if states is None:
                raise ValueError("Could not infer the length of the Markov chain")
            N = int(states)
        elif P_plates[-2] == 1:
            N = 2 if states is None else int(states)
        else:
            if states is not None and P_plates[-2] + 1 != states:
                raise ValueError("Given length of the Markov chain is inconsistent with the "
                                 "transition probability matrix")
            N = P_plates[-2] + 1
        if D != DP:
            raise ValueError("Initial state probability vector and state transition "
                             "probability matrix have different size")
        if len(P_plates) < 1 or P_plates[-1] != D:
            raise ValueError("Transition probability matrix is not square")
        if N < 2:
            raise ValueError("The chain needs at least two time instances")
        self.categories = int(D)
        self.states = int(N)
        self.dims = ((D,), (N - 1, D, D))
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, p0_plates, tuple(P_plates[:-2]))
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))
        self._categorical = None

    def _check_value_shape(self, x):
        x = np.asarray(x)
        ok = x.ndim >= 1 and x.shape[-1] == self.states
        if ok:
            try:
                ok = broadcasted_shape(x.shape[:-1], self.plates) == self.plates
            except ValueError:
                ok = False
        if not ok:
            raise ValueError('State sequences of shape %s do not match plates %s and %d time '
                             'instances' % (x.shape, self.plates, self.states))
        if np.any(x != np.round(x)):
            raise ValueError("Values must be integers")
        if np.any(x < 0) or np.any(x >= self.categories):
            raise ValueError("Invalid category index")

    def observe(self, x, mask=True):
        # the reference has no f(x) for this node (categorical_markov_chain.py:126-130)
        raise NotImplementedError('CategoricalMarkovChain cannot be observed')

    def as_categorical(self):
        """The chain as categorical variables with the time axis as the last plate."""
        if self._categorical is None:
            self._categorical = CategoricalMarkovChainToCategorical(self)
        return self._categorical


class CategoricalMarkovChainToCategorical(Node):

    def __init__(self, Z, name=None):
        if not isinstance(Z, CategoricalMarkovChain):
            raise ValueError('The parent must be a CategoricalMarkovChain')
        K = Z.categories
        super().__init__(Z, plates=Z.plates + (Z.states,), dims=((K,),), name=name)
        self.categories = K
