"""
Beta node and its complement (reference: bayespy/inference/vmp/nodes/beta.py:25-214).

``Beta(alpha)`` with ``alpha = [..., (a, b)]`` is a two-class Dirichlet whose realisations are
scalars p: the moments stay the two-vector [<log p>, <log(1-p)>] (beta.py:25-43), observing /
initialising takes probabilities.  ``p.complement()`` is the deterministic node 1 - p, i.e. the
same moments with the two entries swapped (beta.py:194-214).
"""
import numpy as np

from .node import Node
from .dirichlet import Dirichlet
from ..utils.shapes import broadcasted_shape


class Beta(Dirichlet):

    def __init__(self, alpha, plates=None, name=None, plates_multiplier=None):
        super().__init__(alpha, plates=plates, name=name, plates_multiplier=plates_multiplier)
        if self.dims != ((2,),):
            raise ValueError("Parent has wrong dimensionality. Must be a two-dimensional vector.")

    def _check_value_shape(self, x):
        shape = tuple(np.shape(x))
        try:
            ok = broadcasted_shape(shape, self.plates) == self.plates
        except ValueError:
            ok = False
        if not ok:
            raise ValueError('Probabilities of shape %s do not match plates %s'
                             % (shape, self.plates))

    def complement(self):
        return Complement(self)


class Complement(Node):
    """1 - p of a beta-like node."""

    def __init__(self, p, name=None):
        if not isinstance(p, (Beta, Complement)):
            raise ValueError('Complement needs a beta-like parent')
        super().__init__(p, plates=p.plates, dims=p.dims, name=name)

    def complement(self):
        return Complement(self)
