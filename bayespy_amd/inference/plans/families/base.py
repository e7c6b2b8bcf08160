"""The family interface of the generic engine: the five VMP formulas per node type
(stochastic.py:16-80, expfamily.py:17-70)."""
import os

import numpy as np

from .... import darray as da
from ....darray import DArray, fuse, contiguous
from ....nodes.node import Constant, Stochastic
from ....nodes.gaussian import is_gaussian_gamma
from ....utils import misc, linalg
from ....utils.shapes import broadcasted_shape, is_shape_subset, multiplier_factor
from ..lazy import (DerivedArray,
                    FactoredMoment,
                    LOG2PI,
                    LazyContract,
                    LazySum,
                    PlateSums,
                    Terms,
                    _CONSTS,
                    _Deferred,
                    _LazyList,
                    _arr,
                    _check_device,
                    _const,
                    _diag2,
                    _eye,
                    _factored_min_plates,
                    _gaussian_gradient,
                    _gaussian_q_term,
                    _inner_second,
                    _is_lazy,
                    _lazy_mvdot,
                    _multigammaln,
                    _ones,
                    _shape,
                    _sum_last,
                    _trail,
                    _wsum)


class Family:

    def __init__(self, node):
        self.node = node

    def plates_to_parent(self, index):
        return self.node.plates

    def mask_to_parent(self, index, mask):
        return mask

    def constant_moments(self, index, value):
        raise NotImplementedError

    def gradient(self, rg, u, phi):
        """Euclidean gradient from the Riemannian one (expfamily.py:64-70)."""
        raise NotImplementedError("Standard gradient not yet implemented for %s"
                                  % type(self.node).__name__)
