"""
Node classes with the reference's construction API
(``bayespy.nodes``, bayespy/nodes/__init__.py:105).
"""
from .node import Node, Constant, Stochastic
from .gamma import Gamma
from .gaussian import GaussianARD
from .dot import SumMultiply, Dot

__all__ = ['Node', 'Constant', 'Stochastic', 'Gamma', 'GaussianARD', 'SumMultiply', 'Dot']
