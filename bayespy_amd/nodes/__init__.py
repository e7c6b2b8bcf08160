"""
Node classes with the reference's construction API
(``bayespy.nodes``, bayespy/nodes/__init__.py:105).
"""
from .node import Node, Constant, Stochastic
from .gamma import Gamma, Exponential
from .gaussian import (GaussianARD, Gaussian, GaussianGamma, GaussianToGaussianGamma,
                       WrapToGaussianGamma)
from .dot import SumMultiply, Dot
from .wishart import Wishart
from .dirichlet import Dirichlet
from .categorical import Categorical
from .multinomial import Multinomial
from .beta import Beta
from .binomial import Binomial, Bernoulli
from .poisson import Poisson
from .add import Add, ConcatGaussian
from .take import Take, Concatenate, Gate, Choose
from .mixture import Mixture, MultiMixture
from .gaussian_markov_chain import (GaussianMarkovChain, SwitchingGaussianMarkovChain,
                                    VaryingGaussianMarkovChain)
from .categorical_markov_chain import CategoricalMarkovChain

__all__ = ['Node', 'Constant', 'Stochastic', 'Gamma', 'GaussianARD', 'Gaussian', 'GaussianGamma',
           'GaussianToGaussianGamma', 'WrapToGaussianGamma', 'SumMultiply',
           'Dot', 'Wishart', 'Dirichlet', 'Categorical', 'Multinomial', 'Mixture', 'MultiMixture',
           'GaussianMarkovChain', 'Exponential', 'Beta', 'Binomial', 'Bernoulli', 'Poisson', 'Add', 'ConcatGaussian',
           'Take', 'Concatenate', 'Gate', 'Choose', 'CategoricalMarkovChain',
           'SwitchingGaussianMarkovChain', 'VaryingGaussianMarkovChain']
