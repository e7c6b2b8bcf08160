"""
Families of the generic engine: the five VMP formulas per node type (``Distribution`` of
stochastic.py:16-80 / expfamily.py:17-70; ``_compute_moments`` / ``_compute_message_to_parent`` of
``Deterministic``, deterministic.py:16-96) restated over device arrays.  ``make_family(node)`` picks
the family of a node: registered ones first (plans/extension.py), the small count / plate nodes
(plans/families_extra.py), then the families of this package.
"""
from ....nodes.node import Constant, Stochastic
from ....nodes.gamma import Gamma
from ....nodes.gaussian import (GaussianARD, Gaussian, GaussianGamma, GaussianToGaussianGamma,
                                WrapToGaussianGamma)
from ....nodes.dot import SumMultiply
from ....nodes.wishart import Wishart
from ....nodes.dirichlet import Dirichlet
from ....nodes.categorical import Categorical
from ....nodes.multinomial import Multinomial
from ....nodes.mixture import Mixture
from ....nodes.gaussian_markov_chain import GaussianMarkovChain, MarkovChainToGaussian
from .base import Family
from .scalar import (GammaFamily, WishartFamily, DirichletFamily, CategoricalFamily,
                     MultinomialFamily)
from .gaussian import (GaussianARDFamily, GaussianFamily, GaussianGammaFamily,
                       GaussianToGaussianGammaFamily, WrapToGaussianGammaFamily)
from .mixture import MixtureFamily
from .chain import GaussianMarkovChainFamily, ChainToGaussianFamily
from .dot import SumMultiplyFamily


def make_family(node):
    from ..extension import registered_family
    from ..families_extra import make_extra_family
    # node types registered from outside the package (plans/extension.py: the reference's
    # Distribution contract, writingnodes.rst) come first: a registration may also replace a
    # built-in family
    fam = registered_family(node)
    if fam is not None:
        return fam
    fam = make_extra_family(node)
    if fam is not None:
        return fam
    if isinstance(node, Mixture):
        return MixtureFamily(node, make_family(node._proto))
    if isinstance(node, Gamma):
        return GammaFamily(node)
    if isinstance(node, GaussianGamma):
        return GaussianGammaFamily(node)
    if isinstance(node, GaussianToGaussianGamma):
        return GaussianToGaussianGammaFamily(node)
    if isinstance(node, WrapToGaussianGamma):
        return WrapToGaussianGammaFamily(node)
    if isinstance(node, GaussianARD):
        return GaussianARDFamily(node)
    if isinstance(node, Gaussian):
        return GaussianFamily(node)
    if isinstance(node, Wishart):
        return WishartFamily(node)
    if isinstance(node, Dirichlet):
        return DirichletFamily(node)
    if isinstance(node, Multinomial):
        return MultinomialFamily(node)
    if isinstance(node, Categorical):
        return CategoricalFamily(node)
    if isinstance(node, SumMultiply):
        return SumMultiplyFamily(node)
    if isinstance(node, GaussianMarkovChain):
        return GaussianMarkovChainFamily(node)
    if isinstance(node, MarkovChainToGaussian):
        return ChainToGaussianFamily(node)
    raise NotImplementedError('no device family for node type %s (a node type defined outside the '
                              'package registers its formulas with '
                              'bayespy_amd.inference.register_family, plans/extension.py)'
                              % type(node).__name__)
