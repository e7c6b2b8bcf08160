"""
Mixture node (reference: bayespy/inference/vmp/nodes/mixture.py:359-545).

``Mixture(z, Dist, *params, cluster_plate=-1, plates=...)``: the variable follows
``Dist(*params[k])`` where ``k`` is the class of the categorical node ``z`` and the
parameters carry the cluster axis among their plates.
"""
import numpy as np

from .node import Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class Mixture(Stochastic):

    def __init__(self, z, node_class, *params, cluster_plate=-1, plates=None, name=None,
                 plates_multiplier=None, **node_kwargs):
        if not isinstance(cluster_plate, (int, np.integer)) or cluster_plate >= 0:
            raise ValueError("Cluster plate axis must be negative")
        if cluster_plate != -1 and node_class is Mixture:
            raise NotImplementedError('nested mixtures are built for cluster_plate=-1')
        # a categorical Markov chain is seen through its categorical view: the time axis
        # becomes the last plate (moment converter of categorical_markov_chain.py:435-438)
        if hasattr(z, 'as_categorical'):
            z = z.as_categorical()
        if node_class is Mixture:
            # nested mixture, Mixture(z1, Mixture, z2, Dist, *params) (mixture.py:398-420 builds
            # the mixed distribution recursively): the inner mixture is the mixed "class", its
            # parents (inner selector first) follow the outer selector
            proto = Mixture(*params, **node_kwargs)
            inner = list(proto.parents)
            super().__init__(z, *inner, plates=(), dims=((), ()), name=name)
        else:
            # positional arguments beyond the parents of the mixed class are constructor
            # arguments of that class (the reference forwards them to ``_constructor``, e.g. the
            # ``ndim`` of ``Mixture(z, GaussianARD, mu, alpha, 1)``)
            npar = getattr(node_class, '_parent_count', None)
            extra = ()
            if npar is not None and len(params) > npar:
                params, extra = params[:npar], params[npar:]
            super().__init__(z, *params, plates=(), dims=((), ()), name=name)
            # a throw-away instance of the mixed node class gives dims and plates
            # (with the cluster axis still among the plates)
            proto = node_class(*self.parents[1:], *extra, **node_kwargs)
        self._plates_multiplier_arg = plates_multiplier
        self.node_class = node_class
        self.cluster_plate = cluster_plate
        for p in self.parents[1:]:
            p.children = [(c, i) for (c, i) in p.children if c is not proto]
        self._proto = proto
        self.dims = proto.dims
        if hasattr(proto, 'shape'):
            self.shape = proto.shape
            self.ndim = proto.ndim
        pp = proto.plates
        zn = self.parents[0]
        if isinstance(zn, Constant):
            # fixed class labels: the number of classes is the cluster plate of the parameters
            # (CategoricalMoments.compute_fixed_moments, categorical.py:30-46)
            if len(pp) < -cluster_plate:
                raise ValueError('The parameters have no cluster plate')
            K = pp[cluster_plate]
            if np.any(zn.value != np.round(zn.value)):
                raise ValueError("Values must be integers")
            if np.any(zn.value < 0) or np.any(zn.value >= K):
                raise ValueError("Invalid category index")
            zplates = zn.value.shape
        else:
            K = zn.dims[0][0]
            zplates = zn.plates
        if len(pp) < -cluster_plate:
            raise ValueError("The mixed distribution does not have a plates axis for the "
                             "cluster plate axis")
        if pp[cluster_plate] not in (1, K):
            raise ValueError('The cluster plate (%s) of the parameters does not match the '
                             'number of categories %d' % (pp[cluster_plate], K))
        self.clusters = K
        given = tuple(plates) if plates is not None else ()
        rest = list(pp)
        rest.pop(cluster_plate)
        self.plates = broadcasted_shape(given, zplates, tuple(rest))
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))

    def _check_value_shape(self, x):
        # values have the form of the mixed class (labels for Categorical, counts for
        # Poisson, ...) on this node's plates
        saved = self._proto.plates
        self._proto.plates = self.plates
        try:
            self._proto._check_value_shape(x)
        finally:
            self._proto.plates = saved


def MultiMixture(thetas, *mixture_args, **kwargs):
    """A mixture over several cluster axes with as many categorical selectors: selector i gets i
    trailing unit plate axes and the mixtures are nested,
    ``Mixture(t0, Mixture, t1, ..., Mixture, t_last, *mixture_args)``  (mixture.py:547-566)."""
    from .node import Node
    thetas = [t if isinstance(t, Node) else np.asanyarray(t) for t in thetas]
    thetas = [t[(Ellipsis,) + i * (None,)] for i, t in enumerate(thetas)]
    args = [thetas[0]]
    for t in thetas[1:]:
        args += [Mixture, t]
    return Mixture(*(args + list(mixture_args)), **kwargs)
