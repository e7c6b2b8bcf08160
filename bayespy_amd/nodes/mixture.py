"""
Mixture node (reference: bayespy/inference/vmp/nodes/mixture.py:359-545).

``Mixture(z, Dist, *params, cluster_plate=-1, plates=...)``: the variable follows
``Dist(*params[k])`` where ``k`` is the class of the categorical node ``z`` and the
parameters carry the cluster axis among their plates.
"""
from .node import Stochastic
from ..utils.shapes import broadcasted_shape


class Mixture(Stochastic):

    def __init__(self, z, node_class, *params, cluster_plate=-1, plates=None, name=None,
                 plates_multiplier=None, **node_kwargs):
        if cluster_plate != -1:
            raise NotImplementedError('only cluster_plate=-1 is built')
        # positional arguments beyond the parents of the mixed class are constructor
        # arguments of that class (the reference forwards them to ``_constructor``, e.g. the
        # ``ndim`` of ``Mixture(z, GaussianARD, mu, alpha, 1)``; mixture.py:398-420)
        # a categorical Markov chain is seen through its categorical view: the time axis
        # becomes the last plate (moment converter of categorical_markov_chain.py:435-438)
        if hasattr(z, 'as_categorical'):
            z = z.as_categorical()
        npar = getattr(node_class, '_parent_count', None)
        extra = ()
        if npar is not None and len(params) > npar:
            params, extra = params[:npar], params[npar:]
        super().__init__(z, *params, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        self.node_class = node_class
        self.cluster_plate = cluster_plate
        # a throw-away instance of the mixed node class gives dims and plates
        # (with the cluster axis still among the plates)
        proto = node_class(*self.parents[1:], *extra, **node_kwargs)
        for p in self.parents[1:]:
            p.children = [(c, i) for (c, i) in p.children if c is not proto]
        self._proto = proto
        self.dims = proto.dims
        if hasattr(proto, 'shape'):
            self.shape = proto.shape
            self.ndim = proto.ndim
        K = self.parents[0].dims[0][0]
        pp = proto.plates
        if len(pp) < 1 or pp[-1] not in (1, K):
            raise ValueError('The cluster plate (%s) of the parameters does not match the '
                             'number of categories %d' % (pp[-1:] or None, K))
        self.clusters = K
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, self.parents[0].plates, pp[:-1])
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
