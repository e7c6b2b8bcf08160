"""
GaussianARD node (reference: bayespy/inference/vmp/nodes/gaussian.py:1559-1774).

``GaussianARD(mu, alpha, shape=(K,), plates=...)``: Gaussian with mean ``mu``
and diagonal (ARD) prior precision ``alpha``; the posterior has a full
``shape x shape`` covariance per plate.  Moments u = [<x>, <x x^T>]
(gaussian.py:42-84); the joint (mu, alpha) parent of the reference
(``WrapToGaussianGamma``, gaussian.py:2299-2371) is folded into the plan's
kernels instead of being a separate deterministic node.
"""
from .node import Node, Stochastic
from ..utils.shapes import broadcasted_shape


class GaussianARD(Stochastic):
    _parent_count = 2

    def __init__(self, mu, alpha, ndim=None, shape=None, plates=None, name=None,
                 plates_multiplier=None):
        super().__init__(mu, alpha, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        mu_node, alpha_node = self.parents
        mu_shape = mu_node.plates + (mu_node.dims[0] if _is_gaussian(mu_node) else ())
        if shape is not None and ndim is not None and ndim != len(shape):
            raise ValueError("Given shape and ndim inconsistent")
        if shape is None:
            if ndim is None:
                # like the reference: scalar-valued unless told otherwise, whatever the mean
                # is; variable axes of a Gaussian mean become plates (gaussian.py:1617-1622)
                shape = ()
            else:
                full = broadcasted_shape(mu_shape, alpha_node.plates)
                if ndim > len(full):
                    raise ValueError("Cannot determine shape for ndim={0} because parent full "
                                     "shape has ndim={1}.".format(ndim, len(full)))
                shape = full[len(full) - ndim:] if ndim > 0 else ()
        shape = tuple(int(s) for s in shape)
        nd = len(shape)
        self.shape = shape
        self.ndim = nd
        self.dims = (shape, shape + shape)
        # plates the parents impose: their shapes minus the trailing `ndim` axes
        # (gaussian.py:756-769)
        def strip(s):
            return tuple(s[:len(s) - nd]) if nd > 0 else tuple(s)
        if _is_gaussian(mu_node):
            mu_plates = mu_node.plates if len(mu_node.dims[0]) == nd else strip(mu_shape)
        else:
            mu_plates = strip(mu_shape)
        alpha_plates = strip(alpha_node.plates)
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, mu_plates, alpha_plates)
        if plates is not None and self.plates != given:
            raise ValueError('Plates %s of the parents do not broadcast to plates %s'
                             % ((mu_plates, alpha_plates), given))


def _is_gaussian(node):
    from .dot import SumMultiply
    return isinstance(node, (GaussianARD, SumMultiply)) or type(node).__name__ in (
        'Gaussian', 'MarkovChainToGaussian', 'Add', 'ConcatGaussian') or getattr(node, '_gaussian_like', False) \
        or is_gaussian_gamma(node)


def is_gaussian_gamma(node):
    """The node's moments are Gaussian-gamma moments [<tau x>, <tau x x^T>, <tau>, <log tau>]
    (GaussianGammaMoments, gaussian.py:161-229)."""
    return getattr(node, '_gaussian_gamma', False)


class Gaussian(Stochastic):
    """``Gaussian(mu, Lambda)``: mean ``mu`` (Gaussian moments or array) and precision
    matrix ``Lambda`` (Wishart node or SPD array) -- reference gaussian.py:1346-1556,
    formulas gaussian.py:293-573; the joint (mu, Lambda) wrapper of the reference
    (``WrapToGaussianWishart``, gaussian.py:2374-2527) is folded into the formulas."""
    _parent_count = 2

    def __init__(self, mu, Lambda, plates=None, name=None, plates_multiplier=None):
        super().__init__(mu, Lambda, plates=(), dims=((), ()), name=name)
        self._plates_multiplier_arg = plates_multiplier
        from .node import Constant
        mu_node, L_node = self.parents
        if isinstance(L_node, Constant):
            Ls = L_node.value.shape
            if len(Ls) < 2 or Ls[-1] != Ls[-2]:
                raise ValueError('Lambda must be a (..., D, D) array')
            D, Lplates = Ls[-1], Ls[:-2]
        else:
            D, Lplates = L_node.dims[0][0], L_node.plates
        if isinstance(mu_node, Constant):
            ms = mu_node.value.shape
            mu_plates = ms[:-1] if len(ms) >= 1 else ()
            if len(ms) >= 1 and ms[-1] not in (1, D):
                raise ValueError('mu has %d components, Lambda is %dx%d' % (ms[-1], D, D))
        else:
            if len(mu_node.dims[0]) != 1 or mu_node.dims[0][0] != D:
                raise ValueError('mu must be a vector variable of length %d' % D)
            mu_plates = mu_node.plates
        self.shape = (D,)
        self.ndim = 1
        self.dims = ((D,), (D, D))
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, mu_plates, Lplates)
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))


class GaussianGamma(Stochastic):
    """``GaussianGamma(mu, Lambda, a, b, ndim=1)``: the JOINT node of a Gaussian variable and its
    precision scale,  p(x | tau) = N(x | mu, (tau Lambda)^-1),  p(tau) = Gamma(a, b)  -- reference
    gaussian.py:1777-1840 (node), :892-1136 (GaussianGammaDistribution), moments
    u = [<tau x>, <tau x x^T>, <tau>, <log tau>] (GaussianGammaMoments, :161-229).  q(x, tau) keeps the
    Gaussian-gamma form: the dependence between the mean and the precision that the factorised
    GaussianARD + Gamma pair gives up.  ``mu``: array or Gaussian node, ``Lambda``: SPD array /
    Wishart node (``ndim=1``) or positive array / Gamma node (``ndim=0``), ``a``: array, ``b``:
    array or Gamma node.  A scalar-valued node (``ndim=0``) is a valid mean parent of
    ``GaussianARD`` (gaussian.py:1621, :1646)."""
    _parent_count = 4
    _gaussian_gamma = True

    def __init__(self, mu, Lambda, a, b, ndim=1, plates=None, name=None):
        super().__init__(mu, Lambda, a, b, plates=(), dims=((), (), (), ()), name=name)
        from .node import Constant
        if ndim not in (0, 1):
            raise NotImplementedError('GaussianGamma with ndim = %d' % ndim)
        mu_n, L_n, a_n, b_n = self.parents
        if is_gaussian_gamma(mu_n):
            raise NotImplementedError('a Gaussian-gamma mean of a GaussianGamma node')
        if not isinstance(a_n, Constant):
            raise NotImplementedError('the shape parameter a of GaussianGamma must be an array')
        if ndim == 0:
            shape = ()
            mu_plates = mu_n.value.shape if isinstance(mu_n, Constant) else mu_n.plates
            if not isinstance(mu_n, Constant) and tuple(mu_n.dims[0]) != ():
                raise ValueError('mu and Lambda have wrong shape')
            L_plates = L_n.value.shape if isinstance(L_n, Constant) else L_n.plates
            if not isinstance(L_n, Constant) and tuple(L_n.dims[0]) != ():
                raise ValueError('mu and Lambda have wrong shape')
        else:
            if isinstance(L_n, Constant):
                Ls = L_n.value.shape
                if len(Ls) < 2 or Ls[-1] != Ls[-2]:
                    raise ValueError('mu and Lambda have wrong shape')
                D, L_plates = Ls[-1], Ls[:-2]
            else:
                if len(L_n.dims[0]) != 2:
                    raise ValueError('mu and Lambda have wrong shape')
                D, L_plates = L_n.dims[0][0], L_n.plates
            if isinstance(mu_n, Constant):
                ms = mu_n.value.shape
                if len(ms) < 1 or ms[-1] != D:
                    raise ValueError('mu and Lambda have wrong shape')
                mu_plates = ms[:-1]
            else:
                if tuple(mu_n.dims[0]) != (D,):
                    raise ValueError('mu and Lambda have wrong shape')
                mu_plates = mu_n.plates
            shape = (D,)
        self.shape = tuple(shape)
        self.ndim = ndim
        self.dims = (self.shape, self.shape + self.shape, (), ())
        given = tuple(plates) if plates is not None else ()
        try:
            self.plates = broadcasted_shape(given, tuple(mu_plates), tuple(L_plates),
                                            a_n.plates, b_n.plates)
        except ValueError:
            raise ValueError('The plates of the parents do not broadcast')
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))


class GaussianToGaussianGamma(Node):
    """Gaussian moments seen as Gaussian-gamma moments with the precision scale fixed to 1:
    u = [<x>, <x x^T>, 1, 0]; the message back keeps the Gaussian part (reference
    gaussian.py:2226-2276).  The reference inserts this converter itself wherever a Gaussian node
    is given for a Gaussian-gamma parent; here that case is folded into the families, and the
    explicit node does the same thing one step at a time."""
    _gaussian_gamma = True

    def __init__(self, X, name=None):
        from .node import GaussianConstant
        if not (_is_gaussian(X) and not is_gaussian_gamma(X)) and not isinstance(X, GaussianConstant):
            raise ValueError("Wrong moments, should be Gaussian")
        shape = tuple(X.dims[0])
        super().__init__(X, plates=X.plates, dims=(shape, shape + shape, (), ()),
                         name=name or (X.name + '_as_gaussian_gamma'))
        self.shape, self.ndim = shape, len(shape)


class WrapToGaussianGamma(Node):
    """The joint parent (X, alpha) of a Gaussian child whose precision is ``alpha`` times the
    scale of X: u = [<tau x> <alpha>, <tau x x^T> <alpha>, <tau><alpha>, <log tau> + <log alpha>]
    (reference gaussian.py:2299-2371).  ``X``: Gaussian-gamma moments (a Gaussian node is
    converted first), ``alpha``: Gamma node or positive array.  The stochastic nodes fold this
    wrapper into their formulas; the explicit node exists for graphs that name it."""
    _gaussian_gamma = True

    def __init__(self, X, alpha, ndim=None, name=None):
        from .node import ensure_node
        X = ensure_node(X)
        if not is_gaussian_gamma(X):
            X = GaussianToGaussianGamma(X)
        if ndim is not None and ndim != len(X.dims[0]):
            raise NotImplementedError("Conversion to different ndim in GaussianMoments not yet "
                                      "implemented.")
        shape = tuple(X.dims[0])
        super().__init__(X, alpha, plates=(), dims=(shape, shape + shape, (), ()), name=name)
        self.plates = broadcasted_shape(self.parents[0].plates, self.parents[1].plates)
        self.shape, self.ndim = shape, len(shape)
