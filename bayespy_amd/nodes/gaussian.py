"""
GaussianARD node (reference: bayespy/inference/vmp/nodes/gaussian.py:1559-1774).

``GaussianARD(mu, alpha, shape=(K,), plates=...)``: Gaussian with mean ``mu``
and diagonal (ARD) prior precision ``alpha``; the posterior has a full
``shape x shape`` covariance per plate.  Moments u = [<x>, <x x^T>]
(gaussian.py:42-84); the joint (mu, alpha) parent of the reference
(``WrapToGaussianGamma``, gaussian.py:2299-2371) is folded into the plan's
kernels instead of being a separate deterministic node.
"""
from .node import Stochastic
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
        'Gaussian', 'MarkovChainToGaussian', 'Add', 'ConcatGaussian') or getattr(node, '_gaussian_like', False)


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
