"""
Gaussian Markov chain node (reference:
bayespy/inference/vmp/nodes/gaussian_markov_chain.py:709-1190, formulas :270-707).

``GaussianMarkovChain(mu, Lambda, A, nu, n=N, plates=...)``: x_0 ~ N(mu, Lambda^-1),
x_n ~ N(A x_{n-1}, diag(nu)^-1).  Moments u = [<x_n> (N,D), <x_n x_n^T> (N,D,D),
<x_{n-1} x_n^T> (N-1,D,D)].  The dynamics matrix ``A`` is a Gaussian variable with
shape (D,) whose LAST plate (D) indexes the rows of A; ``nu`` has last plate D.  Only
time-constant dynamics are built (A, nu without the N-1 plate).  A chain used as a
Gaussian parent (e.g. of SumMultiply) is seen through :class:`MarkovChainToGaussian`,
which turns the time axis into the last plate (reference :1988-2098).
"""
import numpy as np

from .node import Node, Stochastic, Constant
from ..utils.shapes import broadcasted_shape


class GaussianMarkovChain(Stochastic):

    def __init__(self, mu, Lambda, A, nu, n=None, inputs=None, plates=None, name=None):
        if inputs is not None:
            raise NotImplementedError('input signals of GaussianMarkovChain are not built')
        super().__init__(mu, Lambda, A, nu, plates=(), dims=((), (), ()), name=name)
        mu_n, L_n, A_n, nu_n = self.parents
        if isinstance(L_n, Constant):
            if L_n.value.ndim < 2 or L_n.value.shape[-1] != L_n.value.shape[-2]:
                raise Exception("Initial state parameters have wrong dimensionality")
            D, Lpl = L_n.value.shape[-1], L_n.value.shape[:-2]
        else:
            D, Lpl = L_n.dims[0][0], L_n.plates
        if isinstance(mu_n, Constant):
            if mu_n.value.ndim < 1 or mu_n.value.shape[-1] != D:
                raise Exception("Initial state parameters have wrong dimensionality")
            mupl = mu_n.value.shape[:-1]
        else:
            if mu_n.dims[0] != (D,):
                raise Exception("Initial state parameters have wrong dimensionality")
            mupl = mu_n.plates
        if isinstance(A_n, Constant):
            if A_n.value.shape[-2:] != (D, D):
                raise Exception("Dynamics matrix has wrong dimensionality")
            Apl = A_n.value.shape[:-1]
        else:
            if A_n.dims[0] != (D,):
                raise Exception("Dynamics matrix has wrong dimensionality")
            Apl = A_n.plates
        nupl = nu_n.value.shape if isinstance(nu_n, Constant) else nu_n.plates
        for pl in (Apl, nupl):
            if len(pl) == 0 or pl[-1] != D:
                raise Exception("Dynamics matrix should have a last plate equal to the "
                                "dimensionality of the system.")
            if len(pl) >= 2 and pl[-2] != 1:
                raise NotImplementedError('time-varying dynamics (an N-1 plate on A / nu) '
                                          'are not built')
        if n is None:
            raise Exception("The number of time instances could not be determined "
                            "automatically. Give the number of time instances.")
        self.N, self.D = int(n), int(D)
        self.dims = ((self.N, D), (self.N, D, D), (self.N - 1, D, D))
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, mupl, Lpl, Apl[:-2], nupl[:-2])
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))

    def _check_value_shape(self, x):
        shape = tuple(x.shape) if hasattr(x, 'shape') else np.shape(x)
        full = self.plates + self.dims[0]
        try:
            ok = broadcasted_shape(shape, full) == full
        except ValueError:
            ok = False
        if not ok:
            raise ValueError('Value of shape %s does not match node %s with plates+dims %s'
                             % (shape, self.name, full))

    def as_gaussian(self):
        if not hasattr(self, '_as_gaussian'):
            self._as_gaussian = MarkovChainToGaussian(self)
        return self._as_gaussian


class MarkovChainToGaussian(Node):
    """Deterministic view of a chain as Gaussian moments with the time axis as the last
    plate (reference ``_MarkovChainToGaussian``, gaussian_markov_chain.py:1988-2098)."""

    def __init__(self, X, name=None):
        super().__init__(X, plates=X.plates + (X.N,), dims=((X.D,), (X.D, X.D)),
                         name=name or (X.name + '_as_gaussian'))
        self.shape = (X.D,)
        self.ndim = 1


def _initial_state(mu_n, L_n):
    """Dimensionality and plates of the initial-state parents (mean, precision matrix) of a
    chain (gaussian_markov_chain.py:1897-1907)."""
    if isinstance(L_n, Constant):
        if L_n.value.ndim < 2 or L_n.value.shape[-1] != L_n.value.shape[-2]:
            raise ValueError("Second parent has wrong dimensionality")
        D, Lpl = L_n.value.shape[-1], L_n.value.shape[:-2]
    else:
        D, Lpl = L_n.dims[0][0], L_n.plates
    if isinstance(mu_n, Constant):
        if mu_n.value.ndim < 1 or mu_n.value.shape[-1] != D:
            raise ValueError("First parent has wrong dimensionality")
        mupl = mu_n.value.shape[:-1]
    else:
        if mu_n.dims[0] != (D,):
            raise ValueError("First parent has wrong dimensionality")
        mupl = mu_n.plates
    return D, mupl, Lpl


class SwitchingGaussianMarkovChain(GaussianMarkovChain):
    """``SwitchingGaussianMarkovChain(mu, Lambda, B, Z, nu, n=N)``: a Gaussian Markov chain whose
    dynamics matrix at every transition is picked from K matrices by a categorical variable,
    x_n ~ N(B[z_{n-1}] x_{n-1}, diag(nu)^-1)  (reference gaussian_markov_chain.py:1454-1985).
    ``B``: Gaussian with shape (D,) and plates (..., K, D) -- the last plate indexes the rows of
    the K matrices; ``Z``: categorical-like with plates (..., N-1) (a CategoricalMarkovChain of
    N-1 states is seen through its categorical view); ``nu``: gamma-like with last plate D,
    constant in time.  Messages go to ``B`` and ``Z`` (and to mu / Lambda); the innovation
    precision is not updated, like in the reference (:1555-1558)."""

    def __init__(self, mu, Lambda, B, Z, nu, n=None, plates=None, name=None):
        if hasattr(Z, 'as_categorical'):
            Z = Z.as_categorical()
        Stochastic.__init__(self, mu, Lambda, B, Z, nu, plates=(), dims=((), (), ()), name=name)
        mu_n, L_n, B_n, Z_n, nu_n = self.parents
        D, mupl, Lpl = _initial_state(mu_n, L_n)
        if isinstance(B_n, Constant):
            if B_n.value.ndim < 3 or B_n.value.shape[-1] != D:
                raise ValueError("Third parent has wrong dimensionality")
            Bpl = B_n.value.shape[:-1]
        else:
            if tuple(B_n.dims[0]) != (D,):
                raise ValueError("Third parent has wrong dimensionality")
            Bpl = B_n.plates
        if len(Bpl) < 2 or Bpl[-1] != D:
            raise ValueError("Third parent should have a last plate equal to the "
                             "dimensionality of the system.")
        K = Bpl[-2]
        if isinstance(Z_n, Constant):
            if np.any(Z_n.value != np.round(Z_n.value)) or np.any(Z_n.value < 0) \
                    or np.any(Z_n.value >= K):
                raise ValueError("Invalid category index")
            Zpl = Z_n.value.shape
        else:
            if tuple(Z_n.dims) != ((K,),):
                raise ValueError("Fourth parent has wrong dimensionality %s, should be %s"
                                 % (Z_n.dims, ((K,),)))
            Zpl = Z_n.plates
        if len(Zpl) == 0:
            raise ValueError("Z must have temporal axis on plates")
        nupl = nu_n.value.shape if isinstance(nu_n, Constant) else nu_n.plates
        if len(nupl) == 0 or nupl[-1] != D:
            raise Exception("Fifth parent should have a last plate equal to the "
                            "dimensionality of the system.")
        if len(nupl) >= 2 and nupl[-2] != 1:
            raise NotImplementedError('a time-dependent innovation precision is not built')
        n_Z = Zpl[-1]
        if n is None:
            if n_Z == 1:
                raise Exception("The number of time instances could not be determined "
                                "automatically. Give the number of time instances.")
            n = n_Z + 1
        if n_Z != n - 1:
            raise ValueError("The last plate of the fourth parent should have length equal to "
                             "N-1, where N is the number of time instances.")
        self.N, self.D, self.K = int(n), int(D), int(K)
        self.dims = ((self.N, D), (self.N, D, D), (self.N - 1, D, D))
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, mupl, Lpl, Bpl[:-2], Zpl[:-1], nupl[:-2])
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))


class VaryingGaussianMarkovChain(GaussianMarkovChain):
    """``VaryingGaussianMarkovChain(mu, Lambda, B, S, nu, n=N)``: a Gaussian Markov chain whose
    dynamics matrix is a time-varying linear combination of K matrices,
    x_n ~ N((sum_k s_{n-1,k} B_k) x_{n-1}, diag(nu)^-1)  (reference
    gaussian_markov_chain.py:930-1452).  ``B``: Gaussian with shape (D, K) and last plate D (the
    rows of the matrices); ``S``: Gaussian with shape (K,) and plates (..., N-1) (e.g. the
    Gaussian view of another chain, sliced ``[1:]``); ``nu``: gamma-like with last plate D,
    constant in time.  Messages go to ``B`` and ``S`` (and to mu / Lambda)."""

    def __init__(self, mu, Lambda, B, S, nu, n=None, plates=None, name=None):
        if isinstance(S, GaussianMarkovChain):
            S = S.as_gaussian()
        Stochastic.__init__(self, mu, Lambda, B, S, nu, plates=(), dims=((), (), ()), name=name)
        mu_n, L_n, B_n, S_n, nu_n = self.parents
        D, mupl, Lpl = _initial_state(mu_n, L_n)
        if isinstance(B_n, Constant) or isinstance(S_n, Constant):
            raise NotImplementedError('the dynamics matrices and their weights must be nodes')
        if len(B_n.dims[0]) != 2 or B_n.dims[0][0] != D:
            raise ValueError("Third parent has wrong dimensionality")
        K = B_n.dims[0][1]
        if len(B_n.plates) == 0 or B_n.plates[-1] != D:
            raise ValueError("Third parent should have a last plate equal to the "
                             "dimensionality of the system.")
        if tuple(S_n.dims[0]) != (K,):
            raise ValueError("Fourth parent has wrong dimensionality")
        if len(S_n.plates) == 0:
            raise ValueError("The weights must have a temporal axis on their plates")
        nupl = nu_n.value.shape if isinstance(nu_n, Constant) else nu_n.plates
        if len(nupl) == 0 or nupl[-1] != D:
            raise Exception("Fifth parent should have a last plate equal to the "
                            "dimensionality of the system.")
        if len(nupl) >= 2 and nupl[-2] != 1:
            raise NotImplementedError('a time-dependent innovation precision is not built')
        n_S = S_n.plates[-1]
        if n is None:
            if n_S == 1:
                raise Exception("The number of time instances could not be determined "
                                "automatically. Give the number of time instances.")
            n = n_S + 1
        if n_S != n - 1:
            raise ValueError("The last plate of the fourth parent should have length equal to "
                             "N-1, where N is the number of time instances.")
        self.N, self.D, self.K = int(n), int(D), int(K)
        self.dims = ((self.N, D), (self.N, D, D), (self.N - 1, D, D))
        given = tuple(plates) if plates is not None else ()
        self.plates = broadcasted_shape(given, mupl, Lpl, B_n.plates[:-1], S_n.plates[:-1],
                                        nupl[:-2])
        if plates is not None and self.plates != given:
            raise ValueError('Plates of the parents do not broadcast to plates %s' % (given,))
