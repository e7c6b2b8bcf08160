"""
Rotation parameter expansion for pairs of Gaussian blocks whose product is observed
(reference: bayespy/inference/vmp/transformations.py:23-224 ``RotationOptimizer``,
:376-1110 ``RotateGaussianARD``; used by demos/pca.py:85-94 as the VB callback).

Rotating q(W) by R and q(X) by R^-T leaves <w^T x> unchanged but can raise the lower bound
by orders of magnitude per iteration.  The split of work follows SURVEY.md 8(f).3:

* the K x K optimisation runs on the host (SciPy nonlinear CG, like the reference's
  utils/optimize.py:15-25) on the sufficient statistics  sum_plates <x x^T>  which the
  execution plan hands over -- a K x K read-back, never plate-sized data;
* the rotation itself is applied to the device state by the plan that owns the node
  (``plan.rotate_node``): means and covariances on the device, the plate-sized array of the
  fused PCA block through the fp64 MFMA contraction kernel.

Supported: ``RotateGaussianARD(X)`` / ``RotateGaussianARD(X, alpha)`` for a zero-mean
GaussianARD rotated along its last axis with the precision shared over the plates
(``alpha`` with plates ``(K,)`` or a constant) -- the PCA / factor-analysis use --, of all
components or of a ``subset`` of them (transformations.py:425-455, :639-690: the statistics are
restricted to the subset, the rotation is the identity elsewhere), and -- for a node whose single
plate axis is rotated as well (``setup(plate_axis=-1)``, ``Q``) -- the dynamics matrix of a linear
state-space model: ``RotateGaussianMarkovChain(X, RotateGaussianARD(A, alpha))`` rotates the state
space of a ``GaussianMarkovChain`` with unit innovation noise (transformations.py:1096-1450; the
speed-up of demos/lssm.py:134-190).  Non-zero prior means raise ``NotImplementedError``.
"""
import warnings

import numpy as np
from scipy import optimize as _sp_optimize

from ..nodes.node import Constant
from ..nodes.gamma import Gamma
from ..nodes.gaussian import GaussianARD


def _minimize(f, x0, maxiter=None, verbose=False):
    """SciPy nonlinear conjugate gradients on (value, gradient) functions
    (utils/optimize.py:15-25)."""
    options = {'disp': verbose}
    if maxiter is not None:
        options['maxiter'] = maxiter
    return _sp_optimize.minimize(f, x0, jac=True, method='CG', options=options).x


class RotateGaussianARD:
    """Rotation of q(X) (and the coupled update of q(alpha)) along the last axis of a
    GaussianARD node with prior  N(0, diag(alpha)^-1)  (transformations.py:376-1110)."""

    def __init__(self, X, *alpha, axis=-1, precompute=False, subset=None):
        if not isinstance(X, GaussianARD) or X.ndim != 1:
            raise NotImplementedError('RotateGaussianARD supports vector-valued GaussianARD nodes')
        if not isinstance(axis, int):
            raise ValueError("Axis must be integer")
        if axis >= 0:
            axis -= X.ndim
        if axis < -X.ndim or axis >= 0:
            raise ValueError("Axis out of bounds")
        if len(alpha) > 1:
            raise ValueError('Too many arguments')
        self.node_X = X
        self.Dfull = X.dims[0][-1]
        # only a subset of the components is rotated (transformations.py:425-455)
        if subset is None:
            self.subset = None
            self.D = self.Dfull
        else:
            self.subset = [int(i) for i in subset]
            if len(set(self.subset)) != len(self.subset) or not all(
                    0 <= i < self.Dfull for i in self.subset):
                raise ValueError('subset must hold distinct component indices')
            self.D = len(self.subset)
        self.update_alpha = len(alpha) == 1
        mu, prec = X.parents
        if not (isinstance(mu, Constant) and np.all(np.asarray(mu.value) == 0)):
            raise NotImplementedError('RotateGaussianARD needs a constant zero prior mean')
        if self.update_alpha:
            self.node_alpha = alpha[0]
            if self.node_alpha is not prec:
                raise ValueError('alpha must be the precision parent of X')
            if not isinstance(prec, Gamma) or prec.plates not in ((self.Dfull,), (1,), ()):
                raise NotImplementedError('alpha must be a Gamma node with plates (K,) or scalar')
        else:
            if not isinstance(prec, Constant):
                raise NotImplementedError('without alpha the precision must be a constant')
            a = np.asarray(prec.value, dtype=np.float64)
            if a.ndim > 1 or a.size not in (1, self.Dfull):
                raise NotImplementedError('constant precision must be a scalar or a (K,) vector')
            self.alpha = self._sub(np.broadcast_to(a, (self.Dfull,)).astype(np.float64))

    def nodes(self):
        return [self.node_X, self.node_alpha] if self.update_alpha else [self.node_X]

    def _sub(self, v):
        """Restrict a per-component vector to the rotated subset."""
        return v if self.subset is None else v[..., self.subset]

    def _embed(self, R):
        """The rotation of all components: ``R`` on the subset, the identity elsewhere."""
        if self.subset is None:
            return R
        full = np.identity(self.Dfull)
        full[np.ix_(self.subset, self.subset)] = R
        return full

    # -- statistics ---------------------------------------------------------------------------
    def setup(self, plate_axis=None):
        """Fetch  XX = sum_plates <x x^T>  (K x K) and the number of plates from the plan that
        owns X (transformations.py:476-640 for mu = 0, axis = -1).  With ``plate_axis`` the rows
        (plates) are rotated too: the per-row means (N, K) and covariances (N, K, K) are kept, and
        XX becomes a function of the plate rotation Q (transformations.py:714-752)."""
        if plate_axis is not None:
            if len(self.node_X.plates) != 1 or plate_axis not in (-1, 0):
                raise NotImplementedError('plate rotations are built for a node with one plate axis')
            if self.subset is not None:
                raise NotImplementedError('a subset cannot be combined with a plate rotation')
        plan = self.node_X._plan
        if plan is None:
            raise RuntimeError('node %s is not part of a VB engine' % self.node_X.name)
        self.plate_axis = plate_axis
        if plate_axis is not None:
            rows = plan.rotation_rows(self.node_X)
            self.Xm = np.asarray(rows['mean'], dtype=np.float64)         # (N, K)
            self.CovX = np.asarray(rows['cov'], dtype=np.float64)        # (N, K, K)
            self.XX = self.CovX.sum(axis=0) + self.Xm.T @ self.Xm
            self.nplates = float(self.Xm.shape[0])
        else:
            st = plan.rotation_statistics(self.node_X)
            self.XX = np.asarray(st['XX'], dtype=np.float64)
            if self.subset is not None:
                self.XX = self.XX[np.ix_(self.subset, self.subset)]
            self.nplates = float(st['nplates'])
        if self.update_alpha:
            K = self.Dfull
            a = self.node_alpha._plan.gamma_posterior_shape(self.node_alpha)
            self.a = self._sub(np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1), (K,)))
            a0 = np.asarray(self.node_alpha.parents[0].value, dtype=np.float64).reshape(-1)
            b0 = np.asarray(self.node_alpha.parents[1].value, dtype=np.float64).reshape(-1)
            self.a0 = self._sub(np.broadcast_to(a0, (K,)))
            self.b0 = self._sub(np.broadcast_to(b0, (K,)))
            if self.node_alpha.plates != (K,):
                raise NotImplementedError('a precision shared over the rotated axis is not '
                                          'supported')

    # -- bound and gradient (transformations.py:693-960) ------------------------------------------
    def _terms(self, R, logdet, inv, Q=None):
        """(bound of X, bound of alpha, gradient w.r.t. R[, gradient w.r.t. Q]).

        With a plate rotation Q (rows i of the node: mean_i <- sum_k Q_ik mean_k exactly, covariance_i
        <- s_i^2 Cov_i with s = column sums of Q -- the approximation of gaussian.py:1743-1772):
        XX(Q) = sum_i s_i^2 Cov_i + X^T Q^T Q X, and the entropy gains K sum_i log|s_i|."""
        if Q is None:
            XX = self.XX
        else:
            sQ = Q.sum(axis=0)
            QX = Q @ self.Xm
            XX = np.einsum('i,ikl->kl', sQ * sQ, self.CovX) + QX.T @ QX
        bx, ba, grad, coef = self._terms_xx(R, logdet, inv, XX)
        if Q is None:
            return bx, ba, grad
        K = self.D
        bx = bx + K * np.sum(np.log(np.abs(sQ)))
        # d bound / d XX = R^T diag(coef / 2) R =: G;  XX depends on Q through both of its terms
        G = R.T @ (0.5 * coef[:, None] * R)
        dQ = 2.0 * (QX @ G @ self.Xm.T) \
            + (2.0 * sQ * np.einsum('kl,ikl->i', G, self.CovX) + K / sQ)[None, :]
        return bx, ba, grad, dQ

    def _terms_xx(self, R, logdet, inv, XX):
        RXX = R @ XX
        v = np.einsum('ik,ik->i', RXX, R)           # <(R x)_k^2> summed over the plates
        N = self.nplates
        if self.update_alpha:
            b = self.b0 + 0.5 * v
            alpha = self.a / b
            logalpha = -np.log(b)
        else:
            alpha = self.alpha
            logalpha = np.zeros(self.D)
        logH_X = N * logdet                          # entropy of q(X)
        logp_X = -0.5 * np.sum(alpha * v) + 0.5 * N * np.sum(logalpha)
        logp_alpha = 0.0
        if self.update_alpha:
            # the entropy of q(alpha) cancels against the log(alpha) term of <log p(alpha)>
            logp_alpha = np.sum(self.a0 * logalpha) - np.sum(self.b0 * alpha)
        # gradient with respect to R, row k multiplies RXX[k]
        if self.update_alpha:
            coef = (-alpha + (0.5 * v + self.b0) * alpha / b - (0.5 * N + self.a0) / b)
        else:
            coef = -alpha
        grad = N * inv.T + coef[:, None] * RXX
        return logp_X + logH_X, logp_alpha, grad, coef

    def _check_q(self, Q):
        if (Q is None) != (getattr(self, 'plate_axis', None) is None):
            raise ValueError('a plate rotation Q goes with setup(plate_axis=...) and only with it')

    def bound(self, R, logdet=None, inv=None, Q=None):
        """(bound, d bound / d R) -- and d bound / d Q as a third item when the plates rotate."""
        self._check_q(Q)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        if inv is None:
            inv = np.linalg.inv(R)
        t = self._terms(R, logdet, inv, Q)
        return (t[0] + t[1],) + tuple(t[2:])

    def get_bound_terms(self, R, logdet=None, inv=None, Q=None):
        self._check_q(Q)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        if inv is None:
            inv = np.linalg.inv(R)
        bx, ba = self._terms(R, logdet, inv, Q)[:2]
        terms = {self.node_X: bx}
        if self.update_alpha:
            terms[self.node_alpha] = ba
        return terms

    # -- apply -------------------------------------------------------------------------------------
    def rotate(self, R, inv=None, logdet=None, Q=None):
        self._check_q(Q)
        R = np.asarray(R, dtype=np.float64)
        if inv is None:
            inv = np.linalg.inv(R)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        if Q is None:
            self.node_X._plan.rotate_node(self.node_X, self._embed(R), self._embed(inv),
                                          float(logdet))
        else:
            self.node_X._plan.rotate_node(self.node_X, R, inv, float(logdet),
                                          Q=np.asarray(Q, dtype=np.float64))
        if self.update_alpha:
            self.node_alpha.update()


class RotateGaussianMarkovChain:
    """Rotation x_t -> R x_t of the state space of a ``GaussianMarkovChain`` with unit innovation
    noise, together with its dynamics matrix A -> R A R^-1 (``A_rotator`` =
    ``RotateGaussianARD(A[, alpha])``, whose columns rotate by R^-T and whose rows by Q = R)
    (transformations.py:1096-1450).  The statistics are K x K plate sums handed over by the plan
    that owns the chain; the plate-sized means are rotated on the device."""

    def __init__(self, X, *args):
        from ..nodes.gaussian_markov_chain import GaussianMarkovChain
        if not isinstance(X, GaussianMarkovChain):
            raise ValueError('RotateGaussianMarkovChain needs a GaussianMarkovChain')
        if len(args) != 1:
            if len(args) == 0:
                raise NotImplementedError()
            raise ValueError("Wrong number of arguments")
        self.X_node = X
        self.A_rotator = args[0]
        mu, Lam, A, nu = X.parents[:4]
        if len(X.parents) > 4:
            raise NotImplementedError('input signals of the chain are not built')
        if self.A_rotator.node_X is not A:
            raise ValueError('the rotator of the dynamics must rotate the A of this chain')
        if not (isinstance(nu, Constant) and np.all(np.asarray(nu.value) == 1)):
            raise NotImplementedError('RotateGaussianMarkovChain assumes unit innovation noise')
        if not (isinstance(mu, Constant) and isinstance(Lam, Constant)):
            raise NotImplementedError('the initial state must have constant mean and precision')
        self.A_node = A
        self.mu0 = np.asarray(mu.value, dtype=np.float64)
        self.Lambda = np.asarray(Lam.value, dtype=np.float64)
        self.D = X.D

    def nodes(self):
        return [self.X_node] + self.A_rotator.nodes()

    def setup(self):
        """K x K sums over time and sequences of the chain's moments, combined with the moments
        of A (transformations.py:1239-1290; A has no time plate here)."""
        st = self.X_node._plan.rotation_statistics(self.X_node)
        self.M = float(st['nvec'])                       # rotated vectors: T x sequences
        self.X0X0 = np.asarray(st['X0X0'], dtype=np.float64)
        self.XnXn = np.asarray(st['XnXn'], dtype=np.float64)
        XpXn = np.asarray(st['XpXn'], dtype=np.float64)      # sum <x_t-1 x_t^T>
        XpXp = np.asarray(st['XpXp'], dtype=np.float64)
        self.Lambda_mu_X0 = np.outer(self.Lambda @ self.mu0, np.asarray(st['X0'], dtype=np.float64))
        self.A_rotator.setup(plate_axis=-1)
        Am, CovA = self.A_rotator.Xm, self.A_rotator.CovX
        self.A_XpXn = Am @ XpXn
        self.A_XpXp_A = Am @ XpXp @ Am.T
        self.CovA_XpXp = np.einsum('dij,ij->d', CovA, XpXp)

    def _x_terms(self, R, logdet, inv):
        """<log p(X | A)> + H(q(X)) after x -> R x, and its gradient (transformations.py:1296-1386)."""
        sumr = R.sum(axis=0)
        R_XnXn = R @ self.XnXn
        L_R_X0X0 = self.Lambda @ R @ self.X0X0
        RA_XpXp_A = R @ self.A_XpXp_A
        yy = np.sum(R_XnXn * R) + np.sum(L_R_X0X0 * R)
        yz = np.sum((R @ self.A_XpXn) * R) + np.sum(self.Lambda_mu_X0 * R)
        zz = np.sum(RA_XpXp_A * R) + np.sum(sumr * sumr * self.CovA_XpXp)
        bound = -0.5 * yy + yz - 0.5 * zz + self.M * logdet
        dyy = 2.0 * (R_XnXn + L_R_X0X0)
        dyz = R @ (self.A_XpXn + self.A_XpXn.T) + self.Lambda_mu_X0
        dzz = 2.0 * (RA_XpXp_A + (sumr * self.CovA_XpXp)[None, :])
        grad = -0.5 * dyy + dyz - 0.5 * dzz + self.M * inv.T
        return bound, grad

    def bound(self, R, logdet=None, inv=None):
        if inv is None:
            inv = np.linalg.inv(R)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        bx, dx = self._x_terms(R, logdet, inv)
        # A: columns by R^-T, rows by Q = R
        ba, dRa, dQa = self.A_rotator.bound(inv.T, inv=R.T, logdet=-logdet, Q=R)
        dRa = -inv.T @ dRa.T @ inv.T
        return bx + ba, dx + dRa + dQa

    def get_bound_terms(self, R, logdet=None, inv=None):
        if inv is None:
            inv = np.linalg.inv(R)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        terms = self.A_rotator.get_bound_terms(inv.T, inv=R.T, logdet=-logdet, Q=R)
        terms[self.X_node] = self._x_terms(R, logdet, inv)[0]
        return terms

    def rotate(self, R, inv=None, logdet=None):
        R = np.asarray(R, dtype=np.float64)
        if inv is None:
            inv = np.linalg.inv(R)
        if logdet is None:
            logdet = np.linalg.slogdet(R)[1]
        self.X_node._plan.rotate_node(self.X_node, R, inv, float(logdet))
        self.A_rotator.rotate(inv.T, inv=R.T, logdet=-logdet, Q=R)


class RotationOptimizer:
    """Jointly optimal rotation of two blocks: block1 by R, block2 by R^-T
    (transformations.py:23-224)."""

    def __init__(self, block1, block2, D):
        self.block1 = block1
        self.block2 = block2
        self.D = D

    def rotate(self, maxiter=10, check_gradient=False, verbose=False, check_bound=False):
        D = self.D

        def cost(r):
            R = np.reshape(r, (D, D))
            invR = np.linalg.inv(R)
            logdetR = np.linalg.slogdet(R)[1]
            b1, db1 = self.block1.bound(R, logdet=logdetR, inv=invR)
            b2, db2 = self.block2.bound(invR.T, logdet=-logdetR, inv=R.T)
            # chain rule for the block rotated by R^-T (transformations.py:90-96)
            db2 = -invR.T @ db2.T @ invR.T
            return -(b1 + b2), -np.ravel(db1 + db2)

        def bound_terms(r):
            R = np.reshape(r, (D, D))
            invR = np.linalg.inv(R)
            logdetR = np.linalg.slogdet(R)[1]
            t = self.block1.get_bound_terms(R, logdet=logdetR, inv=invR)
            t.update(self.block2.get_bound_terms(invR.T, logdet=-logdetR, inv=R.T))
            return t

        def true_bound_terms():
            nodes = set(self.block1.nodes()) | set(self.block2.nodes())
            return {n: n.lower_bound_contribution() for n in nodes}

        self.block1.setup()
        self.block2.setup()

        if check_gradient:
            R0 = np.random.randn(D, D)
            g = cost(np.ravel(R0))[1]
            gn = _sp_optimize.approx_fprime(np.ravel(R0), lambda x: cost(x)[0],
                                            np.sqrt(np.finfo(float).eps))
            err = np.linalg.norm(g - gn) / max(np.linalg.norm(gn), 1e-300)
            if err > 1e-5:
                warnings.warn("Rotation gradient has relative error %g" % err)

        r0 = np.ravel(np.identity(D))
        cost_begin = cost(r0)[0]
        if check_bound:
            terms_begin = bound_terms(r0)
            true_begin = true_bound_terms()

        r = _minimize(cost, r0, maxiter=maxiter, verbose=verbose)
        cost_end = cost(r)[0]

        R = np.reshape(r, (D, D))
        invR = np.linalg.inv(R)
        logdetR = np.linalg.slogdet(R)[1]
        self.block1.rotate(R, inv=invR, logdet=logdetR)
        self.block2.rotate(invR.T, inv=R.T, logdet=-logdetR)

        if cost_end - cost_begin > 0:
            warnings.warn("Rotation optimization made the cost function worse by %g. Probably a "
                          "bug in the gradient of the rotation functions."
                          % (cost_end - cost_begin,))
        if check_bound:
            terms_end = bound_terms(r)
            true_end = true_bound_terms()
            change = 0.0
            for node in terms_begin:
                d = terms_end[node] - terms_begin[node]
                dt = true_end[node] - true_begin[node]
                change += d
                if not np.allclose(d, dt):
                    warnings.warn("Rotation cost function is not consistent with the true lower "
                                  "bound for node %s. Bound changed %g but optimized function "
                                  "changed %g." % (node.name, dt, d))
            if change < 0:
                warnings.warn("Rotation made the true lower bound worse by %g. Probably a bug "
                              "in the rotation functions." % (change,))
        return R
