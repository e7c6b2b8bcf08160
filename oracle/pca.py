"""
CPU oracle for the fully-observed PCA / factor-analysis VB iteration.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: PINNED against
the live reference through tests/golden/pca_*.npz (made by oracle/make_golden.py).

Model (reference: bayespy/demos/pca.py:22-61)::

    alpha_k ~ Gamma(a0, b0)              plates (K,)
    w_d     ~ N(0, diag(alpha)^-1)       plates (D,1), shape (K,)
    x_n     ~ N(0, I)                    plates (1,N), shape (K,)
    tau     ~ Gamma(a0, b0)
    y_dn    ~ N(w_d . x_n, 1/tau)        observed, scalar mask

One VB iteration = the node updates in the constructor order W, X, tau, alpha
followed by the full lower bound, which is what ``VB.update`` does
(bayespy/inference/vmp/vmp.py:154-172, :693-764).

This file restates, in "sufficient statistics" form and chunked over the big
plate N, the arithmetic of

* ``GaussianARDDistribution.compute_phi_from_parents`` / ``compute_moments_and_cgf``
  (bayespy/inference/vmp/nodes/gaussian.py:649-706),
* ``SumMultiply._message_to_parent`` (bayespy/inference/vmp/nodes/dot.py:425-633)
  and ``SumMultiply._compute_moments`` (dot.py:316-415),
* ``WrapToGaussianGamma._compute_message_to_parent`` (gaussian.py:2352-2371),
* ``GammaDistribution.compute_moments_and_cgf`` (bayespy/inference/vmp/nodes/gamma.py:124-148),
* ``ExponentialFamily.lower_bound_contribution`` (bayespy/inference/vmp/nodes/expfamily.py:400-480).

Everything is IEEE float64.  Layout: ``y`` is (D, N) C-order (N contiguous),
which is the reference's plate layout for Y (plates (D, N)).
"""
import numpy as np
from scipy import special


def gamma_moments(a, b):
    """<x>, <log x> of Gamma(a, b).  gamma.py:142-146."""
    return a / b, special.digamma(a) - np.log(b)


def gamma_elbo(a0, b0, a, b):
    """
    E[log p(x|a0,b0) - log q(x)] for q = Gamma(a, b), summed over elements.

    expfamily.py:400-480 with phi = [-b, a], u = [<x>, <log x>],
    g = a log b - lnGamma(a) (gamma.py:147, :160).
    """
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    x, logx = gamma_moments(a, b)
    g_p = a0 * np.log(b0) - special.gammaln(a0)
    g_q = a * np.log(b) - special.gammaln(a)
    L = g_p - g_q + (-b0 + b) * x + (a0 - a) * logx
    return float(np.sum(L))


def spd_inv_logdet(M):
    """Inverse and log-determinant of an SPD matrix via Cholesky
    (bayespy/utils/linalg.py:31-63, :174-223)."""
    L = np.linalg.cholesky(M)
    Linv = np.linalg.solve(L, np.eye(M.shape[0]))
    return Linv.T @ Linv, 2.0 * float(np.sum(np.log(np.diag(L))))


def init_stats(y, x0, chunk=1 << 16):
    """
    Statistics of the injected initial state of X.

    ``X.initialize_from_value(x0)`` makes the moments the delta moments
    u = [x0, x0 x0^T] (expfamily.py:193-204, gaussian.py:74-84).

    y: (D, N); x0: (N, K).  Returns dict(Sxx (K,K), Syx (D,K), Syy float).
    """
    D, N = y.shape
    K = x0.shape[1]
    Sxx = np.zeros((K, K))
    Syx = np.zeros((D, K))
    Syy = 0.0
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        xc = x0[s:e]
        yc = y[:, s:e]
        Sxx += xc.T @ xc
        Syx += yc @ xc
        Syy += float(np.einsum('dn,dn->', yc, yc))
    return dict(Sxx=Sxx, Syx=Syx, Syy=Syy)


class PCAOracle:
    """Chunked float64 VB for fully observed PCA.  See module docstring."""

    mu = None          # constant prior mean of W (None = 0); set by __init__

    def __init__(self, y, x0, a0=1e-2, b0=1e-2, chunk=1 << 16, keep_x=True, mu=None):
        """``mu``: a constant prior mean of W, broadcastable to (D, K) (GaussianARD(mu, alpha):
        phi0 = <alpha> mu, gaussian.py:805-830); None = 0."""
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.D, self.N = self.y.shape
        self.K = x0.shape[1]
        self.mu = (None if mu is None else
                   np.array(np.broadcast_to(mu, (self.D, self.K)), dtype=np.float64))
        self.a0 = float(a0)
        self.b0 = float(b0)
        self.chunk = int(chunk)
        self.keep_x = keep_x
        st = init_stats(self.y, np.asarray(x0, dtype=np.float64), chunk)
        self.Sxx, self.Syx, self.Syy = st['Sxx'], st['Syx'], st['Syy']
        # initialize_from_prior: Gamma prior moments (expfamily.py:168-184)
        self.tau_a, self.tau_b = self.a0, self.b0
        self.alpha_a = np.full(self.K, self.a0)
        self.alpha_b = np.full(self.K, self.b0)
        self.W = None
        self.CW = None
        self.CX = None
        self.X = np.array(x0, dtype=np.float64) if keep_x else None
        self.L = []
        self.L_terms = []

    # -- node updates ------------------------------------------------------

    def update_W(self):
        """gaussian.py:649-706 with the messages of dot.py:581 (E3/E4)."""
        tau, _ = gamma_moments(self.tau_a, self.tau_b)
        alpha, _ = gamma_moments(self.alpha_a, self.alpha_b)
        Lam = np.diag(alpha) + tau * self.Sxx
        self.CW, self.logdet_LamW = spd_inv_logdet(Lam)
        rhs = tau * self.Syx
        if self.mu is not None:
            rhs = rhs + alpha * self.mu                        # prior term <alpha_k> mu_dk
        self.W = rhs @ self.CW                                 # (D,K)
        self.Sww = self.D * self.CW + self.W.T @ self.W

    def _ww(self):
        """sum_d <(w_dk - mu_dk)^2> (the message to alpha, gaussian.py:862-880)."""
        ww = np.diag(self.Sww).copy()
        if self.mu is not None:
            ww += np.sum(self.mu * self.mu - 2.0 * self.mu * self.W, axis=0)
        return ww

    def update_X(self):
        """gaussian.py:649-706 with the messages of dot.py:581 (E5/E6): the
        single streaming pass over Y."""
        tau, _ = gamma_moments(self.tau_a, self.tau_b)
        Lam = np.eye(self.K) + tau * self.Sww
        self.CX, self.logdet_LamX = spd_inv_logdet(Lam)
        A = tau * self.CX @ self.W.T                           # (K,D)
        Sxx = np.zeros((self.K, self.K))
        Syx = np.zeros((self.D, self.K))
        for s in range(0, self.N, self.chunk):
            e = min(self.N, s + self.chunk)
            yc = self.y[:, s:e]
            xc = A @ yc                                        # (K,n)
            Sxx += xc @ xc.T
            Syx += yc @ xc.T
            if self.keep_x:
                self.X[s:e] = xc.T
        self.Sxx_mean = Sxx
        self.Sxx = self.N * self.CX + Sxx
        self.Syx = Syx

    def _residual(self):
        """sum_dn <(y - f)^2>; dot.py:355,403 (E1/E2) collapsed to traces."""
        Syf = float(np.sum(self.W * self.Syx))
        Sff = float(np.sum(self.Sww * self.Sxx))
        return self.Syy - 2.0 * Syf + Sff

    def update_tau(self):
        """gamma.py:116-148 with the message gaussian.py:2363-2369."""
        self.tau_a = self.a0 + 0.5 * self.D * self.N
        self.tau_b = self.b0 + 0.5 * self._residual()

    def update_alpha(self):
        self.alpha_a = np.full(self.K, self.a0 + 0.5 * self.D)
        self.alpha_b = self.b0 + 0.5 * self._ww()

    # -- lower bound ---------------------------------------------------------

    def lower_bound(self):
        """expfamily.py:400-480 specialised per node."""
        D, N, K = self.D, self.N, self.K
        tau, logtau = gamma_moments(self.tau_a, self.tau_b)
        alpha, logalpha = gamma_moments(self.alpha_a, self.alpha_b)
        L_Y = (D * N * (-0.5 * np.log(2 * np.pi) + 0.5 * logtau)
               - 0.5 * tau * self._residual())
        L_X = -0.5 * np.trace(self.Sxx) + N * (-0.5 * self.logdet_LamX + 0.5 * K)
        L_W = (0.5 * D * np.sum(logalpha)
               - 0.5 * np.sum(alpha * self._ww())
               + D * (-0.5 * self.logdet_LamW + 0.5 * K))
        L_tau = gamma_elbo(self.a0, self.b0, self.tau_a, self.tau_b)
        L_alpha = gamma_elbo(self.a0, self.b0, self.alpha_a, self.alpha_b)
        terms = dict(Y=float(L_Y), X=float(L_X), W=float(L_W),
                     tau=float(L_tau), alpha=float(L_alpha))
        return float(L_Y + L_X + L_W + L_tau + L_alpha), terms

    # -- driver -----------------------------------------------------------------

    def iterate(self, n=1):
        for _ in range(n):
            self.update_W()
            self.update_X()
            self.update_tau()
            self.update_alpha()
            L, terms = self.lower_bound()
            self.L.append(L)
            self.L_terms.append(terms)
        return self.L[-1]

    def moments(self):
        tau = gamma_moments(self.tau_a, self.tau_b)
        alpha = gamma_moments(self.alpha_a, self.alpha_b)
        return dict(W=self.W, CW=self.CW, X=self.X, CX=self.CX,
                    tau=np.array(tau), alpha=np.array(alpha),
                    Sxx=self.Sxx, Syx=self.Syx)


def make_pca_data(N, D, K, seed=42, noise=0.1, chunk=1 << 18):
    """
    Synthetic inputs of BASELINE.md section 3 / SURVEY.md section 8(d)
    (follows bayespy/demos/pca.py:70-74).  Returns y (D,N) and x0 (N,K).
    """
    rs = np.random.RandomState(seed)
    w = rs.normal(0, 1, (D, K))
    y = np.empty((D, N))
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        x = rs.normal(0, 1, (e - s, K))
        y[:, s:e] = w @ x.T + noise * rs.normal(size=(D, e - s))
    x0 = rs.normal(0, 1, (N, K))
    return y, x0
