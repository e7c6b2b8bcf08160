"""
CPU oracle for the full-covariance Gaussian mixture VB iteration.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: PINNED against the
live reference through tests/golden/gmm_*.npz (made by oracle/make_golden.py).

Model (reference: bayespy/demos/mog.py:17-64)::

    pi       ~ Dirichlet(alpha0)                       (K,)
    z_n      ~ Categorical(pi)                         plates (N,)
    mu_k     ~ N(0, beta0^-1 I)                        plates (K,), shape (D,)
    Lambda_k ~ Wishart(n0, V0)   (V0 = inverse scale)  plates (K,)
    y_n      ~ N(mu_{z_n}, Lambda_{z_n}^-1)            observed

One VB iteration = updates in the constructor order mu, Lambda, z, alpha followed
by the lower bound (vmp.py:154-172, :693-764).  Restated in sufficient-statistics
form, chunked over N, following

* ``MixtureDistribution`` messages / phi / cgf (mixture.py:53-293),
* ``GaussianDistribution`` + ``WrapToGaussianWishart`` (gaussian.py:293-573, :2374-2527),
* ``GaussianARDDistribution`` for mu (gaussian.py:576-741),
* ``WishartDistribution`` (wishart.py:118-225), ``multidigamma`` (utils/misc.py:1146-1151),
* ``DirichletDistribution`` (dirichlet.py:107-231),
* ``MultinomialDistribution`` / ``CategoricalDistribution`` softmax moments
  (multinomial.py:83-128, utils/misc.py:1366-1401),
* ``ExponentialFamily.lower_bound_contribution`` (expfamily.py:400-480).
"""
import numpy as np
from scipy import special

LOG2PI = np.log(2 * np.pi)


def multidigamma(a, d):
    return np.sum(special.digamma(np.asarray(a)[..., None] - 0.5 * np.arange(d)), axis=-1)


def multigammaln(a, d):
    a = np.asarray(a, dtype=np.float64)
    return (d * (d - 1) / 4.0 * np.log(np.pi)
            + np.sum(special.gammaln(a[..., None] - 0.5 * np.arange(d)), axis=-1))


class GMMOracle:

    def __init__(self, y, lab0, K, alpha0=1e-3, beta0=1e-3, n0=None, V0=None, chunk=1 << 15):
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.N, self.D = self.y.shape
        self.K = K
        self.alpha0 = np.broadcast_to(np.asarray(alpha0, dtype=np.float64), (K,)).copy()
        self.beta0 = float(beta0)
        self.n0 = float(self.D if n0 is None else n0)
        self.V0 = 0.01 * np.eye(self.D) if V0 is None else np.asarray(V0, dtype=np.float64)
        self.chunk = int(chunk)
        D = self.D
        # statistics of the initial one-hot responsibilities (categorical.py:30-46)
        self.r = None
        self._stats_from_labels(np.asarray(lab0))
        # priors (initialize_from_prior, expfamily.py:168-184)
        self.alpha = self.alpha0.copy()
        self.logpi = special.digamma(self.alpha) - special.digamma(self.alpha.sum())
        self.mu = np.zeros((K, D))
        self.Cmu = np.tile(np.eye(D) / self.beta0, (K, 1, 1))
        self.nk = np.full(K, self.n0)
        self.Vk = np.tile(self.V0, (K, 1, 1))
        self._lambda_moments()
        self.sum_lse = -np.inf     # z has delta moments: g = inf
        self.sum_rphi = 0.0
        self.L, self.L_terms = [], []

    # -- statistics ---------------------------------------------------------------
    def _stats_from_labels(self, lab):
        K, D = self.K, self.D
        self.R = np.bincount(lab, minlength=K).astype(np.float64)
        self.S1 = np.zeros((K, D))
        self.S2 = np.zeros((K, D, D))
        for s in range(0, self.N, self.chunk):
            e = min(self.N, s + self.chunk)
            oh = np.zeros((e - s, K))
            oh[np.arange(e - s), lab[s:e]] = 1
            yc = self.y[s:e]
            self.S1 += oh.T @ yc
            self.S2 += np.einsum('nk,ni,nj->kij', oh, yc, yc)

    def _lambda_moments(self):
        D = self.D
        self.Lam = self.nk[:, None, None] * np.linalg.inv(self.Vk)                # wishart.py:184
        self.logdetV = np.linalg.slogdet(self.Vk)[1]
        self.logdetLam = multidigamma(0.5 * self.nk, D) + D * np.log(2.0) - self.logdetV

    # -- node updates ----------------------------------------------------------------
    def update_mu(self):
        D = self.D
        Lmu = self.beta0 * np.eye(D) + self.R[:, None, None] * self.Lam
        self.Cmu = np.linalg.inv(Lmu)
        self.logdet_Lmu = np.linalg.slogdet(Lmu)[1]
        self.mu = np.einsum('kij,kj->ki', self.Cmu, np.einsum('kij,kj->ki', self.Lam, self.S1))

    def _mumu(self):
        return self.Cmu + self.mu[:, :, None] * self.mu[:, None, :]

    def update_Lambda(self):
        mm = self._mumu()
        sm = self.S1[:, :, None] * self.mu[:, None, :]
        self.nk = self.n0 + self.R
        self.Vk = self.V0 + self.S2 - sm - np.swapaxes(sm, 1, 2) + self.R[:, None, None] * mm
        self._lambda_moments()

    def _coefficients(self):
        """ell_nk = c_k + b_k . y_n - 1/2 y_n^T Lam_k y_n."""
        D = self.D
        mm = self._mumu()
        b = np.einsum('kij,kj->ki', self.Lam, self.mu)
        c = 0.5 * self.logdetLam - 0.5 * D * LOG2PI - 0.5 * np.einsum('kij,kij->k', self.Lam, mm)
        return c, b

    def update_z(self, keep_r=True):
        K, D = self.K, self.D
        c, b = self._coefficients()
        R = np.zeros(K)
        S1 = np.zeros((K, D))
        S2 = np.zeros((K, D, D))
        lse_sum, rphi = 0.0, 0.0
        if keep_r:
            self.r = np.empty((self.N, K))
        for s in range(0, self.N, self.chunk):
            e = min(self.N, s + self.chunk)
            yc = self.y[s:e]
            # y^T Lam_k y as one GEMM over the D^2 products y_i y_j (BLAS; the three-operand
            # einsum of the same contraction runs a scalar loop)
            yy = (yc[:, :, None] * yc[:, None, :]).reshape(e - s, D * D)
            phi = (self.logpi + c)[None, :] + yc @ b.T - 0.5 * (yy @ self.Lam.reshape(K, D * D).T)
            m = phi.max(axis=1, keepdims=True)
            lse = np.log(np.exp(phi - m).sum(axis=1, keepdims=True)) + m
            p = np.exp(phi - lse)
            p /= p.sum(axis=1, keepdims=True)              # utils/misc.py:1399
            R += p.sum(axis=0)
            S1 += p.T @ yc
            S2 += (p.T @ yy).reshape(K, D, D)
            lse_sum += float(lse.sum())
            rphi += float((p * phi).sum())
            if keep_r:
                self.r[s:e] = p
        self.R, self.S1, self.S2 = R, S1, S2
        self.sum_lse, self.sum_rphi = lse_sum, rphi

    def update_alpha(self):
        self.alpha = self.alpha0 + self.R
        self.logpi = special.digamma(self.alpha) - special.digamma(self.alpha.sum())

    # -- lower bound -------------------------------------------------------------------
    def lower_bound(self):
        K, D = self.K, self.D
        c, b = self._coefficients()
        L_Y = float(np.sum(self.R * c) + np.sum(b * self.S1)
                    - 0.5 * np.einsum('kij,kij->', self.Lam, self.S2))
        L_z = float(self.sum_lse - self.sum_rphi + np.sum(self.R * self.logpi))
        a0, a = self.alpha0, self.alpha
        L_pi = float(special.gammaln(a0.sum()) - special.gammaln(a0).sum()
                     - special.gammaln(a.sum()) + special.gammaln(a).sum()
                     + np.sum((a0 - a) * self.logpi))
        mm = self._mumu()
        L_mu = float(np.sum(-0.5 * self.beta0 * np.einsum('kii->k', mm) + 0.5 * D * np.log(self.beta0)
                            - 0.5 * self.logdet_Lmu + 0.5 * D))
        logdetV0 = np.linalg.slogdet(self.V0)[1]
        g_p = 0.5 * self.n0 * logdetV0 - 0.5 * D * self.n0 * np.log(2.0) - multigammaln(0.5 * self.n0, D)
        g_q = 0.5 * self.nk * self.logdetV - 0.5 * D * self.nk * np.log(2.0) \
            - multigammaln(0.5 * self.nk, D)
        t = -0.5 * np.einsum('ij,kij->k', self.V0, self.Lam) + 0.5 * self.n0 * self.logdetLam \
            + 0.5 * np.einsum('kij,kij->k', self.Vk, self.Lam) - 0.5 * self.nk * self.logdetLam
        L_Lam = float(np.sum(g_p - g_q + t))
        terms = dict(Y=L_Y, z=L_z, alpha=L_pi, mu=L_mu, Lambda=L_Lam)
        return L_Y + L_z + L_pi + L_mu + L_Lam, terms

    def iterate(self, n=1, keep_r=True):
        for _ in range(n):
            self.update_mu()
            self.update_Lambda()
            self.update_z(keep_r=keep_r)
            self.update_alpha()
            L, t = self.lower_bound()
            self.L.append(L)
            self.L_terms.append(t)
        return self.L[-1]


def make_gmm_data(N, D, K, seed=42):
    """centers = 3 N(0,1), y = centers[lab] + 0.5 eps, random initial labels."""
    rs = np.random.RandomState(seed)
    centers = 3 * rs.normal(size=(K, D))
    lab = rs.randint(K, size=N)
    y = centers[lab] + 0.5 * rs.normal(size=(N, D))
    lab0 = rs.randint(K, size=N)
    return y, lab0
