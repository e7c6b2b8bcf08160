"""
CPU oracle for the PCA / factor-analysis VB iteration WITH MISSING VALUES.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: PINNED against the live
reference through tests/golden/masked_pca.npz (oracle/make_golden.py masked_pca_case; checked
by tests/test_oracle_golden.py).

Model: bayespy/demos/pca.py:22-61 with ``Y.observe(y, mask=mask)`` (demos/pca.py:80-82), node
order W, X, tau, alpha, then the lower bound (vmp.py:154-172, :693-764).  With an array mask
every plate has its own posterior covariance:

* W (gaussian.py:649-706, messages dot.py:425-633 masked by node.py:457-526, :650):
  ``Lam_d = diag<alpha> + <tau> sum_n m_dn <x x^T>_n``, ``w_d = Lam_d^-1 <tau> sum_n m_dn y_dn <x_n>``;
* X: ``Lam_n = c I + <tau> sum_d m_dn <w w^T>_d``, ``x_n = Lam_n^-1 <tau> sum_d m_dn y_dn <w_d>``;
* tau (gamma.py:116-148, message gaussian.py:2352-2371): ``a = a0 + 1/2 sum m``,
  ``b = b0 + 1/2 sum_dn m_dn (y^2 - 2 y <w_d>.<x_n> + tr(<ww>_d <xx>_n))``;
* alpha: ``a = a0 + D'/2``, ``b_k = b0 + 1/2 sum_d' <ww>_d[k,k]`` over the D' rows of W with at least
  one observation (rows without any are ignored plates of W: node.py:457-526, :624-650);
* bound terms: expfamily.py:400-480 per node, masked sums.

Sufficient-statistics form, chunked over the plate N: the only plate-sized state is <x_n>
(N, K); the (N, K, K) second moments live one chunk at a time.  ``M_d = sum_n m_dn <xx>_n``
(D, K, K) and ``r_d = sum_n m_dn y_dn <x_n>`` (D, K) are what W, tau and the bound consume --
exactly what the fused HIP block accumulates (bayespy_amd/csrc/vmp_mpca.hip).
Everything is IEEE float64; ``y`` is (D, N) with arbitrary (NaN) values where ``mask`` is False.
"""
import numpy as np

from .pca import gamma_moments, gamma_elbo

LOG2PI = np.log(2 * np.pi)


def _inv_logdet_block(M):
    # log-determinant from the Cholesky factor (as the reference, utils/linalg.py:209-223); the
    # inverse through LAPACK's general solver, which NumPy batches in C (the Cholesky-based
    # batched form costs 3x the time in NumPy's per-matrix matmul/solve dispatch) -- both agree
    # with cho_solve to rounding, which the golden traces check
    L = np.linalg.cholesky(M)
    logdet = 2.0 * np.sum(np.log(np.diagonal(L, axis1=-2, axis2=-1)), axis=-1)
    return np.linalg.inv(M), logdet


def batched_spd_inv_logdet(M, threads=None):
    """Inverses and log-determinants of a stack of SPD matrices (utils/linalg.py:31-63,
    :174-223; the reference loops over the stack in Python).  Large stacks are split over a
    thread pool (NumPy's linalg kernels release the GIL) with the BLAS library held to one
    thread per worker; ``threads`` defaults to the host's cores."""
    n = M.shape[0] if M.ndim == 3 else 0
    if n < 4096:
        return _inv_logdet_block(M)
    import os
    from concurrent.futures import ThreadPoolExecutor
    nt = int(threads or os.cpu_count() or 1)
    bounds = np.linspace(0, n, min(4 * nt, n // 1024) + 1).astype(np.int64)
    inv = np.empty_like(M)
    logdet = np.empty(n)

    def work(i):
        s, e = bounds[i], bounds[i + 1]
        inv[s:e], logdet[s:e] = _inv_logdet_block(M[s:e])
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except ImportError:        # noqa: BLE001
        limit = None
    try:
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(work, range(len(bounds) - 1)))
    finally:
        if limit is not None:
            limit.restore_original_limits()
    return inv, logdet


class MaskedPCAOracle:

    def __init__(self, y, mask, x0, a0=1e-2, b0=1e-2, x_prec=1.0, chunk=1 << 14):
        self.mask = np.ascontiguousarray(np.broadcast_to(np.asarray(mask, dtype=bool), np.shape(y)))
        self.m = self.mask.astype(np.float64)
        self.y = np.where(self.mask, np.asarray(y, dtype=np.float64), 0.0)   # m * y, NaN-safe
        self.D, self.N = self.y.shape
        self.K = x0.shape[1]
        self.a0, self.b0, self.c = float(a0), float(b0), float(x_prec)
        self.chunk = int(chunk)
        D, K = self.D, self.K
        self.nobs = float(self.m.sum())
        # rows of W that no observation reaches are "ignored" plates of W (mask propagation,
        # node.py:457-526): still updated (to their prior given <alpha>), but left out of the message
        # to alpha (node.py:624-650) and of W's bound term (expfamily.py:470-474)
        self.wm = self.mask.any(axis=1)
        self.D_eff = float(self.wm.sum())
        self.Syy = float(np.sum(self.y * self.y))
        self.tau_a, self.tau_b = self.a0, self.b0
        self.alpha_a = np.full(K, self.a0)
        self.alpha_b = np.full(K, self.b0)
        # X: delta moments of the injected value (expfamily.py:193-204)
        self.X = np.array(x0, dtype=np.float64)
        self.M = np.zeros((D, K, K))
        self.r = np.zeros((D, K))
        self.trXX = 0.0
        for s in range(0, self.N, self.chunk):
            e = min(self.N, s + self.chunk)
            xc = self.X[s:e]
            xx = xc[:, :, None] * xc[:, None, :]
            self.M += (self.m[:, s:e] @ xx.reshape(e - s, K * K)).reshape(D, K, K)
            self.r += self.y[:, s:e] @ xc
            self.trXX += float(np.einsum('nkk->', xx))
        self.logdetCX = None
        # W: prior moments (initialize_from_prior): mean 0, covariance diag(1 / <alpha>)
        self.W = np.zeros((D, K))
        self.WW = np.broadcast_to(np.diag(self.alpha_b / self.alpha_a), (D, K, K)).copy()
        self.logdetCW = None
        self.L, self.L_terms = [], []

    # -- node updates --------------------------------------------------------------------------
    def update_W(self):
        tau, _ = gamma_moments(self.tau_a, self.tau_b)
        alpha, _ = gamma_moments(self.alpha_a, self.alpha_b)
        Lam = np.diag(alpha)[None] + tau * self.M
        CW, logdetLam = batched_spd_inv_logdet(Lam)
        self.W = np.einsum('dij,dj->di', CW, tau * self.r)
        self.WW = CW + self.W[:, :, None] * self.W[:, None, :]
        self.logdetCW = -logdetLam                                  # (D,)

    def update_X(self):
        D, K = self.D, self.K
        tau, _ = gamma_moments(self.tau_a, self.tau_b)
        M = np.zeros((D, K, K))
        r = np.zeros((D, K))
        tr, ld = 0.0, 0.0
        WWf = self.WW.reshape(D, K * K)
        for s in range(0, self.N, self.chunk):
            e = min(self.N, s + self.chunk)
            mc, yc = self.m[:, s:e], self.y[:, s:e]
            Lam = self.c * np.eye(K)[None] + tau * (mc.T @ WWf).reshape(e - s, K, K)
            CX, logdetLam = batched_spd_inv_logdet(Lam)
            xc = np.einsum('nij,nj->ni', CX, tau * (yc.T @ self.W))
            xx = CX + xc[:, :, None] * xc[:, None, :]
            self.X[s:e] = xc
            M += (mc @ xx.reshape(e - s, K * K)).reshape(D, K, K)
            r += yc @ xc
            tr += float(np.einsum('nkk->', xx))
            ld -= float(np.sum(logdetLam))
        self.M, self.r, self.trXX, self.logdetCX = M, r, tr, ld

    def _residual(self):
        """sum_dn m_dn <(y - f)^2> from the statistics (dot.py:355,403 masked)."""
        Syf = float(np.sum(self.W * self.r))
        Sff = float(np.sum(self.WW * self.M))
        return self.Syy - 2.0 * Syf + Sff

    def update_tau(self):
        self.tau_a = self.a0 + 0.5 * self.nobs
        self.tau_b = self.b0 + 0.5 * self._residual()

    def update_alpha(self):
        self.alpha_a = np.full(self.K, self.a0 + 0.5 * self.D_eff)
        self.alpha_b = self.b0 + 0.5 * np.einsum('dkk->k', self.WW[self.wm])

    # -- lower bound ----------------------------------------------------------------------------
    def lower_bound(self):
        D, N, K = self.D, self.N, self.K
        tau, logtau = gamma_moments(self.tau_a, self.tau_b)
        alpha, logalpha = gamma_moments(self.alpha_a, self.alpha_b)
        L_Y = self.nobs * (-0.5 * LOG2PI + 0.5 * logtau) - 0.5 * tau * self._residual()
        L_X = (-0.5 * self.c * self.trXX + 0.5 * self.logdetCX
               + N * (0.5 * K * np.log(self.c) + 0.5 * K))
        De = self.D_eff
        L_W = (0.5 * De * np.sum(logalpha)
               - 0.5 * np.sum(alpha * np.einsum('dkk->k', self.WW[self.wm]))
               + 0.5 * float(np.sum(self.logdetCW[self.wm])) + 0.5 * De * K)
        L_tau = gamma_elbo(self.a0, self.b0, self.tau_a, self.tau_b)
        L_alpha = gamma_elbo(self.a0, self.b0, self.alpha_a, self.alpha_b)
        terms = dict(Y=float(L_Y), X=float(L_X), W=float(L_W), tau=float(L_tau),
                     alpha=float(L_alpha))
        return float(L_Y + L_X + L_W + L_tau + L_alpha), terms

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

    def predictive_Y(self):
        """q of the missing entries of the partially observed leaf Y: <f>, <f^2> + 1/<tau>
        at the masked-out positions (the reference updates them, stochastic.py:276-282)."""
        tau, _ = gamma_moments(self.tau_a, self.tau_b)
        f = self.W @ self.X.T
        return f, tau
