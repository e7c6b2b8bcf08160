"""
CPU oracle for the linear state-space model VB iteration with a plate of sequences.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity: PINNED against the live reference
through tests/golden/lssm.npz (oracle/make_golden.py lssm_cases; tests/test_oracle_golden.py).

Model (bayespy/demos/lssm.py:34-103 with a sequence plate B; BASELINE.json config 5)::

    alpha_j ~ Gamma(a0, b0)                       A[i, :] ~ N(0, diag(alpha)^-1)      plates (D,)
    x_b0 ~ N(mu0, Lam0^-1),  x_bt ~ N(A x_b,t-1, diag(nu)^-1)     GaussianMarkovChain, plates (B,), n=T
    gamma_j ~ Gamma,  c_m ~ N(0, diag(gamma)^-1) plates (M,1,1);  tau ~ Gamma
    y_mbt ~ N(c_m . x_bt, 1/tau)                  observed, scalar mask;  nu fixed or ~ Gamma plates (D,)

node order C, gamma, X, A, alpha, tau[, nu], then the lower bound (vmp.py:154-172, :693-764).

Sufficient-statistics form.  With dynamics, noise and mask shared by all sequences the block
tridiagonal precision of q(X_b) is THE SAME for every b (gaussian_markov_chain.py:89-123,
utils/linalg.py:468-575: SURVEY.md 8f.2), so the covariance recursion runs once over T and only
the means are per-sequence.  Everything the other nodes and the bound read are plate sums:

    Sxx  = sum_bt <x_t x_t^T>          Spp = sum_b sum_{t<T-1} <x_t x_t^T>     Snn = sum_b sum_{t>=1} <x_t x_t^T>
    Snp  = sum_b sum_{t>=1} <x_t x_{t-1}^T>     Syx[m] = sum_bt y_mbt <x_bt>   S00 = sum_b <x_0 x_0^T>, s0 = sum_b <x_0>

Restates GaussianMarkovChainDistribution (gaussian_markov_chain.py:270-707), the chain -> Gaussian
view (:1988-2098), SumMultiply messages (dot.py:425-633), GaussianARD (gaussian.py:576-741) with
the Gamma wrapper (:2299-2371), Gamma (gamma.py:90-211), the bound (expfamily.py:400-480).
"""
import numpy as np

from .pca import gamma_moments, gamma_elbo, spd_inv_logdet

LOG2PI = np.log(2 * np.pi)


def chain_covariances(Dg, E):
    """Block tridiagonal SPD matrix with diagonal blocks Dg[t] (T,D,D) and super-diagonal blocks
    E[t] = Phi[t, t+1] (T-1,D,D): block LDL^T forward, inverse blocks backward
    (linalg.block_banded_solve, utils/linalg.py:468-575).  Returns S^-1 (T,D,D), J (T-1,D,D) with
    x_t = S_t^-1 z_t - J_t x_{t+1}, G (T-1,D,D) with z_{t+1} = h_{t+1} - G_t z_t, the diagonal
    blocks V_t of the inverse, the blocks Cov(x_t, x_{t+1}) and log|Phi|."""
    T, D = Dg.shape[0], Dg.shape[1]
    Sinv = np.empty((T, D, D))
    G = np.empty((max(T - 1, 0), D, D))
    J = np.empty((max(T - 1, 0), D, D))
    logdet = 0.0
    S = Dg[0]
    for t in range(T):
        Sinv[t], ld = spd_inv_logdet(S)
        logdet += ld
        if t < T - 1:
            J[t] = Sinv[t] @ E[t]
            G[t] = J[t].T
            S = Dg[t + 1] - E[t].T @ J[t]
    V = np.empty((T, D, D))
    Cn = np.empty((max(T - 1, 0), D, D))
    V[T - 1] = Sinv[T - 1]
    for t in range(T - 2, -1, -1):
        Cn[t] = -J[t] @ V[t + 1]                       # Cov(x_t, x_{t+1})
        V[t] = Sinv[t] - Cn[t] @ J[t].T
    return Sinv, J, G, V, Cn, logdet


class LSSMOracle:

    def __init__(self, y, x0, c0, mu0=None, Lam0=None, nu=None, nu_prior=None, a_init=None,
                 gamma_init=1e-2, tau_init=1e2, prior=(1e-5, 1e-5)):
        """y (M,B,T); x0 (B,T,D); c0 (M,D).  nu: fixed innovation precisions (D,) or None with
        nu_prior=(a0,b0) for a Gamma node."""
        self.y = np.asarray(y, dtype=np.float64)
        self.M, self.B, self.T = self.y.shape
        self.D = x0.shape[-1]
        D = self.D
        self.mu0 = np.zeros(D) if mu0 is None else np.asarray(mu0, dtype=np.float64)
        self.Lam0 = 1e-3 * np.eye(D) if Lam0 is None else np.asarray(Lam0, dtype=np.float64)
        self.a0, self.b0 = prior
        self.nu_prior = nu_prior
        if nu_prior is None:
            self.nu = np.ones(D) if nu is None else np.asarray(nu, dtype=np.float64)
            self.lognu = np.log(self.nu)
        else:
            self.nu_a = np.full(D, nu_prior[0])
            self.nu_b = np.full(D, nu_prior[1])
            self.nu, self.lognu = gamma_moments(self.nu_a, self.nu_b)
        # initialize_from_value: delta moments (expfamily.py:193-204)
        self.Am = np.eye(D) if a_init is None else np.asarray(a_init, dtype=np.float64)
        self.AA = self.Am[:, :, None] * self.Am[:, None, :]          # <a_i a_i^T> (D,D,D)
        self.logdetCA = None
        self.Cm = np.asarray(c0, dtype=np.float64).reshape(self.M, D)
        self.CC = None
        self.gamma, self.loggamma = np.full(D, gamma_init), np.full(D, np.log(gamma_init))
        self.gamma_a = self.gamma_b = None
        self.tau, self.logtau = tau_init, np.log(tau_init)
        self.tau_a = self.tau_b = None
        self.alpha_a, self.alpha_b = np.full(D, self.a0), np.full(D, self.b0)
        self.alpha, self.logalpha = gamma_moments(self.alpha_a, self.alpha_b)
        self.Syy = float(np.sum(self.y * self.y))
        self.X = np.array(x0, dtype=np.float64)
        self._stats(np.zeros((self.T, D, D)), np.zeros((max(self.T - 1, 0), D, D)))
        self.logdetPhi = None
        self.L, self.L_terms = [], []

    def _stats(self, V, Cn):
        x, B = self.X, self.B
        xx = np.einsum('bti,btj->ij', x, x, optimize=True)
        self.Sxx = B * V.sum(axis=0) + xx
        self.Spp = B * V[:-1].sum(axis=0) + np.einsum('bti,btj->ij', x[:, :-1], x[:, :-1], optimize=True)
        self.Snn = B * V[1:].sum(axis=0) + np.einsum('bti,btj->ij', x[:, 1:], x[:, 1:], optimize=True)
        # <x_t x_{t-1}^T> = Cov(x_{t-1}, x_t)^T + mean outer product
        self.Snp = B * Cn.sum(axis=0).T + np.einsum('bti,btj->ij', x[:, 1:], x[:, :-1], optimize=True)
        self.Syx = np.einsum('mbt,btd->md', self.y, x, optimize=True)
        self.S00 = B * V[0] + x[:, 0].T @ x[:, 0]
        self.s0 = x[:, 0].sum(axis=0)

    # -- node updates ---------------------------------------------------------------------------
    def update_C(self):
        Lam = np.diag(self.gamma) + self.tau * self.Sxx
        self.CovC, ld = spd_inv_logdet(Lam)
        self.logdetCC = -ld
        self.Cm = self.tau * self.Syx @ self.CovC
        self.CC = self.M * self.CovC + self.Cm.T @ self.Cm          # sum_m <c_m c_m^T>

    def _sum_cc(self):
        return self.Cm.T @ self.Cm if self.CC is None else self.CC

    def update_gamma(self):
        self.gamma_a = np.full(self.D, self.a0 + 0.5 * self.M)
        self.gamma_b = self.b0 + 0.5 * np.diag(self._sum_cc())
        self.gamma, self.loggamma = gamma_moments(self.gamma_a, self.gamma_b)

    def update_X(self):
        T, D = self.T, self.D
        AnuA = np.einsum('i,ijk->jk', self.nu, self.AA)
        obs = self.tau * self._sum_cc()
        Dg = np.empty((T, D, D))
        for t in range(T):
            Dg[t] = obs + (self.Lam0 if t == 0 else np.diag(self.nu)) + (AnuA if t < T - 1 else 0.0)
        E = np.broadcast_to(-(self.nu[:, None] * self.Am).T, (max(T - 1, 0), D, D))
        Sinv, J, G, V, Cn, self.logdetPhi = chain_covariances(Dg, E)
        # time-major work arrays (T, B, D): each step of the two recursions then touches one
        # contiguous (B, D) slab instead of B rows 8*T*D bytes apart
        h = self.tau * np.einsum('mbt,md->tbd', self.y, self.Cm, optimize=True)
        h = np.ascontiguousarray(h)
        h[0] += self.Lam0 @ self.mu0
        z = np.empty_like(h)
        z[0] = h[0]
        for t in range(1, T):
            z[t] = h[t] - z[t - 1] @ G[t - 1].T
        xt = np.empty_like(h)
        xt[T - 1] = z[T - 1] @ Sinv[T - 1].T
        for t in range(T - 2, -1, -1):
            xt[t] = z[t] @ Sinv[t].T - xt[t + 1] @ J[t].T
        x = np.ascontiguousarray(xt.transpose(1, 0, 2))
        self.X, self.V, self.Cn = x, V, Cn
        self._stats(V, Cn)

    def update_A(self):
        D = self.D
        self.logdetCA = np.empty(D)
        for i in range(D):
            Lam = np.diag(self.alpha) + self.nu[i] * self.Spp
            Cov, ld = spd_inv_logdet(Lam)
            self.logdetCA[i] = -ld
            self.Am[i] = Cov @ (self.nu[i] * self.Snp[i])
            self.AA[i] = Cov + np.outer(self.Am[i], self.Am[i])

    def update_alpha(self):
        self.alpha_a = np.full(self.D, self.a0 + 0.5 * self.D)
        self.alpha_b = self.b0 + 0.5 * np.einsum('ijj->j', self.AA)
        self.alpha, self.logalpha = gamma_moments(self.alpha_a, self.alpha_b)

    def _residual(self):
        return self.Syy - 2.0 * float(np.sum(self.Cm * self.Syx)) + float(np.sum(self._sum_cc() * self.Sxx))

    def update_tau(self):
        self.tau_a = self.a0 + 0.5 * self.M * self.B * self.T
        self.tau_b = self.b0 + 0.5 * self._residual()
        self.tau, self.logtau = gamma_moments(self.tau_a, self.tau_b)

    def _innovation(self):
        """sum_b sum_{t>=1} <(x_t,i - a_i . x_t-1)^2> per component i."""
        return (np.diag(self.Snn) - 2.0 * np.sum(self.Am * self.Snp, axis=1)
                + np.einsum('ijk,jk->i', self.AA, self.Spp))

    def update_nu(self):
        self.nu_a = np.full(self.D, self.nu_prior[0] + 0.5 * self.B * (self.T - 1))
        self.nu_b = self.nu_prior[1] + 0.5 * self._innovation()
        self.nu, self.lognu = gamma_moments(self.nu_a, self.nu_b)

    # -- lower bound ----------------------------------------------------------------------------
    def lower_bound(self):
        M, B, T, D = self.M, self.B, self.T, self.D
        L_Y = M * B * T * (-0.5 * LOG2PI + 0.5 * self.logtau) - 0.5 * self.tau * self._residual()
        cc = self._sum_cc()
        L_C = (0.5 * M * np.sum(self.loggamma) - 0.5 * np.sum(self.gamma * np.diag(cc))
               + M * (0.5 * self.logdetCC + 0.5 * D))
        L_A = (0.5 * D * np.sum(self.logalpha) - 0.5 * np.sum(self.alpha * np.einsum('ijj->j', self.AA))
               + 0.5 * np.sum(self.logdetCA) + 0.5 * D * D)
        x0dev = (self.S00 - np.outer(self.s0, self.mu0) - np.outer(self.mu0, self.s0)
                 + B * np.outer(self.mu0, self.mu0))
        L_X = (B * (0.5 * T * D + 0.5 * np.linalg.slogdet(self.Lam0)[1]
                    + 0.5 * (T - 1) * np.sum(self.lognu) - 0.5 * self.logdetPhi)
               - 0.5 * np.sum(self.Lam0 * x0dev) - 0.5 * np.sum(self.nu * self._innovation()))
        terms = dict(Y=float(L_Y), C=float(L_C), A=float(L_A), X=float(L_X),
                     gamma=gamma_elbo(self.a0, self.b0, self.gamma_a, self.gamma_b),
                     alpha=gamma_elbo(self.a0, self.b0, self.alpha_a, self.alpha_b),
                     tau=gamma_elbo(self.a0, self.b0, self.tau_a, self.tau_b))
        if self.nu_prior is not None:
            terms['nu'] = gamma_elbo(self.nu_prior[0], self.nu_prior[1], self.nu_a, self.nu_b)
        return float(sum(terms.values())), terms

    def iterate(self, n=1):
        for _ in range(n):
            self.update_C()
            self.update_gamma()
            self.update_X()
            self.update_A()
            self.update_alpha()
            self.update_tau()
            if self.nu_prior is not None:
                self.update_nu()
            L, terms = self.lower_bound()
            self.L.append(L)
            self.L_terms.append(terms)
        return self.L[-1]


def _batched_spd_inv_logdet(S):
    """(..., D, D) SPD matrices: inverses and log-determinants via Cholesky."""
    L = np.linalg.cholesky(S)
    Linv = np.linalg.solve(L, np.broadcast_to(np.eye(S.shape[-1]), S.shape))
    inv = np.swapaxes(Linv, -1, -2) @ Linv
    ld = 2.0 * np.sum(np.log(np.einsum('...ii->...i', L)), axis=-1)
    return inv, ld


class MaskedLSSMOracle(LSSMOracle):
    """The same model observed through an ARRAY mask (``Y.observe(y, mask=mask)``,
    bayespy/demos/lssm.py:132, :239-246), mask broadcastable to (M, B, T).  Pinned on the live
    reference through tests/golden/lssm_masked.npz (oracle/make_golden.py lssm_masked_cases).

    What changes against the fully observed block (node.py:457-526, :570-655: a message is
    multiplied by the child's mask before the plate sum; expfamily.py:343-366: every latent plate
    is updated, also an ignored one; expfamily.py:470-480: the bound skips ignored plates):

    * every sequence has its OWN block-tridiagonal precision: diagonal blocks
      prior + <tau> sum_m mask_mbt <c_m c_m^T>  (gaussian_markov_chain.py:542-627 with the message of
      dot.py:425-633), hence its own covariance recursion (utils/linalg.py:468-575) and log|Phi_b|;
    * every row of C has its own posterior: Lam_m = diag<gamma> + <tau> sum_bt mask_mbt <x_bt x_bt^T>;
    * plates without any observation are IGNORED plates: a row m of C never observed is updated
      to its prior-only posterior but sends nothing to gamma and adds nothing to the bound; a
      sequence b never observed likewise for A / nu / the bound of X (mask of the parent = any()
      over the child's plates, node.py:486-526).
    Values at masked entries are never read (NaN placeholders are fine)."""

    def __init__(self, y, mask, x0, c0, **kw):
        y = np.asarray(y, dtype=np.float64)
        self.mask = np.ascontiguousarray(np.broadcast_to(np.asarray(mask, dtype=bool), y.shape))
        y = np.where(self.mask, y, 0.0)
        self.m_obs = self.mask.any(axis=(1, 2))            # rows of C that see data
        self.b_obs = self.mask.any(axis=(0, 2))            # sequences that see data
        self.n_obs = float(self.mask.sum())
        super().__init__(y, x0, c0, **kw)
        self.CovCm = None                                   # (M, D, D) after the first C.update()
        self.logdetCCm = None

    # -- plate sums ---------------------------------------------------------------------------------
    def _stats(self, V, Cn):
        """V (B,T,D,D) / Cn (B,T-1,D,D) per sequence (zeros for delta moments)."""
        x, B, T, D = self.X, self.B, self.T, self.D
        if V.ndim == 3:
            V = np.broadcast_to(V, (B,) + V.shape)
            Cn = np.broadcast_to(Cn, (B,) + Cn.shape)
        P = V + x[:, :, :, None] * x[:, :, None, :]                         # <x_bt x_bt^T>
        w = self.b_obs.astype(np.float64)
        self.Beff = float(w.sum())
        self.Spp = np.einsum('b,btij->ij', w, P[:, :-1])
        self.Snn = np.einsum('b,btij->ij', w, P[:, 1:])
        # <x_t x_{t-1}^T> = Cov(x_{t-1}, x_t)^T + mean outer product
        self.Snp = np.einsum('b,btji->ij', w, Cn) + np.einsum('b,bti,btj->ij', w, x[:, 1:], x[:, :-1])
        self.S00 = np.einsum('b,bij->ij', w, P[:, 0])
        self.s0 = w @ x[:, 0]
        mk = self.mask.astype(np.float64)
        self.XXm = np.einsum('mbt,btij->mij', mk, P)                        # per row of C
        self.Syx = np.einsum('mbt,btd->md', self.y, x, optimize=True)       # y is zero where masked
        self.P = P

    # -- node updates -------------------------------------------------------------------------------
    def _ccm(self):
        """<c_m c_m^T> per row (M, D, D)."""
        cc = self.Cm[:, :, None] * self.Cm[:, None, :]
        return cc if self.CovCm is None else cc + self.CovCm

    def _sum_cc(self):
        return np.einsum('m,mij->ij', self.m_obs.astype(np.float64), self._ccm())

    def update_C(self):
        Lam = np.diag(self.gamma)[None] + self.tau * self.XXm
        self.CovCm, ld = _batched_spd_inv_logdet(Lam)
        self.logdetCCm = -ld
        self.Cm = self.tau * np.einsum('mij,mj->mi', self.CovCm, self.Syx)

    def update_gamma(self):
        self.gamma_a = np.full(self.D, self.a0 + 0.5 * self.m_obs.sum())
        self.gamma_b = self.b0 + 0.5 * np.diag(self._sum_cc())
        self.gamma, self.loggamma = gamma_moments(self.gamma_a, self.gamma_b)

    def update_X(self):
        T, D, B = self.T, self.D, self.B
        AnuA = np.einsum('i,ijk->jk', self.nu, self.AA)
        base = np.empty((T, D, D))
        for t in range(T):
            base[t] = (self.Lam0 if t == 0 else np.diag(self.nu)) + (AnuA if t < T - 1 else 0.0)
        obs = self.tau * np.einsum('mbt,mij->btij', self.mask.astype(np.float64), self._ccm())
        Dg = base[None] + obs                                                # (B, T, D, D)
        E = -(self.nu[:, None] * self.Am).T                                 # Phi[t, t+1]
        h = self.tau * np.einsum('mbt,md->btd', self.y, self.Cm, optimize=True)
        h[:, 0] += self.Lam0 @ self.mu0
        Sinv = np.empty((B, T, D, D))
        z = np.empty((B, T, D))
        logdet = np.zeros(B)
        S = Dg[:, 0]
        for t in range(T):
            Sinv[:, t], ld = _batched_spd_inv_logdet(S)
            logdet += ld
            if t == 0:
                z[:, 0] = h[:, 0]
            if t < T - 1:
                J = Sinv[:, t] @ E                                           # (B, D, D)
                S = Dg[:, t + 1] - E.T @ J
                z[:, t + 1] = h[:, t + 1] - np.einsum('bji,bj->bi', J, z[:, t])
        x = np.empty((B, T, D))
        V = np.empty((B, T, D, D))
        Cn = np.empty((B, max(T - 1, 0), D, D))
        x[:, T - 1] = np.einsum('bij,bj->bi', Sinv[:, T - 1], z[:, T - 1])
        V[:, T - 1] = Sinv[:, T - 1]
        for t in range(T - 2, -1, -1):
            J = Sinv[:, t] @ E
            x[:, t] = np.einsum('bij,bj->bi', Sinv[:, t], z[:, t]) - np.einsum('bij,bj->bi', J, x[:, t + 1])
            Cn[:, t] = -J @ V[:, t + 1]
            V[:, t] = Sinv[:, t] - Cn[:, t] @ np.swapaxes(J, -1, -2)
        self.X, self.V, self.Cn = x, V, Cn
        self.logdetPhi_b = logdet
        self.logdetPhi = float(np.sum(logdet[self.b_obs]))
        self._stats(V, Cn)

    def _residual(self):
        return (self.Syy - 2.0 * float(np.sum(self.Cm * self.Syx))
                + float(np.sum(self._ccm() * self.XXm)))

    def update_tau(self):
        self.tau_a = self.a0 + 0.5 * self.n_obs
        self.tau_b = self.b0 + 0.5 * self._residual()
        self.tau, self.logtau = gamma_moments(self.tau_a, self.tau_b)

    def update_nu(self):
        self.nu_a = np.full(self.D, self.nu_prior[0] + 0.5 * self.Beff * (self.T - 1))
        self.nu_b = self.nu_prior[1] + 0.5 * self._innovation()
        self.nu, self.lognu = gamma_moments(self.nu_a, self.nu_b)

    def lower_bound(self):
        T, D = self.T, self.D
        Mo, Bo = float(self.m_obs.sum()), self.Beff
        L_Y = self.n_obs * (-0.5 * LOG2PI + 0.5 * self.logtau) - 0.5 * self.tau * self._residual()
        cc = self._sum_cc()
        L_C = (0.5 * Mo * np.sum(self.loggamma) - 0.5 * np.sum(self.gamma * np.diag(cc))
               + 0.5 * float(np.sum(self.logdetCCm[self.m_obs])) + 0.5 * Mo * D)
        L_A = (0.5 * D * np.sum(self.logalpha) - 0.5 * np.sum(self.alpha * np.einsum('ijj->j', self.AA))
               + 0.5 * np.sum(self.logdetCA) + 0.5 * D * D)
        x0dev = (self.S00 - np.outer(self.s0, self.mu0) - np.outer(self.mu0, self.s0)
                 + Bo * np.outer(self.mu0, self.mu0))
        L_X = (Bo * (0.5 * T * D + 0.5 * np.linalg.slogdet(self.Lam0)[1]
                     + 0.5 * (T - 1) * np.sum(self.lognu)) - 0.5 * self.logdetPhi
               - 0.5 * np.sum(self.Lam0 * x0dev) - 0.5 * np.sum(self.nu * self._innovation()))
        terms = dict(Y=float(L_Y), C=float(L_C), A=float(L_A), X=float(L_X),
                     gamma=gamma_elbo(self.a0, self.b0, self.gamma_a, self.gamma_b),
                     alpha=gamma_elbo(self.a0, self.b0, self.alpha_a, self.alpha_b),
                     tau=gamma_elbo(self.a0, self.b0, self.tau_a, self.tau_b))
        if self.nu_prior is not None:
            terms['nu'] = gamma_elbo(self.nu_prior[0], self.nu_prior[1], self.nu_a, self.nu_b)
        return float(sum(terms.values())), terms
