"""
TEST DOUBLE for ``bayespy_amd.inference.plans.pca.HIPKernels``.

Implements the same method set on CPU torch tensors with the NumPy oracle
(oracle/pca.py), honouring the device state layout of include/vmp_hip.h
(obtained from the real library's host-only ``vmp_pca_get_layout``).  It lets
the host logic of the plan -- pattern matching, update order, staleness,
sharding + all-reduce -- run in CPU-only tests (world_size-2 gloo included).
It lives under tests/ and is never imported by the product.
"""
import ctypes

import numpy as np
from scipy import special

from bayespy_amd import _lib
from oracle.pca import gamma_elbo, spd_inv_logdet


class CPURuntimeKernels:

    def __init__(self, rt):
        self.rt = rt
        self.lib = _lib.load()
        self.calls = []

    # -- layout ------------------------------------------------------------------
    def layout(self, D, K):
        L = _lib.PCALayout()
        rc = self.lib.vmp_pca_get_layout(D, K, ctypes.byref(L))
        _lib.raise_for_status(rc)
        return L

    def workspace_doubles(self, D, K):
        return 16

    def _v(self, state, D, K):
        L = self.layout(D, K)
        s = state.numpy()
        DP, KP = int(L.DP), int(L.KP)
        v = dict(
            L=L, KP=KP, DP=DP,
            S=s[L.off_S:L.off_S + L.len_S].reshape(DP + KP, KP),
            Syy=s[L.off_Syy:L.off_Syy + 1],
            tau=s[L.off_tau:L.off_tau + 4],
            alpha=s[L.off_alpha:L.off_alpha + 4 * KP].reshape(4, KP),
            W=s[L.off_W:L.off_W + D * KP].reshape(D, KP),
            CW=s[L.off_CW:L.off_CW + KP * KP].reshape(KP, KP),
            Sww=s[L.off_Sww:L.off_Sww + KP * KP].reshape(KP, KP),
            CX=s[L.off_CX:L.off_CX + KP * KP].reshape(KP, KP),
            A=s[L.off_A:L.off_A + KP * DP].reshape(KP, DP),
            G=s[L.off_G:L.off_G + DP * DP].reshape(DP, DP),
            scal=s[L.off_scal:L.off_scal + 8],
            Lt=s[L.off_L:L.off_L + 8],
            mu=s[L.off_mu:L.off_mu + D * KP].reshape(D, KP),
            mstat=s[L.off_mstat:L.off_mstat + 2 * KP].reshape(2, KP),
        )
        return v

    has_mean = False    # set by small_ops: W has a constant non-zero prior mean (state[off_mu])

    def _ww(self, v, K):
        """sum_d <(w_dk - mu_dk)^2>: the diagonal of Sww, centred with the sums update_w keeps."""
        ww = np.diag(v['Sww'][:K, :K]).copy()
        if self.has_mean:
            ww += v['mstat'][1, :K] - 2.0 * v['mstat'][0, :K]
        return ww

    # -- kernels --------------------------------------------------------------------
    def init_state(self, D, K, a0t, b0t, a0a, b0a, state):
        state.zero_()
        v = self._v(state, D, K)
        v['tau'][:] = [a0t, b0t, a0t / b0t, special.digamma(a0t) - np.log(b0t)]
        v['alpha'][0, :K] = a0a
        v['alpha'][1, :K] = b0a
        v['alpha'][2, :K] = a0a / b0a
        v['alpha'][3, :K] = special.digamma(a0a) - np.log(b0a)

    def syy(self, Y, ldy, N, D, K, state, ws):
        y = Y.numpy()[:, :N]
        self._v(state, D, K)['Syy'][0] = float(np.sum(y * y))

    def stats_from_x(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.calls.append('stats_from_x')
        v = self._v(state, D, K)
        y, x = Y.numpy()[:, :N], X.numpy()[:K, :N]
        v['S'][:] = 0
        v['S'][:D, :K] = y @ x.T
        v['S'][v['DP']:v['DP'] + K, :K] = x @ x.T

    def _sxx(self, v, K, n_total):
        return n_total * v['CX'][:K, :K] + v['S'][v['DP']:v['DP'] + K, :K]

    def update_w(self, D, K, n_total, state):
        self.calls.append('update_w')
        v = self._v(state, D, K)
        tau = v['tau'][2]
        Lam = np.diag(v['alpha'][2, :K]) + tau * self._sxx(v, K, n_total)
        C, logdet = spd_inv_logdet(Lam)
        rhs = tau * v['S'][:D, :K]
        if self.has_mean:
            rhs = rhs + v['alpha'][2, :K] * v['mu'][:, :K]
        W = rhs @ C
        v['CW'][:K, :K] = C
        v['W'][:, :K] = W
        v['Sww'][:K, :K] = D * C + W.T @ W
        v['scal'][0] = logdet
        if self.has_mean:
            v['mstat'][0, :K] = np.sum(v['mu'][:, :K] * W, axis=0)
            v['mstat'][1, :K] = np.sum(v['mu'][:, :K] ** 2, axis=0)

    def prepare_x(self, D, K, x_prec, state):
        self.calls.append('prepare_x')
        v = self._v(state, D, K)
        tau = v['tau'][2]
        C, logdet = spd_inv_logdet(x_prec * np.eye(K) + tau * v['Sww'][:K, :K])
        v['CX'][:K, :K] = C
        v['A'][:K, :D] = tau * C @ v['W'][:, :K].T
        v['scal'][1] = logdet

    def pass_(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.calls.append('pass')
        v = self._v(state, D, K)
        y = Y.numpy()[:, :N]
        x = v['A'][:K, :D] @ y
        X.numpy()[:K, :N] = x
        v['S'][:] = 0
        v['S'][:D, :K] = y @ x.T
        v['S'][v['DP']:v['DP'] + K, :K] = x @ x.T

    def gram(self, Y, ldy, N, D, K, state, ws):
        self.calls.append('gram')
        v = self._v(state, D, K)
        y = Y.numpy()[:, :N]
        v['G'][:D, :D] = y @ y.T

    def xpass(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.calls.append('xpass')
        v = self._v(state, D, K)
        A = v['A'][:K, :D]
        X.numpy()[:K, :N] = A @ Y.numpy()[:, :N]
        syx = v['G'][:D, :D] @ A.T
        v['S'][:] = 0
        v['S'][:D, :K] = syx
        v['S'][v['DP']:v['DP'] + K, :K] = A @ syx

    def xjoin(self):
        self.calls.append('xjoin')

    def tile_y(self, Y, ldy, N, D, K):
        """vmp_pca_tile_y: [tile][DP][32], zero padded."""
        self.calls.append('tile_y')
        import torch
        DP = int(self.layout(D, K).DP)
        nt = (N + 31) // 32
        buf = np.zeros((DP, nt * 32))
        buf[:D, :N] = Y.numpy()[:, :N]
        return torch.from_numpy(np.ascontiguousarray(
            buf.reshape(DP, nt, 32).transpose(1, 0, 2)).reshape(-1))

    # set by a test to exercise the tile-major <x> of the plan (PCAPlan.Xd property)
    x_tiles = False

    def tiled_x_doubles(self, D, K, N):
        return int(self.layout(D, K).KP) * ((N + 31) // 32) * 32

    def tile_x(self, to_tiled, X, ldx, N, D, K, Xt):
        """vmp_pca_tile_x: Xt [tile][KP][32] <-> X (KP, ldx) row-major."""
        self.calls.append('tile_x')
        KP = int(self.layout(D, K).KP)
        nt = (N + 31) // 32
        t = Xt.numpy().reshape(nt, KP, 32)
        x = X.numpy()
        if to_tiled:
            t[:] = x[:KP, :nt * 32].reshape(KP, nt, 32).transpose(1, 0, 2)
        else:
            x[:KP, :nt * 32] = t.transpose(1, 0, 2).reshape(KP, nt * 32)

    def xpass_tiled(self, Yt, N, D, K, X, ldx, state, ws, x_tiled=False):
        self.calls.append('xpass_tiled')
        import torch
        DP = int(self.layout(D, K).DP)
        nt = (N + 31) // 32
        y = Yt.numpy().reshape(nt, DP, 32).transpose(1, 0, 2).reshape(DP, nt * 32)[:D, :N]
        if x_tiled:
            KP = int(self.layout(D, K).KP)
            rows = torch.zeros(KP, nt * 32, dtype=torch.float64)
            self.xpass(torch.from_numpy(np.ascontiguousarray(y)), N, N, D, K, rows, nt * 32, state,
                       ws)
            self.tile_x(True, rows, nt * 32, N, D, K, X)
            self.calls.pop()
        else:
            self.xpass(torch.from_numpy(np.ascontiguousarray(y)), N, N, D, K, X, ldx, state, ws)
        self.calls.pop()

    def small_ops(self, D, K, n_total, x_prec, a0t, b0t, a0a, b0a, ops, state, has_mean=False):
        """vmp_pca_small_ops[_mean]: the operations in order (launch fusion is not modelled)."""
        self.has_mean = bool(has_mean)
        for op in ops:
            if op == 1:
                self.update_w(D, K, n_total, state)
            elif op == 2:
                self.prepare_x(D, K, x_prec, state)
            elif op == 3:
                self.update_tau(D, K, n_total, a0t, b0t, state)
            elif op == 4:
                self.update_alpha(D, K, a0a, b0a, state)
            elif op == 5:
                self.lower_bound(D, K, n_total, x_prec, a0t, b0t, a0a, b0a, state)
            else:
                raise ValueError(op)

    def _resid(self, v, D, K, n_total):
        return (v['Syy'][0] - 2 * np.sum(v['W'][:, :K] * v['S'][:D, :K])
                + np.sum(v['Sww'][:K, :K] * self._sxx(v, K, n_total)))

    def update_tau(self, D, K, n_total, a0, b0, state):
        self.calls.append('update_tau')
        v = self._v(state, D, K)
        a = a0 + 0.5 * D * n_total
        b = b0 + 0.5 * self._resid(v, D, K, n_total)
        v['tau'][:] = [a, b, a / b, special.digamma(a) - np.log(b)]

    def update_alpha(self, D, K, a0, b0, state):
        self.calls.append('update_alpha')
        v = self._v(state, D, K)
        a = a0 + 0.5 * D
        b = b0 + 0.5 * self._ww(v, K)
        v['alpha'][0, :K] = a
        v['alpha'][1, :K] = b
        v['alpha'][2, :K] = a / b
        v['alpha'][3, :K] = special.digamma(a) - np.log(b)

    def lower_bound(self, D, K, n_total, x_prec, a0t, b0t, a0a, b0a, state):
        v = self._v(state, D, K)
        tau, logtau = v['tau'][2], v['tau'][3]
        resid = self._resid(v, D, K, n_total)
        Sxx = self._sxx(v, K, n_total)
        LY = D * n_total * (-0.5 * np.log(2 * np.pi) + 0.5 * logtau) - 0.5 * tau * resid
        LX = (-0.5 * x_prec * np.trace(Sxx)
              + n_total * (0.5 * K * np.log(x_prec) - 0.5 * v['scal'][1] + 0.5 * K))
        LW = (0.5 * D * np.sum(v['alpha'][3, :K])
              - 0.5 * np.sum(v['alpha'][2, :K] * self._ww(v, K))
              + D * (-0.5 * v['scal'][0] + 0.5 * K))
        Lt = gamma_elbo(a0t, b0t, v['tau'][0], v['tau'][1])
        La = gamma_elbo(a0a, b0a, v['alpha'][0, :K], v['alpha'][1, :K])
        v['Lt'][:6] = [LY, LX, LW, Lt, La, LY + LX + LW + Lt + La]

    def rotate_rows(self, R, X, N):
        K = R.shape[0]
        X.numpy()[:K, :N] = R @ X.numpy()[:K, :N]

    def set_timing(self, on):
        pass

    def last_pass_ms(self):
        return 0.0, 0.0


class CPUGMMKernels:
    """TEST DOUBLE for ``bayespy_amd.inference.plans.gmm.GMMKernels``: the same method set on CPU
    torch tensors, the formulas of oracle/gmm.py on the packed state of ``vmp_gmm_layout``
    (host-only ``vmp_gmm_get_layout`` of the real library).  The statistics T = [R, S1, S2] and
    the two softmax sums are LOCAL after ``pass_`` / ``stats_from_labels`` -- the plan all-reduces
    them -- so the sharded host logic runs in CPU-only tests (world_size-2 gloo)."""

    def __init__(self, rt):
        self.rt = rt
        self.lib = _lib.load()
        self.calls = []
        self._prior_only = False

    def layout(self, D, K):
        L = _lib.GMMLayout()
        _lib.raise_for_status(self.lib.vmp_gmm_get_layout(D, K, ctypes.byref(L)))
        return L

    def workspace_doubles(self, D, K):
        return 16

    def _v(self, state, D, K):
        L = self.layout(D, K)
        s = state.numpy()
        KP, FS = int(L.KP), int(L.FS)
        T = s[L.off_T:L.off_T + KP * FS].reshape(KP, FS)
        pr = s[L.off_prior:]
        return dict(
            L=L, s=s, T=T, R=T[:K, 0], S1=T[:K, 1:1 + D], S2=T[:K, 1 + D:].reshape(K, D, D),
            zs=s[L.off_zs:L.off_zs + 2], alpha=s[L.off_alpha:L.off_alpha + K],
            logpi=s[L.off_alpha + KP:L.off_alpha + KP + K],
            mu=s[L.off_mu:L.off_mu + K * D].reshape(K, D),
            Cmu=s[L.off_Cmu:L.off_Cmu + K * D * D].reshape(K, D, D),
            ldLmu=s[L.off_logdetLmu:L.off_logdetLmu + K], nk=s[L.off_nk:L.off_nk + K],
            Vk=s[L.off_Vk:L.off_Vk + K * D * D].reshape(K, D, D),
            Lam=s[L.off_Lam:L.off_Lam + K * D * D].reshape(K, D, D),
            ldLam=s[L.off_logdetLam:L.off_logdetLam + K], ldV=s[L.off_logdetV:L.off_logdetV + K],
            alpha0=pr[:K], hdr=pr[KP:KP + 8], V0=pr[KP + 8:KP + 8 + D * D].reshape(D, D),
            scal=s[L.off_scal:L.off_scal + 8], Lout=s[L.off_L:L.off_L + 8])

    @staticmethod
    def _multidigamma(a, d):
        return sum(special.digamma(a - 0.5 * i) for i in range(d))

    @staticmethod
    def _multigammaln(a, d):
        return d * (d - 1) / 4.0 * np.log(np.pi) + sum(special.gammaln(a - 0.5 * i) for i in range(d))

    def _lambda_moments(self, v, D):
        v['Lam'][:] = v['nk'][:, None, None] * np.linalg.inv(v['Vk'])
        v['ldV'][:] = np.linalg.slogdet(v['Vk'])[1]
        v['ldLam'][:] = self._multidigamma(0.5 * v['nk'], D) + D * np.log(2.0) - v['ldV']

    def init_state(self, D, K, alpha0, beta0, n0, V0, state):
        self.calls.append('init_state')
        state.zero_()
        v = self._v(state, D, K)
        v['alpha0'][:] = alpha0
        v['hdr'][0], v['hdr'][1] = beta0, n0
        v['V0'][:] = V0
        v['hdr'][2] = np.linalg.slogdet(np.asarray(V0))[1]
        v['alpha'][:] = alpha0
        v['logpi'][:] = special.digamma(v['alpha']) - special.digamma(v['alpha'].sum())
        v['mu'][:] = 0.0
        v['Cmu'][:] = np.eye(D) / beta0
        v['ldLmu'][:] = D * np.log(beta0)
        v['nk'][:] = n0
        v['Vk'][:] = V0
        self._lambda_moments(v, D)

    def _set_stats(self, v, r, y):
        v['T'][:] = 0.0
        v['R'][:] = r.sum(axis=0)
        v['S1'][:] = r.T @ y
        v['S2'][:] = np.einsum('nk,ni,nj->kij', r, y, y)

    def stats_from_labels(self, Y, N, D, K, labels, R, state, ws):
        self.calls.append('stats_from_labels')
        v = self._v(state, D, K)
        r = np.zeros((N, K))
        r[np.arange(N), labels.numpy()[:N]] = 1.0
        R.numpy()[:N, :K] = r
        self._set_stats(v, r, Y.numpy()[:N, :D])

    def update_mu(self, D, K, state):
        self.calls.append('update_mu')
        v = self._v(state, D, K)
        Lmu = v['hdr'][0] * np.eye(D) + v['R'][:, None, None] * v['Lam']
        v['Cmu'][:] = np.linalg.inv(Lmu)
        v['ldLmu'][:] = np.linalg.slogdet(Lmu)[1]
        v['mu'][:] = np.einsum('kij,kj->ki', v['Cmu'], np.einsum('kij,kj->ki', v['Lam'], v['S1']))

    def update_lambda(self, D, K, state):
        self.calls.append('update_lambda')
        v = self._v(state, D, K)
        mm = v['Cmu'] + v['mu'][:, :, None] * v['mu'][:, None, :]
        sm = v['S1'][:, :, None] * v['mu'][:, None, :]
        v['nk'][:] = v['hdr'][1] + v['R']
        v['Vk'][:] = v['V0'] + v['S2'] - sm - np.swapaxes(sm, 1, 2) + v['R'][:, None, None] * mm
        self._lambda_moments(v, D)

    def _coefficients(self, v, D):
        mm = v['Cmu'] + v['mu'][:, :, None] * v['mu'][:, None, :]
        b = np.einsum('kij,kj->ki', v['Lam'], v['mu'])
        c = 0.5 * v['ldLam'] - 0.5 * D * np.log(2 * np.pi) - 0.5 * np.einsum('kij,kij->k', v['Lam'], mm)
        return c, b

    def prepare_z(self, D, K, prior_only, state):
        self.calls.append('prepare_z')
        self._prior_only = bool(prior_only)

    def pass_(self, Y, N, D, K, R, state, ws):
        self.calls.append('pass')
        v = self._v(state, D, K)
        y = Y.numpy()[:N, :D]
        if self._prior_only:
            phi = np.tile(v['logpi'][None, :], (N, 1))
        else:
            c, b = self._coefficients(v, D)
            phi = (v['logpi'] + c)[None, :] + y @ b.T \
                - 0.5 * np.einsum('ni,kij,nj->nk', y, v['Lam'], y)
        m = phi.max(axis=1, keepdims=True) if N else phi
        lse = np.log(np.exp(phi - m).sum(axis=1, keepdims=True)) + m
        p = np.exp(phi - lse)
        p /= p.sum(axis=1, keepdims=True)
        R.numpy()[:N, :K] = p
        self._set_stats(v, p, y)
        v['zs'][0] = float(lse.sum())
        v['zs'][1] = float((p * phi).sum())

    def update_alpha(self, D, K, state):
        self.calls.append('update_alpha')
        v = self._v(state, D, K)
        v['alpha'][:] = v['alpha0'] + v['R']
        v['logpi'][:] = special.digamma(v['alpha']) - special.digamma(v['alpha'].sum())

    def lower_bound(self, D, K, state):
        self.calls.append('lower_bound')
        v = self._v(state, D, K)
        c, b = self._coefficients(v, D)
        beta0, n0, ldV0 = v['hdr'][0], v['hdr'][1], v['hdr'][2]
        L_Y = float(np.sum(v['R'] * c) + np.sum(b * v['S1'])
                    - 0.5 * np.einsum('kij,kij->', v['Lam'], v['S2']))
        L_z = float(v['zs'][0] - v['zs'][1] + np.sum(v['R'] * v['logpi']))
        a0, a = v['alpha0'], v['alpha']
        L_pi = float(special.gammaln(a0.sum()) - special.gammaln(a0).sum()
                     - special.gammaln(a.sum()) + special.gammaln(a).sum()
                     + np.sum((a0 - a) * v['logpi']))
        mm = v['Cmu'] + v['mu'][:, :, None] * v['mu'][:, None, :]
        L_mu = float(np.sum(-0.5 * beta0 * np.einsum('kii->k', mm) + 0.5 * D * np.log(beta0)
                            - 0.5 * v['ldLmu'] + 0.5 * D))
        g_p = 0.5 * n0 * ldV0 - 0.5 * D * n0 * np.log(2.0) - self._multigammaln(0.5 * n0, D)
        g_q = 0.5 * v['nk'] * v['ldV'] - 0.5 * D * v['nk'] * np.log(2.0) \
            - self._multigammaln(0.5 * v['nk'], D)
        t = -0.5 * np.einsum('ij,kij->k', v['V0'], v['Lam']) + 0.5 * n0 * v['ldLam'] \
            + 0.5 * np.einsum('kij,kij->k', v['Vk'], v['Lam']) - 0.5 * v['nk'] * v['ldLam']
        L_Lam = float(np.sum(g_p - g_q + t))
        v['Lout'][:6] = [L_Y, L_z, L_pi, L_mu, L_Lam, L_Y + L_z + L_pi + L_mu + L_Lam]
        v['scal'][3] = 0.0

    def set_timing(self, on):
        pass

    def last_pass_ms(self):
        return 0.0, 0.0

    def pass_times_ms(self, cap=64):
        return []


class CPULSSMKernels:
    """TEST DOUBLE for ``bayespy_amd.inference.plans.lssm.LSSMKernels``: time-major arrays and the
    packed ``vmp_lssm_layout`` state as in the library (host-only ``vmp_lssm_get_layout`` /
    ``vmp_lssm_workspace_doubles``), arithmetic in NumPy: the covariance recursion of
    oracle/lssm.py (chain_covariances), the per-sequence recursions and LOCAL plate sums of
    vmp_lssm_smooth, and the replicated-node algebra of vmp_lssm_small_ops operation by
    operation.  The plan all-reduces the raw sums -- world_size-2 gloo tests run on it."""

    def __init__(self, rt):
        self.rt = rt
        self.lib = _lib.load()
        self.calls = []

    def layout(self, D, M):
        L = _lib.LSSMLayout()
        _lib.raise_for_status(self.lib.vmp_lssm_get_layout(D, M, ctypes.byref(L)))
        return L

    def workspace_doubles(self, D, M, B, T):
        return 16

    def relayout_y(self, Y, M, B, T, BL, Yt, syy, ws):
        self.calls.append('relayout_y')
        y = Y.numpy().reshape(M, B, T)
        yt = Yt.numpy().reshape(T, M, BL)
        yt[:] = 0.0
        yt[:, :, :B] = y.transpose(2, 0, 1)
        syy.numpy()[0] = float(np.sum(y * y))

    def x_layout(self, X, D, B, T, BL, Z, to_time_major):
        self.calls.append('x_layout')
        x = X.numpy().reshape(B, T, D)
        z = Z.numpy().reshape(T, D, BL)
        if to_time_major:
            z[:, :, :B] = x.transpose(1, 2, 0)
        else:
            x[:] = z[:, :, :B].transpose(2, 0, 1)

    def _cov(self, T, D, Dg, Sinv, J, sums):
        from oracle.lssm import chain_covariances
        d = Dg.numpy()[:4 * D * D].reshape(4, D, D)
        dg = np.empty((T, D, D))
        dg[:] = d[1]
        dg[0] = d[0]
        dg[T - 1] = d[2] if T > 1 else d[0]
        E = np.broadcast_to(d[3], (max(T - 1, 0), D, D))
        si, j, _, V, Cn, logdet = chain_covariances(dg, E)
        Sinv.numpy()[:T * D * D] = si.reshape(-1)
        if T > 1:
            J.numpy()[:(T - 1) * D * D] = j.reshape(-1)
        s = sums.numpy()
        DD = D * D
        s[0:DD] = V.sum(axis=0).reshape(-1)
        s[DD:2 * DD] = V[0].reshape(-1)
        s[2 * DD:3 * DD] = V[T - 1].reshape(-1)
        s[3 * DD:4 * DD] = (Cn.sum(axis=0) if T > 1 else np.zeros((D, D))).reshape(-1)
        s[5 * DD] = logdet
        s[5 * DD + 1] = 0.0

    def smooth(self, given, Yt, M, B, T, BL, D, Cm, tau, h0, Sinv, J, Z, stats, ws):
        self.calls.append('smooth_given' if given else 'smooth')
        yt = Yt.numpy().reshape(T, M, BL)[:, :, :B]
        z = Z.numpy().reshape(T, D, BL)
        if given:
            x = z[:, :, :B].copy()                                  # (T, D, B)
        else:
            cm = Cm.numpy()[:M * D].reshape(M, D)
            t_ = float(tau.numpy()[0])
            si = Sinv.numpy()[:T * D * D].reshape(T, D, D)
            j = J.numpy()[:max(T - 1, 0) * D * D].reshape(-1, D, D)
            h = t_ * np.einsum('tmb,md->tdb', yt, cm)
            h[0] += h0.numpy()[:D, None]
            zz = np.empty_like(h)
            zz[0] = h[0]
            for t in range(1, T):
                zz[t] = h[t] - j[t - 1].T @ zz[t - 1]
            x = np.empty_like(h)
            x[T - 1] = si[T - 1] @ zz[T - 1]
            for t in range(T - 2, -1, -1):
                x[t] = si[t] @ zz[t] - j[t] @ x[t + 1]
            z[:, :, :B] = x
        raw = stats.numpy()
        DD = D * D
        raw[0:DD] = np.einsum('tib,tjb->ij', x, x).reshape(-1)
        raw[DD:2 * DD] = np.einsum('tib,tjb->ij', x[1:], x[:-1]).reshape(-1)
        raw[2 * DD:3 * DD] = (x[0] @ x[0].T).reshape(-1)
        raw[3 * DD:4 * DD] = (x[T - 1] @ x[T - 1].T).reshape(-1)
        raw[4 * DD:4 * DD + D] = x[0].sum(axis=1)
        raw[4 * DD + D:4 * DD + D + M * D] = np.einsum('tmb,tib->mi', yt, x).reshape(-1)

    def x_update(self, T, D, Dg, Sinv, J, sums, Yt, M, B, BL, Cm, tau, h0, Z, stats, ws):
        self.calls.append('x_update')
        self._cov(T, D, Dg, Sinv, J, sums)
        self.smooth(False, Yt, M, B, T, BL, D, Cm, tau, h0, Sinv, J, Z, stats, ws)
        self.calls.pop()

    def rotate_x(self, D, T, B, BL, R, Z):
        self.calls.append('rotate_x')
        z = Z.numpy().reshape(T, D, BL)
        z[:] = np.einsum('ij,tjb->tib', R.numpy().reshape(D, D), z)

    def small_ops(self, D, M, T, B_total, priors, nu_latent, ops, state):
        """vmp_lssm_small_ops (csrc/vmp_lssm.hip: lssm_small_body), operation by operation."""
        self.calls.extend('op%d' % o for o in ops)
        L = self.layout(D, M)
        st = state.numpy()
        DD = D * D
        pri = list(priors)
        tau = st[L.off_tau:L.off_tau + 4]
        gam = st[L.off_gamma:L.off_gamma + 4 * D].reshape(4, D)
        alp = st[L.off_alpha:L.off_alpha + 4 * D].reshape(4, D)
        nu = st[L.off_nu:L.off_nu + 4 * D].reshape(4, D)
        Cm = st[L.off_Cm:L.off_Cm + M * D].reshape(M, D)
        CovC = st[L.off_CovC:L.off_CovC + DD].reshape(D, D)
        SCC = st[L.off_SCC:L.off_SCC + DD].reshape(D, D)
        Am = st[L.off_Am:L.off_Am + DD].reshape(D, D)
        AA = st[L.off_AA:L.off_AA + DD * D].reshape(D, D, D)
        ldA = st[L.off_ldA:L.off_ldA + D]
        S = st[L.off_S:]
        Sxx, Spp, Snn, Snp, S00 = (S[i * DD:(i + 1) * DD].reshape(D, D) for i in range(5))
        s0 = S[5 * DD:5 * DD + D]
        Syx = S[5 * DD + D:5 * DD + D + M * D].reshape(M, D)
        sc = st[L.off_scal:L.off_scal + 8]
        Lam0 = st[L.off_Lam0:L.off_Lam0 + DD].reshape(D, D)
        mu0 = st[L.off_mu0:L.off_mu0 + D]
        Bt = float(B_total)

        def set_gamma(g, k, a, b):
            g[0][k], g[1][k], g[2][k] = a, b, a / b
            g[3][k] = special.digamma(a) - np.log(b)

        def gamma_term(a0, b0, g, k):
            a, b = g[0][k], g[1][k]
            return (a0 * np.log(b0) - special.gammaln(a0)) - (a * np.log(b) - special.gammaln(a)) \
                + (b - b0) * g[2][k] + (a0 - a) * g[3][k]

        def residual_and_innovation():
            resid = sc[0] - 2.0 * np.sum(Cm * Syx) + np.sum(SCC * Sxx)
            innov = np.array([Snn[i, i] - 2.0 * Am[i] @ Snp[i] + np.sum(AA[i] * Spp)
                              for i in range(D)])
            return resid, innov

        for op in ops:
            if op == 1:                                                    # STATS
                raw = st[L.off_raw:L.off_raw + int(L.len_raw)]
                cs = st[L.off_covsums:L.off_covsums + 5 * DD + 8]
                xx, npm, x0, xT = (raw[i * DD:(i + 1) * DD].reshape(D, D) for i in range(4))
                cV, cV0, cVT, cC = (cs[i * DD:(i + 1) * DD].reshape(D, D) for i in range(4))
                Sxx[:] = Bt * cV + xx
                Spp[:] = Bt * (cV - cVT) + xx - xT
                Snn[:] = Bt * (cV - cV0) + xx - x0
                Snp[:] = Bt * cC.T + npm
                S00[:] = Bt * cV0 + x0
                s0[:] = raw[4 * DD:4 * DD + D]
                Syx[:] = raw[4 * DD + D:4 * DD + D + M * D].reshape(M, D)
                sc[1] = cs[5 * DD]
            elif op == 2:                                                  # C
                lam = tau[2] * Sxx + np.diag(gam[2])
                CovC[:] = np.linalg.inv(lam)
                sc[4] = -np.linalg.slogdet(lam)[1]
                Cm[:] = tau[2] * Syx @ CovC.T
                SCC[:] = M * CovC + Cm.T @ Cm
            elif op == 3:                                                  # GAMMA
                for j in range(D):
                    set_gamma(gam, j, pri[2] + 0.5 * M, pri[3] + 0.5 * SCC[j, j])
            elif op == 4:                                                  # XPREP
                Dg = st[L.off_Dg:L.off_Dg + 4 * DD].reshape(4, D, D)
                anua = np.einsum('i,ijk->jk', nu[2], AA)
                obs = tau[2] * SCC
                dn = np.diag(nu[2])
                Dg[0] = obs + Lam0 + (anua if T > 1 else 0.0)
                Dg[1] = obs + dn + anua
                Dg[2] = obs + (dn if T > 1 else Lam0)
                Dg[3] = -(nu[2][:, None] * Am).T
                st[L.off_h0:L.off_h0 + D] = Lam0 @ mu0
                sc[3] = tau[2]
            elif op == 5:                                                  # A
                for i in range(D):
                    lam = nu[2][i] * Spp + np.diag(alp[2])
                    cov = np.linalg.inv(lam)
                    ldA[i] = -np.linalg.slogdet(lam)[1]
                    Am[i] = cov @ (nu[2][i] * Snp[i])
                    AA[i] = cov + np.outer(Am[i], Am[i])
            elif op == 6:                                                  # ALPHA
                for j in range(D):
                    set_gamma(alp, j, pri[4] + 0.5 * D, pri[5] + 0.5 * np.sum(AA[:, j, j]))
            elif op == 7:                                                  # TAU
                resid, _ = residual_and_innovation()
                tg = tau.reshape(4, 1)
                set_gamma(tg, 0, pri[0] + 0.5 * M * Bt * T, pri[1] + 0.5 * resid)
            elif op == 8:                                                  # NU
                _, innov = residual_and_innovation()
                for i in range(D):
                    set_gamma(nu, i, pri[6] + 0.5 * Bt * (T - 1), pri[7] + 0.5 * innov[i])
            elif op == 9:                                                  # ELBO
                resid, innov = residual_and_innovation()
                Lo = st[L.off_L:L.off_L + 16]
                LOG2PI = np.log(2 * np.pi)
                Lo[0] = M * Bt * T * (-0.5 * LOG2PI + 0.5 * tau[3]) - 0.5 * tau[2] * resid
                Lo[1] = M * (0.5 * sc[4] + 0.5 * D) + np.sum(0.5 * M * gam[3] - 0.5 * gam[2] * np.diag(SCC))
                saa = np.array([np.sum(AA[:, j, j]) for j in range(D)])
                Lo[2] = 0.5 * D * D + 0.5 * np.sum(ldA) + np.sum(0.5 * D * alp[3] - 0.5 * alp[2] * saa)
                lx = Bt * (0.5 * T * D + 0.5 * st[L.off_ldLam0] + 0.5 * (T - 1) * np.sum(nu[3])
                           - 0.5 * sc[1])
                lx -= 0.5 * np.sum(Lam0 * (S00 - np.outer(s0, mu0) - np.outer(mu0, s0)
                                           + Bt * np.outer(mu0, mu0)))
                lx -= 0.5 * np.sum(nu[2] * innov)
                Lo[3] = lx
                Lo[4] = sum(gamma_term(pri[2], pri[3], gam, j) for j in range(D))
                Lo[5] = sum(gamma_term(pri[4], pri[5], alp, j) for j in range(D))
                Lo[6] = gamma_term(pri[0], pri[1], tau.reshape(4, 1), 0)
                Lo[7] = sum(gamma_term(pri[6], pri[7], nu, j) for j in range(D)) if nu_latent else 0.0
                Lo[8] = float(np.sum(Lo[:8]))

    def set_timing(self, on):
        pass

    def pass_times_ms(self, cap=64):
        return []


class CPUMaskedKernels:
    """TEST DOUBLE for ``bayespy_amd.inference.plans.masked_pca.MaskedHIPKernels``: the packed state
    of ``vmp_mpca_layout`` (host-only ``vmp_mpca_get_layout``; Mst = packed M_d | r_d with
    p(i,j) = i(i+1)/2 + j, tile-major Ymt), arithmetic of oracle/masked_pca.py in NumPy.  The
    statistics a pass leaves in the state are LOCAL; the plan all-reduces them."""

    SC_SYY, SC_NOBS, SC_TRXX, SC_LDX, SC_N, SC_STATUS, SC_TAUX, SC_RESID = range(8)

    def __init__(self, rt):
        self.rt = rt
        self.lib = _lib.load()
        self.calls = []

    def layout(self, D, K):
        L = _lib.MPCALayout()
        _lib.raise_for_status(self.lib.vmp_mpca_get_layout(D, K, ctypes.byref(L)))
        return L

    def sizes(self, D, K, N, chunk):
        L = self.layout(D, K)
        nt = max((N + 31) // 32, 1)
        s = _lib.MPCASizes()
        s.ymt_doubles = nt * int(L.DP) * 32
        s.mask_words = nt * int(L.DP)
        s.xm_doubles = nt * 32 * int(L.KP)
        s.lam_doubles = 16
        s.xxf_doubles = 16
        s.workspace_doubles = 16
        return s

    def _v(self, state, D, K):
        L = self.layout(D, K)
        s = state.numpy()
        DP, KP, LR, PT = int(L.DP), int(L.KP), int(L.LR), int(L.PT)
        return dict(L=L, s=s, tau=s[L.off_tau:L.off_tau + 4],
                    alpha=s[L.off_alpha:L.off_alpha + 4 * KP].reshape(4, KP),
                    sc=s[L.off_scal:L.off_scal + 16], Lo=s[L.off_L:L.off_L + 8],
                    W=s[L.off_W:L.off_W + DP * KP].reshape(DP, KP),
                    WW=s[L.off_WW:L.off_WW + DP * KP * KP].reshape(DP, KP, KP),
                    ldW=s[L.off_ldW:L.off_ldW + DP],
                    Mst=s[L.off_M:L.off_M + DP * LR].reshape(DP, LR),
                    Sxx=s[L.off_Sxx:L.off_Sxx + KP * KP].reshape(KP, KP),
                    rowobs=s[L.off_rowobs:L.off_rowobs + DP], roff=16 * PT)

    @staticmethod
    def _tri(K):
        i, j = np.tril_indices(K)
        return i, j, i * (i + 1) // 2 + j

    def _get_M(self, v, D, K):
        i, j, p = self._tri(K)
        M = np.zeros((D, K, K))
        M[:, i, j] = v['Mst'][:D][:, p]
        M[:, j, i] = v['Mst'][:D][:, p]
        return M, v['Mst'][:D, v['roff']:v['roff'] + K]

    def init_state(self, D, K, a0t, b0t, a0a, b0a, state):
        self.calls.append('init_state')
        state.zero_()
        v = self._v(state, D, K)
        v['tau'][:] = [a0t, b0t, a0t / b0t, special.digamma(a0t) - np.log(b0t)]
        v['alpha'][0, :K], v['alpha'][1, :K] = a0a, b0a
        v['alpha'][2, :K] = a0a / b0a
        v['alpha'][3, :K] = special.digamma(a0a) - np.log(b0a)

    def prepare(self, Y, ldy, mask, ldm, N, D, K, Ymt, Mb1, Mb2, state, ws):
        self.calls.append('prepare')
        v = self._v(state, D, K)
        DP = int(v['L'].DP)
        y = Y.numpy()[:D, :N]
        m = (mask.numpy()[:D, :N] != 0) if mask is not None else np.ones((D, N), dtype=bool)
        self._m = m.astype(np.float64)
        self._y = np.where(m, y, 0.0)                    # values at masked entries are never read
        nt = max((N + 31) // 32, 1)
        buf = np.zeros((DP, nt * 32))
        buf[:D, :N] = self._y
        Ymt.numpy()[:nt * DP * 32] = buf.reshape(DP, nt, 32).transpose(1, 0, 2).reshape(-1)
        v['sc'][self.SC_SYY] = float(np.sum(self._y ** 2))
        v['sc'][self.SC_NOBS] = float(self._m.sum())
        v['rowobs'][:] = 0.0
        v['rowobs'][:D] = self._m.sum(axis=1)
        self._xx = None

    def x_begin(self, D, K, n_total, state):
        self.calls.append('x_begin')
        v = self._v(state, D, K)
        v['sc'][self.SC_N] = n_total
        v['sc'][self.SC_TRXX] = v['sc'][self.SC_LDX] = 0.0
        v['Mst'][:] = 0.0
        v['Sxx'][:] = 0.0

    def _posterior_x(self, v, D, K, N, flags, x_prec, Xm):
        x = Xm.numpy()[:N, :K]
        if flags & 2:                                    # FROM_VALUE: delta moments
            xm = x.copy()
            cov = np.zeros((N, K, K))
            ld = 0.0
        elif flags & 4:                                  # PRIOR
            xm = np.zeros((N, K))
            cov = np.broadcast_to(np.eye(K) / x_prec, (N, K, K)).copy()
            ld = -N * K * np.log(x_prec)
        else:
            tau = v['tau'][2]
            WWf = v['WW'][:D, :K, :K].reshape(D, K * K)
            lam = x_prec * np.eye(K)[None] + tau * (self._m.T @ WWf).reshape(N, K, K)
            cov = np.linalg.inv(lam) if N else np.zeros((0, K, K))
            ld = -float(np.sum(np.linalg.slogdet(lam)[1])) if N else 0.0
            xm = np.einsum('nij,nj->ni', cov, tau * (self._y.T @ v['W'][:D, :K]))
            v['sc'][self.SC_TAUX] = tau
        return xm, cov + xm[:, :, None] * xm[:, None, :], ld

    def x_pass(self, D, K, N, chunk, nsets, flags, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state, ws):
        self.calls.append('x_pass')
        v = self._v(state, D, K)
        xm, xx, ld = self._posterior_x(v, D, K, N, flags, x_prec, Xm)
        Xm.numpy()[:N, :K] = xm
        self._xx = xx
        i, j, p = self._tri(K)
        M = (self._m @ xx.reshape(N, K * K)).reshape(D, K, K)
        v['Mst'][:D][:, p] = M[:, i, j]
        v['Mst'][:D, v['roff']:v['roff'] + K] = self._y @ xm
        v['Sxx'][:K, :K] = xx.sum(axis=0)
        v['sc'][self.SC_TRXX] = float(np.einsum('nkk->', xx))
        v['sc'][self.SC_LDX] = ld

    def x_chunk(self, D, K, n0, nplates, flags, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state, ws):
        """Only the INSPECT use of the plan (x_second_moments): <xx> of plates [n0, n0 + nplates)
        as the last pass left them."""
        self.calls.append('x_chunk')
        assert flags & 8
        self._chunk = self._xx[n0:n0 + nplates]

    def unpack_xx(self, D, K, nplates, XXf, out):
        out.numpy()[:nplates] = self._chunk

    def update_w(self, D, K, mode, state):
        self.calls.append('update_w%d' % mode)
        v = self._v(state, D, K)
        alpha = v['alpha'][2, :K]
        if mode == 2:                                    # prior moments given <alpha>
            v['W'][:D, :K] = 0.0
            v['WW'][:D, :K, :K] = np.diag(1.0 / alpha)
            v['ldW'][:D] = -np.sum(np.log(alpha))
        elif mode == 1:                                  # delta moments of the value in W
            w = v['W'][:D, :K]
            v['WW'][:D, :K, :K] = w[:, :, None] * w[:, None, :]
            v['ldW'][:D] = 0.0
        else:
            M, r = self._get_M(v, D, K)
            tau = v['tau'][2]
            lam = np.diag(alpha)[None] + tau * M
            cw = np.linalg.inv(lam)
            w = np.einsum('dij,dj->di', cw, tau * r)
            v['W'][:D, :K] = w
            v['WW'][:D, :K, :K] = cw + w[:, :, None] * w[:, None, :]
            v['ldW'][:D] = -np.linalg.slogdet(lam)[1]

    def small_ops(self, D, K, x_prec, a0t, b0t, a0a, b0a, ops, state):
        self.calls.extend('op%d' % o for o in ops)
        v = self._v(state, D, K)
        sc = v['sc']
        wm = v['rowobs'][:D] > 0                          # rows some rank observes
        De = float(wm.sum())

        def resid():
            M, r = self._get_M(v, D, K)
            return sc[self.SC_SYY] - 2.0 * np.sum(v['W'][:D, :K] * r) \
                + np.sum(v['WW'][:D, :K, :K] * M)
        for op in ops:
            if op == 1:                                  # TAU
                a, b = a0t + 0.5 * sc[self.SC_NOBS], b0t + 0.5 * resid()
                v['tau'][:] = [a, b, a / b, special.digamma(a) - np.log(b)]
            elif op == 2:                                # ALPHA
                a = a0a + 0.5 * De
                b = b0a + 0.5 * np.einsum('dkk->k', v['WW'][:D, :K, :K][wm])
                v['alpha'][0, :K], v['alpha'][1, :K] = a, b
                v['alpha'][2, :K] = a / b
                v['alpha'][3, :K] = special.digamma(a) - np.log(b)
            else:                                        # ELBO
                tau, logtau = v['tau'][2], v['tau'][3]
                alpha, logalpha = v['alpha'][2, :K], v['alpha'][3, :K]
                N = sc[self.SC_N]
                LOG2PI = np.log(2 * np.pi)
                L_Y = sc[self.SC_NOBS] * (-0.5 * LOG2PI + 0.5 * logtau) - 0.5 * tau * resid()
                L_X = -0.5 * x_prec * sc[self.SC_TRXX] + 0.5 * sc[self.SC_LDX] \
                    + N * (0.5 * K * np.log(x_prec) + 0.5 * K)
                L_W = 0.5 * De * np.sum(logalpha) \
                    - 0.5 * np.sum(alpha * np.einsum('dkk->k', v['WW'][:D, :K, :K][wm])) \
                    + 0.5 * float(np.sum(v['ldW'][:D][wm])) + 0.5 * De * K
                L_tau = gamma_elbo(a0t, b0t, v['tau'][0], v['tau'][1])
                L_alpha = gamma_elbo(a0a, b0a, v['alpha'][0, :K], v['alpha'][1, :K])
                v['Lo'][:6] = [L_Y, L_X, L_W, L_tau, L_alpha, L_Y + L_X + L_W + L_tau + L_alpha]
                sc[self.SC_STATUS] = 0.0

    def set_timing(self, on):
        pass

    def pass_times_ms(self, cap=64):
        return []


def attach_cpu(Q, stats=None):
    """Give every fused plan of a VB object its CPU kernel double (and a CPU runtime)."""
    from bayespy_amd.device import Runtime
    doubles = {'PCAPlan': CPURuntimeKernels, 'GMMPlan': CPUGMMKernels, 'LSSMPlan': CPULSSMKernels,
               'MaskedPCAPlan': CPUMaskedKernels}
    rt = Runtime(device='cpu')
    for p in Q.plans:
        if type(p).__name__ == 'MaskedLSSMPlan':
            # the double of this block is its own device code compiled for the host
            import host_build
            from bayespy_amd.inference.plans.lssm_masked import MaskedLSSMKernels
            p._rt, p._kernels = rt, MaskedLSSMKernels(rt, lib=host_build.lssmm_host())
            continue
        p._rt, p._kernels = rt, doubles[type(p).__name__](rt)
        if stats is not None and type(p).__name__ == 'PCAPlan':
            p.stats = stats
    return Q
