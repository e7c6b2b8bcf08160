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
        )
        return v

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
        W = tau * v['S'][:D, :K] @ C
        v['CW'][:K, :K] = C
        v['W'][:, :K] = W
        v['Sww'][:K, :K] = D * C + W.T @ W
        v['scal'][0] = logdet

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

    def small_ops(self, D, K, n_total, x_prec, a0t, b0t, a0a, b0a, ops, state):
        """vmp_pca_small_ops: the operations in order (launch fusion is not modelled)."""
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
        b = b0 + 0.5 * np.diag(v['Sww'][:K, :K])
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
              - 0.5 * np.sum(v['alpha'][2, :K] * np.diag(v['Sww'][:K, :K]))
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
