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
