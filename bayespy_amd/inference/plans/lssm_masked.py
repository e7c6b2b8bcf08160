"""
Execution plan of the linear state-space model block observed through an ARRAY mask

    Q['Y'].observe(y, mask=mask)        # bayespy/demos/lssm.py:132; mask = random.mask(M, N, p=0.3),
                                        # mask[:, 30:80] = False (:239-246)

-- the same graph as plans/lssm.py (with or without a plate of sequences), the mask broadcastable
to (M, [B,] T).  The reference multiplies every message by the child's mask before the plate sum
(node.py:570-655), so the chain precision differs per sequence and every row of C has its own
posterior; rows / sequences without any observation are ignored plates (node.py:486-526,
expfamily.py:470-480).  A group of four lanes per sequence (the rows of the D x D blocks dealt over
them) runs the covariance AND the mean recursion of linalg.block_banded_solve
(utils/linalg.py:468-575) in registers
(bayespy_amd/csrc/vmp_lssmm.hip; formulas pinned in oracle/lssm.py:MaskedLSSMOracle).

HBM: ``Yt`` (T, M, BL) data time-major, zero where masked; ``Mw`` (T, BL) one 64-bit mask word per
(step, sequence); ``F`` (T, NS + D, BL) forward sweep (S_t^-1 packed | z_t); ``Z`` (T, D, BL) <x>;
``P`` (T, NS, BL) <x x^T> packed; one small state block (``vmp_lssmm_layout``).
"""
import ctypes

import numpy as np

from . import _delta
from .lssm import (LSSMPlan, OP_C, OP_GAMMA, OP_XPREP, OP_A, OP_ALPHA, OP_TAU, OP_NU, OP_ELBO)
from ... import _lib
from ...device import ptr


def _sym_unpack(v, D):
    """packed lower triangle (..., NS) -> symmetric (..., D, D)."""
    out = np.empty(v.shape[:-1] + (D, D))
    for i in range(D):
        for j in range(i + 1):
            out[..., i, j] = out[..., j, i] = v[..., i * (i + 1) // 2 + j]
    return out


def _sym_pack(a, D):
    out = np.empty(a.shape[:-2] + (D * (D + 1) // 2,))
    for i in range(D):
        for j in range(i + 1):
            out[..., i * (i + 1) // 2 + j] = 0.5 * (a[..., i, j] + a[..., j, i])
    return out


class MaskedLSSMKernels:
    """ctypes front of the vmp_lssmm_* entry points (include/vmp_hip.h).  ``lib`` is
    libvmp_hip.so; the CPU suite passes the host build of the same device code
    (tests/host_build.py) together with a CPU runtime."""

    def __init__(self, rt, lib=None):
        self.rt = rt
        self.lib = lib if lib is not None else rt.lib
        self.ctx = rt.ctx
        if lib is not None:
            _lib.bind_lssmm(lib)

    def _check(self, rc):
        if rc != _lib.VMP_OK:
            if self.rt.lib is not None and self.lib is self.rt.lib:
                self.rt.check(rc)
            _lib.raise_for_status(rc, 'vmp_lssmm call failed')

    def layout(self, D, M):
        L = _lib.LSSMMLayout()
        rc = self.lib.vmp_lssmm_get_layout(D, M, ctypes.byref(L))
        if rc != _lib.VMP_OK:
            _lib.raise_for_status(rc, 'the masked state-space block supports D <= 8 states, '
                                      'M <= 64 observed dimensions, M D^2 <= 2048')
        return L

    def workspace_doubles(self, D, M, B, T):
        n = ctypes.c_int64()
        self._check(self.lib.vmp_lssmm_workspace_doubles(D, M, B, T, ctypes.byref(n)))
        return n.value

    def prepare(self, Y, mask, strides, M, B, T, BL, D, Yt, Mw, seqobs, state, ws):
        self._check(self.lib.vmp_lssmm_prepare(self.ctx, ptr(Y), ptr(mask), strides[0], strides[1],
                                               strides[2], M, B, T, BL, D, ptr(Yt), ptr(Mw),
                                               ptr(seqobs), ptr(state), ptr(ws)))

    def x_update(self, given, Yt, Mw, seqobs, M, B, T, BL, D, state, F, Z, P, ws):
        self._check(self.lib.vmp_lssmm_x_update(self.ctx, int(given), ptr(Yt), ptr(Mw),
                                                ptr(seqobs), M, B, T, BL, D, ptr(state), ptr(F),
                                                ptr(Z), ptr(P), ptr(ws)))

    def small_ops(self, D, M, T, priors, nu_latent, ops, state):
        pr = (ctypes.c_double * 8)(*priors)
        arr = (ctypes.c_int32 * len(ops))(*ops)
        self._check(self.lib.vmp_lssmm_small_ops(self.ctx, D, M, T, pr, 1 if nu_latent else 0,
                                                 len(ops), arr, ptr(state)))

    # Z (T, D, BL) <-> X (B, T, D), <x> <- R <x>: the layout of the fully observed block
    def x_layout(self, X, D, B, T, BL, Z, to_time_major):
        if self.rt.lib is None:             # CPU runtime of the test-suite: torch indexing
            z = Z.view(T, D, BL)
            if to_time_major:
                z[:, :, :B] = X.permute(1, 2, 0)
            else:
                X.copy_(z[:, :, :B].permute(2, 0, 1))
            return
        self.rt.check(self.rt.lib.vmp_lssm_x_layout(self.ctx, ptr(X), D, B, T, BL, ptr(Z),
                                                    1 if to_time_major else 0))

    def rotate_x(self, D, T, B, BL, R, Z):
        if self.rt.lib is None:
            z = Z.view(T, D, BL)
            z.copy_(self.rt.torch.einsum('ij,tjb->tib', R, z))
            return
        self.rt.check(self.rt.lib.vmp_lssm_rotate_x(self.ctx, D, T, B, BL, ptr(R), ptr(Z)))

    def rotate_p(self, D, T, B, BL, R, P):
        self._check(self.lib.vmp_lssmm_rotate_p(self.ctx, D, T, B, BL, ptr(R), ptr(P)))

    def set_timing(self, on):
        if self.rt.lib is not None:
            self.rt.check(self.rt.lib.vmp_ctx_set_timing(self.ctx, 1 if on else 0))

    def pass_times_ms(self, cap=64):
        if self.rt.lib is None:
            return []
        a = (ctypes.c_double * cap)()
        b = (ctypes.c_double * cap)()
        n = ctypes.c_int32()
        self.rt.check(self.rt.lib.vmp_pass_times_ms(self.ctx, a, b, cap, ctypes.byref(n)))
        return [(a[i], b[i]) for i in range(n.value)]


class MaskedLSSMPlan(LSSMPlan):

    _label = 'fused state-space block with array masks'

    @staticmethod
    def _accepts_mask(Y):
        return Y._mask is not True

    @staticmethod
    def describe():
        return ("GaussianARD(SumMultiply('i,i', C, GaussianMarkovChain(mu0, Lam0, A, nu)), tau) "
                "observed through an array mask, shared dynamics, D <= 8 states")

    @staticmethod
    def _limits():
        mx_d, mx_m = ctypes.c_int32(), ctypes.c_int32()
        _lib.load().vmp_lssmm_limits(ctypes.byref(mx_d), ctypes.byref(mx_m))
        return mx_d.value, mx_m.value

    _dims_note = ', M D^2 <= 2048'

    @staticmethod
    def _dims_ok(D, M):
        return M * D * D <= 2048          # the tables of the sweeps in LDS (lssmm_dims_ok)

    @staticmethod
    def unsupported_state(r):
        Y = r['Y']
        if not Y.observed or Y._mask is True:
            return 'Y must be observed through an array mask'
        for key in ('C', 'gamma', 'X', 'A', 'alpha', 'tau', 'nu'):
            n = r.get(key)
            if n is None:
                continue
            if getattr(n, 'observed', False):
                return '%s is observed' % n.name
            init = n._init
            if init is not None and init[0] != 'value' \
                    and not (init[0] == 'random' and key in ('C', 'A')):
                return '%s.initialize_from_%s' % (n.name, init[0])
        return None

    @property
    def kernels(self):
        if self._kernels is None:
            self._kernels = MaskedLSSMKernels(self.rt)
        return self._kernels

    def invalidate(self, node):
        if node is self.Y and self._ready and self._version > 1 \
                and self.unsupported_state(self.roles) is None:
            self._reobserve()
            self._version += 1
            return
        _delta.warn_state_discarded(self, node)
        self._ready = False
        self._version += 1
        self._pending = []
        if self.unsupported_state(self.roles) is not None:
            roles = LSSMPlan.match(self.nodes())
            if roles is not None:
                LSSMPlan(roles)
            else:
                from .generic import GenericPlan
                GenericPlan(self.nodes())

    # -- device state -------------------------------------------------------------------------------
    def _materialize(self):
        if self._ready:
            return
        self._delta = _delta.delta_roles(self.roles)
        rt, k = self.rt, self.kernels
        torch = rt.torch
        D, M, B, T = self.D, self.M, self.B, self.T
        why = self.unsupported_state(self.roles)
        if why is not None:
            raise NotImplementedError('the masked state-space block does not cover this model '
                                      "state (%s); use VB(..., engine='generic')" % why)
        rt.sync_stream()
        self.layout = L = k.layout(D, M)
        NS = self.NS = int(L.NS)
        self.sharded = any(getattr(n, '_shard_axis', None) is not None
                           for n in (self.X, self.G, self.F, self.Y))
        self.BL = BL = (B + 63) // 64 * 64 if B > 0 else 64
        self.ws = rt.empty(int(k.workspace_doubles(D, M, B, T)))
        st = np.zeros(int(L.total))
        pr = self.priors
        st[L.off_tau:L.off_tau + 4] = self._gamma_init(self.tau, pr[0], pr[1], 1)
        st[L.off_gamma:L.off_gamma + 4 * D] = self._gamma_init(self.gamma, pr[2], pr[3], D)
        st[L.off_alpha:L.off_alpha + 4 * D] = self._gamma_init(self.alpha, pr[4], pr[5], D)
        if self.nu is not None:
            st[L.off_nu:L.off_nu + 4 * D] = self._gamma_init(self.nu, pr[6], pr[7], D)
        else:
            st[L.off_nu + 2 * D:L.off_nu + 3 * D] = self.nu_const
            st[L.off_nu + 3 * D:L.off_nu + 4 * D] = np.log(self.nu_const)
        st[L.off_mu0:L.off_mu0 + D] = self.mu0
        st[L.off_Lam0:L.off_Lam0 + D * D] = self.Lam0.reshape(-1)
        st[L.off_ldLam0] = np.linalg.slogdet(self.Lam0)[1]
        gmean = st[L.off_gamma + 2 * D:L.off_gamma + 3 * D]
        amean = st[L.off_alpha + 2 * D:L.off_alpha + 3 * D]
        if self.C._init is None:
            cm, covc = np.zeros((M, D)), np.diag(1.0 / gmean)
        elif self.C._init[0] == 'random':
            cm, covc = np.random.normal(size=(M, D)) / np.sqrt(gmean), np.zeros((D, D))
        else:
            cm = np.broadcast_to(np.asarray(self.C._init[1], dtype=np.float64),
                                 self.C.plates + (D,)).reshape(M, D)
            covc = np.zeros((D, D))
        st[L.off_Cm:L.off_Cm + M * D] = cm.reshape(-1)
        st[L.off_CovC:L.off_CovC + M * D * D] = np.broadcast_to(covc, (M, D, D)).reshape(-1)
        if self.A._init is None:
            am = np.zeros((D, D))
            aa = np.broadcast_to(np.diag(1.0 / amean), (D, D, D)).copy()
        elif self.A._init[0] == 'random':
            am = np.random.normal(size=(D, D)) / np.sqrt(amean)
            aa = am[:, :, None] * am[:, None, :]
        else:
            am = np.broadcast_to(np.asarray(self.A._init[1], dtype=np.float64), (D, D)).copy()
            aa = am[:, :, None] * am[:, None, :]
        st[L.off_Am:L.off_Am + D * D] = am.reshape(-1)
        st[L.off_AA:L.off_AA + D * D * D] = aa.reshape(-1)
        self.state = torch.from_numpy(st).to(rt.device)
        self.Yt = rt.empty(T * M * BL)
        self.Mw = torch.zeros(T * BL, dtype=torch.int64, device=rt.device)
        self.seqobs = rt.zeros(BL)
        self.Fw = rt.zeros(T * (NS + D) * BL)
        self.Z = rt.zeros(T * D * BL)
        self.Pm = rt.zeros(T * NS * BL)
        self._upload_y()
        if self.X._init is None:
            # q(X) = p(X | <A>, <nu>, mu0, Lam0): the smoother without the message from the
            # observations (expfamily.py:168-184), i.e. <tau> taken as 0 in the tables
            tau_mean = self.state[L.off_tau + 2].clone()
            self.state[L.off_tau + 2] = 0.0
            self._ops([OP_XPREP])
            self._smooth(given=False)
            self.state[L.off_tau + 2] = tau_mean
            prior_init = True
        else:
            prior_init = False
            x0 = self.X._init[1]
            if isinstance(x0, torch.Tensor):
                xd = x0.to(device=rt.device, dtype=torch.float64).reshape(B, T, D).contiguous()
            else:
                xd = torch.from_numpy(np.array(np.broadcast_to(np.asarray(x0, dtype=np.float64),
                                                               self.X.plates + (T, D))
                                               .reshape(B, T, D), order='C')).to(rt.device)
            k.x_layout(xd, D, B, T, BL, self.Z, True)
            del xd
            self._smooth(given=True)
        self._x_updated = prior_init
        self._x_rot = None
        self._ready = True
        self._version += 1

    def _upload_y(self):
        """data (M, [B,] T) + mask -> time-major Yt (zero where masked), mask words, counts (summed
        over the ranks).  Values at masked entries are never read: NaN placeholders are fine."""
        rt, k, L = self.rt, self.kernels, self.layout
        torch = rt.torch
        M, B, T, D = self.M, self.B, self.T, self.D
        y = self.Y._data
        if isinstance(y, torch.Tensor):
            yd = y.to(device=rt.device, dtype=torch.float64).expand(self.Y.plates) \
                .reshape(M, B, T).contiguous()
        else:
            yd = torch.from_numpy(np.array(np.broadcast_to(np.asarray(y, dtype=np.float64),
                                                           self.Y.plates).reshape(M, B, T),
                                           order='C')).to(rt.device)
        mask = self.Y._mask
        if hasattr(mask, 'tensor'):                        # DeviceMask: already in HBM
            md = mask.tensor.to(device=rt.device)
        else:
            md = torch.from_numpy(np.ascontiguousarray(np.asarray(mask, dtype=bool))).to(rt.device)
        # broadcastable to the plates (M, [B,] T): a view with stride 0 on the broadcast axes
        md = md.to(torch.uint8)
        while md.dim() < len(self.Y.plates):
            md = md.unsqueeze(0)
        if len(self.Y.plates) == 2:
            md = md.unsqueeze(1)
        md = md.expand(M, B, T)
        strides = [int(s) for s in md.stride()]
        k.prepare(yd, md, strides, M, B, T, self.BL, D, self.Yt, self.Mw, self.seqobs, self.state,
                  self.ws)
        del yd
        self._reduce(self.state[L.off_setup:L.off_setup + int(L.len_setup)])
        setup = self.state[L.off_setup:L.off_setup + int(L.len_setup)].cpu().numpy()
        self.n_obs, self.B_obs = float(setup[2:].sum()), float(setup[1])
        self.row_observed = setup[2:] > 0
        self.B_total = rt.all_reduce_int(B) if self.sharded else B

    def _reobserve(self):
        """Y.observe(new data / mask) AFTER updates: only Y changes (stochastic.py:223-273).  q(X) and
        every other posterior stay; the plate sums that involve the data and the mask (sum mask y^2,
        the counts, XX_m, Syx_m) are taken again from the stored <x>, <x x^T>.  The chain sums are
        weighted by "sequence has data": if that set changed the block restarts (with the warning
        of every plan whose state is discarded)."""
        self._flush()
        self.rt.sync_stream()
        before = self.seqobs.clone()
        self._upload_y()
        changed = int(not bool((before == self.seqobs).all().item()))
        if self.sharded:
            # every rank restarts or none does: the collectives of the two paths differ
            changed = self.rt.all_reduce_int(changed)
        if changed:
            _delta.warn_state_discarded(self, self.Y)
            self._ready = False
            self._pending = []
            return
        self._smooth(given=2)

    def _smooth(self, given):
        k, L = self.kernels, self.layout
        D, M, B, T = self.D, self.M, self.B, self.T
        k.x_update(given, self.Yt, self.Mw, self.seqobs, M, B, T, self.BL, D, self.state, self.Fw,
                   self.Z, self.Pm, self.ws)
        o = self._raw_offsets()
        lo = L.off_raw + (o['XX'] if given == 2 else 0)         # given = 2: chain sums untouched
        self._reduce(self.state[lo:L.off_raw + int(L.len_raw)])

    def _ops(self, ops):
        self.kernels.small_ops(self.D, self.M, self.T, self.priors, self.nu is not None, ops,
                               self.state)

    def update(self, node):
        self._materialize()
        if node is self.Y:
            return                 # the latent entries of Y are read-outs (get_moments)
        _delta.updated(self._delta, self.roles, node)
        code = {id(self.C): OP_C, id(self.gamma): OP_GAMMA, id(self.A): OP_A,
                id(self.alpha): OP_ALPHA, id(self.tau): OP_TAU}
        if self.nu is not None:
            code[id(self.nu)] = OP_NU
        if node is self.X:
            self._pending.append(OP_XPREP)
            self._flush()
            self.rt.sync_stream()
            self._smooth(given=False)
            self._x_updated = True
            self._x_rot = None
        elif id(node) in code:
            self._pending.append(code[id(node)])
        else:
            return
        self._version += 1

    def _lower_bound_terms(self):
        self._materialize()
        if self._L_version != self._version:
            L = self.layout
            self._pending.append(OP_ELBO)
            self._flush()
            host = self.state[L.off_scal:L.off_L + 16].cpu().numpy()
            status = host[0]
            if status != 0:
                self.state[L.off_scal] = 0.0
                _lib.raise_for_status(int(status))
            t = host[8:]
            self._L = dict(Y=t[0], C=t[1], A=t[2], X=t[3], gamma=t[4], alpha=t[5], tau=t[6], nu=t[7],
                           total=t[8])
            self._L_version = self._version
        return _delta.bound_terms(self._L, self._delta)

    # -- host views (reference shapes) ----------------------------------------------------------------
    def _plate_array(self, buf, nf):
        """(T, nf, BL) device array -> host (B, T, nf)."""
        a = buf.view(self.T, nf, self.BL)[:, :, :self.B].permute(2, 0, 1).contiguous()
        return a.cpu().numpy()

    def x_means(self):
        self._materialize()
        self._flush()
        return self._plate_array(self.Z, self.D)

    def x_second_moments(self):
        """<x_bt x_bt^T> (B, T, D, D)."""
        self._materialize()
        self._flush()
        return _sym_unpack(self._plate_array(self.Pm, self.NS), self.D)

    def chain_covariances(self):
        """Per sequence: V (B, T, D, D) and Cov(x_t, x_t+1) (B, T-1, D, D), from the forward
        quantities of the last X.update() (a read-out on the host; zero for point masses)."""
        B, T, D, NS = self.B, self.T, self.D, self.NS
        if not self._x_updated:
            return np.zeros((B, T, D, D)), np.zeros((B, max(T - 1, 0), D, D))
        L = self.layout
        x = self.x_means()
        P = self.x_second_moments()
        R = self._x_rot
        if R is not None:
            Ri = np.linalg.inv(R)
            x = x @ Ri.T
            P = np.einsum('ik,btkl,jl->btij', Ri, P, Ri)
        V = P - x[..., :, None] * x[..., None, :]
        Sinv = _sym_unpack(self._plate_array(self.Fw, NS + D)[..., :NS], D)
        to = self.state[L.off_tab:L.off_tab + int(L.len_tab)].cpu().numpy()
        E = to[3 * D * D:4 * D * D].reshape(D, D)
        Cn = np.empty((B, max(T - 1, 0), D, D))
        for t in range(T - 1):
            Cn[:, t] = -(Sinv[:, t] @ E) @ V[:, t + 1]
        if R is not None:
            V = np.einsum('ik,btkl,jl->btij', R, V, R)
            Cn = np.einsum('ik,btkl,jl->btij', R, Cn, R)
        return V, Cn

    def get_moments(self, node):
        self._materialize()
        self._flush()
        L = self.layout
        D, M, B, T = self.D, self.M, self.B, self.T
        st = self.state
        if node is self.C:
            cm = st[L.off_Cm:L.off_Cm + M * D].cpu().numpy().reshape(M, D)
            cov = st[L.off_CovC:L.off_CovC + M * D * D].cpu().numpy().reshape(M, D, D)
            u1 = cov + cm[:, :, None] * cm[:, None, :]
            return [cm.reshape(self.C.plates + (D,)), u1.reshape(self.C.plates + (D, D))]
        if node is self.X:
            if 8.0 * B * T * D * D > 8e9:
                raise MemoryError('X.u[1] would take %.0f GB on the host; use plan.x_means() and '
                                  'plan.x_second_moments()' % (8e-9 * B * T * D * D))
            x = self.x_means()
            u1 = self.x_second_moments()
            _, Cn = self.chain_covariances()
            u2 = Cn + x[:, :-1, :, None] * x[:, 1:, None, :]
            pl = self.X.plates
            return [x.reshape(pl + (T, D)), u1.reshape(pl + (T, D, D)),
                    u2.reshape(pl + (T - 1, D, D))]
        return super().get_moments(node)

    def get_mask(self, node):
        m = np.broadcast_to(np.asarray(self.Y._mask, dtype=bool), self.Y.plates)
        if node is self.Y or node is self.F:
            return m
        if node is self.C:
            return m.reshape(self.M, -1).any(axis=1).reshape(self.C.plates)
        if node is self.X:
            return m.reshape(self.M, self.B, self.T).any(axis=(0, 2)).reshape(self.X.plates) \
                if self.X.plates else np.array(bool(m.any()))
        if node is self.G:
            return m.any(axis=0)
        return np.array(bool(m.any()))

    # -- persistence ------------------------------------------------------------------------------------
    def save_state(self, put, nodes, index):
        self._materialize()
        self._flush()
        base = 'plans/%d/' % index
        _delta.save(put, base, self._delta)
        put(base + 'kind', np.array([ord(c) for c in 'lssm_masked'], dtype=np.uint8))
        put(base + 'dims', np.array([self.D, self.M, self.B, self.T, int(self.layout.total)],
                                    dtype=np.int64))
        put(base + 'state', self.state.cpu().numpy())
        put(base + 'X', self.x_means())
        put(base + 'P', self._plate_array(self.Pm, self.NS))
        put(base + 'F', self._plate_array(self.Fw, self.NS + self.D))
        put(base + 'x_updated', bool(self._x_updated))
        if self._x_rot is not None:
            put(base + 'x_rot', self._x_rot)

    def _put_plate_array(self, buf, nf, arr):
        torch = self.rt.torch
        a = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.rt.device)
        buf.view(self.T, nf, self.BL)[:, :, :self.B] = a.permute(1, 2, 0)

    def load_state(self, reader, nodes, index):
        self._materialize()
        base = 'plans/%d/' % index
        self._delta = _delta.load(reader, base)
        if not reader.has(base + 'state'):
            raise Exception("File does not contain the state of the masked state-space block")
        dims = tuple(int(v) for v in reader.get(base + 'dims'))
        want = (self.D, self.M, self.B, self.T, int(self.layout.total))
        if dims != want:
            raise ValueError('checkpoint is for (D, M, B, T, state length) = %s, the model has %s '
                             '(another model, or a checkpoint of an incompatible library build)'
                             % (dims, want))
        torch = self.rt.torch
        self.state.copy_(torch.from_numpy(np.array(reader.get(base + 'state'), dtype=np.float64)))
        self._put_plate_array(self.Z, self.D, np.array(reader.get(base + 'X')))
        self._put_plate_array(self.Pm, self.NS, np.array(reader.get(base + 'P')))
        self._put_plate_array(self.Fw, self.NS + self.D, np.array(reader.get(base + 'F')))
        self._x_updated = bool(reader.get(base + 'x_updated'))
        self._x_rot = np.array(reader.get(base + 'x_rot')) if reader.has(base + 'x_rot') else None
        self._version += 1

    # -- rotations (inference/transformations.py) ---------------------------------------------------------
    def _raw(self, st):
        L = self.layout
        return st[L.off_raw:L.off_raw + int(L.len_raw)]

    def _raw_offsets(self):
        D, M, NS = self.D, self.M, self.NS
        o = dict(sumP=0, Snp=NS, P0=NS + D * D, PT=2 * NS + D * D, x0=3 * NS + D * D)
        o['ld'] = o['x0'] + D
        o['XX'] = o['ld'] + 1
        o['Syx'] = o['XX'] + M * NS
        return o

    def rotation_statistics(self, node):
        st, L = self._host_state(), self.layout
        D, M, T, NS = self.D, self.M, self.T, self.NS
        if node is self.C:
            # XX over the rows that see data (the mask of C), the plate count over ALL rows: what
            # RotateGaussianARD uses (transformations.py:241-248: ``self.N = self.X.plates[0]
            # #np.sum(mask)``)
            return dict(XX=st[L.off_SCC:L.off_SCC + D * D].reshape(D, D).copy(), nplates=float(M))
        if node is self.X:
            raw, o = self._raw(st), self._raw_offsets()
            sumP = _sym_unpack(raw[o['sumP']:o['sumP'] + NS], D)
            P0 = _sym_unpack(raw[o['P0']:o['P0'] + NS], D)
            PT = _sym_unpack(raw[o['PT']:o['PT'] + NS], D)
            Snp = raw[o['Snp']:o['Snp'] + D * D].reshape(D, D)
            return dict(nvec=float(T) * self.B_obs, X0=raw[o['x0']:o['x0'] + D].copy(), X0X0=P0,
                        XnXn=sumP - P0, XpXn=Snp.T.copy(), XpXp=sumP - PT)
        raise NotImplementedError('rotation of %s' % node.name)

    def rotate_node(self, node, R, invR, logdetR, Q=None):
        st, L = self._host_state(), self.layout
        D, M, T, NS, DD = self.D, self.M, self.T, self.NS, self.D * self.D
        if node is self.X:
            if Q is not None:
                raise ValueError('the chain has no plate rotation')
            raw, o = self._raw(st), self._raw_offsets()
            for key in ('sumP', 'P0', 'PT'):
                a = _sym_unpack(raw[o[key]:o[key] + NS], D)
                raw[o[key]:o[key] + NS] = _sym_pack(R @ a @ R.T, D)
            raw[o['Snp']:o['Snp'] + DD] = (R @ raw[o['Snp']:o['Snp'] + DD].reshape(D, D) @ R.T).reshape(-1)
            raw[o['x0']:o['x0'] + D] = R @ raw[o['x0']:o['x0'] + D]
            xx = _sym_unpack(raw[o['XX']:o['XX'] + M * NS].reshape(M, NS), D)
            raw[o['XX']:o['XX'] + M * NS] = _sym_pack(np.einsum('ik,mkl,jl->mij', R, xx, R), D).reshape(-1)
            raw[o['Syx']:o['Syx'] + M * D] = (raw[o['Syx']:o['Syx'] + M * D].reshape(M, D) @ R.T).reshape(-1)
            # log|Phi_b| -> log|Phi_b| - 2 T log|R| for every sequence with data
            raw[o['ld']] -= 2.0 * T * logdetR * self.B_obs
            self._put_state(st)
            torch = self.rt.torch
            Rd = torch.from_numpy(np.ascontiguousarray(R, dtype=np.float64)).to(self.rt.device)
            self.kernels.rotate_x(D, T, self.B, self.BL, Rd, self.Z)
            # <x x^T> <- R <x x^T> R^T on the packed plate array, in place (vmp_lssmm_rotate_p)
            self.kernels.rotate_p(D, T, self.B, self.BL, Rd, self.Pm)
            self._x_rot = R if self._x_rot is None else R @ self._x_rot
            return
        if node is self.C:
            cm = st[L.off_Cm:L.off_Cm + M * D].reshape(M, D) @ R.T
            cov = np.einsum('ik,mkl,jl->mij', R, st[L.off_CovC:L.off_CovC + M * DD].reshape(M, D, D), R)
            st[L.off_Cm:L.off_Cm + M * D] = cm.reshape(-1)
            st[L.off_CovC:L.off_CovC + M * DD] = cov.reshape(-1)
            st[L.off_ldC:L.off_ldC + M] += 2.0 * logdetR
            ob = self.row_observed
            st[L.off_SCC:L.off_SCC + DD] = (cov[ob].sum(axis=0) + cm[ob].T @ cm[ob]).reshape(-1)
            self._put_state(st)
            return
        if node is self.A:
            am = st[L.off_Am:L.off_Am + DD].reshape(D, D)
            aa = st[L.off_AA:L.off_AA + D * DD].reshape(D, D, D)
            cov = aa - am[:, :, None] * am[:, None, :]
            am = am @ R.T
            cov = np.einsum('ik,dkl,jl->dij', R, cov, R)
            ld = st[L.off_ldA:L.off_ldA + D] + 2.0 * logdetR
            if Q is not None:
                sQ = Q.sum(axis=0)
                am = Q @ am
                cov = cov * (sQ * sQ)[:, None, None]
                ld = ld + 2.0 * D * np.log(np.abs(sQ))
            st[L.off_Am:L.off_Am + DD] = am.reshape(-1)
            st[L.off_AA:L.off_AA + D * DD] = (cov + am[:, :, None] * am[:, None, :]).reshape(-1)
            st[L.off_ldA:L.off_ldA + D] = ld
            self._put_state(st)
            return
        raise NotImplementedError('rotation of %s' % node.name)

    # -- measurement -------------------------------------------------------------------------------------
    def kernel_times_ms(self):
        t = self.kernels.pass_times_ms(64)
        if not t:
            return None
        n = float(len(t))
        return dict(lssmm_forward=sum(a for a, _ in t) / n,
                    lssmm_backward_and_stats=sum(b for _, b in t) / n)

    def cov_stationary_from(self):
        return None
