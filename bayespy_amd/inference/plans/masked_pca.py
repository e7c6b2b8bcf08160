"""
Execution plan of the probabilistic-PCA / factor-analysis block WITH MISSING VALUES

    Y = GaussianARD(SumMultiply('i,i', W, X), tau);  Y.observe(y, mask=array)

(bayespy/demos/pca.py:22-61 with the demo's default use, :80-82).  An array mask gives every
plate n and every row d its own K x K posterior.  The reference builds (1,N,K,K) second moments,
contracts them with einsum for both messages (dot.py:355,403,581), applies the mask by
multiplication (node.py:457-526) and factorises the N + D precision matrices in a Python loop
(utils/linalg.py:31-63).  Here ``X.update()`` walks the plates chunk by chunk, three kernels per
chunk on the fp64 matrix cores (``vmp_mpca_x_chunk``, bayespy_amd/csrc/vmp_mpca.hip); what
survives a chunk are the statistics ``M_d = sum_n m_dn <x x^T>_n`` (packed) and
``r_d = sum_n m_dn y_dn <x_n>`` -- all that ``W.update()``, ``tau.update()`` and the bound read,
and what ranks all-reduce when the plate is sharded.

HBM: ``Ymt`` (m*y, tile-major; values at masked entries -- NaN placeholders included -- are never
read), two bit layouts of the mask (1 bit per entry each), ``Xm`` (N, KP) posterior means, the
scratch of one chunk (``chunk`` x (LR + P) doubles), one state block (``vmp_mpca_layout``).
"""
import ctypes
import os

import numpy as np

from . import _delta

from ... import _lib
from ...device import get_runtime, ptr
from ...nodes.node import DeviceMask
from .pca import PCAPlan

OP_TAU, OP_ALPHA, OP_ELBO = 1, 2, 3
FIRST, FROM_VALUE, PRIOR, INSPECT = 1, 2, 4, 8
SC_SYY, SC_NOBS, SC_TRXX, SC_LDX, SC_N, SC_STATUS, SC_TAUX, SC_RESID = range(8)


class MaskedHIPKernels:
    """The C-ABI entry points of the block, bound to a runtime."""

    def __init__(self, rt):
        self.rt, self.lib, self.ctx = rt, rt.lib, rt.ctx

    def layout(self, D, K):
        L = _lib.MPCALayout()
        rc = self.lib.vmp_mpca_get_layout(D, K, ctypes.byref(L))
        if rc != _lib.VMP_OK:
            _lib.raise_for_status(rc, 'the fused missing-data PCA block supports D <= 128, K <= 32')
        return L

    def sizes(self, D, K, N, chunk):
        s = _lib.MPCASizes()
        self.rt.check(self.lib.vmp_mpca_sizes(self.ctx, D, K, N, chunk, ctypes.byref(s)))
        return s

    def init_state(self, D, K, a0t, b0t, a0a, b0a, state):
        self.rt.check(self.lib.vmp_mpca_init_state(self.ctx, D, K, a0t, b0t, a0a, b0a, ptr(state)))

    def prepare(self, Y, ldy, mask, ldm, N, D, K, Ymt, Mb1, Mb2, state, ws):
        self.rt.check(self.lib.vmp_mpca_prepare(
            self.ctx, ptr(Y), ldy, None if mask is None else ptr(mask), ldm, N, D, K, ptr(Ymt),
            ptr(Mb1), ptr(Mb2), ptr(state), ptr(ws)))

    def x_begin(self, D, K, n_total, state):
        self.rt.check(self.lib.vmp_mpca_x_begin(self.ctx, D, K, n_total, ptr(state)))

    def x_chunk(self, D, K, n0, nplates, flags, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state, ws):
        self.rt.check(self.lib.vmp_mpca_x_chunk(
            self.ctx, D, K, n0, nplates, flags, x_prec, ptr(Ymt), ptr(Mb1), ptr(Mb2), ptr(Xm),
            ptr(Lam), ptr(XXf), ptr(state), ptr(ws)))

    def x_pass(self, D, K, N, chunk, nsets, flags, x_prec, Ymt, Mb1, Mb2, Xm, Lam, XXf, state, ws):
        self.rt.check(self.lib.vmp_mpca_x_pass(
            self.ctx, D, K, N, chunk, nsets, flags, x_prec, ptr(Ymt), ptr(Mb1), ptr(Mb2), ptr(Xm),
            ptr(Lam), ptr(XXf), ptr(state), ptr(ws)))

    def update_w(self, D, K, mode, state):
        self.rt.check(self.lib.vmp_mpca_update_w(self.ctx, D, K, mode, ptr(state)))

    def small_ops(self, D, K, x_prec, a0t, b0t, a0a, b0a, ops, state):
        arr = (ctypes.c_int32 * len(ops))(*ops)
        self.rt.check(self.lib.vmp_mpca_small_ops(self.ctx, D, K, x_prec, a0t, b0t, a0a, b0a,
                                                  len(ops), arr, ptr(state)))

    def unpack_xx(self, D, K, nplates, XXf, out):
        self.rt.check(self.lib.vmp_mpca_unpack_xx(self.ctx, D, K, nplates, ptr(XXf), ptr(out)))

    def set_timing(self, on):
        self.rt.check(self.lib.vmp_ctx_set_timing(self.ctx, 1 if on else 0))

    def pass_times_ms(self, cap=64):
        a = (ctypes.c_double * cap)()
        b = (ctypes.c_double * cap)()
        n = ctypes.c_int32()
        self.rt.check(self.lib.vmp_pass_times_ms(self.ctx, a, b, cap, ctypes.byref(n)))
        return [(a[i], b[i]) for i in range(n.value)]


class MaskedPCAPlan:

    MAX_D, MAX_K = 128, 32

    @staticmethod
    def describe():
        return ("GaussianARD(SumMultiply('i,i', W, X), Gamma) observed with an array mask "
                "(missing values), D <= 128, K <= 32")

    # -- pattern matching: the graph of PCAPlan, an array mask on Y ------------------------------
    @staticmethod
    def unsupported_state(roles):
        if roles['Y']._mask is True:
            return 'Y is fully observed'
        for key in ('W', 'X', 'tau', 'alpha', 'F'):
            if getattr(roles[key], 'observed', False):
                return '%s is observed' % roles[key].name
        for key in ('W', 'X'):
            init = roles[key]._init
            if init is not None and init[0] not in ('value', 'random'):
                return '%s.initialize_from_%s' % (roles[key].name, init[0])
        for key in ('tau', 'alpha'):
            if roles[key]._init is not None:
                return '%s.initialize_from_%s' % (roles[key].name, roles[key]._init[0])
        return None

    @staticmethod
    def match(nodes, why=None):
        roles = PCAPlan.match_graph(nodes)
        if roles is None:
            return None
        D, N = roles['Y'].plates
        K = roles['W'].shape[0]
        if not roles['Y'].observed or roles['Y']._mask is True:
            return None
        if D > MaskedPCAPlan.MAX_D or K > MaskedPCAPlan.MAX_K:
            if why is not None:
                why.append('fused missing-data PCA block: D = %d, K = %d exceed its limits D <= %d, '
                           'K <= %d' % (D, K, MaskedPCAPlan.MAX_D, MaskedPCAPlan.MAX_K))
            return None
        bad = MaskedPCAPlan.unsupported_state(roles)
        if bad is not None:
            if why is not None:
                why.append('fused missing-data PCA block: %s' % bad)
            return None
        return roles

    # -- construction ------------------------------------------------------------------------------
    def __init__(self, roles, runtime=None, kernels=None, chunk=None):
        self.roles = roles
        self.Y, self.F, self.W, self.X = roles['Y'], roles['F'], roles['W'], roles['X']
        self.tau, self.alpha = roles['tau'], roles['alpha']
        self.D, self.N = self.Y.plates
        self.K = self.W.shape[0]
        self.a0t = self.tau.parents[0].scalar()
        self.b0t = self.tau.parents[1].scalar()
        self.a0a = self.alpha.parents[0].scalar()
        self.b0a = self.alpha.parents[1].scalar()
        self.x_prec = self.X.parents[1].scalar()
        if chunk is None:
            chunk = int(os.environ.get('BAYESPY_AMD_MPCA_CHUNK', str(1 << 20)))
        self.chunk = max(32, (int(chunk) + 31) // 32 * 32)
        self._rt, self._kernels = runtime, kernels
        self._ready = False
        self._version = 0
        self._L_version = -1
        self._L = None
        self._pending = []
        self.timing = False
        for n in roles.values():
            n._plan = self

    @property
    def rt(self):
        if self._rt is None:
            self._rt = get_runtime()
        return self._rt

    @property
    def kernels(self):
        if self._kernels is None:
            self._kernels = MaskedHIPKernels(self.rt)
        return self._kernels

    def nodes(self):
        return list(self.roles.values())

    def has_state(self):
        """Device state exists (a recompilation would discard it)."""
        return bool(self._ready)

    def invalidate(self, node):
        _delta.warn_state_discarded(self, node)
        self._ready = False
        self._version += 1
        self._pending = []
        if self.unsupported_state(self.roles) is not None:
            roles = dict(self.roles)
            if roles['Y']._mask is True and PCAPlan.unsupported_state(roles) is None:
                PCAPlan(roles)              # the mask is gone: the fully observed block
            else:
                from .generic import GenericPlan
                GenericPlan(self.nodes())

    # -- device state --------------------------------------------------------------------------------
    def _materialize(self):
        if self._ready:
            return
        self._delta = _delta.delta_roles(self.roles)    # point masses until their first update
        rt, k = self.rt, self.kernels
        torch = rt.torch
        D, N, K = self.D, self.N, self.K
        why = self.unsupported_state(self.roles)
        if why is not None:
            raise NotImplementedError('the fused missing-data PCA block does not cover this model '
                                      "state (%s); use VB(..., engine='generic')" % why)
        rt.sync_stream()
        self.layout = L = k.layout(D, K)
        self.KP, self.LR = int(L.KP), int(L.LR)
        self.sharded = any(getattr(n, '_shard_axis', None) is not None
                           for n in (self.X, self.F, self.Y))
        self.n_total = rt.all_reduce_int(N) if self.sharded else N
        self.chunk_eff = min(self.chunk, max(32, (N + 31) // 32 * 32))
        sz = k.sizes(D, K, N, self.chunk_eff)
        self.ntile = (N + 31) // 32
        # ---- data and mask: resident tensors are used in place, host arrays are staged --------
        y = self.Y._data
        if isinstance(y, torch.Tensor) and y.device == rt.device and y.dtype == torch.float64 \
                and tuple(y.shape) == (D, N) and y.stride(1) == 1:
            Yd, ldy = y, y.stride(0)
        else:
            if isinstance(y, torch.Tensor):
                Yd = y.to(device=rt.device, dtype=torch.float64).expand(D, N).contiguous()
            else:
                ya = np.array(np.broadcast_to(np.asarray(y, dtype=np.float64), (D, N)), order='C')
                Yd = torch.from_numpy(ya).to(rt.device)
            ldy = N
        m = self.Y._mask
        if N == 0:
            # an empty local plate (a rank of a sharded run without observations): one tile of
            # padding keeps every pointer valid; nothing of it is read
            Yd, ldy = rt.zeros(D, 32), 32
            Md = torch.zeros(D, 32, dtype=torch.uint8, device=rt.device)
        elif isinstance(m, DeviceMask):
            Md = m.tensor.expand(D, N).to(torch.uint8).contiguous()
        else:
            Md = torch.from_numpy(np.ascontiguousarray(
                np.broadcast_to(np.asarray(m, dtype=bool), (D, N)).astype(np.uint8))).to(rt.device)
        self.state = rt.zeros(int(L.total))
        self.ws = rt.empty(int(sz.workspace_doubles))
        self.Ymt = rt.empty(int(sz.ymt_doubles))
        self.Mb1 = torch.empty(int(sz.mask_words), dtype=torch.int32, device=rt.device)
        self.Mb2 = torch.empty(int(sz.mask_words), dtype=torch.int32, device=rt.device)
        self.Xm = rt.zeros(int(sz.xm_doubles)).view(-1, self.KP)
        # the scratch must hold finite values everywhere: pad plates are read with a zero mask
        # one chunk of scratch: the chunks of a pass run in order.  (Two sets + three streams,
        # vmp_mpca_x_pass with nsets = 2, pipeline the GEMMs of neighbouring chunks beside the
        # sweep; measured on MI355X it is 5-10 % SLOWER than in order -- the three kernels
        # compete for the same issue slots -- so it stays an opt-in for experiments.)
        self.nsets = 2 if (N > self.chunk_eff
                           and os.environ.get('BAYESPY_AMD_MPCA_PIPELINE') == '1') else 1
        self.Lam = rt.zeros(self.nsets * int(sz.lam_doubles))
        self.XXf = rt.zeros(self.nsets * int(sz.xxf_doubles))
        k.init_state(D, K, self.a0t, self.b0t, self.a0a, self.b0a, self.state)
        k.prepare(Yd, ldy, Md, N, N, D, K, self.Ymt, self.Mb1, self.Mb2, self.state, self.ws)
        del Md
        self._Ydev = Yd if Yd is y else None     # own staging copies are dropped after set-up
        sc = self.state[L.off_scal:L.off_scal + 2]
        self._reduce(sc)
        # observations per dimension: a row of W is an ignored plate only if NO rank observes it
        self._reduce(self.state[L.off_rowobs:L.off_rowobs + int(L.DP)])
        self._scal_host = None
        # ---- X: delta moments (value / random) or the prior; their statistics -----------------
        init = self.X._init
        flags = PRIOR
        if init is not None:
            flags = FROM_VALUE
            if init[0] == 'value':
                x0 = init[1]
                if isinstance(x0, torch.Tensor) and x0.device == rt.device:
                    self.Xm[:N, :K].copy_(x0.to(torch.float64).expand(self.X.plates + (K,))
                                          .reshape(N, K))
                else:
                    if isinstance(x0, torch.Tensor):
                        x0 = x0.detach().cpu().numpy()
                    x0 = np.array(np.broadcast_to(np.asarray(x0, dtype=np.float64),
                                                  self.X.plates + (K,)).reshape(N, K), order='C')
                    self.Xm[:N, :K].copy_(torch.from_numpy(x0))
            else:
                self.Xm[:N, :K].copy_(torch.randn(N, K, dtype=torch.float64, device=rt.device)
                                      * self.x_prec ** -0.5)
        self._x_updated = False
        self._x_rot = None          # rotation applied to q(X) since its last update
        k.x_begin(D, K, self.n_total, self.state)
        self._x_pass(flags)
        # ---- W: prior moments or a given value -----------------------------------------------------
        init = self.W._init
        if init is None:
            k.update_w(D, K, 2, self.state)
        else:
            if init[0] == 'value':
                w0 = init[1]
                if isinstance(w0, torch.Tensor):
                    w0 = w0.detach().cpu().numpy()
                w0 = np.broadcast_to(np.asarray(w0, dtype=np.float64),
                                     self.W.plates + (K,)).reshape(D, K)
            else:
                w0 = np.random.normal(size=(D, K)) * np.sqrt(self.b0a / self.a0a)
            wp = np.zeros((int(L.DP), self.KP))
            wp[:D, :K] = w0
            self.state[L.off_W:L.off_W + wp.size].copy_(torch.from_numpy(wp.reshape(-1)))
            k.update_w(D, K, 1, self.state)
        self._ready = True
        self._version += 1

    def _reduce(self, view):
        """Plate sum over the ranks (node.py:650, dot.py:581, expfamily.py:470-480)."""
        if self.sharded:
            self.rt.all_reduce_sum_(view)

    def _x_pass(self, flags):
        """All chunks of the plate; then the statistics are summed over ranks."""
        k, L = self.kernels, self.layout
        D, N, K = self.D, self.N, self.K
        k.x_pass(D, K, N, self.chunk_eff, self.nsets, flags, self.x_prec, self.Ymt, self.Mb1,
                 self.Mb2, self.Xm, self.Lam, self.XXf, self.state, self.ws)
        if self.sharded:
            self._reduce(self.state[L.off_M:L.off_M + int(L.DP) * self.LR])
            self._reduce(self.state[L.off_scal + SC_TRXX:L.off_scal + SC_LDX + 1])
            self._reduce(self.state[L.off_Sxx:L.off_Sxx + self.KP * self.KP])

    # -- node operations ---------------------------------------------------------------------------------
    def update(self, node):
        self._materialize()
        _delta.updated(self._delta, self.roles, node)
        k = self.kernels
        D, K = self.D, self.K
        if node is self.W:
            self._flush()
            self.rt.sync_stream()
            k.update_w(D, K, 0, self.state)
        elif node is self.X:
            self._flush()
            self.rt.sync_stream()
            k.x_begin(D, K, self.n_total, self.state)
            self._x_pass(0)
            self._x_updated = True
            self._x_rot = None
        elif node is self.tau:
            self._pending.append(OP_TAU)
        elif node is self.alpha:
            self._pending.append(OP_ALPHA)
        else:
            return          # Y: its latent entries are evaluated when read (get_moments)
        self._version += 1

    def _flush(self):
        if not self._pending:
            return
        ops, self._pending = self._pending, []
        self.rt.sync_stream()
        for i in range(0, len(ops), 8):
            self.kernels.small_ops(self.D, self.K, self.x_prec, self.a0t, self.b0t, self.a0a,
                                   self.b0a, ops[i:i + 8], self.state)

    def finish(self):
        if self._ready:
            self._flush()

    def _read_scalars(self):
        L = self.layout
        view = self.state[L.off_scal:L.off_L + 8]
        if self.rt.device.type != 'cuda':          # CPU kernel double of the host-logic tests
            return view.numpy().copy()
        if self._scal_host is None:
            self._scal_host = self.rt.torch.empty(view.numel(), dtype=self.rt.torch.float64,
                                                  pin_memory=True)
        self._scal_host.copy_(view, non_blocking=True)
        self.rt.torch.cuda.current_stream(self.rt.device).synchronize()
        return self._scal_host.numpy()

    def _lower_bound_terms(self):
        self._materialize()
        if self._L_version != self._version:
            self._pending.append(OP_ELBO)
            self._flush()
            host = self._read_scalars()
            status = host[SC_STATUS]
            if status != 0:
                self.state[self.layout.off_scal + SC_STATUS] = 0.0
                _lib.raise_for_status(int(status) if status < 0 else _lib.VMP_ERR_NOT_POSDEF)
            t = host[16:]
            self._L = dict(Y=float(t[0]), X=float(t[1]), W=float(t[2]), tau=float(t[3]),
                           alpha=float(t[4]), total=float(t[5]))
            self._L_version = self._version
        return _delta.bound_terms(self._L, self._delta)

    def lower_bound_contribution(self, node):
        terms = self._lower_bound_terms()
        for key in ('Y', 'X', 'W', 'tau', 'alpha'):
            if node is self.roles[key]:
                return terms[key]
        return 0.0

    def lower_bound(self):
        return self._lower_bound_terms()['total']

    # -- host views (reference shapes) ---------------------------------------------------------------------
    def _w_moments(self):
        L = self.layout
        D, K, KP = self.D, self.K, self.KP
        w = self.state[L.off_W:L.off_W + D * KP].cpu().numpy().reshape(D, KP)[:, :K].copy()
        ww = self.state[L.off_WW:L.off_WW + D * KP * KP].cpu().numpy() \
            .reshape(D, KP, KP)[:, :K, :K].copy()
        return w, ww

    def x_second_moments(self, n0=0, n1=None):
        """<x x^T>_n of the plates [n0, n1) as a host array (n, K, K), re-derived on the device from
        what the last X.update() saw of its Markov blanket (or from the initial value)."""
        self._materialize()
        self._flush()
        k = self.kernels
        D, N, K = self.D, self.N, self.K
        n1 = N if n1 is None else n1
        need = 8.0 * (max(n1 - n0, 0) + min(self.chunk_eff, N)) * K * K
        free = self.rt.torch.cuda.mem_get_info(self.rt.device)[0] if self.rt.device.type == 'cuda' \
            else float('inf')
        if need > 0.9 * free:
            raise MemoryError('the (n, K, K) second moments of plates [%d, %d) need %.1f GB on the '
                              'device (%.1f GB free); ask for a smaller plate range (ADVICE r02)'
                              % (n0, n1, need / 1e9, free / 1e9))
        out = self.rt.empty(max(n1 - n0, 0), K, K)
        init = self.X._init
        fresh = getattr(self, '_x_updated', False)
        flags = INSPECT | (0 if fresh else (PRIOR if init is None else FROM_VALUE))
        c0 = n0 // self.chunk_eff * self.chunk_eff
        for s in range(c0, n1, self.chunk_eff):
            npl = min(self.chunk_eff, N - s)
            k.x_chunk(D, K, s, npl, flags, self.x_prec, self.Ymt, self.Mb1, self.Mb2, self.Xm,
                      self.Lam, self.XXf, self.state, self.ws)
            tmp = self.rt.empty(npl, K, K)
            k.unpack_xx(D, K, npl, self.XXf, tmp)
            a, b = max(n0, s), min(n1, s + npl)
            out[a - n0:b - n0].copy_(tmp[a - s:b - s])
        out = out.cpu().numpy()
        if self._x_rot is not None and out.size:
            # q(X) was rotated after its update: <xx>_n -> R <xx>_n R^T
            out = np.einsum('ik,nkl,jl->nij', self._x_rot, out, self._x_rot)
        return out

    def get_moments(self, node):
        self._materialize()
        self._flush()
        L = self.layout
        D, N, K, KP = self.D, self.N, self.K, self.KP
        if node is self.W:
            w, ww = self._w_moments()
            return [w.reshape(self.W.plates + (K,)), ww.reshape(self.W.plates + (K, K))]
        if node is self.X:
            x = self.Xm[:N, :K].cpu().numpy()
            if 8.0 * N * K * K > 8e9:
                raise MemoryError('X.u[1] would take %.0f GB on the host; read slices with '
                                  'plan.x_second_moments(n0, n1)' % (8e-9 * N * K * K))
            xx = self.x_second_moments()
            return [x.reshape(self.X.plates + (K,)), xx.reshape(self.X.plates + (K, K))]
        if node is self.tau:
            t = self.state[L.off_tau:L.off_tau + 4].cpu().numpy()
            return [np.reshape(t[2], self.tau.plates), np.reshape(t[3], self.tau.plates)]
        if node is self.alpha:
            a = self.state[L.off_alpha:L.off_alpha + 4 * KP].cpu().numpy().reshape(4, KP)
            return [a[2, :K].reshape(self.alpha.plates), a[3, :K].reshape(self.alpha.plates)]
        if node is self.Y:
            # observed entries: the data; latent entries: q from the CURRENT <w>, <x>, <tau>
            # (the reference's value is as of Y's last update, which VB.update visits first)
            y = self.Ymt.view(self.ntile, int(L.DP), 32).permute(1, 0, 2) \
                .reshape(int(L.DP), -1)[:D, :N].cpu().numpy()
            m = np.broadcast_to(np.asarray(self.Y._mask, dtype=bool), (D, N))
            w, ww = self._w_moments()
            x = self.Xm[:N, :K].cpu().numpy()
            xx = self.x_second_moments()
            tau = float(self.state[L.off_tau + 2].item())
            f = w @ x.T
            f2 = np.einsum('dij,nij->dn', ww, xx)
            return [np.where(m, y, f), np.where(m, y * y, f2 + 1.0 / tau)]
        if node is self.F:
            # <f_dn> = <w_d>.<x_n>, <f_dn^2> = <ww>_d : <xx>_n (dot.py:316-415) -- a read-out on the
            # host (predictions at the missing entries); the updates never form these arrays
            if 8.0 * N * K * K > 8e9:
                raise MemoryError('F.u would need %.0f GB of <xx> on the host; form predictions from '
                                  'W.u[0] and X.u[0]' % (8e-9 * N * K * K))
            w, ww = self._w_moments()
            x = self.Xm[:N, :K].cpu().numpy()
            xx = self.x_second_moments()
            f = (w @ x.T).reshape(self.F.plates)
            f2 = np.einsum('dij,nij->dn', ww, xx).reshape(self.F.plates)
            return [f, f2]
        raise NotImplementedError('moments of %s are never materialised by the fused block'
                                  % node.name)

    def posterior_parameters(self, node):
        self._materialize()
        self._flush()
        L = self.layout
        K, KP = self.K, self.KP
        if node is self.tau:
            t = self.state[L.off_tau:L.off_tau + 2].cpu().numpy()
            return float(t[0]), float(t[1])
        if node is self.alpha:
            a = self.state[L.off_alpha:L.off_alpha + 2 * KP].cpu().numpy().reshape(2, KP)
            return a[0, :K].copy(), a[1, :K].copy()
        if node is self.W:
            w, ww = self._w_moments()
            return w, ww - w[:, :, None] * w[:, None, :]
        raise NotImplementedError

    def get_mask(self, node):
        m = np.asarray(self.Y._mask, dtype=bool)
        if node is self.Y or node is self.F:
            return np.array(np.broadcast_to(m, self.Y.plates))
        if node is self.W:
            return np.any(np.broadcast_to(m, self.Y.plates), axis=1, keepdims=True)
        if node is self.X:
            return np.any(np.broadcast_to(m, self.Y.plates), axis=0, keepdims=True)
        return np.array(True)

    # -- persistence ---------------------------------------------------------------------------------------
    def save_state(self, put, nodes, index):
        self._materialize()
        self._flush()
        base = 'plans/%d/' % index
        _delta.save(put, base, self._delta)
        put(base + 'kind', np.array([ord(c) for c in 'mpca'], dtype=np.uint8))
        put(base + 'dims', np.array([self.D, self.N, self.K], dtype=np.int64))
        put(base + 'state', self.state.cpu().numpy())
        put(base + 'X', self.Xm[:self.N, :self.K].cpu().numpy())
        put(base + 'x_updated', bool(getattr(self, '_x_updated', False)))
        for node in nodes:
            if node in (self.W, self.tau, self.alpha):
                for i, ui in enumerate(self.get_moments(node)):
                    put('nodes/%s/u%d' % (node.name, i), ui)
                put('nodes/%s/observed' % node.name, False)

    def load_state(self, reader, nodes, index):
        self._materialize()
        base = 'plans/%d/' % index
        self._delta = _delta.load(reader, base)
        if not reader.has(base + 'state'):
            raise Exception("File does not contain the state of the fused missing-data PCA block")
        dims = tuple(int(v) for v in reader.get(base + 'dims'))
        if dims != (self.D, self.N, self.K):
            raise ValueError('checkpoint is for (D, N, K) = %s, the model has %s'
                             % (dims, (self.D, self.N, self.K)))
        torch = self.rt.torch
        self.state.copy_(torch.from_numpy(np.array(reader.get(base + 'state'), dtype=np.float64)))
        self.Xm[:self.N, :self.K].copy_(torch.from_numpy(np.array(reader.get(base + 'X'),
                                                                 dtype=np.float64)))
        self._x_updated = bool(reader.get(base + 'x_updated'))
        self._version += 1

    # -- rotations (inference/transformations.py; demos/pca.py:85-94 rotates a model with missing
    #    values in the VB callback) ---------------------------------------------------------------
    def gamma_posterior_shape(self, node):
        return self.posterior_parameters(node)[0]

    def _tri_index(self):
        K = self.K
        i, j = np.meshgrid(np.arange(K), np.arange(K), indexing='ij')
        a, b = np.maximum(i, j), np.minimum(i, j)
        return a * (a + 1) // 2 + b                      # packed position of (i, j)

    def rotation_statistics(self, node):
        """sum over the plates of <x x^T> (K x K, global over ranks) and the plate count."""
        self._materialize()
        self._flush()
        L = self.layout
        K, KP = self.K, self.KP
        if node is self.W:
            _, ww = self._w_moments()
            return dict(XX=ww.sum(axis=0), nplates=self.D)
        if node is self.X:
            sxx = self.state[L.off_Sxx:L.off_Sxx + KP * KP].cpu().numpy().reshape(KP, KP)[:K, :K]
            return dict(XX=0.5 * (sxx + sxx.T), nplates=self.n_total)
        raise NotImplementedError('rotation of %s' % node.name)

    def _panel_from_moments(self, w, ww):
        """The B-operand panel of the precision GEMM (fragment order, vmp_mpca.hip) from <w_d>,
        <w_d w_d^T> on the host (set-up / rotation only; W.update() writes it on the device)."""
        L = self.layout
        D, K, KP, PT = self.D, self.K, self.KP, int(L.PT)
        DQ = int(L.DP) // 4
        CT = PT + KP // 16
        panel = np.zeros(CT * (DQ // 2) * 128)
        d = np.arange(D)
        q = d >> 2

        def put(c, col16, vals):                          # vals[d] -> element (d, 16 c + col16)
            idx = ((c * (DQ // 2) + (q >> 1)) * 64 + (d & 3) * 16 + col16) * 2 + (q & 1)
            panel[idx] = vals
        for i in range(K):
            for j in range(i + 1):
                p = i * (i + 1) // 2 + j
                put(p >> 4, p & 15, ww[:, i, j])
        for k in range(K):
            put(PT + (k >> 4), k & 15, w[:, k])
        return panel

    def rotate_node(self, node, R, invR, logdetR):
        """q(node) <- the distribution of R x (gaussian.py:1693-1741).  The K x K / D x K state is
        rotated on the host (O(D K^3)); the (N, K) array of <x_n> on the device through the fp64
        MFMA contraction kernel.  <xx>_n itself is never stored: its plate sums M_d, sum_n <xx>_n
        rotate exactly, and later read-outs of X.u[1] apply the accumulated rotation."""
        self._materialize()
        self._flush()
        rt, L = self.rt, self.layout
        torch = rt.torch
        D, K, KP, DP, LR, PT = self.D, self.K, self.KP, int(L.DP), self.LR, int(L.PT)
        if node is self.W:
            w, ww = self._w_moments()
            w = w @ R.T
            ww = np.einsum('ik,dkl,jl->dij', R, ww, R)
            wp = np.zeros((DP, KP))
            wp[:D, :K] = w
            wwp = np.zeros((DP, KP, KP))
            wwp[:D, :K, :K] = ww
            self.state[L.off_W:L.off_W + wp.size].copy_(torch.from_numpy(wp.reshape(-1)))
            self.state[L.off_WW:L.off_WW + wwp.size].copy_(torch.from_numpy(wwp.reshape(-1)))
            ld = self.state[L.off_ldW:L.off_ldW + D].cpu().numpy() + 2.0 * logdetR
            self.state[L.off_ldW:L.off_ldW + D].copy_(torch.from_numpy(ld))
            panel = self._panel_from_moments(w, ww)
            self.state[L.off_panel:L.off_panel + panel.size].copy_(torch.from_numpy(panel))
        elif node is self.X:
            from ...darray import DArray
            from ...utils import linalg
            N = self.N
            if N:
                xn = linalg.mmdot(DArray(self.Xm[:N, :K]), DArray.from_host(np.ascontiguousarray(R.T)))
                self.Xm[:N, :K].copy_(xn.t)
            tri = self._tri_index()
            mst = self.state[L.off_M:L.off_M + DP * LR].cpu().numpy().reshape(DP, LR).copy()
            M = mst[:D][:, tri]                                      # (D, K, K)
            M = np.einsum('ik,dkl,jl->dij', R, M, R)
            r = mst[:D, 16 * PT:16 * PT + K] @ R.T
            iu = np.tril_indices(K)
            mst[:D, iu[0] * (iu[0] + 1) // 2 + iu[1]] = M[:, iu[0], iu[1]]
            mst[:D, 16 * PT:16 * PT + K] = r
            self.state[L.off_M:L.off_M + DP * LR].copy_(torch.from_numpy(mst.reshape(-1)))
            sxx = self.state[L.off_Sxx:L.off_Sxx + KP * KP].cpu().numpy().reshape(KP, KP)
            sx = np.zeros((KP, KP))
            sx[:K, :K] = R @ sxx[:K, :K] @ R.T
            self.state[L.off_Sxx:L.off_Sxx + KP * KP].copy_(torch.from_numpy(sx.reshape(-1)))
            sc = self.state[L.off_scal:L.off_scal + 8].cpu().numpy()
            sc[SC_TRXX] = float(np.trace(sx))
            sc[SC_LDX] += 2.0 * self.n_total * logdetR
            self.state[L.off_scal:L.off_scal + 8].copy_(torch.from_numpy(sc))
            self._x_rot = R if self._x_rot is None else R @ self._x_rot
        else:
            raise NotImplementedError('rotation of %s' % node.name)
        self._version += 1

    # -- measurement -----------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._materialize()
        self.kernels.set_timing(on)
        self.timing = on

    def kernel_times_ms(self):
        """Average duration per chunk of the three plate kernels (HIP events on their stream)."""
        t = self.kernels.pass_times_ms(64)
        if not t:
            return None
        lam = [a for (a, b) in t[0::2]]
        swp = [b for (a, b) in t[0::2]]
        sts = [a for (a, b) in t[1::2]]
        red = [b for (a, b) in t[1::2]]
        n = float(len(lam))
        return dict(mpca_lambda=sum(lam) / n, mpca_sweep=sum(swp) / n,
                    mpca_stats=sum(sts) / max(len(sts), 1), reduce=sum(red) / max(len(red), 1),
                    chunk_plates=self.chunk_eff)
