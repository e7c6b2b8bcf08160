"""
Execution plan of the linear state-space model block (BASELINE.json config 5)

    A = GaussianARD(0, alpha, shape=(D,), plates=(D,));  X = GaussianMarkovChain(mu0, Lam0, A, nu, n=T, plates=(B,))
    C = GaussianARD(0, gamma, shape=(D,), plates=(M,1,1));  Y = GaussianARD(SumMultiply('i,i', C, X), tau)

(bayespy/demos/lssm.py:34-103 with a plate of sequences), Y fully observed.  Dynamics, noise and
mask are shared by the sequences, so the block-tridiagonal precision of q(X_b) is one matrix: its
covariance recursion (linalg.block_banded_solve, utils/linalg.py:468-575) runs ONCE per X.update()
(``vmp_lssm_cov``), the per-sequence mean recursions run one thread per sequence over time-major
arrays (``vmp_lssm_smooth``), and every other node and the bound read plate sums
(bayespy_amd/csrc/vmp_lssm.hip; formulas pinned in oracle/lssm.py).  No (B,T,D,D) array exists.

HBM: ``Yt`` (T, M, BL) the data re-laid-out time-major once, ``Z`` (T, D, BL) the posterior means
(forward pass writes z_t, backward pass overwrites with <x_t>), (T, D, D) shared recursion
matrices, one small state block (``vmp_lssm_layout``).
"""
import ctypes

import numpy as np

from . import _delta

from ... import _lib
from ...device import get_runtime, ptr
from ...nodes.node import Constant
from ...nodes.gamma import Gamma
from ...nodes.gaussian import GaussianARD
from ...nodes.dot import SumMultiply
from ...nodes.gaussian_markov_chain import GaussianMarkovChain, MarkovChainToGaussian

(OP_STATS, OP_C, OP_GAMMA, OP_XPREP, OP_A, OP_ALPHA, OP_TAU, OP_NU, OP_ELBO) = range(1, 10)


def _const_scalar(node):
    return isinstance(node, Constant) and node.is_scalar()


def _gamma_const(node, plates):
    return (isinstance(node, Gamma) and _const_scalar(node.parents[0])
            and _const_scalar(node.parents[1]) and tuple(node.plates) == tuple(plates))


def _const_zero(node):
    return isinstance(node, Constant) and not np.any(node.value)


class LSSMKernels:

    def __init__(self, rt):
        self.rt, self.lib, self.ctx = rt, rt.lib, rt.ctx

    def layout(self, D, M):
        L = _lib.LSSMLayout()
        rc = self.lib.vmp_lssm_get_layout(D, M, ctypes.byref(L))
        if rc != _lib.VMP_OK:
            _lib.raise_for_status(rc, 'the fused LSSM block supports D <= 16 states and M <= 64 '
                                      'observed dimensions')
        return L

    def workspace_doubles(self, D, M, B, T):
        n = ctypes.c_int64()
        self.rt.check(self.lib.vmp_lssm_workspace_doubles(D, M, B, T, ctypes.byref(n)))
        return n.value

    def relayout_y(self, Y, M, B, T, BL, Yt, syy, ws):
        self.rt.check(self.lib.vmp_lssm_relayout_y(self.ctx, ptr(Y), M, B, T, BL, ptr(Yt), ptr(syy),
                                                   ptr(ws)))

    def x_layout(self, X, D, B, T, BL, Z, to_time_major):
        self.rt.check(self.lib.vmp_lssm_x_layout(self.ctx, ptr(X), D, B, T, BL, ptr(Z),
                                                 1 if to_time_major else 0))

    def cov(self, T, D, Dg, Sinv, J, sums):
        DD = D * D
        self.rt.check(self.lib.vmp_lssm_cov(self.ctx, T, D, ptr(Dg[0:]), ptr(Dg[DD:]),
                                            ptr(Dg[2 * DD:]), ptr(Dg[3 * DD:]), ptr(Sinv), ptr(J),
                                            ptr(sums)))

    def smooth(self, given, Yt, M, B, T, BL, D, Cm, tau, h0, Sinv, J, Z, stats, ws):
        self.rt.check(self.lib.vmp_lssm_smooth(self.ctx, 1 if given else 0, ptr(Yt), M, B, T, BL, D,
                                               ptr(Cm), ptr(tau), ptr(h0), ptr(Sinv), ptr(J),
                                               ptr(Z), ptr(stats), ptr(ws)))

    def x_update(self, T, D, Dg, Sinv, J, sums, Yt, M, B, BL, Cm, tau, h0, Z, stats, ws):
        DD = D * D
        self.rt.check(self.lib.vmp_lssm_x_update(
            self.ctx, T, D, ptr(Dg[0:]), ptr(Dg[DD:]), ptr(Dg[2 * DD:]), ptr(Dg[3 * DD:]),
            ptr(Sinv), ptr(J), ptr(sums), ptr(Yt), M, B, BL, ptr(Cm), ptr(tau), ptr(h0), ptr(Z),
            ptr(stats), ptr(ws)))

    def rotate_x(self, D, T, B, BL, R, Z):
        self.rt.check(self.lib.vmp_lssm_rotate_x(self.ctx, D, T, B, BL, ptr(R), ptr(Z)))

    def small_ops(self, D, M, T, B_total, priors, nu_latent, ops, state):
        pr = (ctypes.c_double * 8)(*priors)
        arr = (ctypes.c_int32 * len(ops))(*ops)
        self.rt.check(self.lib.vmp_lssm_small_ops(self.ctx, D, M, T, float(B_total), pr,
                                                  1 if nu_latent else 0, len(ops), arr, ptr(state)))

    def set_timing(self, on):
        self.rt.check(self.lib.vmp_ctx_set_timing(self.ctx, 1 if on else 0))

    def pass_times_ms(self, cap=64):
        a = (ctypes.c_double * cap)()
        b = (ctypes.c_double * cap)()
        n = ctypes.c_int32()
        self.rt.check(self.lib.vmp_pass_times_ms(self.ctx, a, b, cap, ctypes.byref(n)))
        return [(a[i], b[i]) for i in range(n.value)]


class LSSMPlan:

    _label = 'fused state-space block'

    @staticmethod
    def _accepts_mask(Y):
        """This block takes the model when Y is fully observed (the masked block reports the
        near misses of models with array masks)."""
        return Y._mask is True

    @staticmethod
    def describe():
        return ("GaussianARD(SumMultiply('i,i', C, GaussianMarkovChain(mu0, Lam0, A, nu)), tau) fully "
                "observed, shared dynamics, D <= 16 states")

    # -- pattern matching ---------------------------------------------------------------------------
    @staticmethod
    def unsupported_state(r):
        if r['Y']._mask is not True or not r['Y'].observed:
            return 'Y must be fully observed'
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

    @staticmethod
    def _limits():
        mx_d, mx_m = ctypes.c_int32(), ctypes.c_int32()
        _lib.load().vmp_lssm_limits(ctypes.byref(mx_d), ctypes.byref(mx_m))
        return mx_d.value, mx_m.value

    _dims_note = ''

    @staticmethod
    def _dims_ok(D, M):
        return True

    @classmethod
    def match(cls, nodes, why=None):
        def no(Y, msg):
            if why is not None:
                why.append('%s, observed node %s: %s'
                           % (cls._label, Y.name or '<unnamed>', msg))
        if any(any(m != 1 for m in n.plates_multiplier) for n in nodes):
            return None
        for Y in nodes:
            if not isinstance(Y, GaussianARD) or Y.ndim != 0:
                continue
            F, tau = Y.parents
            if not isinstance(F, SumMultiply) or not _gamma_const(tau, tau.plates) \
                    or any(p != 1 for p in tau.plates):
                continue
            if len(F.parents) != 2 or F.out_keys != [] or F.in_keys[0] != F.in_keys[1] \
                    or len(F.in_keys[0]) != 1:
                continue
            P, Q = F.parents
            if isinstance(P, MarkovChainToGaussian):
                P, Q = Q, P
            if not (isinstance(Q, MarkovChainToGaussian) and isinstance(P, GaussianARD)):
                continue
            G, C = Q, P
            X = G.parents[0]
            if type(X) is not GaussianMarkovChain or len(X.plates) > 1:
                no(Y, 'the chain is a %s with plates %s (a plain GaussianMarkovChain with at most '
                      'one plate axis is needed)' % (type(X).__name__, tuple(X.plates)))
                continue
            D, T = X.D, X.N
            mu, Lam, A, nu = X.parents
            if not (isinstance(mu, Constant) and isinstance(Lam, Constant)):
                continue
            if np.shape(mu.value) != (D,) or np.shape(Lam.value) != (D, D):
                continue
            if not (isinstance(A, GaussianARD) and A.ndim == 1 and A.shape == (D,)
                    and tuple(A.plates) == (D,) and _const_zero(A.parents[0])):
                continue
            alpha = A.parents[1]
            if not _gamma_const(alpha, (D,)):
                continue
            if isinstance(nu, Constant):
                if np.shape(nu.value) != (D,):
                    continue
                nu_node = None
            elif _gamma_const(nu, (D,)):
                nu_node = nu
            else:
                continue
            Bp = tuple(X.plates)
            if C.ndim != 1 or C.shape != (D,) or not _const_zero(C.parents[0]):
                continue
            M = C.plates[0] if len(C.plates) == len(Bp) + 2 else None
            if M is None or tuple(C.plates) != (M,) + (1,) * (len(Bp) + 1):
                continue
            gamma = C.parents[1]
            if not _gamma_const(gamma, (D,)):
                continue
            if tuple(Y.plates) != (M,) + Bp + (T,):
                continue
            try:
                mx_d, mx_m = cls._limits()
            except Exception:       # noqa: BLE001
                continue
            if D > mx_d or M > mx_m or not cls._dims_ok(D, M):
                if cls._accepts_mask(Y):
                    no(Y, 'D = %d states, M = %d observed dimensions exceed the limits of the '
                          'block (D <= %d, M <= %d%s)' % (D, M, mx_d, mx_m, cls._dims_note))
                continue
            priv = [C, gamma, X, A, alpha, tau, F, G] + ([nu_node] if nu_node is not None else [])
            if any(len(n.children) != 1 for n in priv):
                continue
            roles = dict(Y=Y, F=F, G=G, C=C, gamma=gamma, X=X, A=A, alpha=alpha, tau=tau)
            if nu_node is not None:
                roles['nu'] = nu_node
            bad = cls.unsupported_state(roles)
            if bad is not None:
                if cls._accepts_mask(Y):
                    no(Y, bad)
                continue
            return roles
        return None

    # -- construction -----------------------------------------------------------------------------------
    def __init__(self, roles, runtime=None, kernels=None):
        self.roles = roles
        for k, v in roles.items():
            setattr(self, k, v)
        self.nu = roles.get('nu')
        self.D, self.T = self.X.D, self.X.N
        self.M = self.C.plates[0]
        self.B = self.X.plates[0] if self.X.plates else 1
        self.mu0 = np.asarray(self.X.parents[0].value, dtype=np.float64)
        self.Lam0 = np.asarray(self.X.parents[1].value, dtype=np.float64)
        self.nu_const = None if self.nu is not None else \
            np.asarray(self.X.parents[3].value, dtype=np.float64)

        def prior(n):
            return [n.parents[0].scalar(), n.parents[1].scalar()]
        self.priors = prior(self.tau) + prior(self.gamma) + prior(self.alpha) + \
            (prior(self.nu) if self.nu is not None else [1.0, 1.0])
        self._rt, self._kernels = runtime, kernels
        self._ready = False
        self._version = 0
        self._L_version = -1
        self._L = None
        self._pending = []
        self._x_updated = False
        self._x_rot = None
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
            self._kernels = LSSMKernels(self.rt)
        return self._kernels

    def nodes(self):
        return list(self.roles.values())

    def has_state(self):
        """Device state exists (a recompilation would discard it)."""
        return bool(self._ready)

    def invalidate(self, node):
        if node is self.Y and self._ready and self._version > 1 and self.Y._mask is True \
                and self.unsupported_state(self.roles) is None:
            self._reobserve()
            self._version += 1
            return
        _delta.warn_state_discarded(self, node)
        self._ready = False
        self._version += 1
        self._pending = []
        if self.unsupported_state(self.roles) is not None:
            # an array mask: the masked block when it covers the sizes; anything else: the
            # generic device message-passing engine
            from .lssm_masked import MaskedLSSMPlan
            roles = MaskedLSSMPlan.match(self.nodes())
            if roles is not None:
                MaskedLSSMPlan(roles)
            else:
                from .generic import GenericPlan
                GenericPlan(self.nodes())

    # -- device state -------------------------------------------------------------------------------------
    def _gamma_init(self, node, a0, b0, n):
        """(a, b, mean, log-mean) rows of a Gamma node: prior or delta moments of a value."""
        from scipy.special import digamma
        out = np.zeros((4, n))
        init = node._init
        if init is None:
            out[0], out[1] = a0, b0
            out[2], out[3] = a0 / b0, digamma(a0) - np.log(b0)
        else:
            v = np.broadcast_to(np.asarray(init[1], dtype=np.float64).reshape(-1), (n,))
            out[0], out[1] = a0, b0                   # unused until the node is updated
            out[2], out[3] = v, np.log(v)
        return out.reshape(-1)

    def _materialize(self):
        if self._ready:
            return
        self._delta = _delta.delta_roles(self.roles)    # point masses until their first update
        rt, k = self.rt, self.kernels
        torch = rt.torch
        D, M, B, T = self.D, self.M, self.B, self.T
        why = self.unsupported_state(self.roles)
        if why is not None:
            raise NotImplementedError('the fused LSSM block does not cover this model state (%s); '
                                      "use VB(..., engine='generic')" % why)
        rt.sync_stream()
        self.layout = L = k.layout(D, M)
        self.sharded = any(getattr(n, '_shard_axis', None) is not None
                           for n in (self.X, self.G, self.F, self.Y))
        self.B_total = rt.all_reduce_int(B) if self.sharded else B
        self.BL = BL = (B + 63) // 64 * 64
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
        # C, A: delta moments of a value or the prior moments (mean 0, covariance diag(1/prec))
        gmean = st[L.off_gamma + 2 * D:L.off_gamma + 3 * D]
        amean = st[L.off_alpha + 2 * D:L.off_alpha + 3 * D]
        if self.C._init is None:
            cm, covc = np.zeros((M, D)), np.diag(1.0 / gmean)
        elif self.C._init[0] == 'random':
            # initialize_from_random (expfamily.py:206-212): delta moments of a draw from the
            # current q = the prior N(0, diag(1 / <gamma>)); host-side set-up, NumPy's global stream
            cm, covc = np.random.normal(size=(M, D)) / np.sqrt(gmean), np.zeros((D, D))
        else:
            cm = np.broadcast_to(np.asarray(self.C._init[1], dtype=np.float64),
                                 self.C.plates + (D,)).reshape(M, D)
            covc = np.zeros((D, D))
        st[L.off_Cm:L.off_Cm + M * D] = cm.reshape(-1)
        st[L.off_CovC:L.off_CovC + D * D] = covc.reshape(-1)
        st[L.off_SCC:L.off_SCC + D * D] = (M * covc + cm.T @ cm).reshape(-1)
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
        self._upload_y()
        self.Z = rt.zeros(T * D * BL)
        self.Sinv = rt.zeros(T * D * D)
        self.J = rt.zeros(max(T - 1, 1) * D * D)
        if self.X._init is None:
            # ---- X from its prior (expfamily.py:168-184): q(X) = p(X | <A>, <nu>, mu0, Lam0), i.e.
            # the smoother without the message from the observations (<tau> taken as 0) -----------
            tau_mean = self.state[L.off_tau + 2].clone()
            self.state[L.off_tau + 2] = 0.0
            self._ops([OP_XPREP])
            self._smooth(given=False)
            self.state[L.off_tau + 2] = tau_mean
            prior_init = True
        else:
            prior_init = False
            # ---- X: delta moments of the given value; their plate sums -----------------------------------
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
        self._x_updated = prior_init          # a proper q(X) (covariances exist), not delta moments
        self._x_rot = None                    # rotation applied to q(X) since its last update
        self._ready = True
        self._version += 1

    def _upload_y(self):
        """data: (M, [B,] T) -> time-major Yt, sum y^2 (summed over the ranks)."""
        rt, k, L = self.rt, self.kernels, self.layout
        torch = rt.torch
        M, B, T = self.M, self.B, self.T
        y = self.Y._data
        if isinstance(y, torch.Tensor):
            yd = y.to(device=rt.device, dtype=torch.float64).reshape(M, B, T).contiguous()
        else:
            yd = torch.from_numpy(np.array(np.broadcast_to(np.asarray(y, dtype=np.float64),
                                                           self.Y.plates).reshape(M, B, T),
                                           order='C')).to(rt.device)
        k.relayout_y(yd, M, B, T, self.BL, self.Yt, self.state[L.off_scal:L.off_scal + 1], self.ws)
        del yd
        self._reduce(self.state[L.off_scal:L.off_scal + 1])

    def _reobserve(self):
        """Y.observe(new data) AFTER updates: only Y changes (stochastic.py:223-273).  q(X) and every
        other posterior stay; the plate sums that involve the data (sum y^2, sum y<x>^T) are taken
        again from the new data and the current <x> (the statistics pass in its "given <x>" form;
        the covariance sums of q(X) are kept)."""
        self._flush()
        self.rt.sync_stream()
        self._upload_y()
        self._smooth(given=True, keep_cov=self._x_updated)

    def _reduce(self, view):
        if self.sharded:
            self.rt.all_reduce_sum_(view)

    def _smooth(self, given, keep_cov=False):
        k, L = self.kernels, self.layout
        D, M, B, T = self.D, self.M, self.B, self.T
        st = self.state
        if given:
            if not keep_cov:
                st[L.off_covsums:L.off_covsums + 5 * D * D + 8].zero_()
            k.smooth(True, self.Yt, M, B, T, self.BL, D, st[L.off_Cm:], st[L.off_scal + 3:],
                     st[L.off_h0:], self.Sinv, self.J, self.Z, st[L.off_raw:], self.ws)
        else:
            # covariance recursion + per-sequence passes; the backward half of the recursion runs
            # beside the passes on a side stream inside the library
            k.x_update(T, D, st[L.off_Dg:L.off_Dg + 4 * D * D], self.Sinv, self.J,
                       st[L.off_covsums:L.off_covsums + 5 * D * D + 8], self.Yt, M, B, self.BL,
                       st[L.off_Cm:], st[L.off_scal + 3:], st[L.off_h0:], self.Z, st[L.off_raw:],
                       self.ws)
        self._reduce(st[L.off_raw:L.off_raw + int(L.len_raw)])
        self._ops([OP_STATS])

    def _ops(self, ops):
        self.kernels.small_ops(self.D, self.M, self.T, self.B_total, self.priors,
                               self.nu is not None, ops, self.state)

    # -- node operations -------------------------------------------------------------------------------------
    def update(self, node):
        self._materialize()
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

    def _flush(self):
        if not self._pending:
            return
        ops, self._pending = self._pending, []
        self.rt.sync_stream()
        for i in range(0, len(ops), 12):
            self._ops(ops[i:i + 12])

    def finish(self):
        if self._ready:
            self._flush()

    def _lower_bound_terms(self):
        self._materialize()
        if self._L_version != self._version:
            L = self.layout
            self._pending.append(OP_ELBO)
            self._flush()
            host = self.state[L.off_scal:L.off_L + 16].cpu().numpy()
            status = host[2]
            if status != 0:
                self.state[L.off_scal + 2] = 0.0
                _lib.raise_for_status(int(status))
            t = host[8:]
            self._L = dict(Y=t[0], C=t[1], A=t[2], X=t[3], gamma=t[4], alpha=t[5], tau=t[6], nu=t[7],
                           total=t[8])
            self._L_version = self._version
        return _delta.bound_terms(self._L, self._delta)

    def lower_bound_contribution(self, node):
        terms = self._lower_bound_terms()
        for key in ('Y', 'C', 'A', 'X', 'gamma', 'alpha', 'tau', 'nu'):
            if node is self.roles.get(key):
                return float(terms[key])
        return 0.0

    def lower_bound(self):
        return float(self._lower_bound_terms()['total'])

    # -- host views (reference shapes) ---------------------------------------------------------------------------
    def _gamma_view(self, off, n, plates):
        g = self.state[off:off + 4 * n].cpu().numpy().reshape(4, n)
        return [g[2].reshape(plates), g[3].reshape(plates)]

    def chain_covariances(self):
        """V_t = Cov(x_t) (T,D,D) and Cov(x_t, x_t+1) (T-1,D,D), shared by the sequences, from the
        recursion matrices of the last X.update() (zero for a delta-initialised X)."""
        T, D = self.T, self.D
        if not self._x_updated:
            return np.zeros((T, D, D)), np.zeros((max(T - 1, 0), D, D))
        Sinv = self.Sinv.cpu().numpy().reshape(T, D, D)
        J = self.J.cpu().numpy().reshape(-1, D, D)
        V = np.empty((T, D, D))
        Cn = np.empty((max(T - 1, 0), D, D))
        V[T - 1] = Sinv[T - 1]
        for t in range(T - 2, -1, -1):
            Cn[t] = -J[t] @ V[t + 1]
            V[t] = Sinv[t] - Cn[t] @ J[t].T
        if self._x_rot is not None:
            # q(X) was rotated after its update (rotate_node): covariances follow
            R = self._x_rot
            V = np.einsum('ik,tkl,jl->tij', R, V, R)
            Cn = np.einsum('ik,tkl,jl->tij', R, Cn, R)
        return V, Cn

    def x_means(self):
        """<x> as a host array (B, T, D)."""
        self._materialize()
        self._flush()
        out = self.rt.empty(self.B, self.T, self.D)
        self.kernels.x_layout(out, self.D, self.B, self.T, self.BL, self.Z, False)
        return out.cpu().numpy()

    def get_moments(self, node):
        self._materialize()
        self._flush()
        L = self.layout
        D, M, B, T = self.D, self.M, self.B, self.T
        st = self.state
        if node is self.tau:
            return self._gamma_view(L.off_tau, 1, self.tau.plates)
        if node is self.gamma:
            return self._gamma_view(L.off_gamma, D, self.gamma.plates)
        if node is self.alpha:
            return self._gamma_view(L.off_alpha, D, self.alpha.plates)
        if self.nu is not None and node is self.nu:
            return self._gamma_view(L.off_nu, D, self.nu.plates)
        if node is self.C:
            cm = st[L.off_Cm:L.off_Cm + M * D].cpu().numpy().reshape(M, D)
            cov = st[L.off_CovC:L.off_CovC + D * D].cpu().numpy().reshape(D, D)
            u1 = cov[None] + cm[:, :, None] * cm[:, None, :]
            return [cm.reshape(self.C.plates + (D,)), u1.reshape(self.C.plates + (D, D))]
        if node is self.A:
            am = st[L.off_Am:L.off_Am + D * D].cpu().numpy().reshape(D, D)
            aa = st[L.off_AA:L.off_AA + D * D * D].cpu().numpy().reshape(D, D, D)
            return [am.copy(), aa.copy()]
        if node is self.X:
            if 8.0 * B * T * D * D > 8e9:
                raise MemoryError('X.u[1] would take %.0f GB on the host; use plan.x_means() and '
                                  'plan.chain_covariances()' % (8e-9 * B * T * D * D))
            x = self.x_means()
            V, Cn = self.chain_covariances()
            u1 = V[None] + x[:, :, :, None] * x[:, :, None, :]
            u2 = Cn[None] + x[:, :-1, :, None] * x[:, 1:, None, :]
            pl = self.X.plates
            return [x.reshape(pl + (T, D)), u1.reshape(pl + (T, D, D)),
                    u2.reshape(pl + (T - 1, D, D))]
        raise NotImplementedError('moments of %s are never materialised by the fused LSSM block'
                                  % node.name)

    def get_mask(self, node):
        return np.array(True)

    # -- persistence ---------------------------------------------------------------------------------------------
    def save_state(self, put, nodes, index):
        self._materialize()
        self._flush()
        base = 'plans/%d/' % index
        _delta.save(put, base, self._delta)
        put(base + 'kind', np.array([ord(c) for c in 'lssm'], dtype=np.uint8))
        put(base + 'dims', np.array([self.D, self.M, self.B, self.T], dtype=np.int64))
        put(base + 'state_len', np.array([int(self.layout.total)], dtype=np.int64))
        put(base + 'state', self.state.cpu().numpy())
        put(base + 'X', self.x_means())
        put(base + 'Sinv', self.Sinv.cpu().numpy())
        put(base + 'J', self.J.cpu().numpy())
        put(base + 'x_updated', bool(self._x_updated))
        if self._x_rot is not None:
            put(base + 'x_rot', self._x_rot)

    def load_state(self, reader, nodes, index):
        self._materialize()
        base = 'plans/%d/' % index
        self._delta = _delta.load(reader, base)
        if not reader.has(base + 'state'):
            raise Exception("File does not contain the state of the fused LSSM block")
        dims = tuple(int(v) for v in reader.get(base + 'dims'))
        if dims != (self.D, self.M, self.B, self.T):
            raise ValueError('checkpoint is for (D, M, B, T) = %s, the model has %s'
                             % (dims, (self.D, self.M, self.B, self.T)))
        torch = self.rt.torch
        st = np.array(reader.get(base + 'state'), dtype=np.float64)
        if st.size != int(self.layout.total):
            # the packed state layout is part of the library build (vmp_lssm_get_layout)
            raise ValueError('incompatible checkpoint: the packed state of the fused state-space '
                             'block has %d doubles in the file, %d in this build of the library '
                             '(written by another version; re-run from the node moments)'
                             % (st.size, int(self.layout.total)))
        self.state.copy_(torch.from_numpy(st))
        xd = torch.from_numpy(np.array(reader.get(base + 'X'), dtype=np.float64)).to(self.rt.device)
        self.kernels.x_layout(xd.contiguous(), self.D, self.B, self.T, self.BL, self.Z, True)
        self.Sinv.copy_(torch.from_numpy(np.array(reader.get(base + 'Sinv'), dtype=np.float64)))
        self.J.copy_(torch.from_numpy(np.array(reader.get(base + 'J'), dtype=np.float64)))
        self._x_updated = bool(reader.get(base + 'x_updated'))
        self._x_rot = np.array(reader.get(base + 'x_rot')) if reader.has(base + 'x_rot') else None
        self._version += 1

    # -- rotation parameter expansion (inference/transformations.py) ------------------------------------------
    # The rotated quantities of this block are the replicated moments and the K x K plate sums of the
    # state vector (host arithmetic, O(D^3)) plus the plate-sized means of the chain (one device pass).
    def _host_state(self):
        self._materialize()
        self._flush()
        self.rt.sync_stream()
        return self.state.cpu().numpy()

    def _put_state(self, st):
        self.state.copy_(self.rt.torch.from_numpy(st))
        self._version += 1

    def rotation_statistics(self, node):
        st, L = self._host_state(), self.layout
        D, M, T, DD = self.D, self.M, self.T, self.D * self.D
        if node is self.C:
            return dict(XX=st[L.off_SCC:L.off_SCC + DD].reshape(D, D).copy(), nplates=float(M))
        if node is self.X:
            S = st[L.off_S:]
            mat = lambda k: S[k * DD:(k + 1) * DD].reshape(D, D).copy()       # noqa: E731
            Sxx, Spp, Snn, Snp, S00 = (mat(k) for k in range(5))
            return dict(nvec=float(T) * float(self.B_total), X0=S[5 * DD:5 * DD + D].copy(),
                        X0X0=S00, XnXn=Snn, XpXn=Snp.T.copy(), XpXp=Spp)
        raise NotImplementedError('rotation of %s' % node.name)

    def rotation_rows(self, node):
        """Per-row means (D, D) and covariances (D, D, D) of the dynamics matrix."""
        if node is not self.A:
            raise NotImplementedError('row statistics of %s' % node.name)
        st, L, D = self._host_state(), self.layout, self.D
        am = st[L.off_Am:L.off_Am + D * D].reshape(D, D).copy()
        aa = st[L.off_AA:L.off_AA + D * D * D].reshape(D, D, D)
        return dict(mean=am, cov=aa - am[:, :, None] * am[:, None, :])

    def gamma_posterior_shape(self, node):
        st, L, D = self._host_state(), self.layout, self.D
        off = {id(self.gamma): L.off_gamma, id(self.alpha): L.off_alpha}.get(id(node))
        if off is None:
            raise NotImplementedError('shape parameter of %s' % node.name)
        return st[off:off + D].copy()

    def rotate_node(self, node, R, invR, logdetR, Q=None):
        st, L = self._host_state(), self.layout
        D, M, T, DD = self.D, self.M, self.T, self.D * self.D
        if node is self.X:
            # the chain: plate sums in the state, the means on the device, log|Phi| of q(X)
            # (gaussian_markov_chain.py:51-65, :167-185: u <- R u R^T, g <- g - T log|R|)
            if Q is not None:
                raise ValueError('the chain has no plate rotation')
            S = st[L.off_S:]
            for k in range(5):
                S[k * DD:(k + 1) * DD] = (R @ S[k * DD:(k + 1) * DD].reshape(D, D) @ R.T).reshape(-1)
            S[5 * DD:5 * DD + D] = R @ S[5 * DD:5 * DD + D]
            o = 5 * DD + D
            S[o:o + M * D] = (S[o:o + M * D].reshape(M, D) @ R.T).reshape(-1)
            st[L.off_scal + 1] -= 2.0 * T * logdetR
            Rd = self.rt.torch.from_numpy(np.ascontiguousarray(R, dtype=np.float64)).to(self.rt.device)
            self.kernels.rotate_x(D, T, self.B, self.BL, Rd, self.Z)
            self._x_rot = R if self._x_rot is None else R @ self._x_rot
        elif node is self.C:
            cm = st[L.off_Cm:L.off_Cm + M * D].reshape(M, D) @ R.T
            covc = R @ st[L.off_CovC:L.off_CovC + DD].reshape(D, D) @ R.T
            st[L.off_Cm:L.off_Cm + M * D] = cm.reshape(-1)
            st[L.off_CovC:L.off_CovC + DD] = covc.reshape(-1)
            st[L.off_SCC:L.off_SCC + DD] = (M * covc + cm.T @ cm).reshape(-1)
            st[L.off_scal + 4] += 2.0 * logdetR
        elif node is self.A:
            am = st[L.off_Am:L.off_Am + DD].reshape(D, D)
            aa = st[L.off_AA:L.off_AA + D * DD].reshape(D, D, D)
            cov = aa - am[:, :, None] * am[:, None, :]
            am = am @ R.T                                           # columns: a_d <- R a_d
            cov = np.einsum('ik,dkl,jl->dij', R, cov, R)
            ld = st[L.off_ldA:L.off_ldA + D] + 2.0 * logdetR
            if Q is not None:
                # rows: means exactly, covariances scaled by the squared column sums of Q
                # (gaussian.py:1743-1772)
                sQ = Q.sum(axis=0)
                am = Q @ am
                cov = cov * (sQ * sQ)[:, None, None]
                ld = ld + 2.0 * D * np.log(np.abs(sQ))
            st[L.off_Am:L.off_Am + DD] = am.reshape(-1)
            st[L.off_AA:L.off_AA + D * DD] = (cov + am[:, :, None] * am[:, None, :]).reshape(-1)
            st[L.off_ldA:L.off_ldA + D] = ld
        else:
            raise NotImplementedError('rotation of %s' % node.name)
        self._put_state(st)

    # -- measurement ---------------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._materialize()
        self.kernels.set_timing(on)

    def cov_stationary_from(self):
        """Time steps from which the forward / backward covariance recursions of the last
        X.update() reused a converged iterate (-1: every step was computed)."""
        L, D = self.layout, self.D
        self.rt.sync_stream()
        fix = self.state[L.off_covsums + 5 * D * D + 2:L.off_covsums + 5 * D * D + 4].cpu().numpy()
        return [int(fix[0]), int(fix[1])]

    def kernel_times_ms(self):
        t = self.kernels.pass_times_ms(64)
        if not t:
            return None
        n = float(len(t))
        return dict(lssm_forward=sum(a for a, _ in t) / n, lssm_backward=sum(b for _, b in t) / n,
                    cov_stationary_from=self.cov_stationary_from())
