"""
Execution plan of the probabilistic-PCA / factor-analysis block

    Y = GaussianARD(SumMultiply('i,i', W, X), tau);  W = GaussianARD(0, alpha);
    X = GaussianARD(0, c);  tau, alpha = Gamma(a0, b0)        (demos/pca.py:22-61)

with a fully observed Y (scalar mask).  The plan owns, in HBM:

* ``Y``  (D, ldy)  fp64, the observation plate N contiguous (the local shard
  when the plate is sharded over ranks);
* ``X``  (K, ldx)  fp64, posterior means, plate contiguous;
* ``state``        one block of doubles (``vmp_pca_layout``): the statistics S
  that ranks all-reduce, and every replicated quantity (tau, alpha, <W>, Cov_W,
  Sww, Cov_X, A, lower-bound terms).

Reference semantics preserved: each ``node.update()`` sees the latest moments
of its Markov blanket, in whatever order the user calls them
(vmp.py:154-172); the only plate-sized work per VB iteration is ONE streaming
pass over Y issued by ``X.update()``.

Two forms of that pass (``stats=``):

* ``'gram'`` (default): ``vmp_pca_xpass`` writes <x_n> = A y_n (read Y once,
  write <x> once -- HBM-bound) and the messages to W come from the constant
  Gram matrix G = Y Y^T, computed and summed over ranks ONCE at set-up; no
  per-iteration collective is left.  Y is constant after ``observe()``, so the
  plan keeps a tile-major copy of it (``layout='tiled'``, 32 plate elements per
  contiguous block, ``vmp_pca_tile_y``) that the pass streams instead of D row
  segments per tile; ``layout='rows'`` passes over the row-major array.
* ``'stream'``: ``vmp_pca_pass`` additionally accumulates sum y<x>^T and
  sum <x><x>^T while streaming (fp64-MFMA-bound); the partial sums are
  all-reduced over ranks every iteration (the reference's plate sums,
  node.py:650 / dot.py:581).
"""
import ctypes
import os

import numpy as np

from . import _delta

from ... import _lib
from ...device import get_runtime, ptr
from ...nodes.node import Constant
from ...nodes.gamma import Gamma
from ...nodes.gaussian import GaussianARD
from ...nodes.dot import SumMultiply


# operation codes of vmp_pca_small_ops (include/vmp_hip.h, enum vmp_pca_op)
OP_W, OP_XPREP, OP_TAU, OP_ALPHA, OP_ELBO = 1, 2, 3, 4, 5


class HIPKernels:
    """The C-ABI entry points used by this plan, bound to a runtime."""

    def __init__(self, rt):
        self.rt = rt
        self.lib = rt.lib
        self.ctx = rt.ctx
        if self.interleave:
            self.x_tiles = True
            self.rt.check(self.lib.vmp_tune_set(b'pca_interleave', 1))

    def layout(self, D, K):
        L = _lib.PCALayout()
        rc = self.lib.vmp_pca_get_layout(D, K, ctypes.byref(L))
        if rc != _lib.VMP_OK:
            _lib.raise_for_status(rc, 'fused PCA block supports D <= 256 and K <= 64 '
                                      '(got D=%d, K=%d)' % (D, K))
        return L

    def workspace_doubles(self, D, K):
        n = ctypes.c_size_t()
        self.rt.check(self.lib.vmp_pca_workspace_bytes(self.ctx, D, K, ctypes.byref(n)))
        return (n.value + 7) // 8

    def init_state(self, D, K, a0t, b0t, a0a, b0a, state):
        self.rt.check(self.lib.vmp_pca_init_state(self.ctx, D, K, a0t, b0t, a0a, b0a, ptr(state)))

    def syy(self, Y, ldy, N, D, K, state, ws):
        self.rt.check(self.lib.vmp_pca_syy(self.ctx, ptr(Y), ldy, N, D, K, ptr(state), ptr(ws)))

    def stats_from_x(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.rt.check(self.lib.vmp_pca_stats_from_x(self.ctx, ptr(Y), ldy, N, D, K, ptr(X), ldx,
                                                    ptr(state), ptr(ws)))

    def small_ops(self, D, K, n_total, x_prec, a0t, b0t, a0a, b0a, ops, state, has_mean=False):
        """Replicated-node operations (OP_* codes) in order, fused into as few
        single-workgroup launches as the library knows sequences for.  ``has_mean``: W has a
        constant non-zero prior mean, kept in state[off_mu] (vmp_pca_small_ops_mean)."""
        arr = (ctypes.c_int32 * len(ops))(*ops)
        if has_mean:
            self.rt.check(self.lib.vmp_pca_small_ops_mean(self.ctx, D, K, n_total, x_prec, a0t,
                                                          b0t, a0a, b0a, len(ops), arr, 1,
                                                          ptr(state)))
            return
        self.rt.check(self.lib.vmp_pca_small_ops(self.ctx, D, K, n_total, x_prec, a0t, b0t, a0a,
                                                 b0a, len(ops), arr, ptr(state)))

    def pass_(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.rt.check(self.lib.vmp_pca_pass(self.ctx, ptr(Y), ldy, N, D, K, ptr(X), ldx,
                                            ptr(state), ptr(ws)))

    def gram(self, Y, ldy, N, D, K, state, ws):
        self.rt.check(self.lib.vmp_pca_gram(self.ctx, ptr(Y), ldy, N, D, K, ptr(state), ptr(ws)))

    def xpass(self, Y, ldy, N, D, K, X, ldx, state, ws):
        self.rt.check(self.lib.vmp_pca_xpass(self.ctx, ptr(Y), ldy, N, D, K, ptr(X), ldx,
                                             ptr(state), ptr(ws)))

    def xjoin(self):
        self.rt.check(self.lib.vmp_pca_xjoin(self.ctx))

    def ensure_gram(self):
        """The Gram-form messages to W are formed lazily by the library (vmp_pca_ensure_gram):
        before the state block is read directly."""
        self.rt.check(self.lib.vmp_pca_ensure_gram(self.ctx))

    def tile_y(self, Y, ldy, N, D, K):
        """Tile-major copy of the constant data (vmp_pca_tile_y): [tile][DP][32]."""
        n = ctypes.c_int64()
        self.rt.check(self.lib.vmp_pca_tiled_doubles(D, K, N, ctypes.byref(n), None))
        Yt = self.rt.empty(max(int(n.value), 1))
        self.rt.check(self.lib.vmp_pca_tile_y(self.ctx, ptr(Y), ldy, N, D, K, ptr(Yt)))
        return Yt

    def xpass_tiled(self, Yt, N, D, K, X, ldx, state, ws, x_tiled=False):
        self.rt.check(self.lib.vmp_pca_xpass_tiled(self.ctx, ptr(Yt), N, D, K, ptr(X), ldx,
                                                   1 if x_tiled else 0, ptr(state), ptr(ws)))

    # <x> may be kept tile-major as well ([tile][KP][32]): the pass then writes one contiguous
    # 8 KB span per tile instead of KP row segments that lie 8*ldx bytes apart
    # (opt-in: measured 2.29 / 2.39 ms against 2.21 / 2.33 ms for the row-major <x> with the
    # placement trials of both, profiles/r03/placement_trials_ab.txt)
    x_tiles = os.environ.get('BAYESPY_AMD_PCA_XTILES', '0') == '1'
    # round 4 experiment: ONE array [tile][DP + KP][32] for the data and <x> (tune key
    # "pca_interleave"): the relative placement of the read and the write stream is then fixed by
    # construction -- no placement trial, no second big allocation
    interleave = os.environ.get('BAYESPY_AMD_PCA_INTERLEAVE', '0') == '1'

    def tiled_x_doubles(self, D, K, N):
        n = ctypes.c_int64()
        self.rt.check(self.lib.vmp_pca_tiled_doubles(D, K, N, None, ctypes.byref(n)))
        return max(int(n.value), 1)

    def tile_x(self, to_tiled, X, ldx, N, D, K, Xt):
        """Xt <- X (row-major (KP, ldx)) or X <- Xt (vmp_pca_tile_x)."""
        self.rt.check(self.lib.vmp_pca_tile_x(self.ctx, 1 if to_tiled else 0, ptr(X), ldx, N, D, K,
                                              ptr(Xt)))

    def rotate_rows(self, R, X, N):
        """X[:, :N] <- R X[:, :N] on the device (vmp_gemm_strided via utils.linalg)."""
        from ...darray import DArray
        from ...utils import linalg
        K = R.shape[0]
        xn = linalg.mmdot(DArray.from_host(np.ascontiguousarray(R)), DArray(X[:K, :N]))
        X[:K, :N].copy_(xn.t)

    def set_timing(self, on):
        self.rt.check(self.lib.vmp_ctx_set_timing(self.ctx, 1 if on else 0))

    def last_pass_ms(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        self.rt.check(self.lib.vmp_pca_last_pass_ms(self.ctx, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def pass_times_ms(self, cap=64):
        """(pass_ms, reduce_ms) of the most recent timed plate passes, oldest first."""
        a = (ctypes.c_double * cap)()
        b = (ctypes.c_double * cap)()
        n = ctypes.c_int32()
        self.rt.check(self.lib.vmp_pass_times_ms(self.ctx, a, b, cap, ctypes.byref(n)))
        return [(a[i], b[i]) for i in range(n.value)]


def _const_scalar(node):
    return isinstance(node, Constant) and node.is_scalar()


def _const_zero(node):
    return isinstance(node, Constant) and not np.any(node.value)


def _const_mean(node, shape):
    """A constant that broadcasts to ``shape`` (the prior mean of W: plates (D, 1) + (K,))."""
    if not isinstance(node, Constant):
        return False
    try:
        return np.broadcast_shapes(node.value.shape, shape) == tuple(shape)
    except ValueError:
        return False


def _gamma_with_const_parents(node):
    return (isinstance(node, Gamma) and _const_scalar(node.parents[0])
            and _const_scalar(node.parents[1]))


def _lead(plates, n):
    """plates right-aligned into n axes (missing axes = 1)."""
    return (1,) * (n - len(plates)) + tuple(plates)


class PCAPlan:

    @staticmethod
    def describe():
        return ("GaussianARD(SumMultiply('i,i', W, X), Gamma) with W=GaussianARD(const, Gamma, "
                "shape=(K,), plates=(D,1)), X=GaussianARD(0, const, shape=(K,), plates=(1,N)), "
                "fully observed")

    # -- pattern matching -----------------------------------------------------------
    @staticmethod
    def unsupported_state(roles):
        """Why the block cannot represent the nodes' current observation / initialisation
        state (None if it can).  The block starts tau, alpha from their priors, W from its
        prior / a value / a draw, X likewise, and updates every role but Y: anything else
        (``tau.observe``, ``alpha.initialize_from_value``, ``initialize_from_parameters``, a
        mask on Y ...) belongs to the generic engine, which keeps per-node state like the
        reference (stochastic.py:223-250, expfamily.py:168-212)."""
        if roles['Y']._mask is not True:
            return 'Y has missing values'
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
        roles = PCAPlan.match_graph(nodes, why, allow_mean=True)
        if roles is None:
            return None
        bad = PCAPlan.unsupported_state(roles)
        if bad is not None:           # e.g. missing values, a fixed tau: other plans
            if why is not None and bad != 'Y has missing values':
                why.append('fused PCA block: %s' % bad)
            return None
        return roles

    @staticmethod
    def match_graph(nodes, why=None, allow_mean=False):
        """The graph pattern alone (shared with the missing-data block, plans/masked_pca.py).
        ``allow_mean``: a constant NON-ZERO prior mean of W is part of the pattern (this block:
        yes, the missing-data block: no).
        ``why``: a list that receives, for every node that looks like the observed node of this
        block, the first condition it fails (compile_model reports them when a model that
        resembles a fused block ends up on the generic engine)."""
        def no(Y, msg):
            if why is not None:
                why.append('fused PCA block, observed node %s: %s' % (Y.name or '<unnamed>', msg))
        # mini-batch multipliers (stochastic VI) go through the generic engine
        if any(any(m != 1 for m in n.plates_multiplier) for n in nodes):
            return None
        for Y in nodes:
            if not isinstance(Y, GaussianARD) or Y.ndim != 0:
                continue
            F, tau = Y.parents
            if not isinstance(F, SumMultiply):
                continue
            if not _gamma_with_const_parents(tau):
                no(Y, 'its precision is not a Gamma node with constant parameters')
                continue
            if len(F.parents) != 2 or F.out_keys != [] or F.in_keys[0] != F.in_keys[1] \
                    or len(F.in_keys[0]) != 1:
                no(Y, "its mean is not SumMultiply('i,i', W, X) / Dot(W, X) of two nodes")
                continue
            if any(p != 1 for p in tau.plates) or len(Y.plates) != 2:
                no(Y, 'it needs plates (D, N) and a scalar precision (got plates %s, precision '
                      'plates %s)' % (tuple(Y.plates), tuple(tau.plates)))
                continue
            D, N = Y.plates
            A, B = F.parents
            if not (isinstance(A, GaussianARD) and isinstance(B, GaussianARD)):
                no(Y, 'both factors must be GaussianARD nodes')
                continue
            if A.ndim != 1 or B.ndim != 1 or A.shape != B.shape:
                no(Y, 'both factors need the same shape (K,)')
                continue
            pa, pb = _lead(A.plates, 2), _lead(B.plates, 2)
            if pa == (D, 1) and pb == (1, N):
                W, X = A, B
            elif pb == (D, 1) and pa == (1, N):
                W, X = B, A
            else:
                no(Y, 'the factors need plates (D, 1) and (1, N) (got %s and %s)'
                      % (tuple(A.plates), tuple(B.plates)))
                continue
            K = W.shape[0]
            alpha = W.parents[1]
            if allow_mean:
                if not _const_mean(W.parents[0], (D, 1, K)):
                    no(Y, 'the prior mean of %s is not a constant of shape (D, 1, K)'
                          % (W.name or 'W'))
                    continue
            elif not _const_zero(W.parents[0]):
                no(Y, 'the prior mean of %s is not the constant 0' % (W.name or 'W'))
                continue
            if not _gamma_with_const_parents(alpha) or _lead(alpha.plates, 1) != (K,):
                no(Y, 'the prior precision of %s is not a Gamma node with plates (K,) and '
                      'constant parameters' % (W.name or 'W'))
                continue
            if not (_const_zero(X.parents[0]) and _const_scalar(X.parents[1])):
                no(Y, 'the prior of %s is not N(0, c I) with constants' % (X.name or 'X'))
                continue
            # every role must be private to this block
            shared = [n.name or type(n).__name__ for n in (W, X, F, tau, alpha)
                      if len(n.children) != 1]
            if shared:
                no(Y, '%s has other children as well' % ', '.join(shared))
                continue
            return dict(Y=Y, F=F, W=W, X=X, tau=tau, alpha=alpha)
        return None

    # -- construction ------------------------------------------------------------------
    def __init__(self, roles, runtime=None, kernels=None, stats=None, layout=None):
        if stats is None:
            stats = os.environ.get('BAYESPY_AMD_PCA_STATS', 'gram')
        if stats not in ('gram', 'stream'):
            raise ValueError("stats must be 'gram' or 'stream'")
        self.stats = stats
        if layout is None:
            layout = os.environ.get('BAYESPY_AMD_PCA_LAYOUT', 'tiled')
        if layout not in ('tiled', 'rows'):
            raise ValueError("layout must be 'tiled' or 'rows'")
        self.plate_layout = layout
        self.Yt = None
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
        # constant prior mean of W (gaussian.py:805-830: phi0 = alpha * mu): None when it is 0
        mu = self.W.parents[0].value
        self.mu0 = (np.array(np.broadcast_to(mu, (self.D, 1, self.K)), dtype=np.float64)
                    .reshape(self.D, self.K) if np.any(mu) else None)
        self._rt = runtime
        self._kernels = kernels
        self._ready = False
        self._version = 0
        self._L_version = -1
        self._L = None
        self._pending = []          # queued replicated-node operations (see _flush)
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
            self._kernels = HIPKernels(self.rt)
        return self._kernels

    def nodes(self):
        return list(self.roles.values())

    def has_state(self):
        """Device state exists (a recompilation would discard it)."""
        return bool(self._ready)

    def invalidate(self, node):
        """Data or initial value of ``node`` changed: rebuild device state lazily."""
        self.finish()
        if node is self.Y and self._ready and self._version > 1 and self.Y._mask is True \
                and self.unsupported_state(self.roles) is None \
                and self.stats == getattr(self, '_stats_built', self.stats):
            self._reobserve()
            self._version += 1
            return
        if node in (self.X, self.W) and self._ready and self._version > 1 \
                and node._init is not None and node._init[0] == 'value' \
                and self.unsupported_state(self.roles) is None:
            self._reinitialise(node)
            self._version += 1
            return
        _delta.warn_state_discarded(self, node)
        self._ready = False
        self._version += 1
        if self.unsupported_state(self.roles) is not None:
            # missing data (per-plate posteriors): the fused missing-data block when it covers
            # the sizes; an observed / specially initialised role: the generic device
            # message-passing engine (Node.shard declarations travel with the nodes)
            from .masked_pca import MaskedPCAPlan
            roles = MaskedPCAPlan.match(self.nodes())
            if roles is not None and not getattr(self, 'generic_only', False):
                MaskedPCAPlan(roles)
            else:
                from .generic import GenericPlan
                GenericPlan(self.nodes())

    # -- device state --------------------------------------------------------------------
    def _upload_y(self):
        """Y: (D, ldy), plate contiguous, 16-byte aligned rows; resident tensors of that form are
        used in place."""
        rt = self.rt
        torch = rt.torch
        D, N = self.D, self.N
        y = self.Y._data
        if isinstance(y, torch.Tensor) and y.device == rt.device and y.dtype == torch.float64 \
                and tuple(y.shape) == (D, N) and y.stride(1) == 1 and y.stride(0) % 2 == 0 \
                and y.data_ptr() % 16 == 0 and N > 0:
            self.Yd, self.ldy = y, y.stride(0)
        else:
            # whole 32-column tiles: no ragged-tail launch.  (An empty local plate -- a rank of a
            # sharded run without any observation -- keeps one tile of padding: valid pointers.)
            ldy = max(32, (N + 31) // 32 * 32)
            self.Yd = rt.zeros(D, ldy)
            if isinstance(y, torch.Tensor):
                src = y
            else:
                ya = np.asarray(y, dtype=np.float64)
                if ya.shape != (D, N) or not ya.flags.c_contiguous or not ya.flags.writeable:
                    ya = np.array(np.broadcast_to(ya, (D, N)), dtype=np.float64, order='C')
                src = torch.from_numpy(ya)
            self.Yd[:, :N].copy_(src)
            self.ldy = ldy
        self.Yt = None

    # -- <x>: row-major (KP, ldx) until the first tile-major pass, tile-major afterwards ---------
    _Xt = None
    _Xrows = None
    _x_form = 'rows'
    _xrows_valid = True

    @property
    def Xd(self):
        """Row-major (KP, ldx) view of <x>.  While <x> lives tile-major (after a tile-major pass)
        this is a copy formed on demand (vmp_pca_tile_x) and kept until the next pass."""
        if self._x_form == 'tiled' and not self._xrows_valid:
            k, rt = self.kernels, self.rt
            k.xjoin()                       # the pass that writes the tiles has finished
            rt.sync_stream()
            if self._Xrows is None:
                self._Xrows = rt.zeros(int(self.layout.KP), self.ldx)
            k.tile_x(False, self._Xrows, self.ldx, self.N, self.D, self.K, self._Xt)
            self._xrows_valid = True
        return self._Xrows

    @Xd.setter
    def Xd(self, value):
        self._Xrows = value
        self._x_form, self._xrows_valid = 'rows', True

    def _x_rows_modified(self):
        """The row-major array was changed in place (a loaded value, a rotation): it is the
        current <x>; the next pass overwrites the tiles anyway."""
        self._x_form, self._xrows_valid = 'rows', True

    def _load_x_value(self, x0):
        """<x> <- a given value, (.., N, K) -> the plate-contiguous (K, N) layout."""
        rt = self.rt
        torch = rt.torch
        N, K = self.N, self.K
        if isinstance(x0, torch.Tensor) and x0.device == rt.device:
            self.Xd[:K, :N].copy_(x0.to(torch.float64).expand(self.X.plates + (K,))
                                  .reshape(N, K).t())
        else:
            if isinstance(x0, torch.Tensor):
                x0 = x0.detach().cpu().numpy()
            x0 = np.broadcast_to(np.asarray(x0, dtype=np.float64),
                                 self.X.plates + (K,)).reshape(N, K)
            self.Xd[:K, :N].copy_(torch.from_numpy(np.array(x0.T, dtype=np.float64, order='C')))
        self._x_rows_modified()

    def _load_w_value(self, w0, cov=None):
        """<w_d> <- the rows of a (D, K) host array, Sww = W^T W (+ D cov; delta moments without
        one); with a prior mean mu also sum_d mu_dk <w_dk> and sum_d mu_dk^2 (state[off_mstat]:
        the sums the alpha update and the bound centre sum_d <w_dk^2> with)."""
        L = self.layout
        D, K, KP = self.D, self.K, int(L.KP)
        self._put_block(L.off_W, np.asarray(w0, dtype=np.float64), KP)
        self._set_block(L.off_Sww, w0.T @ w0 + (0.0 if cov is None else D * cov))
        if self.mu0 is not None:
            self._put_block(L.off_mstat, np.stack([np.sum(self.mu0 * w0, axis=0),
                                                   np.sum(self.mu0 ** 2, axis=0)]), KP)

    def _reinitialise(self, node):
        """initialize_from_value on X or W AFTER updates: that node becomes the point mass of the
        value, every other posterior stays (expfamily.py:193-204)."""
        self.rt.sync_stream()
        L = self.layout
        K = self.K
        if node is self.X:
            self._load_x_value(node._init[1])
            self._set_block(L.off_CX, np.zeros((K, K)))
            self.kernels.stats_from_x(self.Yd, self.ldy, self.N, self.D, K, self.Xd, self.ldx,
                                      self.state, self.ws)
            self._reduce(self.state[L.off_S:L.off_S + L.len_S])
            self._delta.add('X')
        else:
            w0 = node._init[1]
            if isinstance(w0, self.rt.torch.Tensor):
                w0 = w0.detach().cpu().numpy()
            w0 = np.broadcast_to(np.asarray(w0, dtype=np.float64),
                                 self.W.plates + (K,)).reshape(self.D, K)
            self._load_w_value(w0)
            self._set_block(L.off_CW, np.zeros((K, K)))
            self._delta.add('W')

    def _data_statistics(self):
        """sum y^2 and (Gram form) G = Y Y^T of the current data, summed over the ranks."""
        k, L = self.kernels, self.layout
        D, N, K = self.D, self.N, self.K
        k.syy(self.Yd, self.ldy, N, D, K, self.state, self.ws)
        self._reduce(self.state[L.off_Syy:L.off_Syy + 1])
        if self.stats == 'gram':
            k.gram(self.Yd, self.ldy, N, D, K, self.state, self.ws)
            DP = int(L.DP)
            self._reduce(self.state[L.off_G:L.off_G + DP * DP])

    def _reobserve(self):
        """Y.observe(new data) AFTER updates: like in the reference only Y changes (stochastic.py:
        223-273) -- q(W), q(X), q(tau), q(alpha) stay, and the messages from Y are those of the NEW
        data with the CURRENT <x>: sum y^2, G and S = [sum y<x>^T ; sum <x><x>^T] are recomputed."""
        self.rt.sync_stream()
        k, L = self.kernels, self.layout
        self._upload_y()
        self._data_statistics()
        k.stats_from_x(self.Yd, self.ldy, self.N, self.D, self.K, self.Xd, self.ldx, self.state,
                       self.ws)
        self._reduce(self.state[L.off_S:L.off_S + L.len_S])

    def _materialize(self):
        if self._ready:
            return
        self._delta = _delta.delta_roles(self.roles)    # point masses until their first update
        self._Xt, self._Xrows = None, None
        self._x_form, self._xrows_valid = 'rows', True
        rt, k = self.rt, self.kernels
        torch = rt.torch
        D, N, K = self.D, self.N, self.K
        if self.Y._data is None:
            raise ValueError('Node %s has not been observed; the fused PCA block needs '
                             'Y.observe(y)' % self.Y.name)
        why = self.unsupported_state(self.roles)
        if why is not None:
            raise NotImplementedError('the fused PCA block does not cover this model state (%s); '
                                      "use VB(..., engine='generic')" % why)
        rt.sync_stream()
        self.layout = L = k.layout(D, K)
        # ONE sharding contract for every plan: the observation plate is
        # partitioned over the ranks iff a node that carries it was declared with Node.shard();
        # an undeclared model under torch.distributed is an independent replica per rank
        self.sharded = any(getattr(n, '_shard_axis', None) is not None
                           for n in (self.X, self.F, self.Y))
        self.n_total = rt.all_reduce_int(N) if self.sharded else N
        self._upload_y()
        self.ldx = max(32, (N + 31) // 32 * 32)
        self.state = rt.zeros(int(L.total))
        self._scal_host = None
        self.ws = rt.empty(int(k.workspace_doubles(D, K)))
        k.init_state(D, K, self.a0t, self.b0t, self.a0a, self.b0a, self.state)
        self._stats_built = self.stats
        self._data_statistics()
        # ---- X: delta moments (initialize_from_value/random) or the prior --------------
        init = self.X._init
        KPx = int(L.KP)       # pad rows: the tile-major pass writes whole 16-row blocks of <x>
        if init is None:
            self.Xd = rt.zeros(KPx, self.ldx)
            self._set_block(L.off_CX, np.eye(K) / self.x_prec)
        else:
            if init[0] == 'value':
                self.Xd = rt.zeros(KPx, self.ldx)
                self._load_x_value(init[1])
            else:
                # a draw from the current q = prior N(0, I/x_prec) (expfamily.py:206-212);
                # RNG streams are not part of the parity contract
                self.Xd = rt.zeros(KPx, self.ldx)
                self.Xd[:K].copy_(torch.randn(K, self.ldx, dtype=torch.float64, device=rt.device))
                if self.x_prec != 1.0:
                    self.Xd.mul_(self.x_prec ** -0.5)
            k.stats_from_x(self.Yd, self.ldy, N, D, K, self.Xd, self.ldx, self.state, self.ws)
            self._reduce(self.state[L.off_S:L.off_S + L.len_S])
        # ---- W: prior (mean mu, Cov diag(1/<alpha>)) or a given value --------------------
        KP = int(L.KP)
        init = self.W._init
        mu0 = self.mu0
        if mu0 is not None:
            self._put_block(L.off_mu, mu0, KP)
        if init is None:
            cw = np.eye(K) * (self.b0a / self.a0a)
            self._set_block(L.off_CW, cw)
            if mu0 is None:
                self._set_block(L.off_Sww, D * cw)
            else:
                self._load_w_value(mu0, cov=cw)
        else:
            if init[0] == 'value':
                w0 = init[1]
                if isinstance(w0, torch.Tensor):
                    w0 = w0.detach().cpu().numpy()
                w0 = np.broadcast_to(np.asarray(w0, dtype=np.float64),
                                     self.W.plates + (K,)).reshape(D, K)
            else:
                w0 = np.random.normal(size=(D, K)) * np.sqrt(self.b0a / self.a0a)
                if mu0 is not None:
                    w0 = w0 + mu0
            self._load_w_value(w0)
        self._ready = True
        self._version += 1

    def _reduce(self, view):
        """Plate sum over the ranks (node.py:650, dot.py:581) -- only for a declared shard."""
        if self.sharded:
            self.rt.all_reduce_sum_(view)

    def _set_block(self, off, mat):
        """Upload a small K x K host matrix into a KP x KP state block (set-up only)."""
        KP = int(self.layout.KP)
        K = mat.shape[0]
        buf = np.zeros((KP, KP))
        buf[:K, :K] = mat
        self.state[off:off + KP * KP].copy_(self.rt.torch.from_numpy(buf.reshape(-1)))

    # -- node operations ---------------------------------------------------------------------
    def update(self, node):
        self._materialize()
        _delta.updated(self._delta, self.roles, node)
        rt, k, L = self.rt, self.kernels, self.layout
        D, N, K = self.D, self.N, self.K
        if node is self.W:
            self._pending.append(OP_W)
        elif node is self.X:
            self._pending.append(OP_XPREP)
            self._flush()
            rt.sync_stream()
            if self.stats == 'gram':
                # messages to W from the global Gram matrix: nothing to exchange
                if self.plate_layout == 'tiled':
                    if self.Yt is None:
                        self.Yt = k.tile_y(self.Yd, self.ldy, N, D, K)
                        if getattr(k, 'interleave', False):
                            # <x> of a tile sits behind its data rows in the same array
                            self._Xt = self.Yt[int(L.DP) * 32:]
                            self._Xrows = None
                        elif getattr(k, 'x_tiles', False):
                            # from here on <x> lives tile-major; the row-major view (self.Xd)
                            # is formed on demand (_x_rows)
                            self._Xt = rt.empty(k.tiled_x_doubles(D, K, N))
                            self._Xrows = None
                        if not getattr(k, 'interleave', False):
                            self._place_plate_arrays()
                    if self._Xt is not None:
                        k.xpass_tiled(self.Yt, N, D, K, self._Xt, self.ldx, self.state, self.ws,
                                      x_tiled=True)
                        self._x_form, self._xrows_valid = 'tiled', False
                    else:
                        k.xpass_tiled(self.Yt, N, D, K, self.Xd, self.ldx, self.state, self.ws)
                else:
                    k.xpass(self.Yd, self.ldy, N, D, K, self.Xd, self.ldx, self.state, self.ws)
            else:
                k.pass_(self.Yd, self.ldy, N, D, K, self.Xd, self.ldx, self.state, self.ws)
                # child -> parent message sum over the sharded plate (node.py:650, dot.py:581)
                self._reduce(self.state[L.off_S:L.off_S + L.len_S])
        elif node is self.tau:
            self._pending.append(OP_TAU)
        elif node is self.alpha:
            self._pending.append(OP_ALPHA)
        else:
            return
        self._version += 1

    def place_plate_arrays(self):
        """The placement trial of :meth:`_place_plate_arrays` as an explicit set-up step (otherwise
        it is part of the first ``X.update()``): builds the device state and the tile-major Y, tries
        the allocations with whatever A the state holds (the time of the pass does not depend on
        the values) and keeps the current <x> -- the trial writes only into fresh candidates and
        the kept one receives a copy."""
        self._materialize()
        if self.stats != 'gram' or self.plate_layout != 'tiled' or self.Yt is not None:
            return
        k = self.kernels
        self.rt.sync_stream()
        self.Yt = k.tile_y(self.Yd, self.ldy, self.N, self.D, self.K)
        if getattr(k, 'interleave', False):
            # the current row-major <x> moves into the <x> rows of the interleaved array
            self._Xt = self.Yt[int(self.layout.DP) * 32:]
            k.tile_x(True, self._Xrows, self.ldx, self.N, self.D, self.K, self._Xt)
            self._Xrows = None
            self._x_form, self._xrows_valid = 'tiled', False
            self.placement = None
            return
        if getattr(k, 'x_tiles', False):
            self._Xt = self.rt.empty(k.tiled_x_doubles(self.D, self.K, self.N))
        self._place_plate_arrays(keep_x=True)

    def _place_plate_arrays(self, keep_x=False):
        """Where the driver puts a multi-GB allocation physically decides how fast the plate pass
        streams it: the same kernel on the same data ran between 2.20 and 2.50 ms (N = 1e7, D = 128,
        K = 32) over fresh allocations of X and of the tile-major Y within ONE process, while
        offsets inside an allocation, the row stride of X or physically contiguous allocations
        changed nothing (tools/xpass_place.hip, profiles/r03/xpass_place.txt).  So the plan tries
        a few allocations of the tile-major Y and of X at set-up -- every pair, with the pass it
        is about to run anyway (same A, same Y: every trial writes the <x> of this update, bit
        for bit) -- and keeps the fastest pair; the others go back to the allocator.  Arrays
        below 0.2 GB per pass (BAYESPY_AMD_PLACEMENT_MIN_BYTES; 1 GB until round 6: BASELINE config 2,
        0.64 GB, ran 0.148 ms per step as a leg of bench.py behind the headline's 40 GB and 0.126 with
        the trial), the CPU test double and BAYESPY_AMD_PLACEMENT_TRIES=1 skip it; with
        little free memory fewer candidates are tried, an allocation that fails ends the list."""
        rt, k = self.rt, self.kernels
        torch = rt.torch
        N, D, K = self.N, self.D, self.K
        tries = int(os.environ.get('BAYESPY_AMD_PLACEMENT_TRIES', '4'))
        self.placement = None
        min_bytes = float(os.environ.get('BAYESPY_AMD_PLACEMENT_MIN_BYTES', 2e8))
        if rt.device.type != 'cuda' or tries <= 1 or 8.0 * N * (D + K) < min_bytes:
            return
        set_bytes = 8 * (self.Yt.numel() + 2 * (self._Xt if self._Xt is not None else self.Xd).numel())
        free = torch.cuda.mem_get_info(rt.device)[0]
        tries = min(tries, 1 + int(max(free - (8 << 30), 0) // set_bytes))
        if tries <= 1:
            return

        xt = self._Xt is not None
        x_cur = self._Xt if xt else self.Xd

        def timed(Yt, X):
            k.set_timing(True)
            for _ in range(2):
                if xt:
                    k.xpass_tiled(Yt, N, D, K, X, self.ldx, self.state, self.ws, x_tiled=True)
                else:
                    k.xpass_tiled(Yt, N, D, K, X, self.ldx, self.state, self.ws)
            k.xjoin()
            ms = min(a for a, _ in k.pass_times_ms(8))
            k.set_timing(bool(getattr(self, 'timing', False)))
            return ms

        # the placement of <x> (the write stream) decides most of the spread and its candidates are
        # cheap (a fifth of the bytes, no re-layout): more of them, all alive to the end.  The
        # candidates of the tile-major Y (10 GB each at the headline size) are a tournament with ONE
        # challenger alive at a time: the loser of a round goes back to the driver, then the next
        # <x> candidate is allocated -- it takes the start of the hole, so that the next Y candidate
        # cannot land where the loser was -- and then the next challenger.  A round times the holder
        # and the challenger on the <x> candidates that exist by then; the kept Y is timed on all
        # of them at the end.  (Round 4 held every candidate and 5 GB spacers to the end: 118 GB of
        # transient memory; now one Y and 2 tries - 3 <x> candidates beside the kept pair.)
        # keep_x: the current row-major <x> still holds values somebody may read (set-up before the
        # first X.update()): it is not among the candidates, the kept one gets a copy of it
        keep = keep_x and not xt
        xs = [] if keep else [x_cur]
        n_new = max(2 * tries - 3, 1)
        # the pass also queues S <- [G A^T; A G A^T] on the state: inside an iteration that is the
        # statistic of this update, at set-up time (keep_x) the state must come back as it was
        saved = self.state.clone() if keep_x else None
        y_best, rows = self.Yt, []
        try:
            for _ in range(max(n_new - (tries - 1), 1)):
                xs.append(rt.empty(*x_cur.shape))
            n_new -= max(n_new - (tries - 1), 1)
            for r in range(tries - 1):
                if r > 0 and n_new > 0:
                    xs.append(rt.empty(*x_cur.shape))       # into the hole of the last loser
                    n_new -= 1
                cand = k.tile_y(self.Yd, self.ldy, N, D, K)
                row_b = [timed(y_best, x) for x in xs]
                row_c = [timed(cand, x) for x in xs]
                if not rows:
                    rows.append(row_b)
                rows.append(row_c)
                if min(row_c) < min(row_b):
                    y_best = self.Yt = cand                 # (the old holder is the loser)
                del cand
                torch.cuda.empty_cache()                    # the loser: back to the driver
        except RuntimeError:            # out of memory: the candidates made so far take part
            torch.cuda.empty_cache()
        if not xs:
            return
        row_best = [timed(y_best, x) for x in xs]
        rows.append(row_best)
        grid = rows
        y_ms = [min(r) for r in rows]
        best = (min(row_best), len(rows) - 1, row_best.index(min(row_best)))
        if saved is not None:
            k.xjoin()
            rt.sync_stream()
            self.state.copy_(saved)
        if keep:
            xs[best[2]].copy_(x_cur)
        x_cur = xs[best[2]]
        if xt:
            self._Xt = x_cur
        else:
            self.Xd = x_cur
        self.Yt = y_best
        peak = torch.cuda.max_memory_allocated(rt.device)
        del xs, y_best
        # the losers go back to the DRIVER, not only to torch's cache: allocations outside the
        # caching allocator (the library's own, RCCL buffers, another process) must see the memory
        torch.cuda.empty_cache()
        x_ms = grid[0]
        self.placement = {'x_ms': x_ms, 'yt_ms': y_ms, 'grid_ms': grid, 'kept': [best[1], best[2]],
                          'transient_peak_bytes': int(peak)}

    def _flush(self):
        """Issue the queued replicated-node updates.  They are queued rather than launched
        one by one so that the sequences of a VB iteration -- (W, X) and (tau, alpha, lower
        bound) -- become single launches; results are identical, only launch count differs."""
        if not self._pending:
            return
        ops, self._pending = self._pending, []
        self.rt.sync_stream()
        for i in range(0, len(ops), 8):
            if self.mu0 is None:
                self.kernels.small_ops(self.D, self.K, self.n_total, self.x_prec, self.a0t,
                                       self.b0t, self.a0a, self.b0a, ops[i:i + 8], self.state)
            else:
                self.kernels.small_ops(self.D, self.K, self.n_total, self.x_prec, self.a0t,
                                       self.b0t, self.a0a, self.b0a, ops[i:i + 8], self.state,
                                       has_mean=True)

    def finish(self):
        """Order the caller's stream after the outstanding latent pass (Gram form runs it on the
        library's plate stream so that it overlaps the next iteration's replicated updates)."""
        if self._ready:
            self._flush()
        self._pending = []
        if (self._Xrows is not None or self._Xt is not None) and self.stats == 'gram':
            self.kernels.xjoin()

    def __del__(self):
        # the plate array must not return to the allocator while a pass still writes it
        try:
            if (self._Xrows is not None or self._Xt is not None) and self.stats == 'gram':
                self.kernels.xjoin()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass

    def _read_scalars(self):
        """state[off_scal : off_L + 8] -> host through a pinned staging buffer (this read-back
        sits on the critical path of every iteration)."""
        L = self.layout
        if self.rt.device.type != 'cuda':
            return self.state[L.off_scal:L.off_L + 8].numpy().copy()
        if getattr(self, '_scal_host', None) is None:
            self._scal_view = self.state[L.off_scal:L.off_L + 8]
            self._scal_host = self.rt.torch.empty(self._scal_view.numel(),
                                                  dtype=self.rt.torch.float64, pin_memory=True)
        self._scal_host.copy_(self._scal_view, non_blocking=True)
        self.rt.torch.cuda.current_stream(self.rt.device).synchronize()
        return self._scal_host.numpy()

    def _lower_bound_terms(self):
        self._materialize()
        if self._L_version != self._version:
            L = self.layout
            self._pending.append(OP_ELBO)
            self._flush()
            host = self._read_scalars()
            status = int(host[3])
            if status != 0:
                _lib.raise_for_status(status)
            t = host[8:]
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

    # -- host views (reference shapes) ---------------------------------------------------------
    def _block(self, off, rows, cols, ld):
        if hasattr(self.kernels, 'ensure_gram'):
            self.rt.sync_stream()
            self.kernels.ensure_gram()
        a = self.state[off:off + rows * ld].cpu().numpy().reshape(rows, ld)
        return a[:, :cols].copy()

    def get_moments(self, node):
        self._materialize()
        self._flush()
        L = self.layout
        D, N, K, KP = self.D, self.N, self.K, int(self.layout.KP)
        if node is self.W:
            w = self._block(L.off_W, D, K, KP)
            cw = self._block(L.off_CW, K, K, KP)
            u1 = w[:, :, None] * w[:, None, :] + cw
            return [w.reshape(self.W.plates + (K,)), u1.reshape(self.W.plates + (K, K))]
        if node is self.X:
            self.finish()
            x = self.Xd[:K, :N].cpu().numpy().T.copy()
            cx = self._block(L.off_CX, K, K, KP)
            u1 = x[:, :, None] * x[:, None, :] + cx
            return [x.reshape(self.X.plates + (K,)), u1.reshape(self.X.plates + (K, K))]
        if node is self.tau:
            t = np.array(self.state[L.off_tau:L.off_tau + 4].cpu().numpy())
            return [np.reshape(t[2], self.tau.plates), np.reshape(t[3], self.tau.plates)]
        if node is self.alpha:
            a = np.array(self.state[L.off_alpha:L.off_alpha + 4 * KP].cpu().numpy()).reshape(4, KP)
            return [a[2, :K].reshape(self.alpha.plates), a[3, :K].reshape(self.alpha.plates)]
        if node is self.Y:
            y = self.Yd[:, :N].cpu().numpy()
            return [y, y * y]
        if node is self.roles.get('F'):
            # <f_dn> = <w_d>.<x_n>; <f_dn^2> = <ww>_d : <xx>_n with the shared covariances C_W, C_X
            # (dot.py:316-415) -- a read-out on the host; the updates never form these arrays
            self.finish()
            w = self._block(L.off_W, D, K, KP)
            cw = self._block(L.off_CW, K, K, KP)
            cx = self._block(L.off_CX, K, K, KP)
            x = self.Xd[:K, :N].cpu().numpy().T
            f = w @ x.T
            f2 = f * f + np.einsum('nk,kl,nl->n', x, cw, x)[None, :] \
                + np.einsum('dk,kl,dl->d', w, cx, w)[:, None] + np.sum(cw * cx)
            F = self.roles['F']
            return [f.reshape(F.plates), f2.reshape(F.plates)]
        raise NotImplementedError('moments of %s are never materialised by the fused PCA block'
                                  % node.name)

    def posterior_parameters(self, node):
        """(a, b) of the Gamma nodes; (mean, covariance) of the Gaussian nodes."""
        self._materialize()
        self._flush()
        L = self.layout
        K, KP = self.K, int(self.layout.KP)
        if node is self.tau:
            t = self.state[L.off_tau:L.off_tau + 2].cpu().numpy()
            return float(t[0]), float(t[1])
        if node is self.alpha:
            a = self.state[L.off_alpha:L.off_alpha + 2 * KP].cpu().numpy().reshape(2, KP)
            return a[0, :K].copy(), a[1, :K].copy()
        if node is self.W:
            return self._block(L.off_W, self.D, K, KP), self._block(L.off_CW, K, K, KP)
        if node is self.X:
            self.finish()
            return (self.Xd[:K, :self.N].cpu().numpy().T.copy(),
                    self._block(L.off_CX, K, K, KP))
        raise NotImplementedError

    def get_parameters(self, node):
        """Natural parameters phi of q(node) in the reference's layout (``node.phi``,
        expfamily.py:314-321), derived on the host from the packed state: [-b, a] for the
        Gamma nodes (gamma.py:116-148); [Lambda m, -Lambda / 2] with the shared K x K precision
        Lambda for the Gaussian nodes (gaussian.py:649-706).  Inspection only."""
        if node in (self.tau, self.alpha):
            a, b = self.posterior_parameters(node)
            return [np.reshape(-np.asarray(b), node.plates), np.reshape(a, node.plates)]
        if node in (self.W, self.X):
            m, cov = self.posterior_parameters(node)
            lam = np.linalg.inv(cov)
            lam = 0.5 * (lam + lam.T)
            K = self.K
            return [(m @ lam).reshape(node.plates + (K,)),
                    (-0.5 * lam).reshape((1,) * len(node.plates) + (K, K))]
        raise ValueError('node %s is observed: it has no variational parameters' % node.name)

    def get_mask(self, node):
        return np.array(True)

    # -- persistence: the packed device state + <x_n>; node moments for inspection -----------------
    def save_state(self, put, nodes, index):
        self._materialize()
        self.finish()
        base = 'plans/%d/' % index
        _delta.save(put, base, self._delta)
        put(base + 'kind', np.array([ord(c) for c in 'pca'], dtype=np.uint8))
        put(base + 'dims', np.array([self.D, self.N, self.K], dtype=np.int64))
        if hasattr(self.kernels, 'ensure_gram'):
            self.rt.sync_stream()
            self.kernels.ensure_gram()
        put(base + 'state', self.state.cpu().numpy())
        put(base + 'X', self.Xd[:self.K, :self.N].cpu().numpy())
        for node in nodes:
            if node in (self.W, self.tau, self.alpha):
                for i, ui in enumerate(self.get_moments(node)):
                    put('nodes/%s/u%d' % (node.name, i), ui)
                put('nodes/%s/observed' % node.name, False)

    def load_state(self, reader, nodes, index):
        self._materialize()
        self.finish()
        base = 'plans/%d/' % index
        self._delta = _delta.load(reader, base)
        if not reader.has(base + 'state'):
            raise Exception("File does not contain the state of the fused PCA block")
        dims = tuple(int(v) for v in reader.get(base + 'dims'))
        if dims != (self.D, self.N, self.K):
            raise ValueError('checkpoint is for (D, N, K) = %s, the model has %s'
                             % (dims, (self.D, self.N, self.K)))
        torch = self.rt.torch
        st = np.array(reader.get(base + 'state'), dtype=np.float64)
        if st.size != int(self.layout.total):
            # the packed state grew in round 5 (the prior mean of W and its sums at the end): a
            # file of the older layout is its prefix, the new blocks are those of mu = 0
            if st.size < int(self.layout.total) and self.mu0 is None \
                    and st.size == int(self.layout.off_mu):
                st = np.concatenate([st, np.zeros(int(self.layout.total) - st.size)])
            else:
                raise ValueError('checkpoint holds %d state values, this build of the fused PCA block '
                                 'keeps %d' % (st.size, int(self.layout.total)))
        L = self.layout
        KP = int(L.KP)
        mu_file = st[int(L.off_mu):int(L.off_mu) + self.D * KP].reshape(self.D, KP)[:, :self.K]
        mu_model = np.zeros((self.D, self.K)) if self.mu0 is None else self.mu0     # (D, K)
        if not np.allclose(mu_file, mu_model, rtol=1e-12, atol=0.0):
            raise ValueError('checkpoint was saved with another prior mean of W than the model has')
        self.state.copy_(torch.from_numpy(st))
        self.Xd[:self.K, :self.N].copy_(torch.from_numpy(np.array(reader.get(base + 'X'),
                                                                 dtype=np.float64)))
        self._x_rows_modified()
        self._version += 1

    # -- rotations (inference/transformations.py) ----------------------------------------------------
    def gamma_posterior_shape(self, node):
        return self.posterior_parameters(node)[0]

    def rotation_statistics(self, node):
        """sum over the plates of <x x^T> (K x K, global over ranks) and the plate count."""
        self._materialize()
        self._flush()
        L = self.layout
        K, KP, DP = self.K, int(L.KP), int(L.DP)
        if node is self.W:
            self._no_rotation_with_mean()
            return dict(XX=self._block(L.off_Sww, K, K, KP), nplates=self.D)
        if node is self.X:
            sxx = self._block(L.off_S + DP * KP, K, K, KP)
            cx = self._block(L.off_CX, K, K, KP)
            return dict(XX=self.n_total * cx + 0.5 * (sxx + sxx.T), nplates=self.n_total)
        raise NotImplementedError('rotation of %s' % node.name)

    def _no_rotation_with_mean(self):
        """The rotation cost of transformations.py is built for a zero prior mean (its mu terms,
        transformations.py:476-640, are not): decline instead of optimising the wrong bound."""
        if self.mu0 is not None:
            raise NotImplementedError('rotation of %s: the fused PCA block rotates a node with '
                                      'prior mean 0 only' % (self.W.name or 'W'))

    def _put_block(self, off, mat, ld):
        """Upload a small host matrix into a row-major state block of leading dimension ld."""
        rows, cols = mat.shape
        buf = np.zeros((rows, ld))
        buf[:, :cols] = mat
        self.state[off:off + rows * ld].copy_(self.rt.torch.from_numpy(buf.reshape(-1)))

    def rotate_node(self, node, R, invR, logdetR):
        """q(node) <- the distribution of R x (gaussian.py:1693-1741): means and covariances of
        the K x K / D x K state on the host (O(K^3)), the (K, N) array of <x_n> on the device
        through the fp64 MFMA contraction kernel."""
        self._materialize()
        self.finish()
        rt, L = self.rt, self.layout
        torch = rt.torch
        D, K, KP, DP = self.D, self.K, int(L.KP), int(L.DP)
        sc = self.state[L.off_scal:L.off_scal + 2].cpu().numpy()
        if node is self.W:
            self._no_rotation_with_mean()
            w = self._block(L.off_W, D, K, KP)
            cw = self._block(L.off_CW, K, K, KP)
            sww = self._block(L.off_Sww, K, K, KP)
            self._put_block(L.off_W, w @ R.T, KP)
            self._put_block(L.off_CW, R @ cw @ R.T, KP)
            self._put_block(L.off_Sww, R @ sww @ R.T, KP)
            sc[0] -= 2.0 * logdetR             # log|Lambda_W| of the rotated covariance
        elif node is self.X:
            syx = self._block(L.off_S, D, K, KP)
            sxx = self._block(L.off_S + DP * KP, K, K, KP)
            cx = self._block(L.off_CX, K, K, KP)
            self._put_block(L.off_S, syx @ R.T, KP)
            self._put_block(L.off_S + DP * KP, R @ sxx @ R.T, KP)
            self._put_block(L.off_CX, R @ cx @ R.T, KP)
            sc[1] -= 2.0 * logdetR
            self.kernels.rotate_rows(R, self.Xd, self.N)
            self._x_rows_modified()
        else:
            raise NotImplementedError('rotation of %s' % node.name)
        self.state[L.off_scal:L.off_scal + 2].copy_(torch.from_numpy(sc))
        self._version += 1

    # -- measurement ---------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._materialize()
        self.kernels.set_timing(on)
        self.timing = on

    def last_pass_ms(self):
        return self.kernels.last_pass_ms()

    def pass_times_ms(self, cap=64):
        return self.kernels.pass_times_ms(cap)
