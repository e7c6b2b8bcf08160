"""
Execution plan of the full-covariance Gaussian-mixture block

    alpha = Dirichlet(a0); z = Categorical(alpha, plates=(N,));
    mu = GaussianARD(0, beta0, shape=(D,), plates=(K,)); Lambda = Wishart(n0, V0, plates=(K,));
    Y = Mixture(z, Gaussian, mu, Lambda)                     (bayespy/demos/mog.py:17-64)

with a fully observed Y, D <= 32, K <= 64.  The plan owns, in HBM: ``Y`` (N, D), the
responsibilities ``R`` (N, K) (= z.u[0]) and one state block (``vmp_gmm_layout``)
holding the statistics T = r^T [1, y, y y^T] that ranks all-reduce and every
replicated quantity.  The only plate-sized work per VB iteration is ONE pass over
Y, issued by ``z.update()`` (``vmp_gmm_pass``).
"""
import ctypes

import numpy as np

from . import _delta

from ... import _lib
from ...device import get_runtime, ptr
from ...nodes.node import Constant
from ...nodes.gaussian import GaussianARD, Gaussian
from ...nodes.wishart import Wishart
from ...nodes.dirichlet import Dirichlet
from ...nodes.categorical import Categorical
from ...nodes.mixture import Mixture


class GMMKernels:

    def __init__(self, rt):
        self.rt, self.lib, self.ctx = rt, rt.lib, rt.ctx

    def layout(self, D, K):
        L = _lib.GMMLayout()
        rc = self.lib.vmp_gmm_get_layout(D, K, ctypes.byref(L))
        if rc != _lib.VMP_OK:
            _lib.raise_for_status(rc, 'fused GMM block supports D <= 32 and K <= 64')
        return L

    def workspace_doubles(self, D, K):
        n = ctypes.c_size_t()
        self.rt.check(self.lib.vmp_gmm_workspace_bytes(self.ctx, D, K, ctypes.byref(n)))
        return (n.value + 7) // 8

    def init_state(self, D, K, alpha0, beta0, n0, V0, state):
        a = np.ascontiguousarray(alpha0, dtype=np.float64)
        v = np.ascontiguousarray(V0, dtype=np.float64)
        self.rt.check(self.lib.vmp_gmm_init_state(
            self.ctx, D, K, a.ctypes.data_as(ctypes.c_void_p), float(beta0), float(n0),
            v.ctypes.data_as(ctypes.c_void_p), ptr(state)))

    def stats_from_labels(self, Y, N, D, K, labels, R, state, ws):
        self.rt.check(self.lib.vmp_gmm_stats_from_labels(self.ctx, ptr(Y), N, D, K, ptr(labels),
                                                         ptr(R), ptr(state), ptr(ws)))

    def update_mu(self, D, K, state):
        self.rt.check(self.lib.vmp_gmm_update_mu(self.ctx, D, K, ptr(state)))

    def update_lambda(self, D, K, state):
        self.rt.check(self.lib.vmp_gmm_update_lambda(self.ctx, D, K, ptr(state)))

    def prepare_z(self, D, K, prior_only, state):
        self.rt.check(self.lib.vmp_gmm_prepare_z(self.ctx, D, K, 1 if prior_only else 0,
                                                 ptr(state)))

    def pass_(self, Y, N, D, K, R, state, ws):
        self.rt.check(self.lib.vmp_gmm_pass(self.ctx, ptr(Y), N, D, K, ptr(R), ptr(state),
                                            ptr(ws)))

    def update_alpha(self, D, K, state):
        self.rt.check(self.lib.vmp_gmm_update_alpha(self.ctx, D, K, ptr(state)))

    def lower_bound(self, D, K, state):
        self.rt.check(self.lib.vmp_gmm_lower_bound(self.ctx, D, K, ptr(state)))

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


class GMMPlan:

    @staticmethod
    def describe():
        return ("Mixture(Categorical(Dirichlet(const)), Gaussian, GaussianARD(0, const, "
                "shape=(D,), plates=(K,)), Wishart(const, const, plates=(K,))), fully observed, "
                "D <= 32, K <= 64")

    @staticmethod
    def match(nodes, why=None):
        def no(Y, msg):
            if why is not None:
                why.append('fused Gaussian-mixture block, observed node %s: %s'
                           % (Y.name or '<unnamed>', msg))
        # mini-batch multipliers (stochastic VI) go through the generic engine
        if any(any(m != 1 for m in n.plates_multiplier) for n in nodes):
            return None
        for Y in nodes:
            if not isinstance(Y, Mixture) or Y.node_class is not Gaussian:
                continue
            if len(Y.parents) != 3 or len(Y.plates) != 1:
                no(Y, 'it needs plates (N,) and parents (z, mu, Lambda)')
                continue
            if Y._mask is not True:
                no(Y, 'it has missing values')
                continue
            z, mu, Lam = Y.parents
            if not (isinstance(z, Categorical) and isinstance(mu, GaussianARD)
                    and isinstance(Lam, Wishart)):
                no(Y, 'its parents are not (Categorical, GaussianARD, Wishart)')
                continue
            alpha = z.parents[0]
            if not (isinstance(alpha, Dirichlet) and all(p == 1 for p in alpha.plates)):
                no(Y, 'the assignment prior is not one Dirichlet node')
                continue
            N = Y.plates[0]
            K, D = Y.clusters, Y.dims[0][0]
            if D > 32 or K > 64:
                no(Y, 'D = %d, K = %d exceed the limits of the block (D <= 32, K <= 64)' % (D, K))
                continue
            if z.plates != (N,) or mu.plates != (K,) or Lam.plates != (K,) or mu.shape != (D,):
                no(Y, 'plates of z / mu / Lambda are not (N,), (K,), (K,)')
                continue
            m0, b0 = mu.parents
            if not (isinstance(m0, Constant) and not np.any(m0.value)
                    and isinstance(b0, Constant) and b0.is_scalar()):
                no(Y, 'the prior of the means is not N(0, c I) with constants')
                continue
            n0, V0 = Lam.parents
            if not (n0.is_scalar() and V0.value.shape == (D, D)):
                no(Y, 'the Wishart prior is not (scalar degrees, one D x D scale)')
                continue
            if any(len(n.children) != 1 for n in (z, mu, Lam, alpha)) or Y.children:
                no(Y, 'one of its roles has other children as well')
                continue
            return dict(Y=Y, z=z, mu=mu, Lambda=Lam, alpha=alpha)
        return None

    def __init__(self, roles, runtime=None, kernels=None):
        self.roles = roles
        self.Y, self.z, self.mu = roles['Y'], roles['z'], roles['mu']
        self.Lam, self.alpha = roles['Lambda'], roles['alpha']
        self.N = self.Y.plates[0]
        self.K, self.D = self.Y.clusters, self.Y.dims[0][0]
        self.alpha0 = np.broadcast_to(self.alpha.parents[0].value, (self.K,)).astype(np.float64)
        self.beta0 = self.mu.parents[1].scalar()
        self.n0 = self.Lam.parents[0].scalar()
        self.V0 = np.array(self.Lam.parents[1].value, dtype=np.float64)
        self._rt, self._kernels = runtime, kernels
        self._ready = False
        self._version = 0
        self._L_version = -1
        self._L = None
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
            self._kernels = GMMKernels(self.rt)
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
        if node is self.Y and node._mask is not True:
            from .generic import GenericPlan
            GenericPlan(self.nodes())

    def _all_reduce_stats(self):
        """Plate sums over the ranks (node.py:650) -- only when the observation plate was
        declared sharded with Node.shard() on z or Y (the same contract as every other plan)."""
        if not any(getattr(n, '_shard_axis', None) is not None for n in (self.z, self.Y)):
            return
        L = self.layout
        self.rt.all_reduce_sum_(self.state[L.off_T:L.off_T + L.len_T])
        self.rt.all_reduce_sum_(self.state[L.off_zs:L.off_zs + 2])

    def _materialize(self):
        if self._ready:
            return
        self._delta = _delta.delta_roles(self.roles)    # point masses until their first update
        rt, k = self.rt, self.kernels
        torch = rt.torch
        N, D, K = self.N, self.D, self.K
        if self.Y._data is None:
            raise ValueError('Node %s has not been observed' % self.Y.name)
        for n in (self.mu, self.Lam, self.alpha):
            if n._init is not None or n.observed:
                raise NotImplementedError('the fused GMM block initialises mu, Lambda and alpha '
                                          'from their priors')
        rt.sync_stream()
        self.layout = L = k.layout(D, K)
        y = self.Y._data
        if isinstance(y, torch.Tensor) and y.device == rt.device and y.dtype == torch.float64 \
                and tuple(y.shape) == (N, D) and y.is_contiguous():
            self.Yd = y
        else:
            ya = np.array(np.broadcast_to(np.asarray(y, dtype=np.float64), (N, D)), order='C')
            self.Yd = torch.from_numpy(ya).to(rt.device)
        self.Rd = rt.empty(N, K)
        self.state = rt.zeros(int(L.total))
        self.ws = rt.empty(int(k.workspace_doubles(D, K)))
        k.init_state(D, K, self.alpha0, self.beta0, self.n0, self.V0, self.state)
        init = self.z._init
        if init is None:
            k.prepare_z(D, K, True, self.state)
            k.pass_(self.Yd, N, D, K, self.Rd, self.state, self.ws)
        else:
            if init[0] == 'value':
                lab = np.asarray(init[1])
                if not np.issubdtype(lab.dtype, np.integer):
                    raise ValueError("Class indices must be integers")
                lab = np.broadcast_to(lab, (N,)).astype(np.int64)
                if lab.size and (lab.min() < 0 or lab.max() >= K):
                    raise ValueError("Class indices out of range [0, %d)" % K)
            else:
                lab = np.random.randint(K, size=N).astype(np.int64)
            self.labels = torch.from_numpy(np.ascontiguousarray(lab)).to(rt.device)
            k.stats_from_labels(self.Yd, N, D, K, self.labels, self.Rd, self.state, self.ws)
        self._all_reduce_stats()
        self._ready = True
        self._version += 1

    def update(self, node):
        self._materialize()
        _delta.updated(self._delta, self.roles, node)
        rt, k = self.rt, self.kernels
        rt.sync_stream()
        D, K = self.D, self.K
        if node is self.mu:
            k.update_mu(D, K, self.state)
        elif node is self.Lam:
            k.update_lambda(D, K, self.state)
        elif node is self.z:
            k.prepare_z(D, K, False, self.state)
            k.pass_(self.Yd, self.N, D, K, self.Rd, self.state, self.ws)
            # child -> parent message sums over the sharded plate (node.py:650)
            self._all_reduce_stats()
        elif node is self.alpha:
            k.update_alpha(D, K, self.state)
        else:
            return
        self._version += 1

    def _lower_bound_terms(self):
        self._materialize()
        if self._L_version != self._version:
            rt, k, L = self.rt, self.kernels, self.layout
            rt.sync_stream()
            k.lower_bound(self.D, self.K, self.state)
            host = self.state[L.off_scal:L.off_L + 8].cpu().numpy()
            status = int(host[3])
            if status != 0:
                _lib.raise_for_status(status)
            t = host[8:]
            self._L = dict(Y=float(t[0]), z=float(t[1]), alpha=float(t[2]), mu=float(t[3]),
                           Lambda=float(t[4]), total=float(t[5]))
            self._L_version = self._version
        return _delta.bound_terms(self._L, self._delta)

    def lower_bound_contribution(self, node):
        terms = self._lower_bound_terms()
        for key in ('Y', 'z', 'alpha', 'mu', 'Lambda'):
            if node is self.roles[key]:
                return terms[key]
        return 0.0

    def _blk(self, off, shape):
        n = int(np.prod(shape))
        return self.state[off:off + n].cpu().numpy().reshape(shape).copy()

    # -- persistence: the packed device state + responsibilities ------------------------------------
    def save_state(self, put, nodes, index):
        self._materialize()
        base = 'plans/%d/' % index
        _delta.save(put, base, self._delta)
        put(base + 'kind', np.array([ord(c) for c in 'gmm'], dtype=np.uint8))
        put(base + 'dims', np.array([self.N, self.D, self.K], dtype=np.int64))
        put(base + 'state', self.state.cpu().numpy())
        put(base + 'R', self.Rd.cpu().numpy())

    def load_state(self, reader, nodes, index):
        self._materialize()
        base = 'plans/%d/' % index
        self._delta = _delta.load(reader, base)
        if not reader.has(base + 'state'):
            raise Exception("File does not contain the state of the fused mixture block")
        dims = tuple(int(v) for v in reader.get(base + 'dims'))
        if dims != (self.N, self.D, self.K):
            raise ValueError('checkpoint is for (N, D, K) = %s, the model has %s'
                             % (dims, (self.N, self.D, self.K)))
        torch = self.rt.torch
        self.state.copy_(torch.from_numpy(np.array(reader.get(base + 'state'), dtype=np.float64)))
        self.Rd.copy_(torch.from_numpy(np.array(reader.get(base + 'R'), dtype=np.float64)))
        self._version += 1

    def get_moments(self, node):
        self._materialize()
        L, D, K = self.layout, self.D, self.K
        if node is self.z:
            return [self.Rd.cpu().numpy()]
        if node is self.mu:
            m = self._blk(L.off_mu, (K, D))
            C = self._blk(L.off_Cmu, (K, D, D))
            return [m, C + m[:, :, None] * m[:, None, :]]
        if node is self.Lam:
            return [self._blk(L.off_Lam, (K, D, D)), self._blk(L.off_logdetLam, (K,))]
        if node is self.alpha:
            return [self._blk(L.off_alpha + L.KP, (K,)).reshape(self.alpha.plates + (K,))]
        if node is self.Y:
            y = self.Yd.cpu().numpy()
            return [y, y[:, :, None] * y[:, None, :]]
        raise NotImplementedError

    def statistics(self):
        """(R_k, sum r y, sum r y y^T) -- host copies of the all-reduced statistics."""
        self._materialize()
        L, D, K = self.layout, self.D, self.K
        T = self._blk(L.off_T, (int(L.KP), int(L.FS)))[:K]
        return T[:, 0], T[:, 1:1 + D], T[:, 1 + D:].reshape(K, D, D)

    def enable_timing(self, on=True):
        self._materialize()
        self.kernels.set_timing(on)

    def last_pass_ms(self):
        return self.kernels.last_pass_ms()

    def pass_times_ms(self, cap=64):
        return self.kernels.pass_times_ms(cap)
