"""
One VB iteration of the generic engine as a HIP graph.

The reference's loop (vmp.py:132-172) visits the nodes one by one and evaluates the lower bound
after every sweep; on the generic device engine a sweep of a PCA-sized model is ~130 kernel
launches issued from Python (elementwise formulas, plate sums, contractions), and at N = 1e6 the
host spends longer issuing them than the device spends running them.  The launches of a sweep are
the same every iteration -- same kernels, same shapes, only the contents of the state arrays
move -- so after two identical sweeps the plan records the third into a graph
(hipStreamBeginCapture through ``torch.cuda.graph``; every kernel of the library launches on the
context's stream, which is the capturing stream inside the recording) and replays it afterwards:

    state of the previous sweep  --copy-->  the graph's input arrays
    graph launch: all node updates, all lower-bound terms, all validity flags
    ONE device -> host read: the bound terms and the flags

The graph reads the arrays the state occupied when the recording started and writes the arrays it
allocated while recording (the graph's private pool), so a replay needs the results of the previous
sweep copied back into the inputs; everything else is the device work of the eager sweep, kernel for
kernel, in the same order, hence with identical results.

Anything the recording cannot hold abandons it before HIP sees the call (``Runtime.host_access``:
host reads, uploads, collectives): the plan restores its state and stays on the eager path.  The
state is coherent at every iteration boundary: read-only operations (moments, bound terms,
checkpoints) work between replays, and operations that only replace state arrays (a single
``update``, rotations, ``set_parameters``) keep the graph -- a replay compares the structure of the
present state with the recorded one and copies whatever arrays it finds into the graph's inputs.
Everything else drops the graph: ``observe``, ``load``, another sweep (node list), a changed
mask / annealing / plate multiplier.

``BAYESPY_AMD_GRAPH=0`` keeps every sweep eager.
"""
import os

import numpy as np

from ...darray import DArray
from ...device import GraphCaptureAbort

# operations that leave the state of the plan as it is
READ_ONLY = frozenset((
    'get_moments', 'get_mask', 'lower_bound_contribution', 'lower_bound_contributions', 'save_state',
    'natural_parameters', 'get_parameters', 'log_normalizer', 'gamma_posterior_shape', 'nodes',
    'has_state', 'describe', 'graph_iteration', 'graph_info', 'rotation_rows',
    'rotation_statistics', 'logpdf', 'riemannian_gradient', 'gradient'))

# operations that replace state arrays by arrays of the same structure and touch nothing else: the
# recorded sweep stays valid, a replay first copies the present state into the graph's inputs
STATE_ONLY = frozenset(('rotate_node', 'set_parameters', 'gradient_step'))

_STATE_FIELDS = ('u', 'phi', 'g', 'f', 'u_obs', 'obs_mask')


def _enabled():
    return os.environ.get('BAYESPY_AMD_GRAPH', '1') != '0'


class _Recording:
    __slots__ = ('key', 'graph', 'copy_graph', 'pairs', 'outvec', 'n_bound', 'bound_index', 'factors',
                 'checks', 'template', 'fresh', 'replays', 'copy_bytes', 'old', 'new', 'sig', 'skip')


class GraphIteration:
    """Mixin of GenericPlan: ``graph_iteration(update_nodes, bound_nodes)`` runs one sweep from the
    recorded graph (returns True) or declines (returns False: the caller runs the eager sweep,
    which this class watches through ``_graph_note``)."""

    # -- bookkeeping ------------------------------------------------------------------------------
    def _graph_init(self):
        self._g_rec = None
        self._g_log = []            # operations since the last graph_iteration call
        self._g_warm = 0            # consecutive eager sweeps that followed the expected pattern
        self._g_last_key = None
        self._g_disabled = None     # reason when recording failed / is not possible
        self._g_stash = None
        self._g_inside = False
        self._g_attempts = 0
        self._g_touched = True
        self._mask_epoch = 0

    def _graph_note(self, name, args):
        """Called by the operation wrapper for every outermost operation on the plan."""
        if self._g_inside:
            return
        if name not in READ_ONLY:
            self._g_touched = True          # the state may no longer be what the last replay wrote
        if name == 'update':
            self._g_log.append(('update', id(args[0]) if args else None))
            self._g_stash = None
            return
        if name in READ_ONLY:
            if name == 'lower_bound_contributions':
                self._g_log.append(('bound', None))
            return
        if name in STATE_ONLY:
            self._g_log.append(('state', name))
            self._g_stash = None
            return
        self._g_log.append(('other', name))
        self._g_stash = None
        if self._g_rec is not None:
            self._graph_drop(name)
        self._g_warm = 0

    def _graph_drop(self, why):
        self._g_rec = None
        self._g_warm = 0
        self._g_last_key = None

    def graph_info(self):
        r = self._g_rec
        return {'recorded': r is not None, 'replays': 0 if r is None else r.replays,
                'disabled': self._g_disabled,
                'state_copy_bytes': None if r is None else r.copy_bytes}

    def _graph_key(self, upd, bound):
        self._update_masks()
        # what the user can change between two sweeps without going through the plan: annealing,
        # plate multipliers (stochastic VI), observed flags.  The multipliers a node SEES are derived
        # from the ones set along its ancestors (node.py:294-301: a walk over the graph per node,
        # 70 us of the host time between two replays at config 2); they are recomputed only when one
        # of the values they derive from -- read here directly -- has changed
        nodes = self.__dict__.get('_g_key_nodes')
        if nodes is None:
            seen, nodes = set(), []
            stack = list(self.all)
            while stack:
                n = stack.pop()
                if id(n) in seen:
                    continue
                seen.add(id(n))
                nodes.append(n)
                stack.extend(getattr(n, 'parents', ()))
            self._g_key_nodes = nodes
        def own_multiplier(n):
            d = getattr(n, '__dict__', {})
            if '_plates_multiplier_arg' in d:         # a Node: the value set on it (None: inherited)
                return d['_plates_multiplier_arg']
            m = getattr(n, 'plates_multiplier', None)  # anything else: whatever it shows
            return None if m is None else tuple(np.ravel(m))
        raw = tuple((getattr(n, 'annealing', 1.0), own_multiplier(n), getattr(n, 'observed', False))
                    for n in nodes)
        cached = self.__dict__.get('_g_key_host')
        if cached is None or cached[0] != raw:
            host = []
            for n in self.all:
                host.append((float(getattr(n, 'annealing', 1.0)),
                             tuple(np.ravel(getattr(n, 'plates_multiplier', ()))),
                             bool(getattr(n, 'observed', False))))
            cached = (raw, tuple(host))
            self._g_key_host = cached
        return (tuple(id(n) for n in upd), tuple(id(n) for n in bound), cached[1],
                self._mask_epoch)

    # -- state leaves ------------------------------------------------------------------------------
    def _graph_states(self):
        from . import generic as G
        return [(n, self.state[id(n)]) for n in self.all
                if isinstance(n, G.Stochastic) and id(n) in self.state]

    @staticmethod
    def _leaves(obj, out, sig, path):
        """Device tensors of a state field in a fixed traversal order (``out``) and the shape of
        the structure around them (``sig``)."""
        from . import generic as G
        if obj is None or isinstance(obj, (bool, int, float, np.floating, np.integer)):
            sig.append((path, 'host', None if obj is None else float(obj)))
        elif isinstance(obj, np.ndarray):
            sig.append((path, 'ndarray', obj.shape, obj.tobytes() if obj.size <= 64 else None))
        elif isinstance(obj, G.FactoredMoment):
            # (consumers read the factors whether or not the dense form has been evaluated)
            sig.append((path, 'factored', obj.nd))
            GraphIteration._leaves(obj.cov, out, sig, path + ('cov',))
            GraphIteration._leaves(obj.mean, out, sig, path + ('mean',))
            if obj.sums is not None:
                # plate sums made by the pass that wrote the means: the next sweep's first
                # message reads them, so they travel like the means
                sm = obj.sums
                sig.append((path, 'sums', sm.n, sm.ydesc))
                GraphIteration._leaves(sm.x, out, sig, path + ('sum_x',))
                GraphIteration._leaves(sm.xx, out, sig, path + ('sum_xx',))
                if sm.yx is not None:
                    GraphIteration._leaves(sm.yx, out, sig, path + ('sum_yx',))
        elif isinstance(obj, G.DerivedArray):
            # a function of other state arrays, formed on demand: nothing of its own to carry
            sig.append((path, 'derived', obj.kind, tuple(obj.shape)))
        elif isinstance(obj, G.LazySum):
            t = obj.t                       # a state array is read as a whole: evaluate it
            out.append(t)
            sig.append((path, 'tensor', tuple(t.shape)))
        elif isinstance(obj, DArray):
            out.append(obj.t)
            sig.append((path, 'tensor', tuple(obj.t.shape)))
        elif isinstance(obj, (list, tuple)):
            sig.append((path, type(obj).__name__, len(obj)))
            for i, o in enumerate(obj):
                GraphIteration._leaves(o, out, sig, path + (i,))
        else:
            raise GraphCaptureAbort('state of type %s' % type(obj).__name__)

    def _graph_snapshot(self):
        leaves, sig = [], []
        for n, st in self._graph_states():
            for f in _STATE_FIELDS:
                self._leaves(getattr(st, f), leaves, sig, (id(n), f))
            sig.append((id(n), 'flags', st.observed, st.partial, st.ready, st.stale))
        return leaves, sig

    @staticmethod
    def _dead_inputs(sig, leaves, upd, reads=None):
        """Per leaf of a state snapshot: is it dead as an INPUT of the recorded sweep?  (a) The
        natural parameters and the log-normaliser of a node the sweep updates (``upd``): written
        before anything in the sweep reads them.  (b) With ``reads`` -- the storages the launches
        of the recording read (Runtime.note_reads) --: any array of an updated node that no launch
        read, e.g. the means of a node whose plate sums serve every message that leaves it before
        its own update writes new means.  The recorded graph never reads the copy it was given."""
        tags = [e[0] for e in sig if len(e) >= 2 and e[1] == 'tensor']
        if len(tags) != len(leaves) or os.environ.get('BAYESPY_AMD_GRAPH_COPY_ALL') == '1':
            return [False] * len(leaves)
        upd_ids = set(id(n) for n in upd)
        dead = [len(t) >= 2 and t[0] in upd_ids and t[1] in ('phi', 'g') for t in tags]
        if reads is not None and os.environ.get('BAYESPY_AMD_GRAPH_READ_LOG', '1') != '0':
            for i, (t, leaf) in enumerate(zip(tags, leaves)):
                if not dead[i] and len(t) >= 2 and t[0] in upd_ids \
                        and leaf.untyped_storage().data_ptr() not in reads:
                    dead[i] = True
        return dead

    @staticmethod
    def _constant_state(st):
        """A fully observed node: its arrays never change -- they keep their wrapper objects, so that
        sums over them remembered by identity (sum y^2 of the data: a pass over (D, N)) stay
        remembered through recordings and replays."""
        return bool(st.observed and not st.partial)

    @staticmethod
    def _rewrap(obj, memo):
        """Fresh wrapper objects around the same device tensors: identity-keyed caches and lazily
        evaluated dense forms made from the previous contents cannot be reached through them."""
        from . import generic as G
        if id(obj) in memo:
            return memo[id(obj)]
        if isinstance(obj, G.FactoredMoment):
            sums = obj.sums
            if sums is not None:
                sums = G.PlateSums(GraphIteration._rewrap(sums.x, memo),
                                   GraphIteration._rewrap(sums.xx, memo),
                                   None if sums.yx is None else GraphIteration._rewrap(sums.yx, memo),
                                   sums.ydesc, sums.ykeep, sums.n)
            new = G.FactoredMoment(GraphIteration._rewrap(obj.cov, memo),
                                   GraphIteration._rewrap(obj.mean, memo), obj.nd,
                                   None if obj.logdet_prec is None
                                   else GraphIteration._rewrap(obj.logdet_prec, memo), sums)
        elif isinstance(obj, G.DerivedArray):
            new = G.DerivedArray(obj.kind, [GraphIteration._rewrap(d, memo) for d in obj.deps],
                                 obj.shape)
        elif isinstance(obj, G.LazySum):
            new = DArray(obj.t)
        elif isinstance(obj, DArray):
            new = DArray(obj.t)
        elif isinstance(obj, list):
            new = [GraphIteration._rewrap(o, memo) for o in obj]
        elif isinstance(obj, tuple):
            new = tuple(GraphIteration._rewrap(o, memo) for o in obj)
        else:
            return obj
        memo[id(obj)] = new
        return new

    def _graph_constant_ids(self):
        """Arrays that no sweep changes: data of fully observed nodes, constants, masks."""
        ids = set()

        def add(o):
            if isinstance(o, DArray):
                ids.add(id(o))
            elif isinstance(o, (list, tuple)):
                for x in o:
                    add(x)
        for n, st in self._graph_states():
            if st.observed and not st.partial:
                add(st.u)
        for v in self._const_cache.values():
            add(v)
        for v in self._dev_masks.values():
            add(v[0])
        return ids

    def _graph_reset_caches(self):
        """Forget everything derived from the state (it is about to change under the same
        objects); sums over constants stay."""
        self.__dict__.pop('_det_cache', None)
        self.__dict__.pop('_msg_cache', None)
        self.__dict__.get('_contract_memo', {}).clear()
        sums = self.__dict__.get('_sum_cache')
        if sums:
            const = self._graph_constant_ids()
            for k in [k for k, v in sums.items()
                      if not all(r() is not None and id(r()) in const for r in v[0])]:
                del sums[k]

    # -- recording ---------------------------------------------------------------------------------
    def _graph_record(self, upd, bound, key):
        from . import generic as G
        rt = self.rt
        torch = rt.torch
        states = self._graph_states()
        saved = [(st, {f: getattr(st, f) for f in _STATE_FIELDS + ('stale',)}) for _, st in states]
        self._graph_reset_caches()
        # fresh wrappers: nothing lazily evaluated from the present contents stays reachable
        memo = {}
        for _, st in states:
            if self._constant_state(st):
                continue
            for f in _STATE_FIELDS:
                setattr(st, f, self._rewrap(getattr(st, f), memo))
        try:
            old, sig_old = self._graph_snapshot()
            for t in old:
                # the graph's inputs are written by the copy-back: they must be ordinary arrays
                if any(s == 0 and e > 1 for s, e in zip(t.stride(), t.shape)):
                    raise GraphCaptureAbort('broadcast view in the state')
            # the recording keeps a second copy of the state and the temporaries of one sweep: not
            # for a model that fills the device (the eager sweep reuses the allocator's blocks)
            state_bytes = sum(t.numel() * t.element_size() for t in old)
            free_bytes = torch.cuda.mem_get_info(rt.device)[0] + \
                torch.cuda.memory_reserved(rt.device) - torch.cuda.memory_allocated(rt.device)
            if 4 * state_bytes > free_bytes:
                raise GraphCaptureAbort('%.1f GB of state, %.1f GB free: no room for a recording'
                                        % (state_bytes / 1e9, free_bytes / 1e9))
            rec = _Recording()
            rec.key = key
            rec.graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(rt.device)
            rt._capturing = True
            # no cyclic garbage collection while the stream records: collecting an older plan
            # would destroy ITS graph (hipGraphExecDestroy, pool release) in the middle of this
            # recording, which HIP answers by aborting the process
            import gc
            gc_was_on = gc.isenabled()
            gc.disable()
            sm_was, ew_was = rt._tune_sm, rt._tune_ew
            try:
                from ...utils import misc
                memo_was = misc._CUR_MEMO[0]
                misc._CUR_MEMO[0] = self.__dict__.setdefault('_contract_memo', {})
                self._seed_sums()
                # the queue of small operations (vmp_queue_*) stays open inside the recording: a run
                # of small operations is ONE node, its records are kept by the library and copied to
                # the device once (queue_commit).  Until round 6 a record cost the interpreter what a
                # node costs the graph -- a dependent round trip through memory -- and the queue was
                # switched off here; with the launch's small arrays and its records in LDS it pays
                # (config 2: 0.86 -> 0.71 ms).  BAYESPY_AMD_GRAPH_QUEUE=0 records every operation as
                # a node of its own
                rt.flush_small()
                if os.environ.get('BAYESPY_AMD_GRAPH_QUEUE', '1') == '0':
                    rt.set_tune('small_queue_ew', 0)
                    rt.set_tune('small_queue_sm', 0)
                rt._read_log = set()
                with torch.cuda.graph(rec.graph, capture_error_mode='thread_local'):
                    with rt.operation():
                        for n in upd:
                            type(self).update.__wrapped__(self, n)
                        parts = [self._lower_bound_device(n) for n in bound]
                        items, rt._deferred = rt._deferred, []
                        dev = [t.t.reshape(1) for t, _ in parts if t is not None]
                        rt.flush_small()        # the bound terms are about to be read
                        rec.outvec = self._pack_outputs(dev, [f for f, _, _ in items])
            finally:
                rt.set_tune('small_queue_ew', int(ew_was))
                rt.set_tune('small_queue_sm', int(sm_was))
                try:
                    rt.queue_commit()
                except Exception:         # noqa: BLE001 -- the error of the recording itself matters more
                    pass
                rt._capturing = False
                reads, rt._read_log = rt._read_log, None
                rt._deferred = []
                misc._CUR_MEMO[0] = memo_was
                self.__dict__.get('_contract_memo', {}).clear()
                if gc_was_on:
                    gc.enable()
            rec.bound_index = [None if t is None else 1 for t, _ in parts]
            rec.factors = [f for _, f in parts]
            rec.n_bound = len(dev)
            rec.checks = [(e, m) for _, e, m in items]
            new, sig_new = self._graph_snapshot()
            if sig_old != sig_new:
                raise GraphCaptureAbort('the sweep changed the structure of the state')
            # natural parameters and log-normalisers of the nodes the sweep UPDATES are written
            # before anything reads them (messages travel as moments; phi and g are read by the
            # node's bound term, after its update): as inputs of the graph they are dead, and the
            # copy-back of a replay leaves them out -- at config 2 of the PCA model 136 of 264 MB.
            # (An array that is also reachable through a live field keeps its copy.)
            dead_tag = self._dead_inputs(sig_old, old, upd, reads)
            live_ptrs = set(o.data_ptr() for o, d in zip(old, dead_tag) if not d)
            seen, pairs = set(), []
            for o, n_, d in zip(old, new, dead_tag):
                ko = (o.data_ptr(), tuple(o.shape), tuple(o.stride()))
                kn = (n_.data_ptr(), tuple(n_.shape), tuple(n_.stride()))
                if ko == kn or ko in seen:
                    continue
                if d and o.data_ptr() not in live_ptrs:
                    continue
                seen.add(ko)
                pairs.append((o, n_))
            rec.pairs = pairs
            rec.skip = [bool(d and o.data_ptr() not in live_ptrs) for o, d in zip(old, dead_tag)]
            rec.old, rec.new, rec.sig = old, new, sig_old
            rec.copy_graph = None
            if pairs:
                rec.copy_graph = torch.cuda.CUDAGraph()
                gc.disable()
                try:
                    with torch.cuda.graph(rec.copy_graph, capture_error_mode='thread_local'):
                        self._copy_pairs(pairs)
                finally:
                    if gc_was_on:
                        gc.enable()
            rec.template = [(st, {f: getattr(st, f) for f in _STATE_FIELDS + ('stale',)})
                            for _, st in states]
            rec.fresh = True
            rec.replays = 0
            rec.copy_bytes = int(sum(o.numel() * o.element_size() for o, _ in pairs))
            return rec
        except Exception as e:       # noqa: BLE001 -- whatever went wrong, the eager path works
            for st, fields in saved:
                for f, v in fields.items():
                    setattr(st, f, v)
            self._graph_reset_caches()
            torch.cuda.synchronize(rt.device)
            # structural reasons end the attempts; "no room" / a one-off host access (an upload past a
            # cache's cap) may be gone next time: up to three tries before the plan stays eager
            transient = isinstance(e, GraphCaptureAbort) and ('no room' in str(e) or 'needs the host' in str(e))
            if transient and self._g_attempts < 3:           # (counted by graph_iteration)
                self._g_disabled = None
                self._g_warm = 0
            elif isinstance(e, GraphCaptureAbort):
                self._g_disabled = str(e)
            else:
                self._g_disabled = '%s: %s' % (type(e).__name__, str(e)[:200])
            if os.environ.get('BAYESPY_AMD_GRAPH_DEBUG'):
                import traceback
                traceback.print_exc()
            return None

    def _pack_outputs(self, vals, flags):
        """One vector [bound terms ..., any(flag) ...] for the single read of a replay: ONE launch of
        the library (vmp_pack_outputs) where it takes the arrays as they are, torch otherwise."""
        import ctypes
        rt = self.rt
        torch = rt.torch
        if not vals and not flags:
            return None
        ok = hasattr(rt.lib, 'vmp_pack_outputs') and all(
            f.is_contiguous() and f.dtype in (torch.int32, torch.float64) for f in flags) and all(
            v.is_contiguous() and v.dtype == torch.float64 for v in vals)
        if not ok:
            fl = [f.reshape(-1).any().reshape(1).to(torch.float64) for f in flags]
            return torch.cat(list(vals) + fl)
        ents = [(v, 1, 0) for v in vals] + \
            [(f, f.numel(), 1 if f.dtype == torch.int32 else 2) for f in flags]
        n = len(ents)
        out = torch.empty(n, dtype=torch.float64, device=rt.device)
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _, _ in ents])
        cnt = (ctypes.c_int64 * n)(*[int(c) for _, c, _ in ents])
        kind = (ctypes.c_int32 * n)(*[k for _, _, k in ents])
        rt.check(rt.lib.vmp_pack_outputs(rt.ctx, n, src, cnt, kind, ctypes.c_void_p(out.data_ptr())))
        self._pack_keep = ents          # (operands stay referenced while the recording lives)
        return out

    def _copy_pairs(self, pairs):
        """dst <- src for (dst, src) tensor pairs: the dense ones by ONE launch of the library
        (vmp_copy_many: a dozen small state arrays were a dozen dependent copy nodes), the rest
        one by one."""
        import ctypes
        rt = self.rt
        dense = [(o, n_) for o, n_ in pairs if o.is_contiguous() and n_.is_contiguous()
                 and o.numel() == n_.numel() and o.dtype == n_.dtype == rt.torch.float64]
        for o, n_ in pairs:
            if not any(o is d for d, _ in dense):
                o.copy_(n_)
        if dense and hasattr(rt.lib, 'vmp_copy_many'):
            m = len(dense)
            src = (ctypes.c_void_p * m)(*[n_.data_ptr() for _, n_ in dense])
            dst = (ctypes.c_void_p * m)(*[o.data_ptr() for o, _ in dense])
            cnt = (ctypes.c_int64 * m)(*[o.numel() for o, _ in dense])
            rt.sync_stream()
            rt.check(rt.lib.vmp_copy_many(rt.ctx, m, src, dst, cnt))
        else:
            for o, n_ in dense:
                o.copy_(n_)

    def _graph_replay(self, rec):
        rt = self.rt
        torch = rt.torch
        if not rec.fresh and not self._g_touched:
            # nothing but read-only operations since the last replay: the state is its output
            if rec.copy_graph is not None:
                rec.copy_graph.replay()
        elif not rec.fresh:
            # the graph reads rec.old: whatever the state is NOW goes there first.  Normally the
            # state is what the previous replay wrote (rec.new: one recorded copy graph); arrays
            # replaced since by eager operations (a single update, a rotation) are copied one by
            # one; a state of another structure ends the recording's life
            cur, sig = self._graph_snapshot()
            if sig != rec.sig:
                return None

            def same(a, b):
                return a.data_ptr() == b.data_ptr() and a.shape == b.shape \
                    and a.stride() == b.stride()
            extra, from_new = [], False
            for c, o, n_, skip in zip(cur, rec.old, rec.new, rec.skip):
                if same(c, o):
                    continue
                if same(c, n_):
                    from_new = True
                elif not skip:            # (a dead input needs no copy wherever it lives now)
                    extra.append((o, c))
            if from_new and rec.copy_graph is not None:
                rec.copy_graph.replay()
            for o, c in extra:
                o.copy_(c)
        rec.graph.replay()
        rec.fresh = False
        self._g_touched = False
        rec.replays += 1
        # the state objects of the recording, as fresh wrappers (same device arrays) -- host work that
        # needs no value of the replay: done while the device runs it, BEFORE the read that waits
        memo = {}
        for st, fields in rec.template:
            const = self._constant_state(st)
            for f, v in fields.items():
                setattr(st, f, v if (f == 'stale' or const) else self._rewrap(v, memo))
        self._graph_reset_caches()
        vals = rec.outvec.cpu().numpy() if rec.outvec is not None else np.zeros(0)
        for bad, (exc_type, message) in zip(vals[rec.n_bound:], rec.checks):
            if bad:
                self._graph_drop('a validity check failed')
                raise exc_type(message)
        it = iter(vals[:rec.n_bound])
        return [f if i is None else float(next(it)) * f
                for i, f in zip(rec.bound_index, rec.factors)]

    # -- entry point ---------------------------------------------------------------------------------
    def graph_iteration(self, upd, bound):
        """One sweep over ``upd`` plus the bound terms of ``bound`` from the recorded graph.  False:
        nothing was done, the caller runs the sweep node by node."""
        if not _enabled() or self._g_disabled is not None:
            return False
        rt = self.rt
        if rt.device.type != 'cuda':
            return False
        rt._refresh_dist()
        if rt.world > 1 or os.environ.get('BAYESPY_AMD_SHARD_WORLD1') == '1':
            # a sharded model: its plate sums are completed over the ranks inside the sweep.  The
            # library's RCCL collective (vmp_allreduce_sum_f64) is enqueued on the context's stream
            # and is recorded like a kernel; torch.distributed's (gloo: CPU tests, several ranks on
            # one GPU) is not -- those worlds keep the eager sweeps
            if os.environ.get('BAYESPY_AMD_GRAPH_SHARDED', '1') == '0' or not rt._ensure_comm():
                return False
        key = self._graph_key(upd, bound)
        log, self._g_log = self._g_log, []
        rec = self._g_rec
        if rec is not None and rec.key == key:
            pass
        else:
            if rec is not None:
                self._graph_drop('another sweep')
            # the eager sweep before this one: the updates of `upd` in order, then possibly a
            # callback (a rotation: state-only operations, the update of a hyperparameter), then
            # the bound -- and nothing that drops a graph
            expect = [('update', id(n)) for n in upd]
            seen = [e for e in log if e[0] != 'state']
            clean = seen[:len(expect)] == expect and seen[len(expect):] \
                and seen[-1] == ('bound', None) \
                and all(e[0] == 'update' for e in seen[len(expect):-1])
            if key == self._g_last_key and clean:
                self._g_warm += 1
            else:
                self._g_warm = 0
            self._g_last_key = key
            if self._g_warm < 2:
                return False
            self._g_attempts += 1
            self._g_inside = True
            try:
                rec = self._graph_record(upd, bound, key)
            finally:
                self._g_inside = False
            if rec is None:
                return False
            self._g_rec = rec
        self._g_inside = True
        try:
            vals = self._graph_replay(rec)
        finally:
            self._g_inside = False
        if vals is None:
            self._graph_drop('the state changed its structure')
            return False
        self._g_stash = (tuple(id(n) for n in bound), vals)
        return True
