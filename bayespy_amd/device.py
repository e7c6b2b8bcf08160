"""
Device runtime: one process per GPU, one caller-visible HIP stream, one ``vmp_ctx``.

PyTorch is used here as plumbing only -- device memory (caching allocator),
the current HIP stream and ``torch.distributed`` (backend "nccl" == RCCL over
xGMI).  All arithmetic on plate-sized arrays goes through libvmp_hip.so.
"""
import ctypes
import os

import numpy as np

from . import _lib


class Runtime:
    """Process-wide device runtime (lazy singleton, see :func:`get_runtime`)."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError(
                    'bayespy_amd needs an AMD GPU (HIP device); none is visible and there is '
                    'no CPU fallback.')
            ndev = torch.cuda.device_count()
            idx = int(os.environ.get('LOCAL_RANK', '0')) % max(ndev, 1)
            torch.cuda.set_device(idx)
            self.device = torch.device('cuda', idx)
        else:
            self.device = torch.device(device)
        self.lib = None
        self.ctx = None
        self._queue_alive = []
        self._queue_env = None
        self._comm_state = None     # None: not decided; True: the library owns an RCCL communicator
        self._comm_reason = None    # why the library communicator is not in use, if it is not
        self.collective_calls = {'library': 0, 'torch': 0}
        self._op_depth = 0
        self._deferred = []
        self._capturing = False          # a plan is recording an iteration into a HIP graph
        if self.device.type == 'cuda':
            self.lib = _lib.load()
            ctx = ctypes.c_void_p()
            # A dedicated non-blocking stream becomes torch's current stream: work issued on
            # the legacy null stream would implicitly serialise with the library's internal
            # plate stream (CU-masked streams are "blocking" streams in HIP).
            self.stream = torch.cuda.Stream(self.device)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            torch.cuda.set_stream(self.stream)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.vmp_ctx_create(self.device.index, ctypes.c_void_p(stream),
                                         ctypes.byref(ctx))
            if rc != _lib.VMP_OK:
                _lib.raise_for_status(rc, self.lib.vmp_last_error(None).decode())
            self.ctx = ctx
        self._refresh_dist()

    # -- distributed ---------------------------------------------------------
    def _refresh_dist(self):
        dist = self.torch.distributed
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank()
            self.world = dist.get_world_size()
        else:
            self.rank, self.world = 0, 1

    def _ensure_comm(self):
        """Give the library its own RCCL communicator over the ranks of the
        ``torch.distributed`` world (``vmp_comm_init_rank``).  torch.distributed is the
        rendezvous only: it ships rank 0's 128-byte id.  Worlds on another backend (gloo: CPU
        tests, several ranks on one GPU) keep torch's collective."""
        if self._comm_state is not None:
            return self._comm_state
        dist = self.torch.distributed
        if self.ctx is None or not (dist.is_available() and dist.is_initialized()):
            return False          # not cached: a later init_process_group still gets RCCL
        self._comm_state = False
        if os.environ.get('BAYESPY_AMD_COLLECTIVE') == 'torch':
            self._comm_reason = 'BAYESPY_AMD_COLLECTIVE=torch'
            return False
        if dist.get_backend() != 'nccl':
            self._comm_reason = 'backend %s' % dist.get_backend()
            return False
        cid = ctypes.create_string_buffer(128)
        box = [None]
        if self.rank == 0:
            # a failure here (librccl missing, ...) must not strand the other ranks in the
            # broadcast below: ship a sentinel and let every rank fall back together
            if self.lib.vmp_comm_unique_id(self.ctx, cid) == 0:
                box = [cid.raw]
        dist.broadcast_object_list(box, src=0)
        rc = -1
        if box[0] is not None:
            cid = ctypes.create_string_buffer(box[0], 128)
            rc = self.lib.vmp_comm_init_rank(self.ctx, cid, self.rank, self.world)
        # every rank must take the same path: agree on the outcome through the rendezvous
        ok = self.torch.tensor([1 if rc == 0 else 0], dtype=self.torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            import warnings
            msg = self.lib.vmp_last_error(self.ctx)
            self._comm_reason = msg.decode() if msg else 'a rank failed to join'
            if os.environ.get('BAYESPY_AMD_COLLECTIVE') != 'torch-fallback':
                # ONE data-path collective: on the nccl backend the plate sums are the library's
                # vmp_allreduce_sum_f64 or nothing.  (BAYESPY_AMD_COLLECTIVE=torch selects
                # torch.distributed.all_reduce explicitly, =torch-fallback permits it when the
                # communicator cannot be created.)
                raise RuntimeError('library RCCL communicator not available (%s); set '
                                   'BAYESPY_AMD_COLLECTIVE=torch to run the plate sums through '
                                   'torch.distributed.all_reduce instead' % self._comm_reason)
            warnings.warn('library RCCL communicator not available (%s): plate sums go through '
                          'torch.distributed.all_reduce' % self._comm_reason)
            if rc == 0:
                self.lib.vmp_comm_destroy(self.ctx)
            return False
        self._comm_state = True
        return True

    def host_access(self, what):
        """Called where device data meets the host (a read-back, an upload, a collective): while a
        plan records an iteration into a HIP graph such a step cannot be recorded, and the
        recording is abandoned BEFORE the HIP call that would invalidate the capture."""
        if self._capturing:
            raise GraphCaptureAbort('needs the host: %s' % what)
        self.flush_small()

    def all_reduce_sum_(self, tensor):
        """In-place sum over ranks: ``vmp_allreduce_sum_f64`` (RCCL all-reduce over xGMI,
        enqueued on the context's stream by the library).  This is the ONLY data-path
        collective: it stands where the reference sums a message over a plate the parent
        lacks (node.py:650, dot.py:581) and where it sums the per-node lower bound
        (expfamily.py:470-480)."""
        if self._capturing and self._comm_state is True:
            # the library's collective is a launch on the context's stream like its kernels: a
            # sweep recording holds it (graph_iter.py establishes the communicator beforehand)
            self.flush_small()
        else:
            self.host_access('all_reduce_sum_')
        self._refresh_dist()
        if self._ensure_comm():
            if tensor.numel() == 0:
                return tensor
            buf = tensor if tensor.is_contiguous() else tensor.contiguous()
            if buf.dtype != self.torch.float64:
                raise TypeError('plate sums are fp64')
            self.sync_stream()
            self.check(self.lib.vmp_allreduce_sum_f64(self.ctx, ptr(buf), buf.numel()))
            self.collective_calls['library'] += 1
            if buf is not tensor:
                tensor.copy_(buf)
            return tensor
        if self.world > 1:
            if os.environ.get('BAYESPY_AMD_COLLECTIVE') == 'library':
                raise RuntimeError('plate sum would go through torch.distributed.all_reduce (%s) '
                                   'but BAYESPY_AMD_COLLECTIVE=library forbids that path'
                                   % (self._comm_reason or 'no library communicator'))
            # reached only on a non-nccl backend (gloo: the CPU test-suite, several ranks on one
            # GPU), with BAYESPY_AMD_COLLECTIVE=torch, or with the permitted fallback
            self.torch.distributed.all_reduce(tensor)
            self.collective_calls['torch'] += 1
        return tensor

    def comm_info(self):
        """Which collective the plate sums of this process use and the world it spans:
        ``path`` = 'vmp_allreduce_sum_f64' (the library's RCCL communicator), 'torch'
        (torch.distributed.all_reduce: CPU test backend, or the fallback) or 'none' (one rank,
        no process group); ``world`` as the LIBRARY's communicator reports it
        (``vmp_comm_info``); ``rccl_loaded`` = librccl is mapped into this process."""
        self._refresh_dist()
        lib_world = 1
        if self.ctx is not None:
            r, w = ctypes.c_int32(0), ctypes.c_int32(1)
            self.lib.vmp_comm_info(self.ctx, ctypes.byref(r), ctypes.byref(w))
            lib_world = int(w.value)
        dist = self.torch.distributed
        have_group = dist.is_available() and dist.is_initialized()
        if have_group and self._ensure_comm():
            path = 'vmp_allreduce_sum_f64'
            r, w = ctypes.c_int32(0), ctypes.c_int32(1)
            self.lib.vmp_comm_info(self.ctx, ctypes.byref(r), ctypes.byref(w))
            lib_world = int(w.value)
        elif have_group and self.world > 1:
            path = 'torch'
        else:
            path = 'none'
        try:
            rccl = any('librccl' in line for line in open('/proc/self/maps'))
        except OSError:
            rccl = None
        return {'path': path, 'world': lib_world, 'torch_world': self.world,
                'rccl_loaded': rccl, 'calls': dict(self.collective_calls),
                'fallback_reason': self._comm_reason if path != 'vmp_allreduce_sum_f64' else None}

    def all_reduce_int(self, value):
        self._refresh_dist()
        if self.world == 1:
            return int(value)
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.device)
        self.torch.distributed.all_reduce(t)
        return int(t.item())

    # -- memory ----------------------------------------------------------------
    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=self.device)

    def to_device(self, array):
        """Host ndarray (any float dtype) or torch tensor -> fp64 device tensor."""
        torch = self.torch
        if isinstance(array, torch.Tensor):
            if not array.is_cuda:
                self.host_access('to_device')
            return array.to(device=self.device, dtype=torch.float64)
        self.host_access('to_device')
        a = np.array(array, dtype=np.float64, order='C', copy=True)   # keeps 0-d arrays 0-d
        return torch.from_numpy(a).to(self.device)

    # -- C ABI helpers -----------------------------------------------------------
    def check(self, rc):
        if rc != _lib.VMP_OK:
            msg = self.lib.vmp_last_error(self.ctx).decode(errors='replace')
            _lib.raise_for_status(rc, msg)

    # -- the library's queue of small operations (include/vmp_hip.h: vmp_queue_*) ---------------
    # open for the duration of a plan operation: the formulas and plate sums on scalars / K x K
    # arrays of a sweep are run by a few launches of one interpreter kernel instead of one each.
    # Whatever reads their outputs outside the library (a host read, a torch operation) flushes
    # first: host_access() does.
    def queue_begin(self):
        if self.ctx is not None and self.lib is not None:
            if self._queue_env is None:
                self._queue_env = os.environ.get('BAYESPY_AMD_SMALL_QUEUE', '1') != '0'
                if not self._queue_env:
                    self.check(self.lib.vmp_tune_set(b'small_queue', 0))
                # formulas, small plate sums and K x K inverses (round 6: the interpreter keeps the
                # small arrays of a launch in LDS, which made queueing pay inside recorded sweeps).
                # The order of the additions of a queued sum depends on its shape alone, so eager
                # and recorded sweeps agree bit for bit; a single launch outside an operation (the
                # stand-alone kernels) agrees to rounding.  BAYESPY_AMD_SMALL_QUEUE=ew: formulas only
                self.set_tune('small_queue_sm',
                              0 if os.environ.get('BAYESPY_AMD_SMALL_QUEUE', '1') == 'ew' else 1)
                # formulas of up to this many elements are records (the library's own default is 2048)
                self.set_tune('small_queue_ew_max', int(os.environ.get('BAYESPY_AMD_QUEUE_EW_MAX', 2048)))
            self.check(self.lib.vmp_queue_begin(self.ctx))

    def queue_end(self):
        if self.ctx is not None and self.lib is not None:
            self.check(self.lib.vmp_queue_end(self.ctx))
        self._queue_alive = []

    def flush_small(self):
        if self.ctx is not None and self.lib is not None:
            self.check(self.lib.vmp_queue_flush(self.ctx))
        self._queue_alive = []

    def queue_commit(self):
        """After a stream capture that recorded flushes: the device copies of their records
        (vmp_queue_commit) -- before the first replay."""
        if self.ctx is not None and self.lib is not None:
            self.check(self.lib.vmp_queue_commit(self.ctx))

    def queue_collects_sums(self):
        """Small contractions go to the queue of small operations right now (it is open and
        "small_queue_sm" is on): the caller then prefers vmp_sum_multiply over a GEMM launch."""
        return self._op_depth > 0 and self._queue_env and self.lib is not None \
            and self._tune_sm

    def keep_until_flush(self, arrays, out, kind='ew'):
        """A queued operation runs LATER: its operands and its result must not go back to the
        allocator before (a block handed out again would be written by something else first).
        References are kept only when the library can have queued the call -- the queue is open,
        the tune of this kind of operation (``'ew'`` formulas, ``'sm'`` sums and inverses) is on,
        and the result is small (the library queues <= 2048 outputs) -- so plate-sized temporaries
        go back to the allocator at once, also inside a sweep recording (where both tunes are off);
        the list is dropped at every flush this side knows of."""
        if self._op_depth == 0 or self.lib is None or not self._queue_env:
            return
        if not (self._tune_sm if kind == 'sm' else self._tune_ew):
            return
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for o in outs:
            t = getattr(o, 't', o)
            if hasattr(t, 'numel') and t.numel() > self._queue_max_out:
                return
        self._queue_alive.append((arrays, out))

    _tune_sm = False
    _tune_ew = True
    _queue_max_out = 2048        # elements of a result the library may have queued (its tune small_queue_ew_max)

    # which device arrays the kernels of a sweep recording READ (graph_iter.py: an input of the
    # recorded graph that no launch reads needs no copy-back before a replay); None = not logging
    _read_log = None

    def note_reads(self, arrays):
        log = self._read_log
        if log is None:
            return
        for a in arrays:
            t = getattr(a, 't', a) if a is not None else None
            if t is not None and hasattr(t, 'untyped_storage'):
                log.add(t.untyped_storage().data_ptr())

    def set_tune(self, key, value):
        if self.lib is not None:
            self.check(self.lib.vmp_tune_set(key.encode(), int(value)))
            if key == 'small_queue_sm':
                self._tune_sm = bool(value)
            elif key == 'small_queue_ew':
                self._tune_ew = bool(value)
            elif key == 'small_queue_ew_max':
                self._queue_max_out = max(2048, int(value))

    def queue_stats(self):
        if self.ctx is None or self.lib is None:
            return None
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self.check(self.lib.vmp_queue_stats(self.ctx, ctypes.byref(a), ctypes.byref(b)))
        return {'launches': a.value, 'operations': b.value}

    def sync_stream(self):
        """Point the context at torch's current stream before launches.  Inside a plan
        operation (``with rt.operation():``) the lookup is done once at entry, not per launch."""
        if self.ctx is not None and self._op_depth == 0:
            s = self.torch.cuda.current_stream(self.device).cuda_stream
            self.lib.vmp_ctx_set_stream(self.ctx, ctypes.c_void_p(s))

    def operation(self):
        """Context of one plan-level operation (a node update, a lower-bound term): the stream
        is resolved once, and validity checks that would need a device->host read each
        (positive definiteness, positivity of natural parameters) are queued on the device and
        read together by :meth:`check_deferred` -- at the latest when the operation ends."""
        return _Operation(self)

    def defer_check(self, flag_tensor, exc_type, message):
        """``flag_tensor``: device tensor, any non-zero element means failure."""
        if self._op_depth == 0:
            self.host_access('defer_check')
            if bool(flag_tensor.any().item()):
                raise exc_type(message)
            return
        # (reduced when the checks are read: the flag may be the result of a queued operation)
        self._deferred.append((flag_tensor, exc_type, message))

    def check_deferred(self):
        if not self._deferred:
            return
        self.host_access('check_deferred')
        items, self._deferred = self._deferred, []
        flags = self.torch.cat([f.reshape(-1).any().reshape(1) for f, _, _ in items]).cpu().numpy()
        for bad, (_, exc_type, message) in zip(flags, items):
            if bad:
                raise exc_type(message)

    def synchronize(self):
        if self.device.type == 'cuda':
            self.torch.cuda.synchronize(self.device)


class GraphCaptureAbort(RuntimeError):
    """An iteration being recorded into a HIP graph needs the host (see Runtime.host_access)."""


class _Operation:
    def __init__(self, rt):
        self.rt = rt

    def __enter__(self):
        rt = self.rt
        if rt._op_depth == 0:
            rt.sync_stream()
            rt.queue_begin()
        rt._op_depth += 1
        return rt

    def __exit__(self, exc_type, exc, tb):
        rt = self.rt
        rt._op_depth -= 1
        if rt._op_depth == 0:
            rt.queue_end()
            if exc_type is None:
                rt.check_deferred()
            else:
                rt._deferred = []
        return False


_runtime = None


def get_runtime():
    global _runtime
    if _runtime is None:
        _runtime = Runtime()
    return _runtime


def set_runtime(rt):
    """Install a runtime explicitly (tests inject a CPU runtime together with a
    kernel test double; the product path never does this)."""
    global _runtime
    _runtime = rt


_EMPTY_SENTINEL = {}


def ptr(tensor):
    """Device address of a tensor for the C ABI.  An EMPTY tensor (an empty local plate: a rank of
    a sharded run without any observation) has no storage; the entry points get the address of a
    small per-device scratch block instead of NULL -- with a zero count nothing of it is accessed,
    and NULL keeps meaning "argument missing"."""
    p = tensor.data_ptr()
    if p == 0 and tensor.numel() == 0 and tensor.is_cuda:
        key = tensor.device.index
        if key not in _EMPTY_SENTINEL:
            import torch
            _EMPTY_SENTINEL[key] = torch.zeros(64, dtype=torch.float64, device=tensor.device)
        p = _EMPTY_SENTINEL[key].data_ptr()
    return ctypes.c_void_p(p)
