"""
Mini-batch streaming from host memory for stochastic variational inference.

``demos/stochastic_inference.py:99-133`` observes a random subset of the data in every step::

    Y.observe(data[subset, :]);  Q.update(Z);  Q.gradient_step(mu, alpha, scale=rate)

When ``data`` does not fit in HBM it stays on the host (an ndarray, or a ``numpy.memmap`` of a
file larger than RAM) and only mini-batches travel.  :class:`HostBatchStream` keeps the GPU fed:
a worker thread gathers the NEXT mini-batch into pinned memory and enqueues its host->HBM copy on
a dedicated HIP stream while the CURRENT one is being used by the update kernels; ``next()`` hands
out a device tensor that ``Node.observe`` uses in place (no further copy).  The arithmetic of a
step is unchanged -- the stream only replaces ``data[subset, :]`` by the same values, already
resident (``plates_multiplier`` / ``gradient_step`` are in ``inference/vb.py``).
"""
import queue
import threading

import numpy as np

from ..device import get_runtime


class HostBatchStream:
    """Iterator over mini-batches ``data[idx]`` of a host array as fp64 device tensors.

    data     : ndarray / memmap, shape (N, ...); batches are taken along axis 0
    batches  : iterable of index arrays (or slices); e.g. ``(rs.choice(N, NB) for _ in range(T))``
    depth    : batches in flight (pinned + device buffers); 2 = double buffering

    ``for y_dev, idx in HostBatchStream(data, batches): Y.observe(y_dev); ...``
    A yielded tensor stays valid until ``depth - 1`` further batches have been requested.
    """

    def __init__(self, data, batches, depth=2, runtime=None):
        self.data = data
        self.rt = runtime or get_runtime()
        self.torch = torch = self.rt.torch
        self.depth = max(2, int(depth))
        self._it = iter(batches)
        self._copy_stream = torch.cuda.Stream(self.rt.device)
        self._slots = []                 # (pinned host buffer, device buffer) per slot, lazily sized
        self._ready = queue.Queue(maxsize=self.depth - 1)
        self._free = [threading.Event() for _ in range(self.depth)]
        self._release = [None] * self.depth      # event on the consumer's stream: slot may be reused
        self._copied = [None] * self.depth       # event of the slot's last host->HBM copy
        for e in self._free:
            e.set()
        self._stop = False
        self._thread = threading.Thread(target=self._worker, daemon=True)
        self._thread.start()
        self._last_slot = None

    def _buffers(self, slot, shape):
        torch = self.torch
        while len(self._slots) <= slot:
            self._slots.append(None)
        cur = self._slots[slot]
        if cur is None or tuple(cur[0].shape) != tuple(shape):
            host = torch.empty(shape, dtype=torch.float64, pin_memory=True)
            dev = torch.empty(shape, dtype=torch.float64, device=self.rt.device)
            self._slots[slot] = cur = (host, dev)
        return cur

    def _worker(self):
        torch = self.torch
        k = 0
        try:
            for idx in self._it:
                slot = k % self.depth
                self._free[slot].wait()
                self._free[slot].clear()
                if self._stop:
                    break
                if isinstance(idx, slice):
                    n = len(range(*idx.indices(self.data.shape[0])))
                else:
                    idx = np.asarray(idx)
                    n = idx.shape[0]
                host, dev = self._buffers(slot, (n,) + tuple(self.data.shape[1:]))
                # the slot's previous host->HBM copy reads the pinned buffer asynchronously: a
                # consumer that never blocks on the host could otherwise let this gather
                # overwrite the source of a copy that is still queued
                prev = self._copied[slot]
                if prev is not None:
                    prev.synchronize()
                # gather into pinned memory (the only host work of a step), then an async copy
                if isinstance(idx, slice):
                    host.numpy()[...] = self.data[idx]
                else:
                    np.take(self.data, idx, axis=0, out=host.numpy())
                with torch.cuda.stream(self._copy_stream):
                    rel = self._release[slot]
                    if rel is not None:
                        self._copy_stream.wait_event(rel)       # the kernels that used this slot
                    dev.copy_(host, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self._copy_stream)
                self._copied[slot] = done
                self._ready.put((slot, dev, idx, done))
                k += 1
        except Exception as e:       # noqa: BLE001 -- surfaces in the consumer
            self._ready.put(e)
            return
        self._ready.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        torch = self.torch
        cur = torch.cuda.current_stream(self.rt.device)
        if self._last_slot is not None:
            # everything enqueued so far on the consumer's stream has finished with the previous
            # batch before its slot is overwritten
            ev = torch.cuda.Event()
            ev.record(cur)
            self._release[self._last_slot] = ev
            self._free[self._last_slot].set()
        item = self._ready.get()
        if item is None:
            self._last_slot = None
            raise StopIteration
        if isinstance(item, Exception):
            raise item
        slot, dev, idx, done = item
        cur.wait_event(done)                 # no host block: the copy is ordered on the device
        self._last_slot = slot
        return dev, idx

    def close(self):
        self._stop = True
        for e in self._free:
            e.set()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
