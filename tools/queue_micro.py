"""Micro-benchmark of the interpreter of queued small operations: R chained formulas / sums per
launch, time per record."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayespy_amd import _lib
from bayespy_amd.darray import OP_IN, OP_MUL, OP_ADD, OP_CONST
lib = _lib.load()
dev = torch.device('cuda', 0)
stream = torch.cuda.Stream(dev)
ctx = ctypes.c_void_p()
assert lib.vmp_ctx_create(0, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(ctx)) == 0
R = int(os.environ.get('R', '64'))
ws = torch.empty(1 << 16, dtype=torch.float64, device=dev)


def ew_chain(total, queued):
    bufs = [torch.ones(total, dtype=torch.float64, device=dev) for _ in range(R + 1)]
    b = torch.full((total,), 0.5, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    shape = (ctypes.c_int64 * 1)(total)
    strides = (ctypes.c_int64 * 2)(1, 1)
    ops = (ctypes.c_int32 * 5)(OP_IN | (0 << 8), OP_IN | (1 << 8), OP_MUL, OP_CONST | (0 << 8), OP_ADD)
    consts = (ctypes.c_double * 1)(0.25)

    def run():
        if queued:
            assert lib.vmp_queue_begin(ctx) == 0
        for i in range(R):
            ins = (ctypes.c_void_p * 2)(bufs[i].data_ptr(), b.data_ptr())
            assert lib.vmp_ewise(ctx, 1, shape, 2, ins, strides, 5, ops, 1, consts,
                                 ctypes.c_void_p(bufs[i + 1].data_ptr())) == 0
        if queued:
            assert lib.vmp_queue_end(ctx) == 0
    return run, bufs


def sum_chain(nkeep, nred, queued):
    a = torch.randn(nkeep, nred, dtype=torch.float64, device=dev)
    outs = [torch.ones(nkeep, dtype=torch.float64, device=dev) for _ in range(R + 1)]
    torch.cuda.synchronize()
    shape = (ctypes.c_int64 * 2)(nkeep, nred)
    strides = (ctypes.c_int64 * 4)(nred, 1, 1, 0)
    ostr = (ctypes.c_int64 * 2)(1, 0)

    def run():
        if queued:
            assert lib.vmp_queue_begin(ctx) == 0
        for i in range(R):
            ins = (ctypes.c_void_p * 2)(a.data_ptr(), outs[i].data_ptr())
            assert lib.vmp_sum_multiply(ctx, 2, shape, 2, ins, strides, ostr, ctypes.c_uint32(2), 1.0,
                                        ctypes.c_void_p(outs[i + 1].data_ptr()),
                                        ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8) == 0
        if queued:
            assert lib.vmp_queue_end(ctx) == 0
    return run, outs


def timeit(run, reps=20):
    with torch.cuda.stream(stream):
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # graph the launches so that host issue time does not count
        g = ctypes.c_void_p()
        assert lib.vmp_graph_begin(ctx) == 0
        run()
        assert lib.vmp_graph_end(ctx, ctypes.byref(g)) == 0
        lib.vmp_graph_launch(ctx, g); torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            lib.vmp_graph_launch(ctx, g)
        e1.record(stream)
        torch.cuda.synchronize()
        lib.vmp_graph_destroy(ctx, g)
    return e0.elapsed_time(e1) / reps * 1e3 / R


for total in (1, 256, 2048):
    for q in (False, True):
        run, _ = ew_chain(total, q)
        print('ew total=%5d %-8s %.2f us/record' % (total, 'queued' if q else 'kernels', timeit(run)))
for nkeep, nred in ((1, 1), (1, 256), (256, 64), (1024, 16), (16, 64)):
    for q in (False, True):
        run, _ = sum_chain(nkeep, nred, q)
        print('sum %4d x %4d %-8s %.2f us/record' % (nkeep, nred, 'queued' if q else 'kernels', timeit(run)))
