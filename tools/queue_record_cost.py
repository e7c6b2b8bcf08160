"""What ONE record costs the interpreter of small operations (vmp_queue_*), per kind: launches of 100
identical records through the raw C ABI, timed with events on the context's stream; every record reads
what the record before it wrote (a dependent chain, like a sweep)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayespy_amd import _lib
from bayespy_amd.darray import OP_IN, OP_MUL, OP_ADD, OP_CONST, OP_LOG, OP_DIGAMMA
lib = _lib.load()
dev = torch.device('cuda', 0)
stream = torch.cuda.Stream(dev)
ctx = ctypes.c_void_p()
assert lib.vmp_ctx_create(0, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(ctx)) == 0
lib.vmp_tune_set(b'small_queue_sm', 1)
K = 16
vp = lambda t: ctypes.c_void_p(t.data_ptr())
ws = torch.empty(1 << 20, dtype=torch.float64, device=dev)
NREC = 100
bufs = [torch.rand(K, K, dtype=torch.float64, device=dev) + 1.0 for _ in range(NREC + 1)]
vecs = [torch.rand(K, dtype=torch.float64, device=dev) + 1.0 for _ in range(NREC + 1)]
scal = [torch.rand((), dtype=torch.float64, device=dev) + 1.0 for _ in range(NREC + 1)]
b = torch.rand(K, dtype=torch.float64, device=dev)


def ew_kk(i):       # bufs[i+1] = bufs[i] * b + 2     (K x K, broadcast operand)
    shape = (ctypes.c_int64 * 2)(K, K)
    ins = (ctypes.c_void_p * 2)(bufs[i].data_ptr(), b.data_ptr())
    strides = (ctypes.c_int64 * 4)(K, 1, 0, 1)
    ops = (ctypes.c_int32 * 5)(OP_IN | (0 << 8), OP_IN | (1 << 8), OP_MUL, OP_CONST | (0 << 8), OP_ADD)
    consts = (ctypes.c_double * 1)(2.0)
    assert lib.vmp_ewise(ctx, 2, shape, 2, ins, strides, 5, ops, 1, consts, vp(bufs[i + 1])) == 0


def ew_scalar(i):   # scal[i+1] = log(scal[i]) + 2
    shape = (ctypes.c_int64 * 1)(1)
    ins = (ctypes.c_void_p * 1)(scal[i].data_ptr())
    strides = (ctypes.c_int64 * 1)(0)
    ops = (ctypes.c_int32 * 4)(OP_IN | (0 << 8), OP_LOG, OP_CONST | (0 << 8), OP_ADD)
    consts = (ctypes.c_double * 1)(2.0)
    assert lib.vmp_ewise(ctx, 1, shape, 1, ins, strides, 4, ops, 1, consts, vp(scal[i + 1])) == 0


def ew_copy(i):     # scal[i+1] = scal[i]         (one program word)
    shape = (ctypes.c_int64 * 1)(1)
    ins = (ctypes.c_void_p * 1)(scal[i].data_ptr())
    strides = (ctypes.c_int64 * 1)(0)
    ops = (ctypes.c_int32 * 1)(OP_IN | (0 << 8))
    consts = (ctypes.c_double * 1)(2.0)
    assert lib.vmp_ewise(ctx, 1, shape, 1, ins, strides, 1, ops, 0, consts, vp(scal[i + 1])) == 0


def ew_long(i):     # scal[i+1] = ((scal[i] * 1 + 0) * 1 + 0) ...   (33 program words)
    shape = (ctypes.c_int64 * 1)(1)
    ins = (ctypes.c_void_p * 1)(scal[i].data_ptr())
    strides = (ctypes.c_int64 * 1)(0)
    words = [OP_IN | (0 << 8)]
    for _ in range(8):
        words += [OP_CONST | (0 << 8), OP_MUL, OP_CONST | (1 << 8), OP_ADD]
    ops = (ctypes.c_int32 * len(words))(*words)
    consts = (ctypes.c_double * 2)(1.0, 0.0)
    assert lib.vmp_ewise(ctx, 1, shape, 1, ins, strides, len(words), ops, 2, consts, vp(scal[i + 1])) == 0


def ew_digamma(i):  # vecs[i+1] = digamma(vecs[i]) + 2
    shape = (ctypes.c_int64 * 1)(K)
    ins = (ctypes.c_void_p * 1)(vecs[i].data_ptr())
    strides = (ctypes.c_int64 * 1)(1)
    ops = (ctypes.c_int32 * 4)(OP_IN | (0 << 8), OP_DIGAMMA, OP_CONST | (0 << 8), OP_ADD)
    consts = (ctypes.c_double * 1)(2.0)
    assert lib.vmp_ewise(ctx, 1, shape, 1, ins, strides, 4, ops, 1, consts, vp(vecs[i + 1])) == 0


def sum_mv(i):      # vecs[i+1][r] = sum_c bufs[0][r, c] * vecs[i][c]
    shape = (ctypes.c_int64 * 2)(K, K)
    ins = (ctypes.c_void_p * 2)(bufs[0].data_ptr(), vecs[i].data_ptr())
    strides = (ctypes.c_int64 * 4)(K, 1, 0, 1)
    ostr = (ctypes.c_int64 * 2)(1, 0)
    assert lib.vmp_sum_multiply(ctx, 2, shape, 2, ins, strides, ostr, ctypes.c_uint32(2), 1.0 / K,
                                vp(vecs[i + 1]), vp(ws), ws.numel() * 8) == 0


def sum_all(i):     # scal[i+1] = sum bufs[0] * bufs[1] * scal[i]
    shape = (ctypes.c_int64 * 2)(K, K)
    ins = (ctypes.c_void_p * 3)(bufs[0].data_ptr(), bufs[1].data_ptr(), scal[i].data_ptr())
    strides = (ctypes.c_int64 * 6)(K, 1, K, 1, 0, 0)
    ostr = (ctypes.c_int64 * 2)(0, 0)
    assert lib.vmp_sum_multiply(ctx, 2, shape, 3, ins, strides, ostr, ctypes.c_uint32(3), 1.0 / (K * K * 4),
                                vp(scal[i + 1]), vp(ws), ws.numel() * 8) == 0


spd = torch.eye(K, dtype=torch.float64, device=dev) * 3 + 0.1
sinv = [torch.empty(K, K, dtype=torch.float64, device=dev) for _ in range(NREC + 1)]
sld = torch.empty(1, dtype=torch.float64, device=dev)
sinfo = torch.empty(1, dtype=torch.int32, device=dev)


def spd_inv(i):
    assert lib.vmp_spd_batched(ctx, K, 1, vp(spd), vp(sinv[i + 1]), vp(sld), vp(sinfo)) == 0


print('library', lib.vmp_version().decode())
BUSY = '--busy' in sys.argv      # the same launches beside a stream of large matrix products: is a lone
                                 # workgroup on an otherwise idle chip clocked down?
side = torch.cuda.Stream(dev)
big = torch.randn(6144, 6144, dtype=torch.float64, device=dev)
for lds in ((1,) if BUSY else (0, 1)):
    lib.vmp_tune_set(b'small_queue_lds', lds)
    for name, fn in (('formula: copy of a scalar', ew_copy), ('formula: 33 words on a scalar', ew_long), ('formula 16 x 16', ew_kk), ('formula scalar log', ew_scalar), ('formula digamma(16)', ew_digamma),
                     ('sum 16 x 16 . 16', sum_mv), ('sum of 256 products of three', sum_all), ('inverse 16 x 16', spd_inv)):
        ts = []
        for rep in range(5):
            assert lib.vmp_queue_begin(ctx) == 0
            for i in range(NREC):
                fn(i)
            if BUSY:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        big @ big
            with torch.cuda.stream(stream):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                assert lib.vmp_queue_end(ctx) == 0
                e1.record(stream)
            assert lib.vmp_ctx_sync(ctx) == 0
            ts.append(e0.elapsed_time(e1) * 1e3 / NREC)
        print('arrays in LDS = %d%s  %-32s %6.2f us per record' % (lds, ' beside GEMMs' if BUSY else '', name, min(ts)))
lib.vmp_ctx_destroy(ctx)
