"""
Device arrays for the generic VMP path.

``DArray`` is an fp64 array in HBM with NumPy's shape/broadcast semantics (the
reference's plate convention: leading axes are plates, unit or missing axes
mean "same for all").  Storage and *views* (reshape / broadcast / basic
indexing, i.e. pure stride metadata) are torch tensors; every arithmetic
operation is a launch of a hand-written HIP kernel through the C ABI:

* ``fuse(lambda a, b: ..., A, B)`` -- one fused elementwise pass (``vmp_ewise``)
* ``sum_multiply`` (bayespy_amd.utils.misc) -- ``vmp_sum_multiply``
* ``linalg.*`` (bayespy_amd.utils.linalg) -- ``vmp_spd_batched``

No arithmetic is delegated to torch or NumPy.
"""
import ctypes
import numbers

import numpy as np

from .device import get_runtime

# opcodes of include/vmp_hip.h
(OP_IN, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_LOG, OP_EXP, OP_SQR, OP_SQRT,
 OP_RECIP, OP_DIGAMMA, OP_LGAMMA, OP_MAX, OP_MIN, OP_WHERE_NZ, OP_DUP, OP_SWAP,
 OP_TRIGAMMA) = range(20)
MAX_OPS, MAX_CONSTS, MAX_IN, MAX_DIMS = 48, 8, 6, 8


class DArray:
    """fp64 device array (possibly a broadcast / strided view)."""

    __slots__ = ('t', '__weakref__')
    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t

    # -- construction -------------------------------------------------------------
    @staticmethod
    def from_host(a):
        rt = get_runtime()
        return DArray(rt.to_device(np.asarray(a, dtype=np.float64)))

    @staticmethod
    def zeros(shape):
        return DArray(get_runtime().zeros(*tuple(shape)) if len(shape) else
                      get_runtime().zeros(()))

    @staticmethod
    def empty(shape):
        rt = get_runtime()
        return DArray(rt.torch.empty(tuple(shape), dtype=rt.torch.float64, device=rt.device))

    # -- metadata ----------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def size(self):
        return int(self.t.numel())

    def numpy(self):
        t = self.t          # (a lazily evaluated array launches here: before the flush below)
        get_runtime().host_access('DArray.numpy')
        return t.detach().cpu().numpy().copy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def item(self):
        t = self.t
        get_runtime().host_access('DArray.item')
        return float(t.reshape(-1)[0].item())

    # -- views (stride metadata only) ---------------------------------------------------
    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        try:
            return DArray(self.t.view(shape))
        except RuntimeError:
            return DArray(contiguous(self).t.view(shape))

    def broadcast_to(self, shape):
        return DArray(self.t.expand(tuple(shape)))

    def __getitem__(self, idx):
        return DArray(self.t[idx])

    def swapaxes(self, a, b):
        return DArray(self.t.transpose(a, b))

    # -- eager operators (one kernel each; use fuse() for formulas) ------------------------
    def __add__(self, o): return fuse(lambda a, b: a + b, self, o)
    def __radd__(self, o): return fuse(lambda a, b: b + a, self, o)
    def __sub__(self, o): return fuse(lambda a, b: a - b, self, o)
    def __rsub__(self, o): return fuse(lambda a, b: b - a, self, o)
    def __mul__(self, o): return fuse(lambda a, b: a * b, self, o)
    def __rmul__(self, o): return fuse(lambda a, b: b * a, self, o)
    def __truediv__(self, o): return fuse(lambda a, b: a / b, self, o)
    def __rtruediv__(self, o): return fuse(lambda a, b: b / a, self, o)
    def __neg__(self): return fuse(lambda a: -a, self)

    def __repr__(self):
        return 'DArray(shape=%s)' % (self.shape,)


def asdarray(x):
    if isinstance(x, DArray):
        return x
    return DArray.from_host(x)


def is_scalar(x):
    return isinstance(x, numbers.Number) or (isinstance(x, np.ndarray) and x.ndim == 0
                                             and not isinstance(x, DArray))


# ---------------------------------------------------------------------------
# expression tracing -> postfix program
# ---------------------------------------------------------------------------
class Expr:
    __slots__ = ('op', 'args', 'val')

    def __init__(self, op, args=(), val=None):
        self.op, self.args, self.val = op, args, val

    @staticmethod
    def wrap(x):
        if isinstance(x, Expr):
            return x
        if isinstance(x, numbers.Number) or (isinstance(x, np.ndarray) and x.ndim == 0):
            return Expr('const', val=float(x))
        raise TypeError('cannot use %r inside a fused expression' % (type(x),))

    def _bin(self, op, o, swap=False):
        o = Expr.wrap(o)
        return Expr(op, (o, self) if swap else (self, o))

    def __add__(self, o): return self._bin(OP_ADD, o)
    def __radd__(self, o): return self._bin(OP_ADD, o, True)
    def __sub__(self, o): return self._bin(OP_SUB, o)
    def __rsub__(self, o): return self._bin(OP_SUB, o, True)
    def __mul__(self, o): return self._bin(OP_MUL, o)
    def __rmul__(self, o): return self._bin(OP_MUL, o, True)
    def __truediv__(self, o): return self._bin(OP_DIV, o)
    def __rtruediv__(self, o): return self._bin(OP_DIV, o, True)
    def __neg__(self): return Expr(OP_NEG, (self,))
    def __pow__(self, p):
        if p == 2:
            return Expr(OP_SQR, (self,))
        raise NotImplementedError('only **2 is supported in fused expressions')


def _un(op):
    return lambda x: Expr(op, (Expr.wrap(x),))


log, exp, sqrt, square = _un(OP_LOG), _un(OP_EXP), _un(OP_SQRT), _un(OP_SQR)
digamma, gammaln, recip = _un(OP_DIGAMMA), _un(OP_LGAMMA), _un(OP_RECIP)
trigamma = _un(OP_TRIGAMMA)


def maximum(a, b): return Expr(OP_MAX, (Expr.wrap(a), Expr.wrap(b)))
def minimum(a, b): return Expr(OP_MIN, (Expr.wrap(a), Expr.wrap(b)))


def where_nonzero(u, v):
    """v where u != 0 else 0 -- the ``0 * -inf`` guard of expfamily.py:463-464."""
    return Expr(OP_WHERE_NZ, (Expr.wrap(u), Expr.wrap(v)))


def _depth(e):
    if e.op in ('in', 'const'):
        return 1
    ds = [_depth(a) for a in e.args]
    if len(ds) == 1:
        return ds[0]
    # evaluate the deeper operand first (Sethi-Ullman)
    return max(max(ds), min(ds) + 1)


def _emit(e, ops, consts):
    if e.op == 'in':
        ops.append(OP_IN | (e.val << 8))
    elif e.op == 'const':
        if e.val in consts:
            i = consts.index(e.val)
        else:
            consts.append(e.val)
            i = len(consts) - 1
        ops.append(OP_CONST | (i << 8))
    elif len(e.args) == 1:
        _emit(e.args[0], ops, consts)
        ops.append(e.op)
    else:
        a, b = e.args
        if _depth(b) > _depth(a):
            _emit(b, ops, consts)
            _emit(a, ops, consts)
            ops.append(OP_SWAP)
        else:
            _emit(a, ops, consts)
            _emit(b, ops, consts)
        ops.append(e.op)


def _strides(t, shape):
    """Element strides of tensor t right-aligned into `shape` (0 on broadcast axes)."""
    nd = len(shape)
    st = [0] * nd
    off = nd - t.dim()
    for d in range(t.dim()):
        st[off + d] = 0 if t.shape[d] == 1 else t.stride(d)
    return st


_PROGRAMS = {}


def _program_key(fn, kinds):
    """Cache key of a traced formula: its code, default arguments, the numbers it closes over
    and which operands are arrays / which constant values were passed; None = do not cache."""
    try:
        cells = tuple(c.cell_contents for c in (fn.__closure__ or ()))
        if not all(isinstance(v, numbers.Number) for v in cells):
            return None
        kw = getattr(fn, '__kwdefaults__', None)
        key = (fn.__code__, fn.__defaults__, None if not kw else tuple(sorted(kw.items())), cells,
               kinds)
        hash(key)
        return key
    except (AttributeError, TypeError, ValueError):
        return None


def _compile(fn, leaves):
    expr = Expr.wrap(fn(*leaves))
    if _depth(expr) > 4:
        raise ValueError('fused expression needs a stack deeper than 4; split it')
    ops, consts = [], []
    _emit(expr, ops, consts)
    return ((ctypes.c_int32 * len(ops))(*ops), len(ops),
            (ctypes.c_double * max(len(consts), 1))(*consts), len(consts))


def fuse(fn, *operands):
    """
    Evaluate the elementwise formula ``fn(*operands)`` in ONE kernel launch.
    Operands are DArrays, host ndarrays (uploaded) or Python scalars (constants);
    the result has the broadcast shape of the array operands.
    """
    from .utils.shapes import broadcasted_shape
    rt = get_runtime()
    arrays, leaves, kinds = [], [], []
    for x in operands:
        if isinstance(x, DArray):
            leaves.append(Expr('in', val=len(arrays)))
            arrays.append(x)
            kinds.append(None)
        elif is_scalar(x):
            leaves.append(Expr('const', val=float(x)))
            kinds.append(float(x))
        else:
            leaves.append(Expr('in', val=len(arrays)))
            arrays.append(DArray.from_host(x))
            kinds.append(None)
    key = _program_key(fn, tuple(kinds))
    prog = _PROGRAMS.get(key) if key is not None else None
    if prog is None:
        prog = _compile(fn, leaves)
        if key is not None and len(_PROGRAMS) < 4096:
            _PROGRAMS[key] = prog
    c_ops, nops, c_consts, nconsts = prog
    if nops > MAX_OPS or nconsts > MAX_CONSTS or len(arrays) > MAX_IN:
        raise ValueError('fused expression too large for one launch')
    shape = broadcasted_shape(*[a.shape for a in arrays]) if arrays else ()
    if len(shape) > MAX_DIMS:
        raise NotImplementedError('more than %d axes' % MAX_DIMS)
    out = DArray.empty(shape)
    nd, nin = len(shape), len(arrays)
    c_shape = (ctypes.c_int64 * max(nd, 1))(*shape)
    c_in = (ctypes.c_void_p * max(nin, 1))(*[a.t.data_ptr() for a in arrays])
    flat = []
    for a in arrays:
        flat += _strides(a.t, shape)
    c_str = (ctypes.c_int64 * max(len(flat), 1))(*flat)
    rt.sync_stream()
    rt.note_reads(arrays)
    rt.check(rt.lib.vmp_ewise(rt.ctx, nd, c_shape, nin, c_in, c_str, nops, c_ops,
                              nconsts, c_consts, ctypes.c_void_p(out.t.data_ptr())))
    rt.keep_until_flush(arrays, out)
    return out


def contiguous(a):
    """A dense copy of a (possibly broadcast / strided) array."""
    if a.t.is_contiguous():
        return a
    return fuse(lambda x: x + 0.0, a)


def full_like_shape(shape, value):
    return fuse(lambda z: z + float(value), DArray.zeros(shape)) if value != 0.0 \
        else DArray.zeros(shape)
