"""
ctypes binding of ``libvmp_hip.so`` (the C ABI declared in include/vmp_hip.h).

There is NO CPU fallback: if the shared library is missing or does not export
a declared symbol, importing this module's :func:`load` raises.  ``import torch``
must precede the ``CDLL`` call so that the HIP runtime already mapped by torch
(``libamdhip64.so.7``) is the one the library binds to.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libvmp_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'vmp_hip.h')

VMP_OK = 0
VMP_ERR_INVALID = -1
VMP_ERR_NOT_POSDEF = -2
VMP_ERR_HIP = -3
VMP_ERR_UNSUPPORTED = -4
VMP_ERR_FLOATING = -5
VMP_ERR_NOT_POSITIVE = -6


class NotPositiveDefiniteError(Exception):
    """The reference raises a bare ``Exception("Matrix not positive definite")``
    (bayespy/utils/linalg.py:58-59)."""


def raise_for_status(rc, msg=''):
    """Map a C-ABI status to the exception type the reference raises."""
    if rc == VMP_OK:
        return
    if rc == VMP_ERR_INVALID:
        raise ValueError(msg or 'invalid argument')
    if rc == VMP_ERR_NOT_POSDEF:
        raise NotPositiveDefiniteError(msg or 'Matrix not positive definite')
    if rc == VMP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg or 'unsupported by the built HIP kernels')
    if rc == VMP_ERR_FLOATING:
        raise FloatingPointError(msg or 'invalid value encountered')
    if rc == VMP_ERR_NOT_POSITIVE:
        raise ValueError(msg or 'Natural parameters should be positive')
    raise RuntimeError(msg or 'HIP runtime error (status %d)' % rc)


class PCALayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'DP', 'KP', 'off_S', 'len_S', 'off_Syy', 'off_tau', 'off_alpha', 'off_W',
        'off_CW', 'off_Sww', 'off_CX', 'off_A', 'off_G', 'off_scal', 'off_L', 'total',
        'off_mu', 'off_mstat')]


class MPCALayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'DP', 'KP', 'P', 'PT', 'LR', 'off_tau', 'off_alpha', 'off_scal', 'off_L', 'off_W',
        'off_WW', 'off_ldW', 'off_M', 'off_panel', 'off_panel_x', 'off_Sxx', 'off_rowobs',
        'total')]


class MPCASizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'ymt_doubles', 'mask_words', 'xm_doubles', 'lam_doubles', 'xxf_doubles',
        'workspace_doubles')]


class LSSMLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'off_tau', 'off_gamma', 'off_alpha', 'off_nu', 'off_mu0', 'off_Lam0', 'off_ldLam0',
        'off_Cm', 'off_CovC', 'off_SCC', 'off_Am', 'off_AA', 'off_ldA', 'off_Dg', 'off_h0',
        'off_covsums', 'off_raw', 'len_raw', 'off_S', 'off_scal', 'off_L', 'total')]


class LSSMMLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'NS', 'off_tau', 'off_gamma', 'off_alpha', 'off_nu', 'off_mu0', 'off_Lam0', 'off_ldLam0',
        'off_Cm', 'off_CovC', 'off_ldC', 'off_SCC', 'off_Am', 'off_AA', 'off_ldA', 'off_tab',
        'len_tab', 'off_setup', 'len_setup', 'off_raw', 'len_raw', 'off_scal', 'off_L', 'total')]


class GMMLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        'DP', 'KP', 'FS', 'FP', 'F2P', 'off_T', 'len_T', 'off_zs', 'off_alpha', 'off_mu',
        'off_Cmu', 'off_logdetLmu', 'off_nk', 'off_Vk', 'off_Lam', 'off_logdetLam',
        'off_logdetV', 'off_C', 'off_prior', 'off_scal', 'off_L', 'total')]


c_i32, c_i64, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
c_vp, c_sz = ctypes.c_void_p, ctypes.c_size_t
P = ctypes.POINTER

# name -> (restype, argtypes); must list every symbol of include/vmp_hip.h
SIGNATURES = {
    'vmp_ctx_create': (c_i32, [c_i32, c_vp, P(c_vp)]),
    'vmp_ctx_destroy': (c_i32, [c_vp]),
    'vmp_ctx_set_stream': (c_i32, [c_vp, c_vp]),
    'vmp_ctx_sync': (c_i32, [c_vp]),
    'vmp_ctx_num_cu': (c_i32, [c_vp]),
    'vmp_last_error': (ctypes.c_char_p, [c_vp]),
    'vmp_version': (ctypes.c_char_p, []),
    'vmp_malloc': (c_i32, [c_vp, c_sz, P(c_vp)]),
    'vmp_free': (c_i32, [c_vp, c_vp]),
    'vmp_memcpy_h2d': (c_i32, [c_vp, c_vp, c_vp, c_sz]),
    'vmp_memcpy_d2h': (c_i32, [c_vp, c_vp, c_vp, c_sz]),
    'vmp_memset_zero': (c_i32, [c_vp, c_vp, c_sz]),
    'vmp_comm_unique_id': (c_i32, [c_vp, c_vp]),
    'vmp_comm_init_rank': (c_i32, [c_vp, c_vp, c_i32, c_i32]),
    'vmp_comm_destroy': (c_i32, [c_vp]),
    'vmp_comm_info': (c_i32, [c_vp, P(c_i32), P(c_i32)]),
    'vmp_allreduce_sum_f64': (c_i32, [c_vp, c_vp, c_i64]),
    'vmp_pca_get_layout': (c_i32, [c_i32, c_i32, P(PCALayout)]),
    'vmp_pca_workspace_bytes': (c_i32, [c_vp, c_i32, c_i32, P(c_sz)]),
    'vmp_pca_init_state': (c_i32, [c_vp, c_i32, c_i32, c_f64, c_f64, c_f64, c_f64, c_vp]),
    'vmp_pca_syy': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp]),
    'vmp_pca_stats_from_x': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64,
                                     c_vp, c_vp]),
    'vmp_pca_update_w': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp]),
    'vmp_pca_prepare_x': (c_i32, [c_vp, c_i32, c_i32, c_f64, c_vp]),
    'vmp_pca_pass': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'vmp_pca_xpass': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    'vmp_pca_tiled_doubles': (c_i32, [c_i32, c_i32, c_i64, P(c_i64), P(c_i64)]),
    'vmp_pca_tile_y': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    'vmp_pca_tile_x': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    'vmp_pca_xpass_tiled': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp,
                                    c_vp]),
    'vmp_tune_set': (c_i32, [ctypes.c_char_p, c_i32]),
    'vmp_graph_begin': (c_i32, [c_vp]),
    'vmp_graph_end': (c_i32, [c_vp, P(c_vp)]),
    'vmp_graph_launch': (c_i32, [c_vp, c_vp]),
    'vmp_graph_destroy': (c_i32, [c_vp, c_vp]),
    'vmp_copy_many': (c_i32, [c_vp, c_i32, P(c_vp), P(c_vp), P(c_i64)]),
    'vmp_pack_outputs': (c_i32, [c_vp, c_i32, P(c_vp), P(c_i64), P(c_i32), c_vp]),
    'vmp_queue_begin': (c_i32, [c_vp]),
    'vmp_queue_flush': (c_i32, [c_vp]),
    'vmp_queue_end': (c_i32, [c_vp]),
    'vmp_queue_commit': (c_i32, [c_vp]),
    'vmp_queue_stats': (c_i32, [c_vp, P(c_i64), P(c_i64)]),
    'vmp_pca_gram': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp]),
    'vmp_pca_update_tau': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_vp]),
    'vmp_pca_update_alpha': (c_i32, [c_vp, c_i32, c_i32, c_f64, c_f64, c_vp]),
    'vmp_pca_lower_bound': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64,
                                    c_f64, c_vp]),
    'vmp_mpca_get_layout': (c_i32, [c_i32, c_i32, P(MPCALayout)]),
    'vmp_mpca_sizes': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, P(MPCASizes)]),
    'vmp_mpca_init_state': (c_i32, [c_vp, c_i32, c_i32, c_f64, c_f64, c_f64, c_f64, c_vp]),
    'vmp_mpca_prepare': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp,
                                 c_vp, c_vp, c_vp]),
    'vmp_mpca_x_begin': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp]),
    'vmp_mpca_x_chunk': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_i32, c_f64, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_mpca_x_pass': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_i32, c_i32, c_f64, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_mpca_update_w': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    'vmp_mpca_small_ops': (c_i32, [c_vp, c_i32, c_i32, c_f64, c_f64, c_f64, c_f64, c_f64, c_i32,
                                   P(c_i32), c_vp]),
    'vmp_mpca_unpack_xx': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp, c_vp]),
    'vmp_lssm_limits': (c_i32, [P(c_i32), P(c_i32)]),
    'vmp_lssm_get_layout': (c_i32, [c_i32, c_i32, P(LSSMLayout)]),
    'vmp_lssm_workspace_doubles': (c_i32, [c_i32, c_i32, c_i64, c_i32, P(c_i64)]),
    'vmp_lssm_relayout_y': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp]),
    'vmp_lssm_x_layout': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_i32]),
    'vmp_lssm_cov': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_lssm_smooth': (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_lssm_rotate_x': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp]),
    'vmp_lssm_x_update': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_lssm_small_ops': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f64, P(c_f64), c_i32, c_i32,
                                   P(c_i32), c_vp]),
    'vmp_lssmm_limits': (c_i32, [P(c_i32), P(c_i32)]),
    'vmp_lssmm_get_layout': (c_i32, [c_i32, c_i32, P(LSSMMLayout)]),
    'vmp_lssmm_workspace_doubles': (c_i32, [c_i32, c_i32, c_i64, c_i32, P(c_i64)]),
    'vmp_lssmm_prepare': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i64, c_i32,
                                  c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_lssmm_x_update': (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i64,
                                   c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_lssmm_rotate_p': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp]),
    'vmp_lssmm_small_ops': (c_i32, [c_vp, c_i32, c_i32, c_i32, P(c_f64), c_i32, c_i32, P(c_i32),
                                    c_vp]),
    'vmp_gmm_get_layout': (c_i32, [c_i32, c_i32, P(GMMLayout)]),
    'vmp_gmm_workspace_bytes': (c_i32, [c_vp, c_i32, c_i32, P(c_sz)]),
    'vmp_gmm_init_state': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_f64, c_f64, c_vp, c_vp]),
    'vmp_gmm_stats_from_labels': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp,
                                          c_vp]),
    'vmp_gmm_update_mu': (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    'vmp_gmm_update_lambda': (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    'vmp_gmm_prepare_z': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    'vmp_gmm_pass': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'vmp_gmm_update_alpha': (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    'vmp_gmm_lower_bound': (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    'vmp_sum_multiply': (c_i32, [c_vp, c_i32, P(c_i64), c_i32, P(c_vp), P(c_i64), P(c_i64),
                                 ctypes.c_uint32, c_f64, c_vp, c_vp, c_sz]),
    'vmp_sum_multiply_workspace_bytes': (c_sz, []),
    'vmp_ewise': (c_i32, [c_vp, c_i32, P(c_i64), c_i32, P(c_vp), P(c_i64), c_i32, P(c_i32),
                          c_i32, P(c_f64), c_vp]),
    'vmp_spd_batched': (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'vmp_gaussian_moments': (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'vmp_gaussian_shared_update_workspace_bytes': (c_sz, [c_i32, c_i32]),
    'vmp_gaussian_shared_update': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp,
                                           c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp,
                                           c_i64, c_i64, c_vp, c_vp, c_sz]),
    'vmp_softmax_moments': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    'vmp_onehot_i64': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    'vmp_alpha_beta_recursion': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64,
                                         c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t]),
    'vmp_take_axis': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64]),
    'vmp_segment_sum_axis': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'vmp_gemm_strided': (c_i32, [c_vp, c_i32, P(c_i64), c_i64, c_i64, c_i64, c_vp, P(c_i64), c_i64,
                                 c_i64, c_vp, P(c_i64), c_i64, c_i64, c_vp, P(c_i64), c_i64, c_i64,
                                 c_f64, c_vp, c_sz]),
    'vmp_block_banded_solve': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp, c_vp, c_vp]),
    'vmp_ctx_set_timing': (c_i32, [c_vp, c_i32]),
    'vmp_pca_xjoin': (c_i32, [c_vp]),
    'vmp_pca_ensure_gram': (c_i32, [c_vp]),
    'vmp_pca_small_ops': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64, c_f64,
                                  c_i32, P(c_i32), c_vp]),
    'vmp_pca_small_ops_mean': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64, c_f64,
                                       c_i32, P(c_i32), c_i32, c_vp]),
    'vmp_pass_times_ms': (c_i32, [c_vp, P(c_f64), P(c_f64), c_i32, P(c_i32)]),
    'vmp_pca_last_pass_ms': (c_i32, [c_vp, P(c_f64), P(c_f64)]),
}

_lib = None


def bind_lssmm(lib):
    """Attach the vmp_lssmm_* signatures to another library that exports that part of the C ABI
    (the host build of the same device code: tests/host_build.py)."""
    for name, (restype, argtypes) in SIGNATURES.items():
        if name.startswith('vmp_lssmm_'):
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    return lib


def header_symbols():
    """All function names declared in include/vmp_hip.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vmp_[a-z0-9_]+)\s*\(', src)))


def load():
    """Load libvmp_hip.so (once) and attach the signatures.  Raises if the
    library has not been built -- the product path has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libvmp_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C bayespy_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    import torch  # noqa: F401  -- maps libamdhip64.so.7 first (see module docstring)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
