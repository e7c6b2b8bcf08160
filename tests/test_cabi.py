"""CPU: the C-ABI library loads and exports every symbol include/vmp_hip.h declares;
host-only entry points behave (no compute calls without a GPU)."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest
from scipy import special

from bayespy_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libvmp_hip.so does not export %s' % name
    # and the Python binding declares a signature for each of them
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_layout_host_functions():
    lib = _lib.load()
    assert b'gfx950' in lib.vmp_version()
    L = _lib.PCALayout()
    assert lib.vmp_pca_get_layout(128, 32, ctypes.byref(L)) == _lib.VMP_OK
    assert (L.DP, L.KP) == (128, 32)
    assert L.len_S == (128 + 32) * 32
    assert L.total >= L.off_L + 8
    assert lib.vmp_pca_get_layout(6, 3, ctypes.byref(L)) == _lib.VMP_OK
    assert (L.DP, L.KP) == (32, 16)
    assert lib.vmp_pca_get_layout(70, 33, ctypes.byref(L)) == _lib.VMP_OK
    assert (L.DP, L.KP) == (128, 64)
    assert lib.vmp_pca_get_layout(0, 3, ctypes.byref(L)) == _lib.VMP_ERR_INVALID
    assert lib.vmp_pca_get_layout(300, 3, ctypes.byref(L)) == _lib.VMP_ERR_UNSUPPORTED


def test_status_to_exception_mapping():
    with pytest.raises(ValueError):
        _lib.raise_for_status(_lib.VMP_ERR_INVALID)
    with pytest.raises(_lib.NotPositiveDefiniteError):
        _lib.raise_for_status(_lib.VMP_ERR_NOT_POSDEF)
    with pytest.raises(NotImplementedError):
        _lib.raise_for_status(_lib.VMP_ERR_UNSUPPORTED)
    with pytest.raises(FloatingPointError):
        _lib.raise_for_status(_lib.VMP_ERR_FLOATING)
    with pytest.raises(RuntimeError):
        _lib.raise_for_status(_lib.VMP_ERR_HIP)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from bayespy_amd.device import Runtime
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Runtime()


def test_device_special_functions_match_scipy():
    """vmp_digamma / vmp_lgamma / vmp_trigamma (csrc/vmp_common.h) compiled for the host."""
    src = os.path.join(ROOT, 'bayespy_amd', 'csrc', 'vmp_common.h')
    text = open(src).read()
    s = text.index('__host__ __device__ inline double vmp_digamma')
    e = text.index('#ifdef __HIPCC__')
    with tempfile.TemporaryDirectory() as d:
        cpp = os.path.join(d, 'sf.cpp')
        with open(cpp, 'w') as f:
            f.write('#include <math.h>\n#define __host__\n#define __device__\n')
            f.write(text[s:e])
            f.write('extern "C" void dg(const double*x,double*y,int n)'
                    '{for(int i=0;i<n;i++)y[i]=vmp_digamma(x[i]);}\n'
                    'extern "C" void lg(const double*x,double*y,int n)'
                    '{for(int i=0;i<n;i++)y[i]=vmp_lgamma(x[i]);}\n'
                    'extern "C" void tg(const double*x,double*y,int n)'
                    '{for(int i=0;i<n;i++)y[i]=vmp_trigamma(x[i]);}\n')
        so = os.path.join(d, 'sf.so')
        subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', cpp, '-o', so])
        lib = ctypes.CDLL(so)
        x = np.concatenate([np.logspace(-6, 8, 4001), np.linspace(0.01, 40, 4001),
                            0.01 + 0.5 * np.arange(1, 200)])
        y = np.empty_like(x)
        for fn, ref in ((lib.dg, special.digamma), (lib.lg, special.gammaln),
                        (lib.tg, lambda v: special.polygamma(1, v))):
            fn(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), len(x))
            r = ref(x)
            err = np.minimum(np.abs(y - r), np.abs(y - r) / np.maximum(np.abs(r), 1e-300))
            assert err.max() < 2e-14
