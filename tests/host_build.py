"""TEST INFRASTRUCTURE: build the host-compiled doubles of device code (tests/host/*.cpp) with g++.

``lssmm_host()`` -> ctypes library exporting the vmp_lssmm_* C ABI, compiled from the very header
the HIP kernels include (bayespy_amd/csrc/vmp_lssmm_dev.h) plus the host+device special
functions sliced out of vmp_common.h.  Cached per source hash under the system temp directory."""
import ctypes
import hashlib
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'bayespy_amd', 'csrc')


def _special_functions_text():
    text = open(os.path.join(CSRC, 'vmp_common.h')).read()
    s = text.index('__host__ __device__ inline double vmp_digamma')
    e = text.index('#ifdef __HIPCC__')
    return '#include <math.h>\n#define __host__\n#define __device__\n' + text[s:e] + \
        '\n#undef __host__\n#undef __device__\n'


def lssmm_host():
    srcs = [os.path.join(ROOT, 'tests', 'host', 'lssmm_host.cpp'),
            os.path.join(CSRC, 'vmp_lssmm_dev.h'), os.path.join(ROOT, 'include', 'vmp_hip.h')]
    sf = _special_functions_text()
    h = hashlib.sha256(sf.encode())
    for p in srcs:
        h.update(open(p, 'rb').read())
    d = os.path.join(tempfile.gettempdir(), 'bayespy_amd_host_%s' % h.hexdigest()[:16])
    so = os.path.join(d, 'liblssmm_host.so')
    if not os.path.exists(so):
        os.makedirs(d, exist_ok=True)
        # per-process names + atomic renames: several pytest workers may build at once
        sfh = os.path.join(d, 'sf.%d.h' % os.getpid())
        with open(sfh, 'w') as f:
            f.write(sf)
        tmp = so + '.%d.tmp' % os.getpid()
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off',
                               '-pthread', '-include', sfh, srcs[0], '-o', tmp])
        os.replace(tmp, so)
    return ctypes.CDLL(so)
