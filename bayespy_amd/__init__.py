"""
bayespy_amd -- MI355X-native variational message passing behind the BayesPy
node-construction / ``VB.update()`` API.

    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    from bayespy_amd.inference import VB

Only the hot path is built; every number is produced by the
hand-written HIP kernels in ``bayespy_amd/csrc`` through the C ABI of
``include/vmp_hip.h``.  There is no CPU fallback.
"""
from . import nodes      # noqa: F401
from . import inference  # noqa: F401

__version__ = '0.1.0'
