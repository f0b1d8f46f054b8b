"""flamo_amd -- MI355X-native (gfx950) implementation of the frequency-sampling hot path of
gdalsanto/flamo: batched rFFT/irFFT, the per-bin complex MIMO product and the Recursion
closed-loop solve, behind flamo's own ``processor.dsp`` / ``processor.system`` operator API.

    from flamo_amd.processor import dsp, system     # instead of: from flamo.processor import dsp, system

Tensors must live on a ROCm device; the hand-written HIP library (flamo_amd/libflamo_hip.so,
C ABI in include/flamo_hip.h) is required -- there is no CPU or eager fallback.
"""
from . import _lib, functional, ops, utils  # noqa: F401
from .processor import dsp, system  # noqa: F401

__version__ = "0.1.0"


def build(force: bool = False) -> str:
    """Compile the HIP kernels for gfx950 into flamo_amd/libflamo_hip.so."""
    return _lib.build(force)
