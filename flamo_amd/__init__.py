"""flamo_amd -- MI355X-native (gfx950) implementation of the frequency-sampling hot path of
gdalsanto/flamo: batched rFFT/irFFT, the per-bin complex MIMO product and the Recursion
closed-loop solve, behind flamo's own ``processor.dsp`` / ``processor.system`` operator API.

    from flamo_amd.processor import dsp, system     # instead of: from flamo.processor import dsp, system

Tensors must live on a ROCm device; the hand-written HIP library (flamo_amd/libflamo_hip.so,
C ABI in include/flamo_hip.h) is required -- there is no CPU or eager fallback.
"""
import os as _os

# ROCm 7.x replays a captured HIP graph from pre-built AQL packets ("graph packet capture").  With that switch on, a captured
# torch reduction (`tensor.sum()` / `.max()`: a memset node + a kernel node) that follows MB-sized temporaries allocated INSIDE
# the capture returns different -- deterministic, wrong -- values after ANY tiny eager launch between two replays.  It is a
# hazard of the platform, not of this library: tools/dbg/replay_min.py reproduces it with torch kernels alone ("torch only,
# MB-sized temporaries in the capture": 6 of 6 replays differ), while this library's kernels on preallocated buffers, and
# trivial kernels with large by-value arguments or 100 KB of dynamic LDS, replay bit-identically; the library's operators hit
# it only because they allocate their scratch arrays from the graph's pool like any torch op (padding those allocations moves
# it away).  Every tensor this library's kernels write, and every gradient, stayed bit-identical over thousands of replays
# either way (tools/dbg/soak_fdn2.py).  Replays are not measurably slower without the pre-built packets (0.541 vs 0.545 ms at
# config 2, 0.3725 vs 0.3709 ms at config 3), so the switch is turned off unless the user has set it; it has to be in the
# environment before the HIP runtime initialises, i.e. before the first CUDA call of the process (importing this package
# before touching the GPU is enough).
_user_setting = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def _graph_packets_off() -> bool:
    """True when the switch above can be relied on: set by the user, or set here before the HIP runtime came up"""
    return _user_setting == "0" or (_user_setting is None and not _RUNTIME_WAS_UP)


def _hip_runtime_loaded_and_used() -> bool:
    """Has this process already initialised the HIP runtime?  torch.cuda.is_initialized() misses torch.cuda.is_available() /
    device_count(), which call hipGetDeviceCount -- that brings the runtime up (and parses the DEBUG_CLR_* switches) without
    setting torch's lazy-init flag.  The runtime opens the compute driver's device node when it comes up and never closes
    it: an open /dev/kfd among this process's descriptors is the tell-tale."""
    try:
        import torch
        if torch.cuda.is_initialized():
            return True
    except Exception:       # pragma: no cover
        pass
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:         # pragma: no cover
        pass
    return False


_RUNTIME_WAS_UP = _hip_runtime_loaded_and_used()
if _RUNTIME_WAS_UP and _user_setting is None:
    import warnings as _warnings
    _warnings.warn("flamo_amd was imported after the HIP runtime came up (a torch.cuda call ran first): "
                   "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 can no longer take effect in this process, and replayed HIP graphs that "
                   "mix this library's kernels with torch reductions may return wrong reduction values after eager launches "
                   "between replays.  Import flamo_amd before the first torch.cuda call, or export the variable.",
                   RuntimeWarning, stacklevel=2)

from . import _lib, functional, ops, utils  # noqa: F401
from .processor import dsp, system  # noqa: F401

__version__ = "0.1.0"


def build(force: bool = False) -> str:
    """Compile the HIP kernels for gfx950 into flamo_amd/libflamo_hip.so."""
    return _lib.build(force)
