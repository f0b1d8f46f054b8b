"""Import the reference (gdalsanto/flamo) from /root/reference in THIS container only.

Used by tools/gen_golden.py (fixture generation) and, when the reference checkout is
present, by tests that cross-check the oracle directly.  Never used on the GPU box: the
reference does not travel.  Five packages the reference imports at module scope are absent
here and are off the hot path (soundfile: utils.save_audio; nnAudio/pyfar: mel/EDR losses;
torchaudio: filterbank; auraloss: an example) -- empty stand-in modules let
``flamo.processor`` import unmodified.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FLAMO_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "flamo", "processor"))


def load():
    """Returns (dsp, system) modules of the reference."""
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    for name in ["soundfile", "nnAudio", "nnAudio.features", "pyfar",
                 "torchaudio", "torchaudio.functional", "auraloss"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["nnAudio"].features = sys.modules["nnAudio.features"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from flamo.processor import dsp, system  # noqa: E402

    return dsp, system
