from . import dsp, system  # noqa: F401
