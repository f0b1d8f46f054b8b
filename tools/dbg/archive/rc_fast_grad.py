"""Gradient of Series(Matrix, <cascade filter>) with the float / double forward evaluation of the cascade (the backward
reuses the saved forward response): relative difference of parameter gradients, and both against a float64 module."""
import os, sys, warnings
from collections import OrderedDict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
from flamo_amd.processor import dsp, system
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
nfft, N, B = 96000, 8, 4
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
def build(kind, dt):
    torch.manual_seed(1)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=dev, dtype=dt, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw)
    if kind == "geq": f = dsp.GEQ(size=(N, N), **kw)
    elif kind == "peq": f = dsp.PEQ(size=(N, N), n_bands=6, **kw)
    elif kind == "svf": f = dsp.SVF(size=(N, N), n_sections=3, **kw)
    else: f = dsp.Biquad(size=(N, N), n_sections=2, filter_type=kind, **kw)
    m = system.Shell(system.Series(OrderedDict(mix=mat, eq=f)), dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
    return m, [mat.param, f.param]
for kind in ("geq", "peq", "svf", "lowpass", "bandpass"):
    torch.manual_seed(0)
    x = torch.randn(B, nfft, N, device=dev)
    m64, p64 = build(kind, torch.float64)
    (m64(x.double()) ** 2).mean().backward()
    res = {}
    for fast in (1, 0):
        _lib.lib().fl_debug_set_rc_fast(fast)
        m, p = build(kind, torch.float32)
        y = m(x)
        (y ** 2).mean().backward()
        res[fast] = [q.grad.clone() for q in p]
    _lib.lib().fl_debug_set_rc_fast(1)
    print(kind, "float-vs-double forward:", [f"{rel(a, b):.1e}" for a, b in zip(res[1], res[0])],
          " float fwd vs f64:", [f"{rel(a, b.grad):.1e}" for a, b in zip(res[1], p64)],
          " double fwd vs f64:", [f"{rel(a, b.grad):.1e}" for a, b in zip(res[0], p64)])
