"""Cascade-times-matrix forward (fl_sos_response_rc_c64, float evaluation): section pairs per loop trip (tuning)."""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.processor import dsp, system  # noqa: E402

dev = torch.device("cuda:0")
nfft, N, B = 96000, 8, 32
torch.manual_seed(0)
kw = dict(nfft=nfft, alias_decay_db=0.0, device=dev, dtype=torch.float32)
mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft), dsp.iFFT(nfft))
x = torch.randn(B, nfft, N, device=dev)
ref = None
for mode in (1, 2, 3, 6, 1):
    _lib.lib().fl_debug_set_rc_fast(mode)
    for _ in range(3):
        ops.mean_square(model(x)).backward()
    torch.cuda.synchronize()
    ops.kernel_timer.reset(True, 200_000)
    for _ in range(8):
        y = model(x)
        ops.mean_square(y).backward()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    sm = ops.kernel_timer.summary()
    if ref is None:
        ref = y.detach().clone()
    print(f"mode {mode}: ", {k: round(v[1] * 1e3, 1) for k, v in sm.items() if "sos" in k or "geq" in k}, f"  out diff {(y.detach() - ref).norm() / ref.norm():.1e}")
_lib.lib().fl_debug_set_rc_fast(1)
