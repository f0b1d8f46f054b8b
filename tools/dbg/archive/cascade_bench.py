"""Cascade-times-matrix kernels of the config-2 step (fl_sos_response_rc_c64 / fl_sos_response_bwd_rc_c64): launch times per
bin-block count, and the parameter gradients against the float64 module.
    python tools/dbg/cascade_bench.py [--blocks 0,12,24,32]"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.processor import dsp, system  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", default="0")
ap.add_argument("--nfft", type=int, default=96000)
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
nfft, N, B = args.nfft, args.n, args.batch


def build(dt):
    torch.manual_seed(0)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=dev, dtype=dt)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    return system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt)), mat, geq


model, mat, geq = build(torch.float32)
m64, mat64, geq64 = build(torch.float64)
with torch.no_grad():
    mat64.param.copy_(mat.param.double())
    geq64.param.copy_(geq.param.double())
torch.manual_seed(1)
x = torch.randn(B, nfft, N, device=dev)
xs = x[:2].double()
ops.mean_square(m64(xs)).backward()
ref = [mat64.param.grad.clone(), geq64.param.grad.clone()]
rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()  # noqa: E731
for blocks in [int(v) for v in args.blocks.split(",")]:
    _lib.lib().fl_debug_set_sos_chunk(100 * blocks)
    for p in (mat.param, geq.param):
        p.grad = None
    ops.mean_square(model(x[:2])).backward()
    errs = (rel(mat.param.grad, ref[0]), rel(geq.param.grad, ref[1]))
    for _ in range(3):
        ops.mean_square(model(x)).backward()
    torch.cuda.synchronize()
    ops.kernel_timer.reset(True, 200_000)
    for _ in range(8):
        ops.mean_square(model(x)).backward()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    sm = ops.kernel_timer.summary()
    print(f"blocks {blocks}: ", {k: round(v[1] * 1e3, 1) for k, v in sm.items() if "sos" in k or "geq" in k},
          f" grad vs float64 module: W {errs[0]:.1e}  gains {errs[1]:.1e}")
_lib.lib().fl_debug_set_sos_chunk(0)
