"""Phase cycle sums of fl_spec_gradh_loop_* (one wavefront per workgroup) and its launch time.
    python tools/dbg/gradloop_dbg.py [--dtype f64] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f64")
ap.add_argument("--nfft", type=int, default=96000)
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float64 if args.dtype == "f64" else torch.float32
cd = torch.complex128 if args.dtype == "f64" else torch.complex64
nfft, N, B = args.nfft, args.n, args.batch
M = nfft // 2 + 1
torch.manual_seed(0)
x = torch.randn(B, nfft, N, device=dev, dtype=dt)
H = ops.permute_bins(torch.randn(M, N, N, device=dev, dtype=cd) / N ** 0.5, nfft).requires_grad_(True)


def step():
    y = ops.spectral_apply(x, H, nfft)
    torch.autograd.grad(ops.mean_square(y), [H])


def timed(n=20):
    for _ in range(5):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for on in (True, False):
    ops.GRADH_LOOP = on
    print(f"one-launch response gradient {on}: {timed():.1f} us per eager step")
    ops.kernel_timer.reset(True)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    print("   ", {k: round(v[1] * 1e3, 1) for k, v in sorted(ops.kernel_timer.summary().items())})
ops.GRADH_LOOP = True
for _ in range(5):
    step()
torch.cuda.synchronize()
ops.kernel_timer.reset(True)
for _ in range(20):
    step()
torch.cuda.synchronize()
ops.kernel_timer.enabled = False
for k, (n, ms) in sorted(ops.kernel_timer.summary().items()):
    print(f"   {k:34s} x{n / 20:.0f}  {ms * 1e3:7.1f} us")
nwg = 1024
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
_lib.lib().fl_debug_set_spec_times(buf.data_ptr())
step()
torch.cuda.synchronize()
_lib.lib().fl_debug_set_spec_times(None)
t = buf.view(nwg, 8).cpu().double()
t = t[t[:, 0] > 0][:, :6]
names = ["P1 (rows from LDS, 8-point residues)", "barrier", "P2 (+ issue of the next rows)", "wait spectrum + barrier", "P3 split + products",
         "wait rows + barrier + issue spectrum"]
tot = t.sum(1).mean().item()
print(f"{len(t)} workgroups, {tot:.0f} cycles per workgroup, {tot / B:.0f} per item")
for n_, v in zip(names, t.mean(0).tolist()):
    print(f"   {n_:44s} {v / B:8.0f} per item ({100 * v / tot:4.1f} %)")
