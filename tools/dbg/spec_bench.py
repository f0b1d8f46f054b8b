"""Per-kernel timings of the fused Shell pipeline at BASELINE configs[1] (HIP events, single stream).
    python tools/dbg/spec_bench.py [--vt 32] [--steps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--vt", type=int, default=0)
ap.add_argument("--rg", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--nfft", type=int, default=96000)
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
nfft, N, B = args.nfft, args.n, args.batch
M = nfft // 2 + 1
_lib.lib().fl_debug_set_spec(args.vt, args.rg)
torch.manual_seed(0)
x = torch.randn(B, nfft, N, device=dev)
H = ops.permute_bins(torch.randn(M, N, N, device=dev, dtype=torch.complex64) / N ** 0.5, nfft).requires_grad_(True)
xg = x.clone().requires_grad_(True)


def step(with_x):
    y = ops.spectral_apply(xg if with_x else x, H, nfft)
    loss = ops.mean_square(y)
    torch.autograd.grad(loss, [H] + ([xg] if with_x else []))


for with_x in (False, True):
    for _ in range(5):
        step(with_x)
    torch.cuda.synchronize()
    ops.kernel_timer.reset(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step(with_x)
    e1.record()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    tot = 0.0
    print(f"--- vt={args.vt} input_grad={with_x}: {e0.elapsed_time(e1) / args.steps * 1e3:.1f} us per step (eager, events)")
    for k, (n, ms) in sorted(ops.kernel_timer.summary().items()):
        print(f"   {k:34s} x{n / args.steps:.0f}  {ms * 1e3:7.1f} us")
        tot += ms * n / args.steps
    print(f"   sum of timed kernels {tot * 1e3:.1f} us")

# ---- mid-kernel variants (tuning): response-row prefetch modes, no-response floor
S = ops._spec_cols_fwd(x, nfft, 0.0)
Hp = ops._h_planar(H.detach(), True)
for pf in (1, 5, 2, 4):
    _lib.lib().fl_debug_set_spec(args.vt, 100 * pf + args.rg)
    for _ in range(3):
        ops._spec_mid(S.clone(), B, N, N, nfft, Hp, False, True, True, 1.0, 0, 0)
    torch.cuda.synchronize()
    ops.kernel_timer.reset(True)
    for _ in range(10):
        ops._spec_mid(S.clone(), B, N, N, nfft, Hp, False, True, True, 1.0, 0, 0)
    torch.cuda.synchronize()
    print(f"mid pf={pf}:", {k: round(v[1] * 1e3, 1) for k, v in ops.kernel_timer.summary().items()})
ops.kernel_timer.reset(True)
for _ in range(10):
    ops._spec_mid(S.clone(), B, N, N, nfft, None, False, True, True, 1.0, 0, 0)
    ops._spec_mid(S.clone(), B, N, N, nfft, None, False, False, True, 1.0, 0, 0)
torch.cuda.synchronize()
print("mid without response:", {k: round(v[1] * 1e3, 1) for k, v in ops.kernel_timer.summary().items()})
ops.kernel_timer.enabled = False

# ---- phase timeline of the mid kernel (cycle stamps of one lane per workgroup)
nwg = ((nfft // 2 // 240 // 2 + 1 + 7) // 8) * 8 * B if nfft == 96000 else 0
for bgv in ((5, 2) if nwg else ()):
    _lib.lib().fl_debug_set_spec(args.vt, 100 * bgv + args.rg)
    print(f"--- batch items per workgroup: {bgv}")
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    _lib.lib().fl_debug_set_spec_times(buf.data_ptr())
    ops._spec_mid(S.clone(), B, N, N, nfft, Hp, False, True, True, 1.0, 0, 0)
    torch.cuda.synchronize()
    _lib.lib().fl_debug_set_spec_times(None)
    t = buf.view(nwg, 8).cpu().double()
    t = t[t[:, 5] > 0]
    d = (t[:, 1:6] - t[:, 0:5])
    names = ["P1 load+fft16", "P2 fft15", "P3 split/product/pre", "P4 ifft16", "P5 ifft15+store"]
    tot = (t[:, 5] - t[:, 0]).mean().item()
    print(f"mid phases (shader cycles per workgroup, mean over {len(t)} workgroups; whole body {tot:.0f}):")
    for n_, v in zip(names, d.mean(0).tolist()):
        print(f"   {n_:24s} {v:9.0f}  ({100 * v / tot:4.1f} %)")
    print(f"   of P3: LDS reads + split step + spectrum store {(t[:, 6] - t[:, 2]).mean().item():9.0f}, product + pre-step {(t[:, 3] - t[:, 6]).mean().item():9.0f}")
    span = (t[:, 5].max() - t[:, 0].min()).item()
    print(f"   kernel span {span:.0f} cycles; sum of workgroup bodies / 256 CUs = {(t[:, 5] - t[:, 0]).sum().item() / 256:.0f}")
