"""Fused Shell pipeline in float64 against the layered float64 operators: time per forward + backward.
    python tools/dbg/spec64_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for nfft, N, B in ((96000, 8, 8), (96000, 8, 32), (192000, 8, 8), (384000, 4, 8), (65536, 8, 8)):
    M = nfft // 2 + 1
    torch.manual_seed(0)
    x = torch.randn(B, nfft, N, device=dev, dtype=torch.float64, requires_grad=True)
    H = (torch.randn(M, N, N, device=dev, dtype=torch.complex128) / N ** 0.5).requires_grad_(True)
    Hrm = ops.permute_bins(H.detach(), nfft).requires_grad_(True)

    def fused():
        y = ops.spectral_apply(x, Hrm, nfft)
        torch.autograd.grad(y.square().mean(), [x, Hrm])

    def layered():
        X = ops.rfft(x, nfft)
        Y = ops.mimo(H, X)
        y = ops.irfft(Y, nfft)
        torch.autograd.grad(y.square().mean(), [x, H])

    for name, fn in (("fused", fused), ("layered", layered)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"nfft={nfft} N={N} B={B} float64 {name}: {e0.elapsed_time(e1) / 10:.3f} ms per forward+backward")
