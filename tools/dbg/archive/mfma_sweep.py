#!/usr/bin/env python
"""Kernel-only time of the per-bin 32x32x32 (or NxNxN) complex product for the tuning variants of
fl_debug_set_mimo_variant: -1 lane-per-bin, -(10*rb+depth) MFMA tiles.   python tools/dbg/mfma_sweep.py [N] [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 192001
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(0)
Hp = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
Xp = ops.to_planar(torch.randn(1, M, N, N, dtype=torch.complex64, device=dev))
flop = 8.0 * N ** 3 * M
byt = 3 * 8.0 * N * N * M
for v in (-1, 0, -14):
    L.fl_debug_set_mimo_variant(v, 0)
    for adj in (False, True):
        for _ in range(2):
            ops._mimo_launch(Hp, True, False, adj, Xp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops._mimo_launch(Hp, True, False, adj, Xp)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"variant {v:4d} adj={int(adj)}: {t * 1e6:8.1f} us  {flop / t / 1e12:6.1f} TFLOP/s  {byt / t / 1e12:5.2f} TB/s")
L.fl_debug_set_mimo_variant(0, 0)
# constant-matrix gradient (sum over bins): MFMA reduction against the lane-per-bin reduction
Gp = ops.to_planar(torch.randn(1, M, N, N, dtype=torch.complex64, device=dev))
for v in (-1, 0):
    L.fl_debug_set_mimo_variant(v, 0)
    for _ in range(2):
        ops._gradw_launch(Gp, Xp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops._gradw_launch(Gp, Xp)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"gradw variant {v:3d}: {t * 1e6:8.1f} us  {flop / t / 1e12:6.1f} TFLOP/s  {2 * 8.0 * N * N * M / t / 1e12:5.2f} TB/s")
L.fl_debug_set_mimo_variant(0, 0)
