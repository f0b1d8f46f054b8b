"""Constant-matrix products (response compositions, Gain/Matrix on signals): scalar-load specialisation on / off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
for (No, Ni, B, M) in ((8, 8, 8, 48001), (16, 1, 1, 96001), (1, 16, 1, 96001), (8, 8, 32, 48001)):
    W = torch.randn(No, Ni, dtype=torch.complex64, device=dev)
    X = ops.to_planar(torch.randn(B, M, Ni, dtype=torch.complex64, device=dev))
    ref = None
    for cap in (-4, 0, -4, 0):
        L.fl_debug_set_mimo_variant(0, cap)
        Y = ops._mimo_launch(W, False, False, False, X)
        if ref is None: ref = Y
        for _ in range(3): ops._mimo_launch(W, False, False, False, X)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops._mimo_launch(W, False, False, False, X)
        e1.record(); torch.cuda.synchronize()
        print(f"W {No}x{Ni} B={B} M={M} {'per-bin addressing' if cap else 'scalar loads      '}: {e0.elapsed_time(e1)/20*1e3:7.1f} us  equal {torch.equal(Y, ref)}")
L.fl_debug_set_mimo_variant(0, 0)
