"""gradh register tile: 4x4 / 8x4 / 8x8 at config 2 (HIP events)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0")
B, M, N = 32, 48001, 8
G = ops._empty_planar((B, M, N), torch.complex64, dev); G.copy_(torch.randn(B, M, N, device=dev, dtype=torch.complex64))
X = ops._empty_planar((B, M, N), torch.complex64, dev); X.copy_(torch.randn(B, M, N, device=dev, dtype=torch.complex64))
big = torch.empty(64 * 1024 * 1024, device=dev)
for name, cap in (("4x4", 0), ("8x4", -84), ("8x8", -88)):
    _lib.lib().fl_debug_set_mimo_variant(0, cap)
    for _ in range(3):
        ops._gradh_launch(G, X, False)
    ts = []
    for _ in range(10):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops._gradh_launch(G, X, False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(name, f"{sorted(ts)[len(ts)//2]:.1f} us")
_lib.lib().fl_debug_set_mimo_variant(0, 0)
