"""the per-bin 8x8 product at config 2: register-tile kernel against the LDS-DMA streaming kernel (HIP events; cold = behind
192 MB of unrelated copies, warm = back to back on alternating inputs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0")
B, M, N = 32, 48001, 8
torch.manual_seed(0)
Xs = [ops._empty_planar((B, M, N), torch.complex64, dev) for _ in range(2)]
for X in Xs:
    X.copy_(torch.randn(B, M, N, device=dev, dtype=torch.complex64))
H = ops._h_planar(torch.randn(M, N, N, device=dev, dtype=torch.complex64), True)
big = torch.empty(48 * 1024 * 1024, device=dev)
ref = None
for name, cap in (("register tile 8x4", 0), ("stream s1 cb4", -1614), ("stream s2 cb4", -1624), ("stream s2 cb2", -1622), ("stream s4 cb2", -1642), ("stream s4 cb4", -1644), ("stream s8 cb2", -1682)):
    _lib.lib().fl_debug_set_mimo_variant(0, cap)
    Y = ops._mimo_launch(H, True, False, False, Xs[0])
    torch.cuda.synchronize()
    if ref is None:
        ref = Y.clone()
    else:
        print("   max |diff| vs register-tile kernel:", (Y - ref).abs().max().item(), " rel l2:", ((Y - ref).norm() / ref.norm()).item())
        Ya = ops._mimo_launch(H, True, False, True, Xs[0])
        _lib.lib().fl_debug_set_mimo_variant(0, 0)
        Yb = ops._mimo_launch(H, True, False, True, Xs[0])
        _lib.lib().fl_debug_set_mimo_variant(0, cap)
        print("   adjoint (H^H) rel l2:", ((Ya - Yb).norm() / Yb.norm()).item())
    for mode in ("cold", "warm"):
        ts = []
        for i in range(12):
            if mode == "cold":
                big.zero_(); big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops._mimo_launch(H, True, False, False, Xs[i & 1]); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        print(f"{name:20s} {mode}: {t:6.1f} us  {221.19 / t:5.2f} TB/s  frac {221.19 / t / 8:.3f}")
_lib.lib().fl_debug_set_mimo_variant(0, 0)
