"""MFMA product kernel: output through LDS with 16-byte stores (default) against direct stores (gradw_cap -2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
N, M = 32, 192001
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
Hp = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
Xp = ops.to_planar(torch.randn(1, M, N, N, dtype=torch.complex64, device=dev))
ref = None
for var in (-1, 0, -14):
    for cap in (0, -2):
        L.fl_debug_set_mimo_variant(var, cap)
        Y = ops._mimo_launch(Hp, True, False, False, Xp)
        if ref is None: ref = Y
        err = ((Y - ref).norm() / ref.norm()).item()
        for _ in range(2): ops._mimo_launch(Hp, True, False, False, Xp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops._mimo_launch(Hp, True, False, False, Xp)
        e1.record(); torch.cuda.synchronize()
        print(f"variant {var:4d} {'direct stores' if cap else 'LDS-staged stores'}: {e0.elapsed_time(e1)/5*1e3:8.1f} us  relerr vs lane {err:.1e}")
L.fl_debug_set_mimo_variant(0, 0)
