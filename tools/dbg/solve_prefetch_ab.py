"""The factored FDN solve (fl_solve_dud_*, N = 16 on 8 lanes x 2 rows) with the right-hand side requested at the top of the kernel
(default) against behind the elimination (fl_debug_set_solve_variant(5)), interleaved on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(0)
for N, M in ((16, 96001), (8, 96001), (32, 96001)):
    cd = torch.complex64
    U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(dev, cd)
    l = (0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, cd)
    r = (0.9 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, cd)
    R = torch.randn(1, M, N, dtype=cd, device=dev)
    res = {0: [], 5: []}
    for rep in range(5):
        for v in (0, 5):
            L.fl_debug_set_solve_variant(v)
            for _ in range(5):
                ops.solve_dud(l, U, r, R)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.solve_dud(l, U, r, R)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 50 * 1e3)
    L.fl_debug_set_solve_variant(0)
    print(f"N={N}: prefetch {min(res[0]):.1f} us (median {sorted(res[0])[2]:.1f}) | behind the elimination {min(res[5]):.1f} us (median {sorted(res[5])[2]:.1f})")
