"""Closed-loop solve with a materialised loop matrix P (M, N, N): variants 0 (default), 1 (shuffle kernel), 3 (two rows
per lane also for P at N > 16); kernel-level time through events around the C-ABI launch only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
for N, M in ((16, 96001), (32, 96001), (24, 48001)):
    P = (0.6 / N ** 0.5) * torch.randn(M, N, N, dtype=torch.complex64, device=dev)
    Pp = ops._h_planar(P, True)
    R = ops.to_planar(torch.randn(2, M, N, dtype=torch.complex64, device=dev))
    A = torch.eye(N, dtype=torch.complex128, device=dev) - P[:3000].to(torch.complex128)
    for adj in (False, True):
        Aref = A.conj().transpose(-1, -2) if adj else A
        ref = torch.linalg.solve(Aref.unsqueeze(0), R[:, :3000].to(torch.complex128).unsqueeze(-1)).squeeze(-1)
        for v in (0, 1, 3):
            L.fl_debug_set_solve_variant(v)
            y = ops._solve_launch(Pp, True, adj, R)
            err = ((y[:, :3000] - ref).norm() / ref.norm()).item()
            for _ in range(2): ops._solve_launch(Pp, True, adj, R)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops._solve_launch(Pp, True, adj, R)
            e1.record(); torch.cuda.synchronize()
            print(f"N={N} M={M} adjoint={int(adj)} variant {v}: {e0.elapsed_time(e1)/5*1e3:8.1f} us  err {err:.1e}")
L.fl_debug_set_solve_variant(0)
# the factored loop with the same number of right-hand sides, for reference
for N, M in ((32, 96001), (24, 96001), (17, 96001)):
    U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(dev, torch.complex64)
    l = ops._h_planar((0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, torch.complex64), True)
    for B in (1, 2):
        R = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
        for v in (0, 1):
            L.fl_debug_set_solve_variant(v)
            for _ in range(2): ops._solve_dud_launch(l, U, None, False, R)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops._solve_dud_launch(l, U, None, False, R)
            e1.record(); torch.cuda.synchronize()
            print(f"factored N={N} M={M} B={B} variant {v}: {e0.elapsed_time(e1)/5*1e3:8.1f} us")
L.fl_debug_set_solve_variant(0)
