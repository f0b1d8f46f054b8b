"""Factored-loop solve at N in (4, 16]: one row per lane (fl_debug_set_solve_variant(4)) against two rows per lane (0, the
default: 8 lanes x 2 rows up to N = 16, 4 lanes x 2 rows up to N = 8); kernel time through events around the launch, error against LAPACK in float64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
for N, M, B, cd in ((16, 96001, 1, torch.complex64), (8, 96001, 1, torch.complex64), (8, 96001, 8, torch.complex64), (6, 96001, 1, torch.complex64), (8, 96001, 1, torch.complex128)):
    U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
    l64 = 0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
    R64 = torch.randn(B, M, N, dtype=torch.complex128)
    A = torch.eye(N, dtype=torch.complex128) - l64[:2000].unsqueeze(-1) * U64
    U = U64.to(dev, cd); l = ops._h_planar(l64.to(dev, cd), True)
    R = ops.to_planar(R64.to(dev, cd))
    for adj in (False, True):
        Aref = A.conj().transpose(-1, -2) if adj else A
        ref = torch.linalg.solve(Aref.unsqueeze(0), R64[:, :2000].unsqueeze(-1)).squeeze(-1)
        ys = {}
        for v in (4, 0):
            L.fl_debug_set_solve_variant(v)
            y = ops._solve_dud_launch(l, U, None, adj, R)
            ys[v] = y.clone()
            err = ((ys[v] - ys[4]).norm() / ys[4].norm()).item()      # against the 16-lane kernel
            for _ in range(3): ops._solve_dud_launch(l, U, None, adj, R)
            torch.cuda.synchronize()
            ops.kernel_timer.reset(True)
            for _ in range(10): ops._solve_dud_launch(l, U, None, adj, R)
            torch.cuda.synchronize(); ops.kernel_timer.enabled = False
            t = list(ops.kernel_timer.summary().values())[0][1] * 1e3
            print(f"{cd} N={N} M={M} B={B} adjoint={int(adj)} variant {v}: {t:7.1f} us  err {err:.1e}")
L.fl_debug_set_solve_variant(0)
