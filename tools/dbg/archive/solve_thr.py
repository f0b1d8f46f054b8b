"""Pivot-threshold sweep of the in-place solve kernels on the FDN loop matrix I - diag(l) U (how often does the rare
row-exchange path run?).   python tools/dbg/solve_thr.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
M = 96001
for N in (16, 32):
    U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(dev, torch.complex64)
    l = (0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, torch.complex64)
    R = torch.randn(1, M, N, dtype=torch.complex64, device=dev)
    A = torch.eye(N, dtype=torch.complex128, device=dev) - l.to(torch.complex128).unsqueeze(-1) * U.to(torch.complex128)
    ref = torch.linalg.solve(A, R[0].to(torch.complex128).unsqueeze(-1)).squeeze(-1)
    for thr in (1, 2, 3, 5, 30):
        L.fl_debug_set_solve_variant(10 + thr)
        y = ops.solve_dud(l, U, None, R)
        err = ((y[0] - ref).norm() / ref.norm()).item()
        for _ in range(3): ops.solve_dud(l, U, None, R)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.solve_dud(l, U, None, R)
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} threshold 2^-{thr}: {e0.elapsed_time(e1)/10*1e3:8.1f} us (op incl. host)  err {err:.1e}")
L.fl_debug_set_solve_variant(0)
