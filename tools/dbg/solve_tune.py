#!/usr/bin/env python
"""Closed-loop solve kernels side by side: variant 0 (rows exchanged in place, DPP broadcasts) against variant 1
(shuffle kernel), for the factored FDN loop matrix and for general matrices that need row exchanges.
    python tools/dbg/solve_tune.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    torch.manual_seed(0)
    for real, cd in ((torch.float32, torch.complex64), (torch.float64, torch.complex128)):
        for N, M in [(int(a), 96001) for a in os.environ.get("SOLVE_NS", "4,8,16,32").split(",")]:
            U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(dev, cd)
            l = (0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, cd)
            R = torch.randn(1, M, N, dtype=cd, device=dev)
            # general matrices: Gaussian, and a permutation plus noise (zero-ish diagonal: exchanges at every step)
            G = torch.randn(2000, N, N, dtype=cd, device=dev)
            perm = torch.eye(N, dtype=cd, device=dev)[torch.randperm(N)]
            Pm = perm.unsqueeze(0) + 0.05 * torch.randn(2000, N, N, dtype=cd, device=dev)
            Rg = torch.randn(3, 2000, N, dtype=cd, device=dev)
            out = {}
            for v in (0, 1):
                L.fl_debug_set_solve_variant(v)
                t = timeit(lambda: ops.solve_dud(l, U, None, R))
                y = ops.solve_dud(l, U, None, R)
                A = torch.eye(N, dtype=torch.complex128, device=dev) - l.to(torch.complex128).unsqueeze(-1) * U.to(torch.complex128)
                ref = torch.linalg.solve(A, R[0].to(torch.complex128).unsqueeze(-1)).squeeze(-1)
                e1 = ((y[0] - ref).norm() / ref.norm()).item()
                errs = []
                for mat in (G, Pm):
                    x = ops.solve(mat, Rg, one_minus=False)
                    rg = torch.linalg.solve(mat.to(torch.complex128).unsqueeze(0), Rg.to(torch.complex128).unsqueeze(-1)).squeeze(-1)
                    errs.append(((x - rg).norm() / rg.norm()).item())
                out[v] = (t, e1, errs)
            L.fl_debug_set_solve_variant(0)
            print(f"{str(real)[6:]} N={N:2d} M={M}: in-place {out[0][0]:7.1f} us (err {out[0][1]:.1e}, general {out[0][2][0]:.1e} {out[0][2][1]:.1e})"
                  f" | shuffle {out[1][0]:7.1f} us (err {out[1][1]:.1e}, general {out[1][2][0]:.1e} {out[1][2][1]:.1e})")


if __name__ == "__main__":
    main()
