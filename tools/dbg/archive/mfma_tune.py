#!/usr/bin/env python
"""Per-bin products of matrix-valued signals: the MFMA kernel against the lane-per-bin kernels (results, times).
    python tools/dbg/mfma_tune.py [N] [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 192001
    dev = torch.device("cuda:0")
    L = _lib.lib()
    torch.manual_seed(0)
    for (No, Ni, B, K) in ((N, N, 1, N), (N, N, 2, 12), (N - 4, N, 1, N - 3), (N, 8, 3, 8)):
        H = torch.randn(M, No, Ni, dtype=torch.complex64, device=dev, requires_grad=True)
        X = torch.randn(B, M, Ni, K, dtype=torch.complex64, device=dev, requires_grad=True)
        C = torch.randn(B, M, No, K, dtype=torch.complex64, device=dev)
        res = {}
        for v in (0, -1):
            L.fl_debug_set_mimo_variant(v, 0)
            Y = ops.mimo(H, X)
            gH, gX = torch.autograd.grad(torch.sum((Y * C.conj()).real), [H, X])
            with torch.no_grad():
                t_f = timeit(lambda: ops.mimo(H, X))
            res[v] = (Y.detach(), gH, gX, t_f)
        L.fl_debug_set_mimo_variant(0, 0)
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()   # noqa: E731
        ref = torch.einsum("fmn,bfnk->bfmk", H.detach()[:4096].to(torch.complex128), X.detach()[:, :4096].to(torch.complex128))
        print(f"No={No} Ni={Ni} B={B} K={K} M={M}: fwd mfma {res[0][3]:8.1f} us  lane {res[-1][3]:8.1f} us | "
              f"Y {rel(res[0][0], res[-1][0]):.1e} (vs f64 {rel(res[0][0][:, :4096], ref):.1e})  gH {rel(res[0][1], res[-1][1]):.1e}  gX {rel(res[0][2], res[-1][2]):.1e}")


if __name__ == "__main__":
    main()
