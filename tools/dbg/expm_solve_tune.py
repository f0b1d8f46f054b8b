"""matrix_exp at N = 16 (matrix-core kernels against the LDS kernels) and the N = 16 solve's tuning variants.
    python tools/dbg/expm_solve_tune.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(0)
X = (torch.randn(16, 16, device=dev) * 1.0).requires_grad_(True)
Cw = torch.randn(16, 16, device=dev)
def fwd_only():
    with torch.no_grad():
        ops.matrix_exp_both(X, skew=True)


def fwd_bwd():
    E, Ec = ops.matrix_exp_both(X, skew=True)
    torch.autograd.grad((E * Cw).sum() + (Ec.real * Cw).sum(), [X])


for on in (1, 0):
    L.fl_debug_set_expm_mfma(on)
    # back-to-back launches on one stream: the launch gaps are in both figures; the difference is the kernels'
    print(f"matrix_exp N=16, matrix cores {on}: forward {timeit(fwd_only, 200):.1f} us, forward + backward (with the two torch sums) {timeit(fwd_bwd, 200):.1f} us")
L.fl_debug_set_expm_mfma(1)

N, M = 16, 96001
cd = torch.complex64
U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(dev, cd)
l = (0.98 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))).to(dev, cd)
R = torch.randn(1, M, N, dtype=cd, device=dev)
A = torch.eye(N, dtype=torch.complex128, device=dev) - l.to(torch.complex128).unsqueeze(-1) * U.to(torch.complex128)
ref = torch.linalg.solve(A, R[0].to(torch.complex128).unsqueeze(-1)).squeeze(-1)
names = {0: "default (8 lanes x 2 rows, DPP broadcasts)", 5: "pivot row through LDS", 6: "3 wavefronts per SIMD", 7: "4 wavefronts per SIMD",
         8: "LDS + 3 per SIMD", 9: "LDS + 4 per SIMD", 4: "one row per lane"}
for v in (0, 4, 0):
    L.fl_debug_set_solve_variant(v)
    t = timeit(lambda: ops.solve_dud(l, U, None, R))
    y = ops.solve_dud(l, U, None, R)
    err = ((y[0] - ref).norm() / ref.norm()).item()
    print(f"solve N=16 M={M} variant {v} ({names[v]}): {t:7.1f} us  err {err:.2e}")
L.fl_debug_set_solve_variant(0)
