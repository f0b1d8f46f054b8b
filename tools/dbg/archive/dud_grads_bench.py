"""fl_solve_dud_grads timing at the 16-channel network's shape (events on the current stream)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M = int(sys.argv[2]) if len(sys.argv) > 2 else 96001
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.manual_seed(0)
cd = torch.complex64
OUT = ops._empty_planar((B, M, N, 1), cd, dev); OUT.copy_(torch.randn(B, M, N, 1, dtype=cd, device=dev))
gR = ops._empty_planar((B, M, N, 1), cd, dev); gR.copy_(torch.randn(B, M, N, 1, dtype=cd, device=dev))
l = ops._h_planar(torch.randn(M, N, dtype=cd, device=dev), True)
U = torch.randn(N, N, dtype=cd, device=dev)
for flags in ((True, True, False), (False, True, False), (True, False, False), (True, True, True)):
    rp = l if flags[2] else None
    for _ in range(3):
        ops._dud_grads_launch(l, U, rp, gR, OUT, *flags)
    torch.cuda.synchronize()
    ops.kernel_timer.reset(True)
    for _ in range(20):
        ops._dud_grads_launch(l, U, rp, gR, OUT, *flags)
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    print(f"N={N} M={M} B={B} need(l,U,r)={flags}:", {k: round(v[1] * 1e3, 1) for k, v in ops.kernel_timer.summary().items()}, "us (kernel + final)")
