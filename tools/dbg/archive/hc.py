"""Constant-matrix products (response compositions, Gain/Matrix on signals): tile / unroll variants, run under
rocprofv3 --kernel-trace and read with ktrace_top.py (eager event timing is host-bound at these sizes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib(); torch.manual_seed(0)
No, Ni, B, M = 8, 8, 8, 48001
W = torch.randn(No, Ni, dtype=torch.complex64, device=dev)
X = ops.to_planar(torch.randn(B, M, Ni, dtype=torch.complex64, device=dev))
for var in (0, 842, 824, 444, 424, 422, 881, 481, 482):
    L.fl_debug_set_mimo_variant(var, 0)
    for _ in range(20): ops._mimo_launch(W, False, False, False, X)
    torch.cuda.synchronize()
L.fl_debug_set_mimo_variant(0, 0)
