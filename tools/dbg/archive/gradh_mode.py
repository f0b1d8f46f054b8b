"""spectral_apply forward + backward (graph replay) with a tuning mode of the backward walking kernel:
python tools/dbg/gradh_mode.py <mode> [nfft N B]   -- run under rocprofv3 --kernel-trace --stats for per-kernel times"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402
mode = int(sys.argv[1])
nfft, N, B = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (96000, 8, 32)
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = nfft // 2 + 1
x = torch.randn(B, nfft, N, device=dev)
H = (torch.randn(M, N, N, device=dev, dtype=torch.complex64) / N ** 0.5)
Hr = ops.permute_bins(H, nfft).requires_grad_(True)
_lib.lib().fl_debug_set_walk(mode, 0, 0, None)
gs = GraphedStep(lambda xx: ops.mean_square(ops.spectral_apply(xx, Hr, nfft)), (x,), [Hr])
for _ in range(60):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    gs.replay()
torch.cuda.synchronize()
print(f"mode {mode}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per step")
