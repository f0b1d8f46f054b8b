"""spectral_apply forward + backward at a length, channels, batch: ms per step from graph replays (A/B of two library builds
through FLAMO_HIP_LIB).   python tools/dbg/plan_ab.py 192000 8 8"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402
nfft, N, B = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = nfft // 2 + 1
x = torch.randn(B, nfft, N, device=dev)
H = (torch.randn(M, N, N, device=dev, dtype=torch.complex64) / N ** 0.5).requires_grad_(True)
Hr = ops.permute_bins(H.detach(), nfft).requires_grad_(True)
gs = GraphedStep(lambda xx: ops.mean_square(ops.spectral_apply(xx, Hr, nfft)), (x,), [Hr])
for _ in range(60):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    gs.replay()
torch.cuda.synchronize()
print(f"nfft {nfft} {N}x{N} batch {B} walk={ops._walk_applies(nfft, B, N, N)}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")
