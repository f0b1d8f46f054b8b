"""rfft / irfft against torch.fft (float64 on the device) for every pair of fast sub-lengths, float32 and float64."""
import os, sys, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
worst = 0.0
for L1, L2 in itertools.product((200, 240, 300, 320, 400, 480), repeat=2):
    nfft = 2 * L1 * L2
    for dt, cd, tol in ((torch.float32, torch.complex64, 3e-6), (torch.float64, torch.complex128, 1e-12)):
        x = torch.randn(2, nfft, 3, dtype=dt, device=dev)
        Z = torch.randn(2, nfft // 2 + 1, 3, dtype=cd, device=dev)
        X = ops.rfft(x, nfft); y = ops.irfft(Z, nfft)
        Xr = torch.fft.rfft(x.double(), n=nfft, dim=1); yr = torch.fft.irfft(Z.to(torch.complex128), n=nfft, dim=1)
        e1 = ((X - Xr).norm() / Xr.norm()).item(); e2 = ((y - yr).norm() / yr.norm()).item()
        worst = max(worst, e1 / tol, e2 / tol)
        if e1 > tol or e2 > tol: print("FAIL", nfft, L1, L2, dt, e1, e2)
print("done; worst error / tolerance = %.3f" % worst)
