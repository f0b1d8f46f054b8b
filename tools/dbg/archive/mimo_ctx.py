"""Does the per-bin product run slower right behind a kernel that has just written its input (as in the step)?"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from flamo_amd import ops
dev = torch.device("cuda:0")
nfft, B, N = 96000, 32, 8
M = nfft // 2 + 1
torch.manual_seed(0)
X = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
H = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
Xbase = X.movedim(1, -1)            # memory-order view (B, N, M)
src = torch.randn_like(torch.view_as_real(Xbase))
other = torch.randn(B * N * 48032 * 2, device=dev)
other2 = torch.empty_like(other)
def run(pre):
    ts = []
    for rep in range(30):
        torch.cuda._sleep(300_000)
        if pre == "write_x":
            torch.view_as_real(Xbase).copy_(src)          # a kernel that has just written X (like the FFT row pass)
        elif pre == "write_other":
            other2.copy_(other)                            # same traffic, unrelated buffers
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops._mimo_launch(H, True, False, False, X)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)
for pre in ("none", "write_x", "write_other", "none"):
    med, mn = run(pre)
    print("%-12s median %.1f us  min %.1f" % (pre, med, mn))
