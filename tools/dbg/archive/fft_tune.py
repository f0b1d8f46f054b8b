"""Time the FFT passes alone (planar input, no transposes): run under rocprofv3 --kernel-trace --stats for per-kernel numbers."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from flamo_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
nfft, B, N = 96000, 32, 8
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L.fl_debug_set_fft_fast(mode)
torch.manual_seed(0)
x = torch.randn(B, N, nfft, device=dev).movedim(-1, 1)      # planar (B, T, N)
ref = torch.fft.rfft(x.double(), n=nfft, dim=1)
X = ops.rfft(x, nfft)
print("mode", mode, "rfft relerr", ((X - ref).norm() / ref.norm()).item())
y = ops.irfft(X, nfft)
print("irfft relerr", ((y - x).norm() / x.norm()).item())
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("rfft us %.1f" % timeit(lambda: ops.rfft(x, nfft)), "irfft us %.1f" % timeit(lambda: ops.irfft(X, nfft)))
