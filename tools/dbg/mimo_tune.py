import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from flamo_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
nfft, B, N = 96000, 32, 8
M = nfft // 2 + 1
torch.manual_seed(0)
X = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
H = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
G = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
# a second set of buffers to defeat the 256 MB infinity cache between repetitions
X2 = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = None
for var in (0, 841, 842, 821, 822, 824, 441, 442, 444, 481, 482, 881, 421, 422, 424):
    L.fl_debug_set_mimo_variant(var, 0)
    try:
        Y = ops._mimo_launch(H, True, False, False, X)
    except RuntimeError as e:
        print(var, "unsupported", e); continue
    if ref is None: ref = Y
    err = ((Y - ref).abs().max() / ref.abs().max()).item()
    flip = [0]
    def go():
        flip[0] ^= 1
        ops._mimo_launch(H, True, False, False, X2 if flip[0] else X)
    us = timeit(go)
    print("variant", var, "us %.1f" % us, "GB/s %.0f" % (221.19e6 / us / 1e3), "err %.1e" % err)
L.fl_debug_set_mimo_variant(0, 0)
Hs = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
Gs = Hs.unsqueeze(0)   # (1, M, N, N) as a signal: the compose-backward shape (8 columns)
for cap in (64, 128, 188):
    L.fl_debug_set_mimo_variant(0, cap)
    us = timeit(lambda: ops._gradw_launch(ops.to_planar(Gs.permute(2, 1, 3, 0)[..., 0] if False else G[:8]), X[:8]))
    print("gradw cap", cap, "us %.1f" % us)
L.fl_debug_set_mimo_variant(0, 0)
