"""Interleaved A/B of per-bin product variants on the config-2 block (fl_debug_set_mimo_variant):
variant = mt*100 + bt*10 + nu (0 = default)."""
import sys, os, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from flamo_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
nfft, B, N = 96000, 32, 8
M = nfft // 2 + 1
torch.manual_seed(0)
X = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
X2 = ops.to_planar(torch.randn(B, M, N, dtype=torch.complex64, device=dev))
H = ops._h_planar(torch.randn(M, N, N, dtype=torch.complex64, device=dev), True)
variants = [int(v) for v in sys.argv[1:]] or [0, 441, 481]
ref = None
for var in variants:
    L.fl_debug_set_mimo_variant(var, 0)
    for _ in range(3): Y = ops._mimo_launch(H, True, False, False, X)
    if ref is None: ref = Y
    print("variant", var, "max err vs first", ((Y - ref).abs().max() / ref.abs().max()).item())
torch.cuda.synchronize()
# back-to-back launches (hot queue), alternating inputs; several rounds interleaved
times = {v: [] for v in variants}
for rnd in range(6):
    for var in variants:
        L.fl_debug_set_mimo_variant(var, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400_000)
        e0.record()
        for i in range(10):
            ops._mimo_launch(H, True, False, False, X2 if i % 2 else X)
        e1.record(); torch.cuda.synchronize()
        times[var].append(e0.elapsed_time(e1) / 10 * 1e3)
for var in variants:
    t = sorted(times[var])
    print("variant", var, "median us %.1f" % statistics.median(t), "min %.1f" % t[0], "GB/s(median) %.0f" % (221.19e6 / statistics.median(t) / 1e3))
L.fl_debug_set_mimo_variant(0, 0)
