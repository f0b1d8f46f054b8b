"""bench.py step with mimo_full workgroup size forced (gradw_cap -64 / -256) vs the default heuristic."""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from flamo_amd import _lib, ops
import bench
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
from flamo_amd.graph import GraphedStep
import time
for cap in (-256, -64, 0, -256, 0):
    L.fl_debug_set_mimo_variant(0, cap)
    gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
    for _ in range(10): gs.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): gs.replay()
    torch.cuda.synchronize()
    print("cap", cap, "ms/step %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
L.fl_debug_set_mimo_variant(0, 0)
