"""bench.py's config-2 step in float64, replayed: run under rocprofv3 --kernel-trace for the timeline / per-kernel times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flamo_amd  # noqa: F401
import torch
import bench
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device("cuda", 0)
dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
torch.manual_seed(130709)
model, params = bench.build_model(dev, dt)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev, dtype=dt)
gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
for _ in range(60):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    gs.replay()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step ({dt})")
