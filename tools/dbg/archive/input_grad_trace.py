"""bench.py's step WITH the gradient of the input, replayed: run under rocprofv3 --kernel-trace --stats for the per-kernel times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flamo_amd  # noqa: F401
import torch
import bench
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device("cuda", 0)
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
xg = x.detach().clone().requires_grad_(True)
gs = GraphedStep(lambda xx: ops.mean_square(model(xg)), (x,), list(params) + [xg], warmup=2)
for _ in range(100):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    gs.replay()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per step with the input's gradient")
