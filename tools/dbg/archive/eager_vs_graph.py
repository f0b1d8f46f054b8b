"""config-2 step: eager (with / without the side-stream response build) against HIP-graph replay, wall clock"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flamo_amd import ops
from flamo_amd.processor import system
from flamo_amd.graph import GraphedStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)

def step():
    for p in params:
        p.grad = None
    ops.mean_square(model(x)).backward()

def timed(fn, n=50):
    for _ in range(15):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for ov in (True, False):
    system.OVERLAP_RESPONSES = ov
    print(f"eager overlap={ov}: {timed(step):.3f} ms/step")
for ov in (True, False):
    system.OVERLAP_RESPONSES = ov
    gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
    print(f"graph overlap={ov}: {timed(gs.replay):.3f} ms/step")
