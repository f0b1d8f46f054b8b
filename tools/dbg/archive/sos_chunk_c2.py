"""config-2 step under graph replay for several section-chunk sizes of the cascade backward (fl_debug_set_sos_chunk)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flamo_amd import _lib, ops
from flamo_amd.graph import GraphedStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)

def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

variants = [0, 12, 8, 6, 4]
graphs = {}
for v in variants:
    _lib.lib().fl_debug_set_sos_chunk(v)
    graphs[v] = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
_lib.lib().fl_debug_set_sos_chunk(0)
for rep in range(3):
    print("  ".join(f"chunk {v}: {timed(graphs[v].replay):.4f}" for v in variants))
