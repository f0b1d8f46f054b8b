"""config-2 step under HIP-graph replay for several column-pass tilings (fl_debug_set_spec), wall clock, interleaved"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flamo_amd import _lib, ops
from flamo_amd.graph import GraphedStep
if len(sys.argv) > 1:
    bench.NCH = int(sys.argv[1])
if len(sys.argv) > 2:
    bench.NFFT = int(sys.argv[2])
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

variants = [(32, 2), (16, 2), (32, 4), (32, 1), (16, 1)]
graphs = {}
for v in variants:
    _lib.lib().fl_debug_set_spec(*v)
    graphs[v] = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
for rep in range(3):
    print("  ".join(f"vt{v[0]}/rg{v[1]}: {timed(graphs[v].replay):.4f}" for v in variants))
