"""config-5 chain with a solve variant forced: python tools/dbg/c5_variant.py <variant> [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402
import bench_fdn  # noqa: E402
v = int(sys.argv[1])
_lib.lib().fl_debug_set_solve_variant(v)
dev = torch.device("cuda:0")
torch.manual_seed(130709)
model, params = bench_fdn.build_config5(dev, torch.float32, 32, 384000)
x = torch.randn(1, 384000, 32, device=dev)
c = torch.randn(1, 384000, 32, device=dev)
gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params)
for _ in range(5):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    l = gs.replay()
torch.cuda.synchronize()
print(f"variant {v}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per step, loss {l.item():.6e}, grads {[float(p.grad.abs().sum()) for p in params]}")
