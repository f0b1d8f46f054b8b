"""Host-side cost of an eager config-2 step (cProfile over 100 steps)."""
import os, sys, cProfile, pstats, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
import bench
from flamo_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
def step():
    for p in params: p.grad = None
    ops.mean_square(model(x)).backward()
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print("eager ms/step: %.3f" % ((time.perf_counter() - t0) / 100 * 1e3))
# host time alone: enqueue without waiting
t0 = time.perf_counter()
for _ in range(100): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host enqueue ms/step: %.3f" % ((t1 - t0) / 100 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
