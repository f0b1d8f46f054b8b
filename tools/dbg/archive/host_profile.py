"""Host-side cost of an eager config-3 FDN step (cProfile over 50 steps)."""
import os, sys, cProfile, pstats, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from bench_fdn import build
dev = torch.device("cuda:0"); torch.manual_seed(130709)
model, params = build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev); c = torch.randn(1, 192000, 1, device=dev)
def step():
    for p in params: p.grad = None
    (model(x) * c).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("eager ms/step: %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
