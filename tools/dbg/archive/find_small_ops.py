"""Which torch (aten) device ops run in a config-3 FDN step, and from where?  (torch.profiler with stacks)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings; warnings.simplefilter("ignore")
from bench_fdn import build
dev = torch.device("cuda:0")
torch.manual_seed(130709)
model, params = build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev); c = torch.randn(1, 192000, 1, device=dev)
def step():
    for p in params: p.grad = None
    (model(x) * c).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_time_total > 0 and e.name.startswith("aten::") and not any(
            c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        st = [s.split("/")[-1] for s in (e.stack or []) if "flamo_amd" in s or "bench_fdn" in s][:2]
        k = (e.name, " <- ".join(st))
        agg[k][0] += 1; agg[k][1] += e.device_time_total
tot = sum(v[1] for v in agg.values())
print("aten device time per step: %.0f us in %d ops" % (tot, sum(v[0] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]:7.1f} us x{v[0]:3d}  {k[0]:26s} {k[1]}")
