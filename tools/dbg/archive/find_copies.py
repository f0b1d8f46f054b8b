"""Where do the large torch copies / fills of a config-5 step come from?  (torch.profiler with stacks)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings; warnings.simplefilter("ignore")
from bench_fdn import build_config5
dev = torch.device("cuda:0")
torch.manual_seed(130709)
model, params = build_config5(dev, torch.float32, 32, 384000)
x = torch.randn(1, 384000, 32, device=dev); c = torch.randn(1, 384000, 32, device=dev)
def step():
    for p in params: p.grad = None
    (model(x) * c).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_time_total > 300 and e.name.startswith("aten::")]
seen = set()
for e in sorted(evs, key=lambda e: -e.device_time_total)[:14]:
    st = [s for s in (e.stack or []) if "flamo_amd" in s or "bench_fdn" in s][:4]
    key = (e.name, tuple(st))
    if key in seen: continue
    seen.add(key)
    print(f"{e.name:28s} {e.device_time_total:9.0f} us  shapes {e.input_shapes if hasattr(e,'input_shapes') else ''}")
    for s in st: print("      ", s)
