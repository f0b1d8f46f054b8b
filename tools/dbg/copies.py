import sys, torch, warnings
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(32, 96000, 8, device=dev)
def step():
    for p in params: p.grad = None
    y = model(x); loss = (y ** 2).mean(); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous") and e.input_shapes and any(len(s) and (s[0] if s else 0) for s in e.input_shapes):
        n = 1
        for d in (e.input_shapes[0] or []): n *= d
        if n > 1_000_000:
            print(e.name, e.input_shapes, [str(f) for f in (e.stack or [])[:6]])
