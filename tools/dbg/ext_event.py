"""Can timing events be recorded inside a captured graph (external events) and read after a replay?"""
import torch
x = torch.randn(1 << 24, device="cuda")
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external events:", ex)
    raise SystemExit
y = x * 2
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e0.record()
    y = torch.sin(x) * 2 + x
    e1.record()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("elapsed in replay:", e0.elapsed_time(e1), "ms")
