import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
torch.manual_seed(1)
model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
for _ in range(2):
    (model(x) * c).sum().backward()
for p in params:
    p.grad = None
torch.cuda.synchronize()
ys = {}
def fn(xx):
    y = model(xx)
    ys["y"] = y
    return (y * c).sum()
gs = GraphedStep(fn, (x,), params, warmup=0)
out = gs.replay(); torch.cuda.synchronize()
v0, y0 = out.item(), ys["y"].clone()
print("first", v0)
junk = []
for sz in (1, 16, 256, 4096, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
    for _ in range(4):
        t = torch.full((sz,), 7.0, device=dev)
        junk.append(t)
torch.cuda.synchronize()
out = gs.replay(); torch.cuda.synchronize()
print("after junk allocations:", out.item(), "y equal:", torch.equal(ys["y"], y0), "max |dy|", (ys["y"] - y0).abs().max().item())
del junk
g0 = [p.grad.clone() for p in params]
same = [torch.equal(p.grad, g) for p, g in zip(params, g0)]
out = gs.replay(); torch.cuda.synchronize()
print("after clones of the grads:", out.item(), "y equal:", torch.equal(ys["y"], y0))
