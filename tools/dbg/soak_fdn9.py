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
ys = {}
def fn(xx):
    y = model(xx)
    if torch.cuda.is_current_stream_capturing():
        ys["y"] = y
        ys["yc"] = y * c
        return ys["yc"].sum()
    return (y * c).sum()
gs = GraphedStep(fn, (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
y0, yc0 = ys["y"].clone(), ys["yc"].clone()
print("first", out0.item(), "sum(yc) recomputed", ys["yc"].sum().item())
for i in range(4):
    out = gs.replay()
    torch.cuda.synchronize()
    print(i, "out", out.item(), "y same", torch.equal(ys["y"], y0), "yc same", torch.equal(ys["yc"], yc0), "eager sum(yc)", ys["yc"].sum().item(),
          "x same", torch.equal(gs.static_inputs[0], x), "c sum", c.sum().item())
    for p, g in zip(params, g0):
        torch.equal(p.grad, g)
