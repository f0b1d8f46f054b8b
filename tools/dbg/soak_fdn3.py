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
    ys["y"] = y
    return (y * c).sum()
for _ in range(2):
    (model(x) * c).sum().backward()
for p in params:
    p.grad = None
torch.cuda.synchronize()
gs = GraphedStep(fn, (x,), params, warmup=0)
vals = []
for i in range(8):
    out = gs.replay(); torch.cuda.synchronize()
    vals.append((out.item(), ys["y"].double().abs().sum().item(), (ys["y"].double() * c.double()).sum().item()))
print(vals)
with torch.no_grad():
    y = model(x)
    print("eager:", (y * c).sum().item(), y.double().abs().sum().item(), (y.double() * c.double()).sum().item())
    print("y finite:", torch.isfinite(y).all().item(), "max |y|", y.abs().max().item())
