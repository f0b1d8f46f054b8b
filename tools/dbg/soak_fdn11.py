import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
mode = sys.argv[1]
torch.manual_seed(1)
model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
vals = []
for i in range(4):
    out = gs.replay()
    torch.cuda.synchronize()
    if mode == "item_first":
        vals.append(out.item())
    cur = [out] + [p.grad for p in params]
    for t, (a, b) in enumerate(zip(cur, [out0] + g0)):
        torch.equal(a, b)
    if mode == "item_last":
        vals.append(out.item())
    if mode == "clone_last":
        vals.append(out.clone())
print(mode, [v if isinstance(v, float) else v.item() for v in vals], "final static_out", gs.static_out.item())
