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
gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
print("first", out0.item())
step = sys.argv[1] if len(sys.argv) > 1 else "all"
for i in range(3):
    out = gs.replay(); torch.cuda.synchronize()
    print(i, "out", out.item())
    if step in ("all", "eq_out"):
        torch.equal(out, out0)
    if step in ("all", "eq_grads"):
        for p, g in zip(params, g0):
            torch.equal(p.grad, g)
    if step == "eq_one":
        torch.equal(params[int(sys.argv[2])].grad, g0[int(sys.argv[2])])
    torch.cuda.synchronize()
with torch.no_grad():
    print("eager", (model(x) * c).sum().item())
