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
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
if mode == "pure_torch":
    w = torch.nn.Parameter(torch.randn(1, device=dev))
    params = [w]
    fn = lambda xx: ((xx * w) * c).sum()
else:
    model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
    fn = (lambda xx: ops.mean_square(model(xx))) if mode == "ms" else (lambda xx: (model(xx) * c).sum())
gs = GraphedStep(fn, (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
vals = []
for i in range(4):
    out = gs.replay()
    torch.cuda.synchronize()
    for a, b in zip([out] + [p.grad for p in params], [out0] + g0):
        torch.equal(a, b)
    vals.append(out.clone())
print(mode, [v.item() for v in vals], "first", out0.item())
