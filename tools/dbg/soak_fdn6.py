import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
torch.manual_seed(1)
model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
print("out0", out0.item(), "static_out now", gs.static_out.item())
for i in range(5):
    out = gs.replay()
    torch.cuda.synchronize()
    print(i, out.item(), torch.equal(out, out0), out.dtype, out.shape, out0.shape)
