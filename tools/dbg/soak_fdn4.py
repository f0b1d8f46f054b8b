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
for wu in (2, 0):
    gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=wu)
    vals = []
    for i in range(6):
        out = gs.replay(); torch.cuda.synchronize()
        vals.append(out.item())
    o0 = gs.replay().clone(); torch.cuda.synchronize()
    o1 = gs.replay(); torch.cuda.synchronize()
    print("warmup", wu, vals, "clone-after-replay", o0.item(), "next", o1.item(), "ptr same:", o1.data_ptr() == gs.static_out.data_ptr())
