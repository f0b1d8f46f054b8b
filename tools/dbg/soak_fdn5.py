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
mode = sys.argv[1] if len(sys.argv) > 1 else "torchsum"
if mode == "torchsum":
    fn = lambda xx: (model(xx) * c).sum()
else:
    fn = lambda xx: ops.mean_square(model(xx))
gs = GraphedStep(fn, (x,), params, warmup=2)
sync_each = len(sys.argv) > 2
prev = None; changes = []
for i in range(3000):
    out = gs.replay()
    if sync_each or i % 50 == 49:
        torch.cuda.synchronize()
        v = out.item()
        if prev is not None and v != prev:
            changes.append((i, prev, v))
        prev = v
print(mode, "sync_each" if sync_each else "sync/50", "changes:", changes[:10], "n =", len(changes), "final", prev)
