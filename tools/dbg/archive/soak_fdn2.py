"""Which tensor of the replayed FDN step changes between replays, and by how much"""
import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd import ops
from flamo_amd.processor import system
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
for flags in ({},):
    for k in ("FDN_CORE", "FDN_DIAGONAL_IN_SOLVE"):
        setattr(system, k, flags.get(k, True))
    torch.manual_seed(1)
    model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
    x = torch.randn(1, 192000, 1, device=dev)
    c = torch.randn(1, 192000, 1, device=dev)
    gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
    out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
    torch.cuda.synchronize()
    ndiff = [0] * (1 + len(params)); worst = [0.0] * (1 + len(params))
    for i in range(2000):
        out = gs.replay()
        torch.cuda.synchronize()
        cur = [out] + [p.grad for p in params]
        for t, (a, b) in enumerate(zip(cur, [out0] + g0)):
            if not torch.equal(a, b):
                if t == 0 and ndiff[0] < 3:
                    print("   replay", i, "out", a.item(), "first", b.item(), flush=True)
                ndiff[t] += 1
                worst[t] = max(worst[t], ((a - b).abs().max() / b.abs().max()).item())
    print(flags, "replays differing from the first (out, in_gain, out_gain, mix, att):", ndiff, ["%.1e" % w for w in worst])
    del gs
