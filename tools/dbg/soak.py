"""Stability check: many replays of the captured config-2 step -- bit-identical loss/gradients throughout,
no growth of device memory."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
torch.manual_seed(1)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
bad = 0
for i in range(3000):
    out = gs.replay()
    if i % 500 == 499:
        torch.cuda.synchronize()
        same = torch.equal(out, out0) and all(torch.equal(p.grad, g) for p, g in zip(params, g0))
        bad += (not same)
        print(i + 1, "replays: identical =", same, "mem delta", torch.cuda.memory_allocated() - m0)
print("OK" if bad == 0 else "MISMATCH")
