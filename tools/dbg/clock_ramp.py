import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
torch.cuda.synchronize()
out = []
for blk in range(60):
    t0 = time.perf_counter()
    for _ in range(100):
        gs.replay()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 100 * 1e6, 1))
print(out)
time.sleep(2.0)
out = []
for blk in range(10):
    t0 = time.perf_counter()
    for _ in range(100):
        gs.replay()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 100 * 1e6, 1))
print("after 2 s idle:", out)
