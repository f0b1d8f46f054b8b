import sys, time, torch, warnings
sys.path.insert(0, '.')
warnings.simplefilter("ignore")
import bench
dev = torch.device("cuda:0")
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(32, 96000, 8, device=dev)
def step():
    for p in params: p.grad = None
    y = model(x); loss = (y ** 2).mean(); loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
