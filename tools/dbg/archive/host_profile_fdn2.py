"""cProfile of the eager 16-channel FDN step, cumulative time of this package's functions (forward side; the backward runs in
autograd's thread and shows as run_backward)"""
import os, sys, cProfile, pstats, torch, warnings, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from bench_fdn import build
dev = torch.device("cuda:0"); torch.manual_seed(130709)
model, params = build(dev, torch.float32, 16, 192000)
x = torch.randn(1, 192000, 1, device=dev); c = torch.randn(1, 192000, 1, device=dev)
def step():
    for p in params: p.grad = None
    (model(x) * c).sum().backward()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
buf = io.StringIO()
st = pstats.Stats(pr, stream=buf); st.sort_stats("cumtime").print_stats(60)
for line in buf.getvalue().splitlines():
    if "flamo_amd" in line or "run_backward" in line or "ncalls" in line:
        print(line[:150])
