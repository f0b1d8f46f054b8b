import torch
dev = torch.device('cuda:0')
x = torch.randn(1 << 24, device=dev); y = torch.empty_like(x)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): torch.mul(x, 2.0, out=y)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        torch.mul(x, 2.0, out=y)
        e0.record()
        torch.mul(x, 3.0, out=y)
        e1.record()
        torch.mul(x, 4.0, out=y)
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        print("elapsed ms", e0.elapsed_time(e1))
except Exception as ex:
    print("FAILED:", type(ex).__name__, str(ex)[:300])
