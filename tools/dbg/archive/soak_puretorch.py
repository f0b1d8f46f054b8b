"""Pure-torch check of the replay hazard seen in tools/dbg/soak_fdn*.py: a captured step of N elementwise kernels followed by
reductions; between replays one tiny eager tensor is created and filled.  No flamo code involved."""
import sys, torch
dev = torch.device('cuda:0')
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(1)
x = torch.randn(1, 192000, 1, device=dev); c = torch.randn(1, 192000, 1, device=dev)
w = torch.nn.Parameter(torch.randn(1, device=dev))
def fn(xx):
    t = xx * w
    for k in range(n_ops):
        t = t * 1.0001 + 0.001 * torch.sin(t)
    yc = t * c
    return yc.sum(), yc.abs().max(), yc
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        w.grad = None
        fn(x)[0].backward()
torch.cuda.current_stream().wait_stream(side)
w.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    s, m, yc = fn(x)
    (gw,) = torch.autograd.grad(s, [w])
res = []
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    res.append((s.item(), m.item(), yc.sum().item(), yc.abs().max().item(), gw.item()))
    j = torch.full((1,), 5.0, device=dev); del j
print(n_ops, res)
