import sys, torch, time
sys.path.insert(0, '.')
from flamo_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
nfft, C, S = 96000, 64, 12
torch.manual_seed(0)
b = torch.randn(3, S, 8, 8, dtype=torch.float64, device=dev, requires_grad=True)
a = (torch.randn(3, S, 8, 8, dtype=torch.float64, device=dev) + torch.tensor([3., 0, 0], dtype=torch.float64, device=dev).view(3,1,1,1)).requires_grad_(True)
Cg = torch.randn(nfft // 2 + 1, 8, 8, dtype=torch.complex64, device=dev)
ref = None
for sch in (6412, 1612, 812, 412, 1606, 806, 3206):
    L.fl_debug_set_sos_chunk(sch)
    H = ops.sos_response(b, a, 0.9999, nfft)
    for _ in range(3):
        g = torch.autograd.grad(torch.sum(torch.real(H * torch.conj(Cg))), [b, a], retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g = torch.autograd.grad(torch.sum(torch.real(H * torch.conj(Cg))), [b, a], retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = g
    err = max(((g[0]-ref[0]).norm()/ref[0].norm()).item(), ((g[1]-ref[1]).norm()/ref[1].norm()).item())
    print("chunk", sch, "ms per bwd (incl. torch sum ops)", e0.elapsed_time(e1) / 10, "diff vs chunk12", err)
