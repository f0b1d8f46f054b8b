import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import relerr
from flamo_amd import functional as F, ops
gpu = torch.device('cuda:0')
torch.manual_seed(3)
nfft = 9600
cf, sc = F.eq_freqs(1)
des = F.GEQDesign(cf, sc)
gdb = torch.rand(12, 3, 2, dtype=torch.float64) * 24 - 12
b_geq, a_geq = des.sections(gdb)
th = torch.rand(14, 5, dtype=torch.float64) * 3.0 + 0.05
r = 0.5 + 0.49 * torch.rand(14, 5, dtype=torch.float64)
b_bp = torch.stack([torch.ones_like(th), torch.zeros_like(th), -torch.ones_like(th)]) * 0.3
a_bp = torch.stack([torch.ones_like(th), -2 * r * torch.cos(th), r * r])
for (b0, a0), gamma in (((b_geq, a_geq), 1.0), ((b_geq, a_geq), 10 ** (-30 / 20 / nfft)),
                        ((b_bp, a_bp), 1.0), ((b_bp[:, :3], a_bp[:, :3]), 0.9999)):
    grads = {}
    for mixed in (True, False):
        ops.SOS_BWD_MIXED = mixed
        b = b0.to(gpu).requires_grad_(True)
        a = a0.to(gpu).requires_grad_(True)
        H = ops.sos_response(b, a, gamma, nfft, dtype=torch.float32)
        torch.manual_seed(17)
        Cw = torch.randn(H.shape, dtype=torch.complex64, device=gpu)
        grads[mixed] = torch.autograd.grad(torch.sum(torch.real(H * torch.conj(Cw))), [b, a])
    # double reference of the same thing in torch (c128)
    for gm, gd in zip(grads[True], grads[False]):
        dm = gm[0] - 2 * gm[1] + gm[2]; dd = gd[0] - 2 * gd[1] + gd[2]
        print(gamma, relerr(gm.cpu(), gd.cpu()), relerr(dm.cpu(), dd.cpu()), (dm-dd).abs().max().item(), dd.abs().max().item(), gd.abs().max().item())
