"""Edge shapes of ops.fdn_core (and through it fl_solve_fdn_* / fl_solve_dud2_grads_* with side reductions) against plain
torch in complex128: one bin, a handful of bins, N = 1 / 2 / 3 / 5, both precisions.  (Empty batches are not accepted by the
per-bin operators: the C ABI rejects null pointers.)"""
import os, sys, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from flamo_amd import ops
dev = torch.device("cuda:0")
rel = lambda a, b: ((a.cpu().to(torch.complex128) - b).norm() / (b.norm() + 1e-300)).item()
bad = 0
for cd in (torch.complex128, torch.complex64):
    tol = 1e-10 if cd == torch.complex128 else 5e-5
    for N in (1, 2, 3, 5):
        for M in (1, 2, 7, 65):
            for B in (1, 2):
                torch.manual_seed(N * 100 + M)
                U = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
                l = 0.9 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
                l2 = 0.95 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
                b = torch.randn(N, 1, dtype=torch.float64); c = torch.randn(1, N, dtype=torch.float64)
                X = torch.randn(B, M, 1, dtype=torch.complex128); C = torch.randn(B, M, 1, dtype=torch.complex128)
                ins = [t.clone().requires_grad_(True) for t in (b, c, l, U, X)]
                A = torch.eye(N, dtype=torch.complex128) - (ins[2] * l2).unsqueeze(-1) * ins[3]
                R = l2.unsqueeze(0).unsqueeze(-1) * (ins[0].to(torch.complex128) @ ins[4].unsqueeze(-1).transpose(-1, -2).transpose(-1, -2))  # (B, M, N, 1)
                R = l2.unsqueeze(0) * (ins[4] * ins[0].to(torch.complex128).view(1, 1, N))
                OUT = torch.linalg.solve(A.unsqueeze(0), R.unsqueeze(-1)).squeeze(-1)
                y = (OUT * ins[1].to(torch.complex128).view(1, 1, N)).sum(-1, keepdim=True)
                gref = torch.autograd.grad((y * C.conj()).real.sum(), ins)
                rd = torch.float64 if cd == torch.complex128 else torch.float32
                dins = [b.to(dev, rd).requires_grad_(True), c.to(dev, rd).requires_grad_(True), l.to(dev, cd).requires_grad_(True),
                        U.to(dev, cd).requires_grad_(True), X.to(dev, cd).requires_grad_(True)]
                yd = ops.fdn_core(dins[0], dins[1], dins[2], l2.to(dev, cd), dins[3], None, dins[4])
                gd = torch.autograd.grad((yd * C.to(dev, cd).conj()).real.sum(), dins)
                errs = [rel(yd.detach(), y.detach())] + [rel(a.to(torch.complex128) if not a.is_complex() else a, (g.to(torch.complex128))) for a, g in zip(gd, gref)]
                if max(errs) > tol:
                    bad += 1
                    print("fdn_core", cd, "N", N, "M", M, "B", B, ["%.1e" % e for e in errs])
print("edge cases:", "OK" if bad == 0 else f"{bad} problems")
