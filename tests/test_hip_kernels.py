"""GPU unit tests of individual C-ABI kernels (layout conversion, per-bin product shapes)."""
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 49, 4), (1, 49, 16), (3, 1000, 8), (2, 8, 1000), (1, 301, 169), (2, 96000, 8),
                                   (1, 5, 3), (1, 257, 1), (1, 49, 2), (1, 3, 70000), (1, 64, 4096), (2, 4096, 64)])
def test_transpose_bit_exact(gpu, shape):
    from flamo_amd import ops
    B, R, C = shape
    for dt in (torch.float32, torch.complex64, torch.complex128):
        x = torch.randn(B, R, C, dtype=dt, device=gpu)
        y = ops._transpose(x, B, R, C).view(B, C, R)
        assert torch.equal(y, x.transpose(1, 2).contiguous())            # pure data movement: bit exact
        assert torch.equal(ops.to_planar(x).contiguous(), x)              # logical tensor unchanged


@pytest.mark.parametrize("No,Ni,K,B", [(1, 1, 1, 1), (3, 2, 1, 5), (8, 8, 1, 32), (5, 7, 3, 2), (16, 16, 16, 1),
                                       (32, 32, 1, 3), (1, 16, 1, 4), (16, 1, 1, 4)])
def test_mimo_shapes_against_einsum(gpu, No, Ni, K, B):
    from flamo_amd import ops
    torch.manual_seed(No * 100 + Ni)
    M = 777
    for cd, tol in ((torch.complex64, 2e-6), (torch.complex128, 1e-13)):
        X = torch.randn(B, M, Ni, K, dtype=cd, device=gpu) if K > 1 else torch.randn(B, M, Ni, dtype=cd, device=gpu)
        H = torch.randn(M, No, Ni, dtype=cd, device=gpu, requires_grad=True)
        W = torch.randn(No, Ni, dtype=cd, device=gpu, requires_grad=True)
        Xg = X.clone().requires_grad_(True)
        for Hm, pat in ((H, "fmn,bfn...->bfm..."), (W, "mn,bfn...->bfm...")):
            Y = ops.mimo(Hm, Xg)
            Yr = torch.einsum(pat, Hm.detach().cpu().to(torch.complex128), X.cpu().to(torch.complex128))
            assert relerr(Y.detach().cpu(), Yr) < tol
            C = torch.randn_like(Y)
            gH, gX = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [Hm, Xg])
            Hr = Hm.detach().cpu().to(torch.complex128).requires_grad_(True)
            Xr = X.cpu().to(torch.complex128).requires_grad_(True)
            Yr2 = torch.einsum(pat, Hr, Xr)
            gHr, gXr = torch.autograd.grad(torch.sum(torch.real(Yr2 * torch.conj(C.cpu().to(torch.complex128)))), [Hr, Xr])
            assert relerr(gH.cpu(), gHr) < tol * 5 and relerr(gX.cpu(), gXr) < tol * 5
        if No == Ni:
            h = torch.randn(M, Ni, dtype=cd, device=gpu)
            Yd = ops.mimo(h, X, diag=True)
            assert relerr(Yd.cpu(), torch.einsum("fn,bfn...->bfn...", h.cpu(), X.cpu())) < tol
