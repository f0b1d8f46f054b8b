"""Objectives on the hot path's output: ops.mse / flamo_amd.optimize.mse_loss against torch's own evaluation of the
reference's lines (flamo/optimize/loss.py:66-103, examples/e7_biquad.py:82), and the fused ops.mean_square's guards."""
import pytest
import torch

from conftest import cc, check_close, relerr


def test_mse_loss_module_matches_reference_lines_on_host():
    """host tensors take the reference's own lines (sum over the last axis, nn.MSELoss against the squeezed target)"""
    from flamo_amd.optimize import mse_loss
    torch.manual_seed(0)
    y = torch.randn(3, 50, 4, dtype=torch.float64, requires_grad=True)
    t = torch.randn(3, 50, 1, dtype=torch.float64)
    crit = mse_loss(nfft=50)
    assert crit.name == "MSE" and crit.nfft == 50 and isinstance(crit.mse_loss, torch.nn.MSELoss)
    ref = torch.nn.functional.mse_loss(y.sum(-1), t.squeeze(-1))
    assert torch.equal(crit(y, t), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(3, 960, 8), (2, 1500, 4), (1, 777, 3), (2, 512, 16), (5, 100, 1)])
def test_mse_kernels_against_torch(gpu, dtype, shape):
    from flamo_amd import ops
    from flamo_amd.optimize import mse_loss
    torch.manual_seed(3)
    y = torch.randn(*shape, device=gpu, dtype=dtype)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for sum_last in (False, True):
        t = torch.randn(*(shape[:-1] + ((1,) if sum_last else shape[-1:])), device=gpu, dtype=dtype)
        y1 = y.clone().requires_grad_(True)
        y0 = y.double().clone().requires_grad_(True)
        if sum_last:
            loss = mse_loss(nfft=shape[1], device="cuda")(y1, t)
            ref = torch.nn.functional.mse_loss(y0.sum(-1), t.double().squeeze(-1))
        else:
            loss = ops.mse(y1, t)
            ref = torch.nn.functional.mse_loss(y0, t.double())
        (3.0 * loss).backward()
        (3.0 * ref).backward()
        tag = f"mse/{'x'.join(map(str, shape))}_{str(dtype)[-2:]}_{int(sum_last)}"
        check_close(tag + "/loss", loss.detach().double().reshape(1), ref.detach().reshape(1), tol)
        check_close(tag + "/grad", y1.grad.double(), y0.grad, tol)


@pytest.mark.gpu
def test_mse_on_the_fused_shell_against_oracle(gpu):
    """config-2 miniature trained against a target with the reference's criterion: loss and every gradient against the
    float64 oracle graph under torch's mse_loss"""
    from collections import OrderedDict
    from flamo_amd.optimize import mse_loss
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    torch.manual_seed(11)
    nfft, N, B = 24000, 4, 5
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft), dsp.iFFT(nfft))
    x = torch.randn(B, nfft, N, device=gpu)
    t = torch.randn(B, nfft, 1, device=gpu)
    loss = mse_loss(nfft=nfft, device="cuda")(model(x), t)
    loss.backward()
    W, G = (p.detach().cpu().double().requires_grad_(True) for p in (mat.param, geq.param))
    yo = O.config2_forward(x.cpu().double(), W, G, nfft)
    ref = torch.nn.functional.mse_loss(yo.sum(-1), t.cpu().double().squeeze(-1))
    gW, gG = torch.autograd.grad(ref, [W, G])
    check_close("mse_shell/loss", loss.detach().cpu().double().reshape(1), ref.detach().reshape(1), 1e-5)
    check_close("mse_shell/g_W", mat.param.grad.cpu().double(), gW, 1e-5)
    check_close("mse_shell/g_geq", geq.param.grad.cpu().double(), gG, 1e-4)


@pytest.mark.gpu
def test_mse_on_the_fused_shell_at_bench_size_against_oracle(gpu):
    """The `mse` legs of the bench line (bench.py::objective_legs) at EXACTLY their size -- nfft = 96000, 8x8, batch 32, float32,
    the walking kernels and the 8-channel / 32-item grid of the column passes (the sum((y - t)^2) partials of
    spec_cols_inv<..., true> and the c (y - t) load of the backward column pass) -- against the float64 oracle graph under
    torch's own mse_loss: the reference's criterion (flamo/optimize/loss.py:101-102, target (32, 96000, 1)) and nn.MSELoss on
    equal shapes (examples/e7_biquad.py:82).  Loss 1e-5, gW 1e-5, gG 1e-4 (the equaliser-gain gradient: DESIGN 5)."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.optimize import mse_loss
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    N, nfft, B = 8, 96000, 32
    torch.manual_seed(130709)
    W = torch.randn(N, N).double()
    G = torch.empty(12, N, N).uniform_(10 ** (-6 / 20), 10 ** (6 / 20)).double()
    x = torch.randn(B, nfft, N).double()
    t_sum = torch.randn(B, nfft, 1).double()
    t_full = torch.randn(B, nfft, N).double()
    Wl, Gl = W.clone().requires_grad_(True), G.clone().requires_grad_(True)
    yref = O.config2_forward(x, Wl, Gl, nfft)
    ref_sum = torch.nn.functional.mse_loss(yref.sum(-1), t_sum.squeeze(-1))
    ref_full = torch.nn.functional.mse_loss(yref, t_full)
    g_sum = torch.autograd.grad(ref_sum, [Wl, Gl], retain_graph=True)
    g_full = torch.autograd.grad(ref_full, [Wl, Gl])
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    mat.assign_value(W.to(gpu, torch.float32))
    geq.assign_value(G.to(gpu, torch.float32))
    model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft), dsp.iFFT(nfft))
    xg = x.to(gpu, torch.float32)
    crit = mse_loss(nfft=nfft, device="cuda")
    loss = crit(model(xg), t_sum.to(gpu, torch.float32))
    g = torch.autograd.grad(loss, [mat.param, geq.param])
    check_close("mse_shell_bench/sum/loss", loss.detach().cpu().double().reshape(1), ref_sum.detach().reshape(1), 1e-5)
    check_close("mse_shell_bench/sum/gW", g[0].cpu().double(), g_sum[0], 1e-5)
    check_close("mse_shell_bench/sum/gG", g[1].cpu().double(), g_sum[1], 1e-4)
    loss = ops.mse(model(xg), t_full.to(gpu, torch.float32))
    g = torch.autograd.grad(loss, [mat.param, geq.param])
    check_close("mse_shell_bench/full/loss", loss.detach().cpu().double().reshape(1), ref_full.detach().reshape(1), 1e-5)
    check_close("mse_shell_bench/full/gW", g[0].cpu().double(), g_full[0], 1e-5)
    check_close("mse_shell_bench/full/gG", g[1].cpu().double(), g_full[1], 1e-4)


@pytest.mark.gpu
def test_fused_mean_square_only_for_the_pipelines_own_output(gpu):
    """ops.mean_square takes its one-node form only for a y that still is the pipeline's differentiable output: a y produced
    under no_grad gives a loss without a graph (as (y ** 2).mean() would), a y with a hook or retain_grad differentiates
    through its own node (the hook fires, y.grad is filled)"""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    torch.manual_seed(2)
    nfft, N, B = 24000, 4, 4
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft), dsp.iFFT(nfft))
    x = torch.randn(B, nfft, N, device=gpu)
    with torch.no_grad():
        y0 = model(x)
    l0 = ops.mean_square(y0)                     # grad mode is on again here
    assert not l0.requires_grad and l0.grad_fn is None
    cc("l0_reshape_1", l0.reshape(1), (y0 ** 2).mean().reshape(1), 1e-06)
    # reference gradients: the fused node
    y = model(x)
    ops.mean_square(y).backward()
    gref = [p.grad.clone() for p in (mat.param, geq.param)]
    for p in (mat.param, geq.param):
        p.grad = None
    # a hook and retain_grad on y: both must be honoured
    y = model(x)
    seen = []
    y.register_hook(lambda g: seen.append(g.shape))
    y.retain_grad()
    loss = ops.mean_square(y)
    loss.backward()
    assert seen and y.grad is not None
    check_close("ms_guard/y_grad", y.grad, 2.0 * y.detach() / y.numel(), 1e-6)
    for p, g in zip((mat.param, geq.param), gref):
        cc("p_grad", p.grad, g, 1e-05)
    # torch.autograd.grad with respect to y itself works on the unfused form
    y = model(x)
    y.retain_grad()
    (gy,) = torch.autograd.grad(ops.mean_square(y), y)
    cc("gy", gy, 2.0 * y.detach() / y.numel(), 1e-06)


@pytest.mark.gpu
@pytest.mark.parametrize("nfft,N,B,with_x,dt", [(96000, 8, 5, False, torch.float32), (96000, 8, 3, True, torch.float32),
                                                (96000, 4, 2, False, torch.float32), (192000, 8, 2, False, torch.float32),
                                                (144000, 4, 2, True, torch.float32), (96000, 16, 2, False, torch.float32),
                                                (96000, 8, 5, False, torch.float64), (96000, 8, 2, True, torch.float64),
                                                (4096, 2, 3, True, torch.float32), (65536, 8, 5, False, torch.float32),
                                                (32000, 4, 2, True, torch.float64), (88200, 4, 2, False, torch.float32)])
def test_gradient_column_pass_inside_the_forward_pass(gpu, nfft, N, B, with_x, dt):
    """fl_spec_cols_inv_sumsq_grad_f32: once the output of an operator of some shape has gone into ops.mean_square and been
    differentiated, the next forward pass of that shape leaves the gradient's first column pass (fl_spec_cols_fwd of y) from the
    tiles of its inverse column pass -- y is not read back.  Same operations on the same values: output, loss and gradients EQUAL
    the two-pass form's in float32 (1e-13 in float64), the backward pass runs no column pass of its own (only the input gradient's inverse one), a second
    backward over the same graph is as good as the first (16 channels: the 32-wide tile and the in-place row kernel), small and
    odd-factor plans, and a plan that is not taken (441-point columns) keeps the separate pass."""
    from flamo_amd import _lib, ops
    torch.manual_seed(nfft % 977 + N)
    M = nfft // 2 + 1
    cdt = torch.complex64 if dt == torch.float32 else torch.complex128
    H0 = ops.permute_bins(torch.randn(M, N, N, device=gpu, dtype=cdt) / N ** 0.5, nfft)
    x0 = torch.randn(B, nfft, N, device=gpu, dtype=dt)
    taken = bool(ops._spec_fn("fl_spec_cols_inv_grad_supported", dt)(nfft, N))
    assert taken or nfft == 88200      # (every plan but the 441- / 800-point columns: 88200 = 2 x 441 x 100)

    def run(twice=False):
        H = H0.clone().requires_grad_(True)
        x = x0.clone().requires_grad_(with_x)
        ops.kernel_timer.reset(True)
        y = ops.spectral_apply(x, H, nfft)
        loss = ops.mean_square(y)
        g = torch.autograd.grad(3.0 * loss, [H] + ([x] if with_x else []), retain_graph=twice)
        if twice:
            g2 = torch.autograd.grad(3.0 * loss, [H] + ([x] if with_x else []))
            for a, b in zip(g, g2):
                assert torch.equal(a, b)
        torch.cuda.synchronize()
        used = [k for k, v in ops.kernel_timer.records.items() for _ in v]      # one entry per launch
        ops.kernel_timer.enabled = False
        return [y.detach(), loss.detach()] + [t.detach() for t in g], used

    ops._GRAD_COLS_SEEN.clear()
    try:
        ops.GRAD_COLS_IN_FORWARD = False
        plain, used0 = run()
        assert "spec_cols_inv+grad_cols" not in used0
        ops._GRAD_COLS_SEEN.clear()
        ops.GRAD_COLS_IN_FORWARD = True
        first, used1 = run()                      # nothing remembered yet: the two-pass form, and the shape is remembered
        assert "spec_cols_inv+grad_cols" not in used1
        fused, used2 = run(twice=True)
        assert ("spec_cols_inv+grad_cols" in used2) == taken, used2
        if taken:
            assert "spec_cols_fwd" in used0 and used2.count("spec_cols_fwd") == used0.count("spec_cols_fwd") - 1, (used0, used2)
        for i, (a, b, c) in enumerate(zip(plain, first, fused)):
            assert torch.equal(a, b)
            if dt == torch.float32:
                assert torch.equal(a, c), float((a - c).abs().max())
            else:       # (float64: the compiler contracts the tile's scaling into the transform's first additions -- rounding apart)
                check_close(f"grad_cols/{nfft}_{N}_{B}_64/{i}", c, a, 1e-13, max_tol=float("inf"))
        # evaluation without a backward pass afterwards: the value is right whichever launch ran
        with torch.no_grad():
            y = ops.spectral_apply(x0, H0, nfft)
            assert torch.equal(y, plain[0]) and torch.equal(ops.mean_square(y), plain[1])
        if taken:
            # another criterion on the same shape: its gradient arrives through the operator's own node, with the right value
            # (the column pass formed ahead is simply not used), and the shape stops forming it
            key = next(iter(ops._GRAD_COLS_SEEN))
            H = H0.clone().requires_grad_(True)
            ops.kernel_timer.reset(True)
            y = ops.spectral_apply(x0, H, nfft)
            (gH,) = torch.autograd.grad((y ** 2).mean() * 3.0, [H])
            torch.cuda.synchronize()
            assert "spec_cols_inv+grad_cols" in ops.kernel_timer.records and key not in ops._GRAD_COLS_SEEN
            ops.kernel_timer.enabled = False
            check_close(f"grad_cols/{nfft}_{N}_{B}_{str(dt)[-2:]}/other_criterion", gH, plain[2], 2e-6 if dt == torch.float32 else 1e-12,
                        max_tol=float("inf"))
    finally:
        ops.GRAD_COLS_IN_FORWARD = True
        ops._GRAD_COLS_SEEN.clear()
