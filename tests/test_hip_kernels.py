"""GPU unit tests of individual C-ABI kernels (layout conversion, per-bin product shapes)."""
import pytest
import torch

from conftest import cc, check_close, relerr

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
                                       (32, 32, 1, 3), (1, 16, 1, 4), (16, 1, 1, 4),
                                       # matrix-valued signals on the MFMA kernels: full, ragged rows/columns/depth,
                                       # batched columns, both tile shapes (<= 16 and > 16 rows or columns)
                                       (32, 32, 32, 1), (20, 9, 11, 2), (16, 24, 8, 1), (33, 17, 5, 4), (40, 32, 36, 1)])
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
            cc("Y_detach_cpu", Y.detach().cpu(), Yr, tol, max_tol=float("inf"))
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
            cc("Yd_cpu", Yd.cpu(), torch.einsum("fn,bfn...->bfn...", h.cpu(), X.cpu()), tol, max_tol=float("inf"))


def test_geq_design_kernel_matches_host_formulas(gpu):
    """Fused GEQ design kernel vs the vectorised host restatement (bit-exact up to pow() ulps) and
    its analytic backward vs autograd."""
    from flamo_amd import functional as F, ops
    torch.manual_seed(5)
    cf, sc = F.eq_freqs(1)
    des = F.GEQDesign(cf, sc)
    gdb = (torch.rand(12, 8, 8, dtype=torch.float64) * 24 - 12).requires_grad_(True)
    b_ref, a_ref = des.sections(gdb)
    gd = gdb.detach().to(gpu).requires_grad_(True)
    b, a = ops.geq_sections(gd, des.device_consts(gpu))
    assert b.shape == b_ref.shape and b.dtype == torch.float64
    # float32-representable outputs, identical to the host evaluation except where a 1-ulp pow()
    # difference flips a float32 rounding (rare)
    assert torch.equal(b.float().double(), b) and torch.equal(a.float().double(), a)
    mism = ((b.cpu() != b_ref.double()) | (a.cpu() != a_ref.double())).float().mean().item()
    assert mism < 0.01
    assert relerr(b.cpu(), b_ref.double()) < 2e-7 and relerr(a.cpu(), a_ref.double()) < 2e-7
    cb, ca = torch.randn_like(b_ref, dtype=torch.float64), torch.randn_like(a_ref, dtype=torch.float64)
    (g_ref,) = torch.autograd.grad((b_ref.double() * cb).sum() + (a_ref.double() * ca).sum(), [gdb])
    (g,) = torch.autograd.grad((b * cb.to(gpu)).sum() + (a * ca.to(gpu)).sum(), [gd])
    cc("g_cpu", g.cpu(), g_ref, 1e-6, max_tol=float("inf"))


def test_series_fusion_matches_module_by_module(gpu):
    """Folding adjacent per-bin modules into one pass (system.FUSE_SERIES) changes nothing but the
    floating-point association: outputs and every gradient agree with the unfused evaluation."""
    from collections import OrderedDict
    from flamo_amd.processor import dsp, system
    torch.manual_seed(3)
    nfft, N, B = 480, 4, 6
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 5e-6)):
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
        mods = OrderedDict(
            g=dsp.parallelGain(size=(N,), requires_grad=True, **kw),
            d=dsp.parallelDelay(size=(N,), max_len=50, isint=True, **kw),
            mix=dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw),
            eq=dsp.GEQ(size=(3, N), requires_grad=True, **kw),
            fir=dsp.parallelFilter(size=(5, 3), requires_grad=True, **kw),
            out=dsp.Gain(size=(2, 3), requires_grad=True, **kw))
        core = system.Series(mods)
        model = system.Shell(core, dsp.FFT(nfft, dtype=dt), dsp.iFFTAntiAlias(nfft, alias_decay_db=30.0, device=gpu, dtype=dt))
        x = torch.randn(B, nfft, N, device=gpu, dtype=dt, requires_grad=True)
        params = [m.param for m in mods.values() if m.param.requires_grad]
        res = {}
        for fuse in (True, False):
            system.FUSE_SERIES = fuse
            try:
                y = model(x)
                res[fuse] = [y.detach()] + list(torch.autograd.grad((y ** 2).mean(), [x] + params))
            finally:
                system.FUSE_SERIES = True
        for a, b in zip(res[True], res[False]):
            cc("a", a, b, tol, max_tol=float("inf"))
        # ext_param routing through a fused run still logs the external values into the module
        ext = {"mix": torch.randn(N, N, device=gpu, dtype=dt)}
        y1 = core(torch.randn(B, nfft // 2 + 1, N, device=gpu, dtype=CDT[dt]), ext)
        assert torch.equal(mods["mix"].param.detach(), ext["mix"]) and y1.shape[2] == 2
        with pytest.raises(ValueError):
            core(torch.zeros(B, nfft // 2 + 1, N + 1, device=gpu, dtype=CDT[dt]))


CDT = {torch.float64: torch.complex128, torch.float32: torch.complex64}


def test_bin_sharded_core_equals_full_core(gpu):
    """Bins are independent: evaluating the core on each rank's bin range (responses generated
    for that range only) and concatenating reproduces the unsharded result bit for bit in the
    per-bin kernels (same arithmetic per bin)."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.dist import shard_bins
    from flamo_amd.processor import dsp, system
    torch.manual_seed(11)
    nfft, N, world = 960, 6, 3
    kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=torch.float64)
    dl = dsp.parallelDelay(size=(N,), max_len=170, isint=True, **kw)
    fb = system.Series(OrderedDict(mix=dsp.Matrix(size=(N, N), matrix_type="orthogonal", **kw),
                                   att=dsp.parallelGEQ(size=(N,), **kw)))
    core = system.Series(OrderedDict(
        input_gain=dsp.Gain(size=(N, 2), **kw), fir=dsp.Filter(size=(7, N, N), **kw),
        frac=dsp.parallelDelay(size=(N,), max_len=20, isint=False, **kw),
        feedback_loop=system.Recursion(fF=dl, fB=fb), output_gain=dsp.Gain(size=(2, N), **kw)))
    with torch.no_grad():
        fb[1].param.mul_(0.5)
    M = nfft // 2 + 1
    X = torch.randn(5, M, 2, dtype=torch.complex128, device=gpu)
    with torch.no_grad():
        full = core(X)
        parts = []
        for r in range(world):
            bin0, ml = shard_bins(M, r, world)
            ops.set_bin_shard(bin0, ml)
            try:
                parts.append(core(X[:, bin0:bin0 + ml]))
            finally:
                ops.set_bin_shard(0, None)
    cc("torch_cat_parts_dim_1", torch.cat(parts, dim=1), full, 1e-14, max_tol=float("inf"))


def test_factored_fdn_loop_matches_generic_recursion(gpu):
    """Recursion with P = diag(l) U diag(r) kept factored (ops.solve_dud) against the generic path
    that materialises P = F(B(I)) and calls ops.solve: outputs and all gradients, vector and
    matrix right-hand sides, diagonal factors on both sides of the mixing matrix."""
    from collections import OrderedDict
    from flamo_amd.processor import dsp, system
    torch.manual_seed(17)
    nfft, N = 960, 6
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 2e-5)):
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
        dl = dsp.parallelDelay(size=(N,), max_len=170, isint=True, **kw)
        pre = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
        mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
        att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
        with torch.no_grad():
            att.param.mul_(0.5)
            pre.param.copy_(torch.rand(N, device=gpu, dtype=dt) * 0.5 + 0.4)
        fb = system.Series(OrderedDict(pre=pre, mix=mix, att=att))
        rec = system.Recursion(fF=dl, fB=fb)
        params = [pre.param, mix.param, att.param]
        M = nfft // 2 + 1
        for shape in ((3, M, N), (1, M, N, N)):
            X = torch.randn(*shape, dtype=CDT[dt], device=gpu, requires_grad=True)
            C = torch.randn(*shape, dtype=CDT[dt], device=gpu)
            res = {}
            for fuse in (True, False):
                system.FUSE_SERIES = fuse
                try:
                    Y = rec(X)
                    res[fuse] = [Y.detach()] + list(torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [X] + params))
                finally:
                    system.FUSE_SERIES = True
            for a, b in zip(res[True], res[False]):
                cc("a", a, b, tol, max_tol=float("inf"))


def test_composed_loop_matches_identity_recursion(gpu):
    """Loops that are not diag-U-diag -- a per-bin full matrix (Delay(N,N), FIR Filter) or two full matrices in
    the loop -- take P from the composition of the modules' responses; against the reference's way (push the
    identity through feedback and feedforward): outputs and every gradient, vector and matrix inputs."""
    from collections import OrderedDict
    from flamo_amd.processor import dsp, system
    torch.manual_seed(19)
    nfft = 480
    M = nfft // 2 + 1
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 3e-5)):
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
        for N, case in ((5, "delay"), (4, "fir"), (17, "two_matrices")):
            gain = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
            mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
            with torch.no_grad():
                gain.param.copy_(torch.rand(N, device=gpu, dtype=dt) * 0.4 / N ** 0.5 + 0.05)
            if case == "delay":
                full = dsp.Delay(size=(N, N), max_len=60, isint=True, **kw)
            elif case == "fir":
                full = dsp.Filter(size=(7, N, N), requires_grad=True, **kw)
                with torch.no_grad():
                    full.param.mul_(0.3)
            else:
                full = dsp.Matrix(size=(N, N), matrix_type="random", requires_grad=True, **kw)
                with torch.no_grad():
                    full.param.mul_(0.5 / N ** 0.5)
            rec = system.Recursion(fF=system.Series(OrderedDict(d=full, g=gain)), fB=mix)
            params = [p for p in rec.parameters() if p.requires_grad]
            for shape in ((2, M, N), (1, M, N, N)):
                X = torch.randn(*shape, dtype=CDT[dt], device=gpu, requires_grad=True)
                C = torch.randn(*shape, dtype=CDT[dt], device=gpu)
                res = {}
                for fuse in (True, False):
                    system.FUSE_SERIES = fuse
                    try:
                        Y = rec(X)
                        res[fuse] = [Y.detach()] + list(torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [X] + params))
                    finally:
                        system.FUSE_SERIES = True
                for a, b in zip(res[True], res[False]):
                    cc("a", a, b, tol, max_tol=float("inf"))


def test_graphed_step_matches_eager(gpu):
    """A forward+backward step replayed from a HIP graph gives the same loss and gradients as eager."""
    from collections import OrderedDict
    from flamo_amd.graph import GraphedStep
    from flamo_amd.processor import dsp, system
    torch.manual_seed(23)
    nfft, N = 960, 4
    kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=torch.float32)
    dl = dsp.parallelDelay(size=(N,), max_len=150, isint=True, **kw)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
    with torch.no_grad():
        att.param.mul_(0.5)
    core = system.Series(OrderedDict(input_gain=dsp.Gain(size=(N, 1), requires_grad=True, **kw),
                                     feedback_loop=system.Recursion(fF=dl, fB=system.Series(OrderedDict(m=mix, a=att))),
                                     output_gain=dsp.Gain(size=(1, N), requires_grad=True, **kw)))
    model = system.Shell(core, dsp.FFT(nfft), dsp.iFFTAntiAlias(nfft, alias_decay_db=30.0, device=gpu))
    params = [p for p in model.parameters() if p.requires_grad]
    x = torch.randn(3, nfft, 1, device=gpu)
    loss_fn = lambda xx: (model(xx) ** 2).mean()  # noqa: E731
    gs = GraphedStep(loss_fn, (x,), params)
    x2 = torch.randn(3, nfft, 1, device=gpu)
    lg = gs(x2).clone()
    gg = [g.clone() for g in gs.grads]
    for p in params:
        p.grad = None
    le = loss_fn(x2)
    le.backward()
    cc("lg", lg, le.detach(), 1e-6, max_tol=float("inf"))
    for a, p in zip(gg, params):
        cc("a", a, p.grad, 1e-5, max_tol=float("inf"))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.float64, 1e-13)])
def test_mean_square_matches_torch_in_every_layout(gpu, dt, tol):
    """ops.mean_square == (y ** 2).mean() with gradient 2 y g / count, for contiguous tensors, the
    signal-planar views the transforms return (dense and padded rows) and odd sizes (scalar path)."""
    from flamo_amd import ops
    torch.manual_seed(11)
    cases = []
    cases.append(torch.randn(3, 1000, 8, dtype=dt, device=gpu))                               # contiguous
    cases.append(torch.randn(2, 8, 96000, dtype=dt, device=gpu).movedim(-1, 1))               # planar, dense
    cases.append(torch.randn(2, 5, 1024, dtype=dt, device=gpu)[..., :1001].movedim(-1, 1))    # planar, padded
    cases.append(torch.randn(1, 3, 1030, dtype=dt, device=gpu)[..., 1:1028].movedim(-1, 1))   # unaligned base
    cases.append(torch.randn(7, 13, dtype=dt, device=gpu))                                    # tiny
    cases.append(torch.randn(4, 600, 6, dtype=dt, device=gpu)[:, ::2])                        # generic strides
    for y0 in cases:
        y = y0.clone(memory_format=torch.preserve_format) if y0.is_contiguous() else y0
        y.requires_grad_(True)
        loss = ops.mean_square(y)
        w = torch.tensor(0.75, dtype=dt, device=gpu)
        (g,) = torch.autograd.grad(loss * w, [y])
        yr = y0.detach().cpu().double().requires_grad_(True)
        lr = (yr ** 2).mean()
        (gr,) = torch.autograd.grad(lr * 0.75, [yr])
        assert abs(loss.item() - lr.item()) <= tol * abs(lr.item())
        assert g.shape == y.shape and relerr(g.cpu(), gr) < tol
    # repeated launches reuse the scratch: same bits every time
    y = cases[1]
    vals = {ops.mean_square(y).item() for _ in range(5)}
    assert len(vals) == 1


def test_sos_backward_mixed_precision_matches_double(gpu):
    """float32 modules: the mixed-precision cascade backward (double section values, float
    quotients, d-basis sums) against the all-double kernel -- graphic-EQ sections with a 31 Hz
    band, band-pass sections whose numerator vanishes at DC and Nyquist, with and without the
    anti-alias radius, S below and above one register chunk."""
    from flamo_amd import functional as F, ops
    torch.manual_seed(3)
    nfft = 9600
    M = nfft // 2 + 1
    cf, sc = F.eq_freqs(1)
    des = F.GEQDesign(cf, sc)
    gdb = torch.rand(12, 3, 2, dtype=torch.float64) * 24 - 12
    b_geq, a_geq = (t.double() for t in des.sections(gdb))   # float32-rounded values, held in double
    th = torch.rand(14, 5, dtype=torch.float64) * 3.0 + 0.05
    r = 0.5 + 0.49 * torch.rand(14, 5, dtype=torch.float64)
    b_bp = torch.stack([torch.ones_like(th), torch.zeros_like(th), -torch.ones_like(th)]) * 0.3
    a_bp = torch.stack([torch.ones_like(th), -2 * r * torch.cos(th), r * r])
    for (b0, a0), gamma in (((b_geq, a_geq), 1.0), ((b_geq, a_geq), 10 ** (-30 / 20 / nfft)),
                            ((b_bp, a_bp), 1.0), ((b_bp[:, :3], a_bp[:, :3]), 0.9999)):
        grads = {}
        for mixed in (True, False):
            ops.SOS_BWD_MIXED = mixed
            try:
                b = b0.to(gpu).requires_grad_(True)
                a = a0.to(gpu).requires_grad_(True)
                H = ops.sos_response(b, a, gamma, nfft, dtype=torch.float32)
                torch.manual_seed(17)
                Cw = torch.randn(H.shape, dtype=torch.complex64, device=gpu)
                grads[mixed] = torch.autograd.grad(torch.sum(torch.real(H * torch.conj(Cw))), [b, a])
            finally:
                ops.SOS_BWD_MIXED = True
        assert H.shape[0] == M
        for gm, gd in zip(grads[True], grads[False]):
            cc("gm_cpu", gm.cpu(), gd.cpu(), 5e-6, max_tol=float("inf"))
            # the second differences the parameter maps take downstream must survive as well
            dm = gm[0] - 2 * gm[1] + gm[2]
            dd = gd[0] - 2 * gd[1] + gd[2]
            cc("dm_cpu", dm.cpu(), dd.cpu(), 5e-5, max_tol=float("inf"))


@pytest.mark.parametrize("N", [1, 4, 16, 23, 32, 64])
def test_matrix_exp_kernel_matches_torch(gpu, N):
    """ops.matrix_exp (one launch each way, fixed schedule) vs torch.matrix_exp on the CPU in float64,
    values and gradients, plain and through the skew map of the orthogonal Matrix."""
    from flamo_amd import ops
    torch.manual_seed(N)
    for dt, tol0 in ((torch.float64, 2e-12), (torch.float32, 3e-6)):
        for skew, amp in ((True, 1.0), (False, 0.3), (True, 25.0), (True, 1e-3)):   # |A|_1 from ~1e-2 to several hundred:
            # the kernel picks its squaring count from the norm it measures itself
            tol = tol0 * (100.0 if amp > 1 else 1.0)
            X0 = torch.randn(N, N, dtype=dt) * amp
            Xr = X0.double().requires_grad_(True)
            A = (torch.triu(Xr, 1) - torch.triu(Xr, 1).mT) if skew else Xr
            Er = torch.matrix_exp(A)
            Cw = torch.randn(N, N, dtype=torch.float64)
            (gr,) = torch.autograd.grad((Er * Cw).sum(), [Xr])
            X = X0.to(gpu).requires_grad_(True)
            E = ops.matrix_exp(X, skew=skew)
            (g,) = torch.autograd.grad((E * Cw.to(gpu, dt)).sum(), [X])
            assert E.dtype == dt and relerr(E.detach().cpu(), Er.detach()) < tol
            cc("g_cpu", g.cpu(), gr, tol * 5, max_tol=float("inf"))
            if skew:   # orthogonal to working precision
                I = torch.eye(N, dtype=torch.float64)
                assert (E.detach().cpu().double() @ E.detach().cpu().double().mT - I).abs().max() < tol * 10


def test_matrix_exp_16_on_matrix_cores_matches_lds_kernels(gpu):
    """N = 16 (the feedback delay networks' mixing matrix, e8_fdn.py / e8_colorless_fdn.py) runs as one wavefront on
    v_mfma_f64_16x16x4_f64 with every matrix and its transpose in registers: against the LDS kernels of the same schedule -- values,
    gradients, and each backward on the OTHER forward's stash (the layout is shared) -- plain and through the skew map, all three
    output forms."""
    from flamo_amd import _lib, ops
    L = _lib.lib()
    torch.manual_seed(16)
    prev = L.fl_debug_set_expm_mfma(-1)
    try:
        for dt, tol in ((torch.float64, 1e-13), (torch.float32, 1e-6)):
            for skew, amp in ((True, 1.0), (False, 0.3), (True, 25.0), (False, 3.0)):
                X0 = (torch.randn(16, 16, dtype=dt) * amp).to(gpu)
                Cw = torch.randn(16, 16, dtype=dt, device=gpu)
                res = {}
                for fwd_on in (1, 0):
                    for bwd_on in (1, 0):
                        X = X0.clone().requires_grad_(True)
                        L.fl_debug_set_expm_mfma(fwd_on)
                        E, Ec = ops.matrix_exp_both(X, skew=skew)
                        L.fl_debug_set_expm_mfma(bwd_on)
                        (g,) = torch.autograd.grad((E * Cw).sum() + (Ec.real * Cw.mT).sum(), [X])
                        res[(fwd_on, bwd_on)] = (E.detach(), Ec.detach(), g)
                E0, Ec0, g0 = res[(0, 0)]
                assert torch.equal(Ec0.real, E0) and float(Ec0.imag.abs().max()) == 0.0
                for key, (E, Ec, g) in res.items():
                    tag = f"expm16/{str(dt)[6:]}_{int(skew)}_{amp}_{key[0]}{key[1]}"
                    assert relerr(E, E0) < tol and relerr(Ec.real, E0) < tol, (tag, relerr(E, E0))
                    assert relerr(g, g0) < 20 * tol, (tag, relerr(g, g0))
    finally:
        L.fl_debug_set_expm_mfma(prev)


def test_response_overlap_is_transparent(gpu):
    """Building the folded response on the side stream (Shell + Series) changes nothing but timing:
    eager outputs and gradients are bit-identical with the overlap on and off, and a captured step
    replays to the same numbers."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.graph import GraphedStep
    from flamo_amd.processor import dsp, system
    torch.manual_seed(21)
    nfft, N = 4800, 4
    kw = dict(nfft=nfft, device=gpu, dtype=torch.float32, requires_grad=True)
    core = system.Series(OrderedDict(mix=dsp.Matrix(size=(N, N), matrix_type="orthogonal", **kw), eq=dsp.GEQ(size=(N, N), **kw),
                                     d=dsp.parallelDelay(size=(N,), max_len=300, isint=True, nfft=nfft, device=gpu,
                                                         dtype=torch.float32)))
    model = system.Shell(core, dsp.FFT(nfft), dsp.iFFT(nfft))
    params = [p for p in model.parameters() if p.requires_grad]
    x = torch.randn(5, nfft, N, device=gpu)
    res = {}
    for on in (True, False):
        system.OVERLAP_RESPONSES = on
        try:
            loss = ops.mean_square(model(x))
            res[on] = (loss.detach().clone(), [g.clone() for g in torch.autograd.grad(loss, params)])
        finally:
            system.OVERLAP_RESPONSES = True
    assert torch.equal(res[True][0], res[False][0])
    for ga, gb in zip(res[True][1], res[False][1]):
        assert torch.equal(ga, gb)
    gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
    for _ in range(2):
        out = gs.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, res[True][0])
    for p, gb in zip(params, res[True][1]):
        assert torch.equal(p.grad, gb)


@pytest.mark.parametrize("N", [2, 3, 5, 8, 16, 32])
def test_eigvals_kernel_matches_lapack(gpu, N):
    """ops.eigvals (Hessenberg + shifted QR, one wavefront per matrix) against torch.linalg.eigvals on the
    CPU in float64: spectra as sets (matched greedily), an order-independent loss and its gradient."""
    from flamo_amd import ops
    torch.manual_seed(100 + N)
    for cd, tol in ((torch.complex128, 1e-9), (torch.complex64, 2e-4)):
        A0 = torch.randn(37, N, N, dtype=torch.complex128) / N ** 0.5
        A0[3] = torch.diag_embed(torch.randn(N, dtype=torch.complex128))               # already triangular
        A0[5] = torch.triu(A0[5])                                                       # upper triangular
        A0[7] = A0[7] + A0[7].mH                                                        # Hermitian
        Ar = A0.clone().requires_grad_(True)
        lr = torch.linalg.eigvals(Ar)
        loss_r = ((lr.abs() - 0.7) ** 2).sum() + (lr.real ** 3).sum() * 0.1
        (gr,) = torch.autograd.grad(loss_r, [Ar])
        A = A0.to(gpu, cd).requires_grad_(True)
        l = ops.eigvals(A)
        assert l.shape == (37, N)
        loss = ((l.abs() - 0.7) ** 2).sum() + (l.real ** 3).sum() * 0.1
        (g,) = torch.autograd.grad(loss, [A])
        # match every computed eigenvalue to the nearest reference one
        lc, lrc = l.detach().cpu().to(torch.complex128), lr.detach()
        d = (lc.unsqueeze(-1) - lrc.unsqueeze(-2)).abs()
        assert d.min(dim=-1).values.max() < tol * 10 and d.min(dim=-2).values.max() < tol * 10
        assert abs(loss.item() - loss_r.item()) < tol * 10 * abs(loss_r.item())
        good = torch.ones(37, dtype=torch.bool)
        good[3] = good[5] = False          # gradients of exactly triangular inputs are fine too, but skip the degenerate ones
        cc("g_cpu_good", g.cpu()[good], gr[good], (1e-7 if cd == torch.complex128 else 2e-3), max_tol=float("inf"))
        assert int(ops.eigvals_info().abs().max()) == 0
    # batched leading dims and the functional wrapper
    from flamo_amd import functional as F
    B4 = torch.randn(2, 5, 4, 4, dtype=torch.complex64, device=gpu)
    l4 = F.get_eigenvalues(B4)
    assert l4.shape == (2, 5, 4)
    tr = torch.diagonal(B4, dim1=-2, dim2=-1).sum(-1)
    cc("l4_sum_1_cpu", l4.sum(-1).cpu(), tr.cpu(), 1e-5, max_tol=float("inf"))


def test_eigvals_large_and_degenerate(gpu):
    """N = 48 and 64 (LDS above the default 64 KB), the identity, a Jordan block and a nilpotent matrix."""
    from flamo_amd import ops
    torch.manual_seed(9)
    for N, cd, tol in ((48, torch.complex64, 5e-4), (64, torch.complex64, 1e-3), (40, torch.complex128, 1e-9)):
        A0 = torch.randn(5, N, N, dtype=torch.complex128) / N ** 0.5
        lr = torch.linalg.eigvals(A0)
        with torch.no_grad():
            l = ops.eigvals(A0.to(gpu, cd)).cpu().to(torch.complex128)
        d = (l.unsqueeze(-1) - lr.unsqueeze(-2)).abs()
        assert d.min(dim=-1).values.max() < tol and d.min(dim=-2).values.max() < tol
        assert int(ops.eigvals_info().abs().max()) == 0
    N = 6
    eye = torch.eye(N, dtype=torch.complex128)
    jordan = 0.5 * eye + torch.diag(torch.ones(N - 1, dtype=torch.complex128), 1)
    nil = torch.diag(torch.ones(N - 1, dtype=torch.complex128), 1)
    with torch.no_grad():
        l = ops.eigvals(torch.stack([eye, jordan, nil]).to(gpu)).cpu()
    assert (l[0] - 1).abs().max() < 1e-12 and (l[1] - 0.5).abs().max() < 1e-6 and l[2].abs().max() < 1e-6


def test_mimo_mfma_matches_lane_kernels(gpu):
    """The MFMA kernels (matrix-valued signals) against the lane-per-bin kernels on the same data, bins not a
    multiple of the 32/64-bin workgroup tiles: forward, adjoint and the per-bin matrix gradient."""
    from flamo_amd import _lib, ops
    L = _lib.lib()
    torch.manual_seed(11)
    for No, Ni, B, K, M in ((32, 32, 1, 32, 1037), (24, 16, 2, 10, 1037), (16, 8, 1, 16, 1037), (32, 32, 1, 32, 5), (17, 9, 1, 9, 1)):
        H = torch.randn(M, No, Ni, dtype=torch.complex64, device=gpu, requires_grad=True)
        X = torch.randn(B, M, Ni, K, dtype=torch.complex64, device=gpu, requires_grad=True)
        C = torch.randn(B, M, No, K, dtype=torch.complex64, device=gpu)
        out = {}
        try:
            for v in (0, -1, -14):       # default (LDS-staged, wide tile where it applies), lane kernels, 16x16 tile
                L.fl_debug_set_mimo_variant(v, 0)
                Y = ops.mimo(H, X)
                out[v] = (Y.detach(),) + torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [H, X])
        finally:
            L.fl_debug_set_mimo_variant(0, 0)
        for v in (0, -14):
            for got, ref in zip(out[v], out[-1]):
                cc("got_cpu", got.cpu(), ref.cpu(), 1e-6, max_tol=float("inf"))
        # frequency-independent matrix: its gradient is the per-bin outer product summed over bins in the accumulators
        W = torch.randn(No, Ni, dtype=torch.complex64, device=gpu, requires_grad=True)
        gw = {}
        try:
            for v in (0, -1):
                L.fl_debug_set_mimo_variant(v, 0)
                Y = ops.mimo(W, X)
                gw[v] = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [W])[0]
        finally:
            L.fl_debug_set_mimo_variant(0, 0)
        ref = torch.einsum("bfmk,bfnk->mn", C.cpu().to(torch.complex128), X.detach().cpu().to(torch.complex128).conj())
        assert relerr(gw[0].cpu(), ref) < 2e-6 and relerr(gw[-1].cpu(), ref) < 2e-6


@pytest.mark.parametrize("N", [3, 5, 8, 9, 13, 16, 17, 24, 32])
def test_factored_solve_all_sizes_and_row_exchanges(gpu, N):
    """(I - diag(l) U diag(r))^-1 R through fl_solve_dud_* for every kernel shape (4/8/16 lanes per bin, two rows per
    lane above 16), with loops that never exchange rows (damped orthogonal) and loops that must (U = 3 x a
    permutation with a zero diagonal + noise: the diagonal of I - P is far below the column maximum at every
    step), forward and gradients (the adjoint solve), float32 and float64, against LAPACK in float64."""
    from flamo_amd import _lib, ops
    torch.manual_seed(100 + N)
    M = 203
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 2e-5)):
        for kind in ("orthogonal", "permutation"):
            if kind == "orthogonal":
                U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
                l64 = 0.97 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
            else:
                perm = torch.roll(torch.arange(N), 1)
                U64 = (3.0 * torch.eye(N, dtype=torch.float64)[perm] + 0.05 * torch.randn(N, N, dtype=torch.float64)).to(torch.complex128)
                l64 = torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
            r64 = (0.8 + 0.2 * torch.rand(N, dtype=torch.float64)).to(torch.complex128)
            R64 = torch.randn(2, M, N, dtype=torch.complex128)
            C64 = torch.randn(2, M, N, dtype=torch.complex128)
            ref_in = [t.clone().requires_grad_(True) for t in (l64, U64, r64, R64)]
            A = torch.eye(N, dtype=torch.complex128) - ref_in[0].unsqueeze(-1) * ref_in[1] * ref_in[2]
            Yr = torch.linalg.solve(A.unsqueeze(0), ref_in[3].unsqueeze(-1)).squeeze(-1)
            gr = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), ref_in)
            dev_in = [t.detach().to(gpu, cd).requires_grad_(True) for t in (l64, U64, r64, R64)]
            outs = {}
            try:
                for variant in (0, 1, 4):      # in-place kernels (two rows per lane up to N = 16) / shuffle kernel / one row per lane
                    _lib.lib().fl_debug_set_solve_variant(variant)
                    Y = ops.solve_dud(dev_in[0], dev_in[1], dev_in[2], dev_in[3])
                    g = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), dev_in)
                    outs[variant] = [Y.detach()] + list(g)
            finally:
                _lib.lib().fl_debug_set_solve_variant(0)
            scale = 30.0 if kind == "permutation" else 1.0      # conditioning of the exchanged systems
            for variant in (0, 1, 4):
                for got, want in zip(outs[variant], [Yr.detach()] + list(gr)):
                    cc("got_cpu_to_torch_complex128", got.cpu().to(torch.complex128), want, tol * scale, max_tol=float("inf"))


@pytest.mark.parametrize("N", [3, 4, 6, 16, 20, 32, 48])
@pytest.mark.parametrize("have", [(True, False), (False, True), (True, True), (False, False)])
def test_factored_solve_one_pass_gradients(gpu, N, have):
    """fl_solve_dud_grads_* (one pass: gl, gU, gr) against the layered backward (five launches) and against autograd
    through LAPACK in float64; per-bin / absent diagonal factors, several batch items and signal columns, a bin count
    that is no multiple of the kernel's bins per iteration."""
    from flamo_amd import ops
    torch.manual_seed(7 * N + have[0] + 2 * have[1])
    M, B, K = 331, 2, 3
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 3e-5)):
        if cd == torch.complex128 and N > 32:
            continue
        U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
        l64 = 0.95 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64)) if have[0] else None
        r64 = (0.7 + 0.25 * torch.rand(M, N, dtype=torch.float64)) * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64)) \
            if have[1] else None
        if not have[0] and not have[1]:
            U64 = 0.9 * U64
        R64 = torch.randn(B, M, N, K, dtype=torch.complex128)
        C64 = torch.randn(B, M, N, K, dtype=torch.complex128)
        names = ["l", "U", "r", "R"]
        ref_in = [None if t is None else t.clone().requires_grad_(True) for t in (l64, U64, r64, R64)]
        P = ref_in[1].unsqueeze(0).expand(M, N, N)
        if have[0]:
            P = ref_in[0].unsqueeze(-1) * P
        if have[1]:
            P = P * ref_in[2].unsqueeze(-2)
        A = torch.eye(N, dtype=torch.complex128) - P
        Yr = torch.linalg.solve(A.unsqueeze(0), ref_in[3])
        live = [i for i, t in enumerate(ref_in) if t is not None]
        gref = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), [ref_in[i] for i in live])
        got = {}
        try:
            for fused in (True, False):
                ops.FUSE_DUD_GRADS = fused
                dev_in = [None if t is None else t.detach().to(gpu, cd).requires_grad_(True) for t in (l64, U64, r64, R64)]
                Y = ops.solve_dud(dev_in[0], dev_in[1], dev_in[2], dev_in[3])
                g = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), [dev_in[i] for i in live])
                got[fused] = [Y.detach()] + list(g)
        finally:
            ops.FUSE_DUD_GRADS = True
        for fused in (True, False):
            for name, a, b in zip(["Y"] + [names[i] for i in live], got[fused], [Yr.detach()] + list(gref)):
                cc("a_cpu_to_torch_complex128", a.cpu().to(torch.complex128), b, tol, max_tol=float("inf"))
        for a, b in zip(got[True], got[False]):
            cc("a", a, b, (1e-12 if cd == torch.complex128 else 3e-6), max_tol=float("inf"))


@pytest.mark.parametrize("shape", [(16, 1), (1, 16), (8, 8), (5, 3), (32, 32)])
def test_real_constant_matrix_product_without_complex_cast(gpu, shape):
    """ops.mimo with a real frequency-independent matrix (fl_mimo_* conj_h bit 1, fl_mimo_gradw_re_*) against the cast
    path (to_complex, complex kernels, real part of the gradient): forward, input gradient, real parameter gradient;
    float32 / float64, contiguous and transposed storage, vector and matrix-valued signals."""
    from flamo_amd import ops
    No, Ni = shape
    M = 1001
    for rd, cd, tol in ((torch.float32, torch.complex64, 2e-6), (torch.float64, torch.complex128, 1e-12)):
        for tail in ((), (3,)):
            for transposed in (False, True):
                torch.manual_seed(No * 7 + Ni)
                W0 = torch.randn(Ni, No, device=gpu, dtype=rd).t() if transposed else torch.randn(No, Ni, device=gpu, dtype=rd)
                X0 = torch.randn(2, M, Ni, *tail, device=gpu, dtype=cd)
                C = torch.randn(2, M, No, *tail, device=gpu, dtype=cd)
                res = {}
                try:
                    for fast in (True, False):
                        ops.REAL_CONST_MIMO = fast
                        W = W0.clone().requires_grad_(True) if not transposed else W0.detach().requires_grad_(True)
                        X = X0.clone().requires_grad_(True)
                        ops.kernel_timer.reset(True)
                        Y = ops.mimo(W, X)
                        gW, gX = torch.autograd.grad((Y * C.conj()).real.sum(), [W, X])
                        torch.cuda.synchronize()
                        ops.kernel_timer.enabled = False
                        names = set(ops.kernel_timer.summary())
                        res[fast] = (Y.detach(), gW, gX, names)
                finally:
                    ops.REAL_CONST_MIMO = True
                    ops.kernel_timer.enabled = False
                big = rd == torch.float32 and No >= 16 and Ni >= 8 and 2 * (3 if tail else 1) >= 8
                assert any("mimo_const_real" in n for n in res[True][3]) == (not big)
                assert not any("mimo_const_real" in n for n in res[False][3])
                assert res[True][1].dtype == rd and res[True][1].shape == (No, Ni)
                for a, b in zip(res[True][:3], res[False][:3]):
                    cc("a", a, b, tol, max_tol=float("inf"))
                ref = torch.einsum("mn,bfn...->bfm...", W0.to(torch.complex128).cpu(), X0.to(torch.complex128).cpu())
                cc("res_True_0_cpu_to_torch_complex128", res[True][0].cpu().to(torch.complex128), ref, (1e-6 if rd == torch.float32 else 1e-13), max_tol=float("inf"))


@pytest.mark.parametrize("N", [4, 6, 16, 32])
def test_matrix_exp_complex_output_matches_real(gpu, N):
    """ops.matrix_exp(..., complex_out=True): the complex matrix (re, 0) written by the same launch, and the backward reading
    the real part of a complex gradient, against the real pair followed by a cast"""
    from flamo_amd import ops
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(N)
        X0 = torch.randn(N, N, device=gpu, dtype=dt)
        cd = torch.complex64 if dt == torch.float32 else torch.complex128
        C = torch.randn(N, N, device=gpu, dtype=cd)
        Xa = X0.clone().requires_grad_(True)
        Ea = ops.matrix_exp(Xa, skew=True, complex_out=True)
        (ga,) = torch.autograd.grad((Ea * C.conj()).real.sum(), [Xa])
        Xb = X0.clone().requires_grad_(True)
        Eb = ops.matrix_exp(Xb, skew=True).to(cd)
        (gb,) = torch.autograd.grad((Eb * C.conj()).real.sum(), [Xb])
        assert Ea.dtype == cd and torch.equal(Ea.real, Eb.real) and float(Ea.detach().imag.abs().max()) == 0.0
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("N", [4, 13, 16, 32])
def test_scaled_loop_solve_matches_composed_loop(gpu, N):
    """ops.solve_scaled_loop -- (I - diag(g) D[f] U)^-1 R with P' = D U formed once, the gains applied as a row scale
    inside the solve (fl_solve_scaled_*), gradients from two per-bin matrix-vector passes -- against LAPACK autograd in
    float64 on the materialised loop, and (Recursion level) against the composed-loop route it replaces."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    torch.manual_seed(N)
    M, B = 257, 2
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 3e-5)):
        D64 = torch.exp(2j * torch.pi * torch.rand(M, N, N, dtype=torch.float64))
        U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
        g64 = ((0.2 + 0.5 * torch.rand(N, dtype=torch.float64)) / N ** 0.5).to(torch.complex128)
        R64 = torch.randn(B, M, N, dtype=torch.complex128)
        C64 = torch.randn(B, M, N, dtype=torch.complex128)
        gr, Ur, Rr = (t.clone().requires_grad_(True) for t in (g64, U64, R64))
        A = torch.eye(N, dtype=torch.complex128) - gr.unsqueeze(-1) * (D64 @ Ur)
        Yr = torch.linalg.solve(A.unsqueeze(0), Rr.unsqueeze(-1)).squeeze(-1)
        want = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), [gr, Ur, Rr])
        gd, Ud, Rd = (t.detach().to(gpu, cd).requires_grad_(True) for t in (g64, U64, R64))
        Y = ops.solve_scaled_loop(gd, D64.to(gpu, cd), Ud, Rd)
        got = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), [gd, Ud, Rd])
        cc("Y_detach_cpu_to_torch_complex128", Y.detach().cpu().to(torch.complex128), Yr.detach(), tol, max_tol=float("inf"))
        for a, b in zip(got, want):
            cc("a_cpu_to_torch_complex128", a.cpu().to(torch.complex128), b, tol, max_tol=float("inf"))
    # the planner: Recursion(fF=Series(Delay((N,N)), parallelGain(N)), fB=Matrix orthogonal) with and without the route
    nfft = 960
    kw = dict(nfft=nfft, alias_decay_db=20.0, device=gpu, dtype=torch.float64)
    dly = dsp.Delay(size=(N, N), max_len=200, isint=True, **kw)
    gain = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
    with torch.no_grad():
        gain.param.copy_(torch.rand_like(gain.param) * 0.5 / N ** 0.5 + 0.01)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    loop = system.Recursion(fF=system.Series(OrderedDict(d=dly, g=gain)), fB=mix)
    X = torch.randn(2, nfft // 2 + 1, N, device=gpu, dtype=torch.complex128)
    Cw = torch.randn(2, nfft // 2 + 1, N, device=gpu, dtype=torch.complex128)
    res = {}
    try:
        for on in (True, False):
            system.SCALED_LOOP = on
            ops.kernel_timer.reset(True)
            Y = loop(X)
            g = torch.autograd.grad(torch.sum(torch.real(Y * Cw.conj())), [gain.param, mix.param])
            torch.cuda.synchronize()
            ops.kernel_timer.enabled = False
            res[on] = (Y.detach(), *g, set(ops.kernel_timer.summary()))
    finally:
        system.SCALED_LOOP = True
        ops.kernel_timer.enabled = False
    for a, b in zip(res[True][:3], res[False][:3]):
        cc("a", a, b, 1e-10, max_tol=float("inf"))
    assert not any(n.startswith("mimo_gradh[") for n in res[True][3]), res[True][3]     # no (M, N, N) gradient tensor
    assert any(n.startswith("mimo_gradh[") for n in res[False][3])


@pytest.mark.parametrize("N", [4, 6, 16, 24])
@pytest.mark.parametrize("have", [(True, False), (False, False), (True, True)])
def test_factored_solve_with_feedforward_diagonal(gpu, N, have):
    """ops.solve_dud2: (I - diag(l . l2) U diag(r))^-1 (l2 . R0) with l2 applied inside the kernels (fl_solve_dud2_*,
    fl_solve_dud2_grads_*) against ops.solve_dud on explicitly formed l . l2 and l2 . R0, and against LAPACK autograd in
    float64: output and the gradients of l, U, r, R0."""
    from flamo_amd import ops
    torch.manual_seed(3 * N + have[0] + 2 * have[1])
    M, B, K = 203, 2, 2
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 3e-5)):
        U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
        l64 = 0.95 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64)) if have[0] else None
        r64 = (0.8 + 0.15 * torch.rand(M, N, dtype=torch.float64)) * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64)) \
            if have[1] else None
        l2_64 = (0.97 if have[0] else 0.9) * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
        R64 = torch.randn(B, M, N, K, dtype=torch.complex128)
        C64 = torch.randn(B, M, N, K, dtype=torch.complex128)
        ref_in = [None if t is None else t.clone().requires_grad_(True) for t in (l64, U64, r64, R64)]
        P = ref_in[1].unsqueeze(0).expand(M, N, N) * l2_64.unsqueeze(-1)
        if have[0]:
            P = ref_in[0].unsqueeze(-1) * P
        if have[1]:
            P = P * ref_in[2].unsqueeze(-2)
        A = torch.eye(N, dtype=torch.complex128) - P
        Yr = torch.linalg.solve(A.unsqueeze(0), l2_64.unsqueeze(-1) * ref_in[3])
        live = [i for i, t in enumerate(ref_in) if t is not None]
        gref = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), [ref_in[i] for i in live])
        dev_in = [None if t is None else t.detach().to(gpu, cd).requires_grad_(True) for t in (l64, U64, r64, R64)]
        Y = ops.solve_dud2(dev_in[0], l2_64.to(gpu, cd), dev_in[1], dev_in[2], dev_in[3])
        g = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), [dev_in[i] for i in live])
        for name, a, b in zip(["Y"] + [["l", "U", "r", "R"][i] for i in live], [Y.detach()] + list(g), [Yr.detach()] + list(gref)):
            cc("a_cpu_to_torch_complex128", a.cpu().to(torch.complex128), b, tol, max_tol=float("inf"))


def test_fdn_feedforward_diagonal_inside_the_solve(gpu):
    """Recursion(fF=parallelDelay, fB=Series(Matrix orthogonal, parallelGEQ)) with and without FDN_DIAGONAL_IN_SOLVE:
    same output and parameter gradients, four diagonal-product launches fewer."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    torch.manual_seed(5)
    nfft, N = 4800, 6
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 2e-5)):
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
        dl = dsp.parallelDelay(size=(N,), max_len=300, isint=True, **kw)
        mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
        att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
        with torch.no_grad():
            att.param.copy_(0.5 + 0.4 * torch.rand_like(att.param))
        loop = system.Recursion(fF=dl, fB=system.Series(OrderedDict(mixing_matrix=mix, attenuation=att)))
        cd = torch.complex128 if dt == torch.float64 else torch.complex64
        X0 = torch.randn(2, nfft // 2 + 1, N, device=gpu, dtype=cd)
        C = torch.randn(2, nfft // 2 + 1, N, device=gpu, dtype=cd)
        res = {}
        try:
            for on in (True, False):
                system.FDN_DIAGONAL_IN_SOLVE = on
                X = X0.clone().requires_grad_(True)
                ops.kernel_timer.reset(True)
                Y = loop(X)
                g = torch.autograd.grad(torch.sum(torch.real(Y * C.conj())), [mix.param, att.param, X])
                torch.cuda.synchronize()
                ops.kernel_timer.enabled = False
                res[on] = (Y.detach(), *g)
        finally:
            system.FDN_DIAGONAL_IN_SOLVE = True
            ops.kernel_timer.enabled = False
        for a, b in zip(res[True], res[False]):
            cc("a", a, b, tol, max_tol=float("inf"))


@pytest.mark.parametrize("N", [4, 7, 16])
def test_matrix_exp_both_forms_and_step_scope(gpu, N):
    """ops.matrix_exp_both (real and complex forms from one launch, one backward launch for both gradients) against the two
    single-form calls; and ops.step_scope(): a Matrix module's complex response and the real map asked for by a criterion
    come from ONE evaluation, with the same gradient as two."""
    from flamo_amd import ops
    from flamo_amd.processor import dsp
    for dt in (torch.float32, torch.float64):
        cd = torch.complex64 if dt == torch.float32 else torch.complex128
        torch.manual_seed(N)
        X0 = torch.randn(N, N, device=gpu, dtype=dt)
        Cr = torch.randn(N, N, device=gpu, dtype=dt)
        Cc = torch.randn(N, N, device=gpu, dtype=cd)
        Xa = X0.clone().requires_grad_(True)
        E, Ec = ops.matrix_exp_both(Xa, skew=True)
        (ga,) = torch.autograd.grad((E * Cr).sum() + (Ec * Cc.conj()).real.sum(), [Xa])
        Xb = X0.clone().requires_grad_(True)
        E1, E2 = ops.matrix_exp(Xb, skew=True), ops.matrix_exp(Xb, skew=True, complex_out=True)
        (gb,) = torch.autograd.grad((E1 * Cr).sum() + (E2 * Cc.conj()).real.sum(), [Xb])
        assert torch.equal(E.detach(), E1.detach()) and torch.equal(Ec.detach(), E2.detach())
        cc("ga", ga, gb, (1e-6 if dt == torch.float32 else 1e-13), max_tol=float("inf"))
        # only one of the two forms used
        Xc = X0.clone().requires_grad_(True)
        E, Ec = ops.matrix_exp_both(Xc, skew=True)
        (gc,) = torch.autograd.grad((E * Cr).sum(), [Xc])
        Xd = X0.clone().requires_grad_(True)
        (gd,) = torch.autograd.grad((ops.matrix_exp(Xd, skew=True) * Cr).sum(), [Xd])
        cc("gc", gc, gd, (1e-6 if dt == torch.float32 else 1e-13), max_tol=float("inf"))
        # module level
        mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, nfft=480, device=gpu, dtype=dt)
        grads = {}
        for scoped in (True, False):
            mix.param.grad = None
            ctx = ops.step_scope() if scoped else torch.enable_grad()
            with ctx:
                H, _ = mix._bin_response(mix.param)
                A = mix.map(mix.param)
                if scoped:
                    H2, _ = mix._bin_response(mix.param)
                    assert H2 is H and mix.map(mix.param) is A        # one evaluation, shared
                ((H * Cc.conj()).real.sum() + (A.abs()).sum()).backward()
            grads[scoped] = mix.param.grad.clone()
        assert ops.step_memo() is None
        cc("grads_True", grads[True], grads[False], (1e-6 if dt == torch.float32 else 1e-13), max_tol=float("inf"))


@pytest.mark.parametrize("N", [4, 6, 16, 24])
def test_fdn_between_its_gains_as_one_operator(gpu, N):
    """Series(Gain(N,1), Recursion(fF=parallelDelay, fB=Series(Matrix orthogonal, parallelGEQ)), Gain(1,N)) on a one-channel
    spectrum: ops.fdn_core (both gains' gradients from the side reductions of fl_solve_dud2_grads_*) against the
    module-by-module route; output, every parameter gradient and the input gradient, float64 and float32, batch 1 and 3."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    nfft = 4800
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 3e-5)):
        torch.manual_seed(N)
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
        ig = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
        og = dsp.Gain(size=(1, N), requires_grad=True, **kw)
        dl = dsp.parallelDelay(size=(N,), max_len=300, isint=True, **kw)
        mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
        att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
        with torch.no_grad():
            att.param.copy_(0.5 + 0.4 * torch.rand_like(att.param))
        core = system.Series(OrderedDict(input_gain=ig, feedback_loop=system.Recursion(
            fF=dl, fB=system.Series(OrderedDict(mixing_matrix=mix, attenuation=att))), output_gain=og))
        cd = torch.complex128 if dt == torch.float64 else torch.complex64
        params = [ig.param, og.param, mix.param, att.param]
        for B in (1, 3):
            X0 = torch.randn(B, nfft // 2 + 1, 1, device=gpu, dtype=cd)
            C = torch.randn(B, nfft // 2 + 1, 1, device=gpu, dtype=cd)
            res = {}
            try:
                for on in (True, "launches", False):        # gains inside the solve / as launches of their own / module by module
                    system.FDN_CORE = bool(on)
                    ops.FDN_GAINS_IN_SOLVE = on is True
                    X = X0.clone().requires_grad_(True)
                    ops.kernel_timer.reset(True)
                    Y = core(X)
                    g = torch.autograd.grad(torch.sum(torch.real(Y * C.conj())), params + [X])
                    torch.cuda.synchronize()
                    ops.kernel_timer.enabled = False
                    res[on] = ([Y.detach()] + list(g), set(ops.kernel_timer.summary()))
            finally:
                system.FDN_CORE = True
                ops.FDN_GAINS_IN_SOLVE = True
                ops.kernel_timer.enabled = False
            assert not any(n.startswith("mimo_gradw") for n in res[True][1]), res[True][1]
            in_solve = N <= (32 if dt == torch.float32 else 16)          # the in-place kernels' range
            assert any(n.startswith("mimo_const_real_fwd") for n in res[True][1]) == (not in_solve), res[True][1]
            assert any(n.startswith("mimo_const_real_fwd") for n in res["launches"][1])
            assert any(n.startswith("mimo_gradw") for n in res[False][1])
            assert res[True][0][1].dtype == dt and res[True][0][1].shape == ig.param.shape
            for k in (True, "launches"):
                for a, b in zip(res[k][0], res[False][0]):
                    cc("a", a, b, tol, max_tol=float("inf"))


def test_replayed_step_is_stable_under_eager_launches(gpu):
    """A captured step (Shell(FFT -> Gain(16,1) -> Gain(1,16) -> iFFTAntiAlias), torch's own `sum` as the loss) replayed with a
    tiny eager tensor created and filled between replays: loss and gradients stay bit-identical.  With ROCm's pre-built graph
    packets the captured memset + reduction pair of `sum()` returned a different value after the first eager launch
    (flamo_amd/__init__.py turns DEBUG_CLR_GRAPH_PACKET_CAPTURE off for that reason; tools/dbg/archive/soak_ops.py, soak_fdn15.py)."""
    import os
    from collections import OrderedDict
    from flamo_amd.graph import GraphedStep
    from flamo_amd.processor import dsp, system
    # (the hazard is the platform's: tools/dbg/archive/replay_min.py reproduces it with torch kernels alone -- MB-sized temporaries
    # allocated inside the capture in front of a captured sum() -- and with no kernel of this library in the graph)
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
    torch.manual_seed(1)
    nfft, N = 192000, 16
    kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=torch.float32)
    ig = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
    og = dsp.Gain(size=(1, N), requires_grad=True, **kw)
    model = system.Shell(system.Series(OrderedDict(input_gain=ig, output_gain=og)), dsp.FFT(nfft),
                         dsp.iFFTAntiAlias(nfft, alias_decay_db=30.0, device=gpu))
    x = torch.randn(1, nfft, 1, device=gpu)
    c = torch.randn(1, nfft, 1, device=gpu)
    params = [ig.param, og.param]
    gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
    out0 = gs.replay().clone()
    g0 = [p.grad.clone() for p in params]
    torch.cuda.synchronize()
    with torch.no_grad():
        want = (model(x) * c).sum()
    cc("out0", out0, want, 1e-5, max_tol=float("inf"))
    for i in range(6):
        out = gs.replay()
        torch.cuda.synchronize()
        junk = torch.full((1,), float(i), device=gpu)
        del junk
        assert torch.equal(out, out0), (i, out.item(), out0.item())
        assert all(torch.equal(p.grad, g) for p, g in zip(params, g0))


def test_captured_step_holds_integer_delay_response_as_a_constant(gpu):
    """A feedback delay network's integer delays never change during training (e8_fdn.py:84-90, e8_colorless_fdn.py: parallelDelay
    with isint=True, requires_grad=False): a GraphedStep takes their response from the cache its eager warm-up runs left instead of
    recording the response's launch and the eight small ones in front of it into every replay.  Replays equal the eager step;
    eager calls with another bin order in between do not disturb them (the step keeps the tensor alive); once the delays are
    assigned a new value the step refuses to replay."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_colorless_fdn as T
    from flamo_amd import ops
    from flamo_amd.graph import GraphedStep
    torch.manual_seed(3)
    nfft, N = 24000, 8
    model = T.build(gpu, torch.float32, N, nfft)
    x, target = T.colorless_batch(1, nfft, gpu, torch.float32)
    params = [p for p in model.parameters() if p.requires_grad]
    delays = model.get_core().feedback_loop.feedforward

    def criteria(xx):
        return T.mse_criterion(model(xx), target) + 0.2 * T.sparsity_criterion(model)

    for p in params:
        p.grad = None
    with ops.step_scope():
        criteria(x).backward()
    want = [p.grad.clone() for p in params]
    step = GraphedStep(criteria, (x,), params, warmup=2)
    assert len(step._constants) >= 1 and all(c[0]() is delays.param for c in step._constants)
    loss0 = step.replay().clone()
    for p, w in zip(params, want):
        cc("p_grad", p.grad, w, 1e-6, max_tol=float("inf"))
    # an eager evaluation of the same module under another key replaces the module's cache entry: the graph's constant survives
    with torch.no_grad():
        ops.set_bin_shard(100, 50)
        try:
            delays.freq_response(delays.param)
        finally:
            ops.set_bin_shard(0, None)
    junk = [torch.randn(1 << 18, device=gpu) for _ in range(8)]
    del junk
    assert torch.equal(step.replay(), loss0)
    with torch.no_grad():
        delays.assign_value(delays.param.detach().clone() * 1.5)
    with pytest.raises(RuntimeError):
        step.replay()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_sparsity_criterion_matches_the_reference_lines(gpu, dt):
    """ops.sparsity / flamo_amd.optimize.sparsity_loss (flamo/optimize/loss.py:12-63) against the reference's torch lines: a
    plain (N, N) matrix, a (C, N, N) stack, values and gradients (sign(0) = 0), and the class on an FDN model."""
    import math
    import sys
    import os
    from flamo_amd import ops
    from flamo_amd.optimize import sparsity_loss
    torch.manual_seed(9)
    tol = 1e-6 if dt == torch.float32 else 1e-14
    for shape in ((16, 16), (3, 8, 8), (1, 4, 4), (2, 2)):
        A0 = torch.randn(*shape, dtype=dt, device=gpu)
        A0.view(-1)[1] = 0.0
        N = shape[-1]
        Ar = A0.clone().requires_grad_(True)
        if len(shape) == 3:
            ref = torch.mean((torch.sum(torch.abs(Ar), dim=(-2, -1)) - N * math.sqrt(N)) / (N * (1 - math.sqrt(N))))
        else:
            ref = -(torch.sum(torch.abs(Ar)) - N * math.sqrt(N)) / (N * (math.sqrt(N) - 1))
        (gr,) = torch.autograd.grad(ref * 1.7, [Ar])
        Ah = A0.clone().requires_grad_(True)
        out = ops.sparsity(Ah)
        (gh,) = torch.autograd.grad(out * 1.7, [Ah])
        assert abs(out.item() - ref.item()) <= tol * max(1.0, abs(ref.item())), (shape, out.item(), ref.item())
        assert relerr(gh, gr) < tol and float(gh.view(-1)[1]) == 0.0
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_colorless_fdn as T
    model = T.build(gpu, dt, 8, 4800)
    mix = model.get_core().feedback_loop.feedback
    got = sparsity_loss()(None, None, model)
    A = mix.map(mix.param)
    want = -(torch.sum(torch.abs(A)) - 8 * math.sqrt(8)) / (8 * (math.sqrt(8) - 1))
    assert abs(got.item() - want.item()) <= 10 * tol
    (g1,) = torch.autograd.grad(got, [mix.param])
    (g2,) = torch.autograd.grad(want, [mix.param])
    cc("g1", g1, g2, 20 * tol, max_tol=float("inf"))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_magnitude_output_layer_in_one_launch(gpu, dt):
    """dsp.Transform(lambda x: torch.abs(x)) -- the output layer of examples/e7_biquad.py:76 and e8_colorless_fdn.py:102 -- is
    recognised by probe (values and gradient of the callable bit for bit torch.abs's) and runs as ops.cabs: against the callable
    itself on a contiguous spectrum and on a bin-planar view (a core's output), zeros included; other callables are left alone."""
    from flamo_amd import ops
    from flamo_amd.processor import dsp
    cd = torch.complex64 if dt == torch.float32 else torch.complex128
    tol = 1e-6 if dt == torch.float32 else 1e-14
    torch.manual_seed(4)
    layer = dsp.Transform(lambda x: torch.abs(x), device=gpu, dtype=dt)
    other = dsp.Transform(lambda x: torch.abs(x) * 1.0000001, device=gpu, dtype=dt)
    assert dsp._is_magnitude_map(layer.transform) and dsp._is_magnitude_map(torch.abs) and not dsp._is_magnitude_map(other.transform)
    B, M, N = 2, 4801, 3
    zc = torch.randn(B, M, N, dtype=cd, device=gpu)
    zc[0, 5, 1] = 0
    planar = ops._empty_planar((B, M, N), cd, gpu)
    planar.copy_(zc)
    assert not planar.is_contiguous()
    w = torch.randn(B, M, N, dtype=dt, device=gpu)
    for z0 in (zc, planar):
        res = []
        for on in (True, False):
            dsp.MAGNITUDE_LAYER = on
            try:
                z = z0.detach().clone() if z0.is_contiguous() else z0.detach()
                z = z.requires_grad_(True)
                y = layer(z)
                (g,) = torch.autograd.grad((y * w).sum(), [z])
                res.append((y.detach(), g))
            finally:
                dsp.MAGNITUDE_LAYER = True
        assert res[0][0].dtype == dt and res[0][0].shape == (B, M, N)
        assert relerr(res[0][0], res[1][0]) < tol and relerr(res[0][1], res[1][1]) < tol
        assert float(res[0][0][0, 5, 1]) == 0.0 and complex(res[0][1][0, 5, 1]) == 0
    y2 = other(zc)
    cc("y2", y2, torch.abs(zc) * 1.0000001, tol, max_tol=float("inf"))


@pytest.mark.parametrize("N,K", [(20, 1), (32, 1), (32, 3), (48, 1)])
def test_scaled_loop_adjoint_from_kept_factors(gpu, N, K):
    """fl_solve_scaled_keep_* + fl_solve_kept_adjoint_*: the backward pass's adjoint system A^H x = g from the LU factors the
    forward solve left (no second elimination) against LAPACK autograd in float64 on the materialised loop
    (flamo/processor/system.py:420-425: torch.linalg.solve and its backward), vector and matrix right-hand sides, and against
    the re-factoring route it replaces (ops.KEEP_LU = False)."""
    from flamo_amd import ops
    torch.manual_seed(100 + N)
    M, B = 193, 2
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 3e-5)):
        if cd == torch.complex128 and N > 32:
            continue
        D64 = torch.exp(2j * torch.pi * torch.rand(M, N, N, dtype=torch.float64))
        U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0].to(torch.complex128)
        g64 = ((0.2 + 0.5 * torch.rand(N, dtype=torch.float64)) / N ** 0.5).to(torch.complex128)
        shape = (B, M, N) if K == 1 else (B, M, N, K)
        R64 = torch.randn(*shape, dtype=torch.complex128)
        C64 = torch.randn(*shape, dtype=torch.complex128)
        gr, Ur, Rr = (t.clone().requires_grad_(True) for t in (g64, U64, R64))
        A = torch.eye(N, dtype=torch.complex128) - gr.unsqueeze(-1) * (D64 @ Ur)
        Yr = torch.linalg.solve(A.unsqueeze(0), Rr.unsqueeze(-1) if K == 1 else Rr)
        Yr = Yr.squeeze(-1) if K == 1 else Yr
        want = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), [gr, Ur, Rr])
        res = {}
        for keep in (True, False):
            ops.KEEP_LU = keep
            try:
                gd, Ud, Rd = (t.detach().to(gpu, cd).requires_grad_(True) for t in (g64, U64, R64))
                ops.kernel_timer.reset(True)
                Y = ops.solve_scaled_loop(gd, D64.to(gpu, cd), Ud, Rd)
                got = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), [gd, Ud, Rd])
                torch.cuda.synchronize()
            finally:
                ops.KEEP_LU = True
                ops.kernel_timer.enabled = False
            res[keep] = (Y.detach(), *got)
        tag = f"kept_lu/N{N}_K{K}_{str(cd)[-3:]}"
        check_close(tag + "/Y", res[True][0].cpu().to(torch.complex128), Yr.detach(), tol)
        for name, a, b in zip(("g_g", "g_U", "g_R"), res[True][1:], want):
            check_close(f"{tag}/{name}", a.cpu().to(torch.complex128), b, tol)
        assert torch.equal(res[True][0], res[False][0])                 # the same forward kernel, with and without the stores
        for a, b in zip(res[True][1:], res[False][1:]):                 # the two adjoint routes: rounding apart
            cc("a", a, b, (1e-11 if cd == torch.complex128 else 3e-5), max_tol=float("inf"))


@pytest.mark.parametrize("N", [5, 6, 8, 9, 12, 13, 16])
@pytest.mark.parametrize("kind", ["damped", "exchanges"])
def test_fdn_adjoint_from_kept_factors(gpu, N, kind):
    """ops.fdn_core's backward system A^H x = c^H g (8 < N <= 16) three ways: x = w g with w = A^-H c^H computed by the FORWARD
    launch from its own factors (fl_solve_fdn_wadj_c64: float32 default, no solve in the backward pass), a substitution over
    factors the forward solve kept (fl_solve_fdn_keep_* + fl_solve_kept_adjoint_rank1_*), and a second elimination
    (fl_solve_fdn_* adjoint) -- output and every gradient against LAPACK autograd
    in float64 on the materialised system (flamo/processor/system.py:420-425 between the two gains), for damped loops (no row
    moves) and loops that exchange rows at every step (the pivot order travels with the factors), odd and even N, batch 1 and
    3, and against the re-factoring route it replaces (ops.KEEP_LU_FDN = False)."""
    from flamo_amd import _lib, ops
    torch.manual_seed(7 * N + (kind == "exchanges"))
    M = 391
    assert _lib.lib().fl_solve_fdn_keep_tile(N, 0) == (32 if N > 8 else 0)      # (ops.KEEP_LU_FDN is off by default: measured slower)
    assert _lib.lib().fl_solve_fdn_wadj_supported(N) == 1
    for cd, tol in ((torch.complex128, 1e-11), (torch.complex64, 3e-5)):
        rd = torch.float64 if cd == torch.complex128 else torch.float32
        if kind == "damped":
            U64 = torch.linalg.qr(torch.randn(N, N, dtype=torch.float64))[0]
            l64 = 0.97 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
            scale = 1.0
        else:
            perm = torch.roll(torch.arange(N), 1)
            U64 = 3.0 * torch.eye(N, dtype=torch.float64)[perm] + 0.05 * torch.randn(N, N, dtype=torch.float64)
            l64 = torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
            scale = 30.0
        l2_64 = 0.9 * torch.exp(2j * torch.pi * torch.rand(M, N, dtype=torch.float64))
        r64 = (0.8 + 0.2 * torch.rand(M, N, dtype=torch.float64)).to(torch.complex128)
        b64, c64 = torch.randn(N, 1, dtype=torch.float64), torch.randn(1, N, dtype=torch.float64)
        for B in (1, 3):
            X64 = torch.randn(B, M, 1, dtype=torch.complex128)
            C64 = torch.randn(B, M, 1, dtype=torch.complex128)
            ref_in = [t.clone().requires_grad_(True) for t in (b64, c64, l64, U64, r64, X64)]
            bR, cR, lR, UR, rR, XR = ref_in
            A = torch.eye(N, dtype=torch.complex128) - (lR * l2_64).unsqueeze(-1) * UR.to(torch.complex128) * rR.unsqueeze(-2)
            rhs = l2_64.unsqueeze(0) * (bR.to(torch.complex128).squeeze(-1) * XR)              # (B, M, N)
            out = torch.linalg.solve(A.unsqueeze(0), rhs.unsqueeze(-1)).squeeze(-1)
            Yr = (out * cR.to(torch.complex128).squeeze(0)).sum(-1, keepdim=True)
            want = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C64))), ref_in)
            res = {}
            # three routes to A^-H c^H gy: w gy with w from the forward launch (float32 default) / a substitution over kept factors /
            # a second elimination
            routes = (("forward", (True, False)), ("kept", (False, True)), ("refactor", (False, False)))
            if N <= 8:
                routes = (routes[0], routes[2])          # (factors are kept above 8 channels only)
            for route, (in_fwd, keep) in routes:
                ops.FDN_ADJOINT_IN_FORWARD, ops.KEEP_LU_FDN = in_fwd, keep
                try:
                    dev_in = [b64.to(gpu, rd), c64.to(gpu, rd), l64.to(gpu, cd), U64.to(gpu, cd), r64.to(gpu, cd), X64.to(gpu, cd)]
                    dev_in = [t.requires_grad_(True) for t in dev_in]
                    ops.kernel_timer.reset(True)
                    Y = ops.fdn_core(dev_in[0], dev_in[1], dev_in[2], l2_64.to(gpu, cd), dev_in[3], dev_in[4], dev_in[5])
                    got = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C64.to(gpu, cd)))), dev_in)
                    torch.cuda.synchronize()
                    used = set(ops.kernel_timer.summary())
                finally:
                    ops.FDN_ADJOINT_IN_FORWARD, ops.KEEP_LU_FDN = True, False
                    ops.kernel_timer.enabled = False
                res[route] = [Y.detach()] + list(got)
                # the forward route runs no adjoint solve at all
                assert ("solve_dud_adj" in used) == (route != "forward"), (route, used)
            tag = f"fdn_kept/{kind}_N{N}_B{B}_{str(cd)[-3:]}"
            names = ("Y", "g_b", "g_c", "g_l", "g_U", "g_r", "g_X")
            for route in [r for r in ("forward", "kept") if r in res]:
                for name, a, w in zip(names, res[route], [Yr.detach()] + list(want)):
                    w = w.real if (not a.is_complex() and w.is_complex()) else w
                    check_close(f"{tag}/{route}/{name}", a.cpu().to(w.dtype), w, tol * scale, max_tol=float("inf"))
            for route in [r for r in ("forward", "kept") if r in res]:
                # ("kept": the same forward kernel with the factor stores -- identical; "forward": an instantiation of its own)
                if route == "kept":
                    assert torch.equal(res[route][0], res["refactor"][0])
                for a, b in zip(res[route], res["refactor"]):                # rounding apart
                    assert relerr(a, b) < tol * scale
