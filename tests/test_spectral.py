"""GPU tests of the fused Shell pipeline (csrc/spectral.hip, ops.spectral_apply):
   y = irfft(H[f] . rfft(x)) in three launches, against torch.fft on the CPU in float64 (the reference's own
   primitives, dsp.py:88 / dsp.py:114 / dsp.py:922-924) and against the layered HIP operators.
Tolerance: float32 kernels, relative l2 error 1e-5 against float64 (BASELINE.json north_star)."""
from collections import OrderedDict

import pytest
import torch

from conftest import cc, check_close, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-5
PLANS = {96000: (200, 240), 192000: (400, 240), 384000: (800, 240)}


def _ref(x, H, nfft, norm_f="backward", norm_i="backward", db_f=0.0, db_i=0.0):
    """float64 CPU: irfft(H . rfft(x e_f)) e_i with the reference's rising envelopes (dsp.py:158-162, 201-205)"""
    x = x.detach().cpu().double()
    H = H.detach().cpu().to(torch.complex128)
    t = torch.arange(nfft, dtype=torch.float64)
    ef = 10.0 ** (abs(db_f) / (20.0 * nfft) * t) if db_f else None
    ei = 10.0 ** (abs(db_i) / (20.0 * nfft) * t) if db_i else None
    xx = x[:, :nfft]
    if ef is not None:
        xx = xx * ef[: xx.shape[1], None]
    X = torch.fft.rfft(xx, n=nfft, dim=1, norm=norm_f)
    Y = torch.einsum("fmn,bfn->bfm", H, X)
    y = torch.fft.irfft(Y, n=nfft, dim=1, norm=norm_i)
    if ei is not None:
        y = y * ei[:, None]
    return X, y


def test_plans_from_the_factorisation(gpu):
    """fl_spec_plan derives (L1, L2) for lengths without a table entry; what it cannot factor over the instantiated column and
    row lengths (odd half-lengths, large primes) reports unsupported and the operators take the layered route."""
    import ctypes
    from flamo_amd import _lib, ops
    L = _lib.lib()
    for nfft, want in ((32000, (50, 320)), (44100, (441, 50)), (88200, (441, 100)), (64000, (125, 256)), (24000, (50, 240)),
                       (16000, (50, 160)), (160000, (250, 320)), (256000, (400, 320)), (176400, (441, 200))):
        L1, L2 = ctypes.c_int(), ctypes.c_int()
        assert L.fl_spec_plan(nfft, ctypes.byref(L1), ctypes.byref(L2)) == 0, nfft
        assert L1.value * L2.value * 2 == nfft and (L1.value, L2.value) == want, (nfft, L1.value, L2.value)
        assert ops.spectral_supported(nfft, 8, 8)
    for nfft in (22050, 95999, 2176, 34):
        assert L.fl_spec_plan(nfft, None, None) != 0 and not ops.spectral_supported(nfft, 8, 8)
    assert not ops.spectral_supported(44100, 2, 2)           # 2 channels: 8-column tiles do not divide the 50-bin rows


@pytest.mark.parametrize("nfft", sorted(PLANS))
def test_plan_and_bin_order(gpu, nfft):
    import ctypes
    from flamo_amd import _lib, ops
    L1, L2 = ctypes.c_int(), ctypes.c_int()
    assert _lib.lib().fl_spec_plan(nfft, ctypes.byref(L1), ctypes.byref(L2)) == 0
    assert (L1.value, L2.value) == PLANS[nfft]
    M = nfft // 2 + 1
    H = torch.arange(M, device=gpu, dtype=torch.float32).to(torch.complex64).view(M, 1, 1).expand(M, 2, 3).contiguous()
    Hr = ops.permute_bins(H, nfft)
    i = torch.arange(M - 1, device=gpu)
    k = i // L2.value + L1.value * (i % L2.value)
    assert torch.equal(Hr[:-1, 1, 2].real.long(), k) and Hr[-1, 0, 0].real.item() == M - 1   # bit exact: an index map
    assert torch.equal(ops.permute_bins(Hr, nfft, inverse=True), H)


@pytest.mark.parametrize("N", [2, 4, 8, 16])
@pytest.mark.parametrize("vt", [32, 16])
def test_transform_round_trip_and_spectrum(gpu, N, vt):
    """K1 + mid + K3 without a response: the spectrum it leaves (row-major bin order) is rfft(x), the output is x"""
    from flamo_amd import _lib, ops
    nfft, B = 96000, 3
    _lib.lib().fl_debug_set_spec(vt, 2)
    try:
        torch.manual_seed(N)
        x = torch.randn(B, nfft, N, device=gpu)
        S = ops._spec_cols_fwd(x, nfft, 0.0)
        S2, Xs = ops._spec_mid(S, B, N, N, nfft, None, False, True, True, 1.0, 0, 0)
        y = ops._spec_cols_inv(S2, B, nfft, nfft, N, nfft, 1.0 / nfft, 0.0)
        Xref = torch.fft.rfft(x.cpu().double(), n=nfft, dim=1)
        X = ops.permute_bins(Xs.movedim(-1, 0), nfft, inverse=True)            # (M, B, N) natural order
        cc("X_permute_1_0_2_cpu", X.permute(1, 0, 2).cpu(), Xref, TOL, max_tol=float("inf"))
        cc("y_cpu", y.cpu(), x.cpu().double(), TOL, max_tol=float("inf"))
    finally:
        _lib.lib().fl_debug_set_spec(0, 0)      # back to the per-shape choice


@pytest.mark.parametrize("nfft,N,B", [(96000, 8, 3), (96000, 4, 2), (96000, 2, 5), (96000, 16, 2), (192000, 8, 2), (384000, 4, 2),
                                      (192000, 16, 1), (384000, 2, 1), (384000, 16, 1),
                                      # the batch-walking row kernels (csrc/specwalk.hip): 8 x 8 channels at nfft = 96000 from 4 items
                                      # on -- a unit or two per workgroup (4, 7), row-pair changes inside a workgroup's range (33)
                                      (96000, 8, 4), (96000, 8, 7), (96000, 8, 33),
                                      # the other planned lengths (csrc/spectral.hip kPlans): the reference's default 2^11, powers of
                                      # two to 2^17, one and three seconds at 48 kHz
                                      (2048, 2, 3), (2048, 16, 2), (4096, 8, 3), (8192, 4, 2), (16384, 16, 1), (32768, 2, 2), (65536, 8, 2),
                                      (131072, 4, 1), (48000, 8, 3), (48000, 16, 1), (144000, 8, 2), (144000, 2, 1),
                                      # lengths PLANNED from the factorisation (no table entry): examples/e4_recursion_nn.py:349's
                                      # 32000 = 2 . 50 . 320, 44100 = 2 . 441 . 50, 88200 = 2 . 441 . 100, and multiples of 8000
                                      (32000, 8, 3), (32000, 2, 2), (44100, 8, 2), (88200, 4, 1), (64000, 8, 2), (24000, 4, 2), (16000, 2, 2),
                                      (160000, 8, 1), (256000, 2, 1)])
def test_spectral_apply_against_torch_fft(gpu, nfft, N, B):
    from flamo_amd import ops
    torch.manual_seed(nfft + N)
    M = nfft // 2 + 1
    x = torch.randn(B, nfft, N, device=gpu, requires_grad=True)
    H = (torch.randn(M, N, N, device=gpu, dtype=torch.complex64) / N ** 0.5).requires_grad_(True)
    y = ops.spectral_apply(x, ops.permute_bins(H, nfft), nfft)
    assert y.shape == (B, nfft, N) and y.is_contiguous()
    c = torch.randn(B, nfft, N, device=gpu)
    gx, gH = torch.autograd.grad((y * c).sum(), [x, H])
    xr = x.detach().cpu().double().requires_grad_(True)
    Hr = H.detach().cpu().to(torch.complex128).requires_grad_(True)
    Y = torch.einsum("fmn,bfn->bfm", Hr, torch.fft.rfft(xr, n=nfft, dim=1))
    yr = torch.fft.irfft(Y, n=nfft, dim=1)
    gxr, gHr = torch.autograd.grad((yr * c.cpu().double()).sum(), [xr, Hr])
    cc("y_detach_cpu", y.detach().cpu(), yr.detach(), TOL, max_tol=float("inf"))
    cc("gx_cpu", gx.cpu(), gxr, TOL, max_tol=float("inf"))
    cc("gH_cpu", gH.cpu(), gHr, TOL, max_tol=float("inf"))


TOL64 = 1e-12     # float64 kernels against torch.fft in float64 on the CPU


@pytest.mark.parametrize("nfft,N,B", [(96000, 8, 3), (96000, 2, 2), (96000, 16, 2), (192000, 4, 2), (192000, 8, 1), (384000, 8, 1), (384000, 2, 2),
                                      (2048, 16, 3), (4096, 8, 2), (65536, 4, 2), (131072, 2, 1), (48000, 8, 2), (144000, 4, 1)])
def test_spectral_apply_float64(gpu, nfft, N, B):
    """The same three launches compiled for double (fl_spec_*_f64): forward and both gradients"""
    from flamo_amd import ops
    assert ops.spectral_supported(nfft, N, N, torch.float64)
    torch.manual_seed(nfft + N)
    M = nfft // 2 + 1
    x = torch.randn(B, nfft - 5, N, device=gpu, dtype=torch.float64, requires_grad=True)
    H = (torch.randn(M, N, N, device=gpu, dtype=torch.complex128) / N ** 0.5).requires_grad_(True)
    ops.kernel_timer.reset(True)
    y = ops.spectral_apply(x, ops.permute_bins(H, nfft), nfft, "ortho", "backward", 20.0, 10.0)
    c = torch.randn(B, nfft, N, device=gpu, dtype=torch.float64)
    gx, gH = torch.autograd.grad((y * c).sum(), [x, H])
    torch.cuda.synchronize()
    used = set(ops.kernel_timer.records)
    ops.kernel_timer.enabled = False
    assert {"spec_cols_fwd", "spec_cols_inv"} <= used and any(k.startswith("spec_mid[") for k in used)
    assert y.dtype == torch.float64 and gx.shape == x.shape and gH.dtype == torch.complex128
    xr = x.detach().cpu().requires_grad_(True)
    Hr = H.detach().cpu().requires_grad_(True)
    t = torch.arange(nfft, dtype=torch.float64)
    xx = xr * (10.0 ** (20.0 / (20.0 * nfft) * t))[: nfft - 5, None]
    Y = torch.einsum("fmn,bfn->bfm", Hr, torch.fft.rfft(xx, n=nfft, dim=1, norm="ortho"))
    yr = torch.fft.irfft(Y, n=nfft, dim=1) * (10.0 ** (10.0 / (20.0 * nfft) * t))[:, None]
    gxr, gHr = torch.autograd.grad((yr * c.cpu()).sum(), [xr, Hr])
    cc("y_detach_cpu", y.detach().cpu(), yr.detach(), TOL64, max_tol=float("inf"))
    cc("gx_cpu", gx.cpu(), gxr, TOL64, max_tol=float("inf"))
    cc("gH_cpu", gH.cpu(), gHr, TOL64, max_tol=float("inf"))


def test_shell_float64_fused_equals_layered(gpu):
    """A float64 Shell (FFT -> Series(Gain, parallelDelay, Matrix) -> iFFT) takes the fused route and agrees with the layered one"""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    nfft, N, dt = 96000, 8, torch.float64
    kw = dict(nfft=nfft, device=gpu, dtype=dt)
    torch.manual_seed(5)
    core = system.Series(OrderedDict(g=dsp.Gain(size=(N, N), requires_grad=True, **kw),
                                     d=dsp.parallelDelay(size=(N,), max_len=1500, isint=True, requires_grad=False, **kw),
                                     m=dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)))
    shell = system.Shell(core, dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
    x = torch.randn(2, nfft, N, device=gpu, dtype=dt)
    params = [p for p in shell.parameters() if p.requires_grad]
    c = torch.randn(2, nfft, N, device=gpu, dtype=dt)      # (a plain energy loss does not depend on the orthogonal matrix)
    ops.kernel_timer.reset(True)
    y = shell(x)
    g1 = torch.autograd.grad((y * c).sum(), params)
    torch.cuda.synchronize()
    used = set(ops.kernel_timer.records)
    ops.kernel_timer.enabled = False
    assert any(k.startswith("spec_mid[") for k in used)
    old = ops.spectral_supported
    ops.spectral_supported = lambda *a, **k: False
    try:
        y2 = shell(x)
        g2 = torch.autograd.grad((y2 * c).sum(), params)
    finally:
        ops.spectral_supported = old
    cc("y", y, y2, TOL64, max_tol=float("inf"))
    for a, b in zip(g1, g2):
        cc("a", a, b, 1e-10, max_tol=float("inf"))


@pytest.mark.parametrize("norm_f,norm_i,db_f,db_i,T", [("backward", "backward", 0.0, 30.0, 96000), ("ortho", "ortho", 30.0, 30.0, 96000),
                                                        ("forward", "forward", 30.0, 0.0, 96000), ("backward", "backward", 0.0, 0.0, 50001),
                                                        ("backward", "backward", 0.0, 0.0, 96017)])
@pytest.mark.parametrize("N,B", [(4, 2), (8, 6)])        # (8, 6): through the batch-walking row kernels
def test_spectral_apply_norms_envelopes_lengths(gpu, norm_f, norm_i, db_f, db_i, T, N, B):
    from flamo_amd import ops
    nfft = 96000
    torch.manual_seed(T)
    M = nfft // 2 + 1
    x = torch.randn(B, T, N, device=gpu, requires_grad=True)
    H = (torch.randn(M, N, N, device=gpu, dtype=torch.complex64) / N ** 0.5).requires_grad_(True)
    y = ops.spectral_apply(x, ops.permute_bins(H, nfft), nfft, norm_f, norm_i, db_f or None, db_i or None)
    c = torch.randn(B, nfft, N, device=gpu)
    gx, gH = torch.autograd.grad((y * c).sum(), [x, H])
    assert gx.shape == x.shape
    xr = x.detach().cpu().double().requires_grad_(True)
    Hr = H.detach().cpu().to(torch.complex128).requires_grad_(True)
    t = torch.arange(nfft, dtype=torch.float64)
    xx = xr[:, :nfft]
    if db_f:
        xx = xx * (10.0 ** (db_f / (20.0 * nfft) * t))[: xx.shape[1], None]
    Y = torch.einsum("fmn,bfn->bfm", Hr, torch.fft.rfft(xx, n=nfft, dim=1, norm=norm_f))
    yr = torch.fft.irfft(Y, n=nfft, dim=1, norm=norm_i)
    if db_i:
        yr = yr * (10.0 ** (db_i / (20.0 * nfft) * t))[:, None]
    gxr, gHr = torch.autograd.grad((yr * c.cpu().double()).sum(), [xr, Hr])
    cc("y_detach_cpu", y.detach().cpu(), yr.detach(), TOL, max_tol=float("inf"))
    cc("gx_cpu", gx.cpu(), gxr, TOL, max_tol=float("inf"))
    cc("gH_cpu", gH.cpu(), gHr, TOL, max_tol=float("inf"))


def _config2(gpu, N, nfft, db=0.0):
    from flamo_amd.processor import dsp, system
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=torch.float32)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    core = system.Series(OrderedDict(mix=mat, eq=geq))
    if db:
        shell = system.Shell(core, dsp.FFTAntiAlias(nfft, alias_decay_db=db, device=gpu), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, device=gpu))
    else:
        shell = system.Shell(core, dsp.FFT(nfft), dsp.iFFT(nfft))
    return shell, [mat.param, geq.param]


@pytest.mark.parametrize("db", [0.0, 30.0])
@pytest.mark.parametrize("grad_in", [False, True])
def test_shell_fused_equals_layered_and_oracle(gpu, db, grad_in):
    """BASELINE configs[1] through Shell.forward: fused operator == layered operators == float64 oracle (outputs, parameter
    gradients, input gradient)."""
    from flamo_amd import ops
    from flamo_amd.processor import system
    from oracle import hotpath as O
    nfft, N, B = 96000, 8, 3
    torch.manual_seed(7)
    shell, params = _config2(gpu, N, nfft, db)
    x = torch.randn(B, nfft, N, device=gpu, requires_grad=grad_in)
    wanted = params + ([x] if grad_in else [])

    def run():
        ops.kernel_timer.reset(True)
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), wanted)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return y.detach(), g, used

    y1, g1, used1 = run()
    assert any(k.startswith("spec_mid") for k in used1), used1            # the fused operator ran
    system.FUSE_SHELL = False
    try:
        y2, g2, used2 = run()
    finally:
        system.FUSE_SHELL = True
    assert not any(k.startswith("spec_") for k in used2)
    cc("y1", y1, y2, TOL, max_tol=float("inf"))
    for a, b in zip(g1, g2):
        cc("a", a, b, 3e-5, max_tol=float("inf"))
    if db == 0.0:
        W, G = (p.detach().cpu().double().requires_grad_(True) for p in params)
        xo = x.detach().cpu().double().requires_grad_(grad_in)
        yo = O.config2_forward(xo, W, G, nfft)
        go = torch.autograd.grad((yo ** 2).mean(), [W, G] + ([xo] if grad_in else []))
        cc("y1_cpu", y1.cpu(), yo.detach(), TOL, max_tol=float("inf"))
        for a, b in zip(g1, go):
            cc("a_cpu", a.cpu(), b, TOL, max_tol=float("inf"))


def test_shell_fused_under_graph_capture(gpu):
    """forward + backward of the fused Shell replayed from a HIP graph == eager"""
    from flamo_amd import ops
    from flamo_amd.graph import GraphedStep
    nfft, N, B = 96000, 8, 4
    torch.manual_seed(11)
    shell, params = _config2(gpu, N, nfft)
    x = torch.randn(B, nfft, N, device=gpu)
    loss = ops.mean_square(shell(x))
    ge = torch.autograd.grad(loss, params)
    gs = GraphedStep(lambda xx: ops.mean_square(shell(xx)), (x,), params, warmup=2)
    for _ in range(3):
        lg = gs.replay()
    torch.cuda.synchronize()
    assert abs(lg.item() - loss.item()) <= 1e-6 * abs(loss.item())
    for p, g in zip(params, ge):
        cc("p_grad", p.grad, g, 1e-6, max_tol=float("inf"))


def test_unsupported_shapes_take_the_layered_path(gpu):
    """odd channel counts, other transform lengths and float64 shapes whose tiles exceed the LDS fall back to the layered
    operators, silently and with the same results"""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    # (22050: an odd half-length has no plan; 2176 = 2^7 17: not even a Stockham length, the transforms take the chirp-z route)
    for nfft, N, dt in ((96000, 3, torch.float32), (96000, 6, torch.float64), (22050, 4, torch.float32), (2176, 2, torch.float32)):
        kw = dict(nfft=nfft, device=gpu, dtype=dt)
        shell = system.Shell(system.Series(dsp.Matrix(size=(N, N), **kw)), dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
        x = torch.randn(1 if N == 16 else 2, nfft, N, device=gpu, dtype=dt)
        ops.kernel_timer.reset(True)
        with torch.no_grad():
            y = shell(x)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        assert not any(k.startswith("spec_") for k in used)
        W = shell.get_core()[0].param.detach().cpu().double()
        yr = torch.fft.irfft(torch.einsum("mn,bfn->bfm", W.to(torch.complex128), torch.fft.rfft(x.cpu().double(), n=nfft, dim=1)), n=nfft, dim=1)
        cc("y_cpu", y.cpu(), yr, (1e-10 if dt == torch.float64 else TOL), max_tol=float("inf"))


def _zoo(kind, gpu, nfft, N):
    from flamo_amd.processor import dsp
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32, requires_grad=True)
    if kind == "fir":
        return [dsp.Filter(size=(5, N, N), **kw)]
    if kind == "pfir+gain":
        return [dsp.parallelFilter(size=(7, N), **kw), dsp.Gain(size=(N, N), **kw)]
    if kind == "biquad":
        return [dsp.Biquad(size=(N, N), n_sections=2, filter_type="lowpass", **kw)]
    if kind == "delay+matrix":
        d = dsp.parallelDelay(size=(N,), max_len=3000, isint=True, nfft=nfft, device=gpu)
        return [d, dsp.Matrix(size=(N, N), matrix_type="random", **kw)]   # (an orthogonal mix after pure delays leaves the loss flat: gradient = rounding noise)
    if kind == "fracdelay":
        return [dsp.Delay(size=(N, N), max_len=500, isint=False, **kw)]
    if kind == "gaindelay+peq":
        return [dsp.parallelGainDelay(size=(N,), max_len=800, isint=True, nfft=nfft, device=gpu, requires_grad=True),
                dsp.parallelGEQ(size=(N,), **kw)]
    if kind == "gain-only":
        return [dsp.Gain(size=(N, N), **kw), dsp.parallelGain(size=(N,), **kw)]
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["fir", "pfir+gain", "biquad", "delay+matrix", "fracdelay", "gaindelay+peq", "gain-only"])
def test_shell_fused_module_zoo(gpu, kind):
    """every kind of per-bin module through the fused Shell: responses generated natively in the pipeline's bin order
    (cascades, integer delays), reordered by a gather (FIR, fractional delays) or bin-independent -- against the
    layered operators on the same modules (which the goldens pin to the reference)"""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    nfft, N, B = 96000, 4, 2
    torch.manual_seed(3)
    mods = _zoo(kind, gpu, nfft, N)
    shell = system.Shell(system.Series(*mods), dsp.FFT(nfft), dsp.iFFT(nfft))
    params = [p for p in shell.parameters() if p.requires_grad]
    x = torch.randn(B, nfft, N, device=gpu, requires_grad=True)

    def run():
        ops.kernel_timer.reset(True)
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), params + [x], allow_unused=True)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return y.detach(), g, used

    y1, g1, used1 = run()
    assert any(k.startswith("spec_mid") for k in used1), used1
    system.FUSE_SHELL = False
    try:
        y2, g2, _ = run()
    finally:
        system.FUSE_SHELL = True
    cc("y1", y1, y2, TOL, max_tol=float("inf"))
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            cc("a", a, b, 1e-4, max_tol=float("inf"))


@pytest.mark.parametrize("kind", ["geq", "biquad", "svf", "geq-orth-4"])
def test_matrix_cascade_operator_equals_composition(gpu, kind):
    """Series(Matrix, cascade filter): the fused pair operator (response = cascade @ matrix, both gradients from the cascade's
    backward kernel) against the generic composition of the two modules' responses"""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    nfft, B = 96000, 2
    N = 4 if kind.endswith("4") else 8
    torch.manual_seed(5)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="orthogonal" if "orth" in kind else "random", **kw)
    if kind.startswith("geq"):
        flt = dsp.GEQ(size=(N, N), **kw)
    elif kind == "biquad":
        flt = dsp.Biquad(size=(N, N), n_sections=3, filter_type="bandpass", **kw)
    else:
        flt = dsp.SVF(size=(N, N), n_sections=2, **kw)
    shell = system.Shell(system.Series(OrderedDict(mix=mat, flt=flt)), dsp.FFT(nfft), dsp.iFFT(nfft))
    params = [mat.param, flt.param]
    x = torch.randn(B, nfft, N, device=gpu)

    def run():
        ops.kernel_timer.reset(True)
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), params)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return y.detach(), g, used

    y1, g1, used1 = run()
    assert "sos_response_bwd_rc" in used1, used1
    system.FUSE_MATRIX_CASCADE = False
    try:
        y2, g2, used2 = run()
    finally:
        system.FUSE_MATRIX_CASCADE = True
    assert "sos_response_bwd_rc" not in used2
    cc("y1", y1, y2, TOL, max_tol=float("inf"))
    for a, b in zip(g1, g2):
        cc("a", a, b, 2e-5, max_tol=float("inf"))


@pytest.mark.parametrize("nfft,N", [(144000, 4), (48000, 8), (2048, 2), (192000, 8)])
def test_matrix_cascade_operator_at_other_lengths(gpu, nfft, N):
    """the bin-PAIR walk of the cascade-times-matrix forward (two adjacent bins per thread: elements one row apart in row-major
    order) at plan lengths with an odd row count (144000: 225 rows), few bins (2048) and the longer BASELINE length, and the
    one-round grid of its backward: fused pair operator against the generic composition, forward and both gradients"""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    torch.manual_seed(nfft % 1000 + N)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=torch.float32, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw)
    flt = dsp.GEQ(size=(N, N), **kw)
    shell = system.Shell(system.Series(OrderedDict(mix=mat, flt=flt)), dsp.FFT(nfft), dsp.iFFT(nfft))
    params = [mat.param, flt.param]
    x = torch.randn(2, nfft, N, device=gpu)

    def run():
        ops.kernel_timer.reset(True)
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), params)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return y.detach(), g, used

    y1, g1, used1 = run()
    assert "sos_response_bwd_rc" in used1 and any(k.startswith("spec_mid") for k in used1), used1
    system.FUSE_MATRIX_CASCADE = False
    try:
        y2, g2, _ = run()
    finally:
        system.FUSE_MATRIX_CASCADE = True
    cc("y1", y1, y2, TOL, max_tol=float("inf"))
    for a, b in zip(g1, g2):
        cc("a", a, b, 2e-5, max_tol=float("inf"))
    # natural bin order, an odd number of bins and a bin shard that starts at an odd bin: the same kernels through ops
    M = nfft // 2 + 1
    b64, a64 = flt._sos_coeffs(flt.map(flt.param.detach().double()))
    Wr = mat.map(mat.param.detach()).float()
    Hn = ops.sos_response_rc(b64, a64, Wr, 1.0, nfft)
    with ops.row_major_bins(nfft):
        Hr = ops.sos_response_rc(b64, a64, Wr, 1.0, nfft)
    assert torch.equal(ops.permute_bins(Hr, nfft, inverse=True), Hn)       # the same arithmetic per bin, whatever the order
    ops.set_bin_shard(3, min(1001, M - 3))
    try:
        Hs = ops.sos_response_rc(b64, a64, Wr, 1.0, nfft)
    finally:
        ops.set_bin_shard(0, None)
    assert Hs.shape[0] == min(1001, M - 3) and torch.equal(Hs, Hn[3:3 + Hs.shape[0]])


@pytest.mark.parametrize("variant", [2, 4, 5])
def test_mid_kernel_variants_agree(gpu, variant):
    """the row kernel's experimental forms (tuning hook: 2 / 4 batch items per workgroup with the three-sweep product, one
    item with the three-sweep product) give the default form's results (odd batch: the tail group is partial)"""
    from flamo_amd import _lib, ops
    nfft, N, B = 96000, 8, 5
    torch.manual_seed(variant)
    M = nfft // 2 + 1
    x = torch.randn(B, nfft, N, device=gpu)
    H = ops.permute_bins(torch.randn(M, N, N, device=gpu, dtype=torch.complex64) / N ** 0.5, nfft)
    y0 = ops.spectral_apply(x, H, nfft)
    _lib.lib().fl_debug_set_spec(0, 100 * variant)
    try:
        y1 = ops.spectral_apply(x, H, nfft)
        xg = x.clone().requires_grad_(True)
        (g1,) = torch.autograd.grad(ops.spectral_apply(xg, H, nfft).square().sum(), [xg])
    finally:
        _lib.lib().fl_debug_set_spec(0, 0)      # back to the per-shape choice
    (g0,) = torch.autograd.grad(ops.spectral_apply(xg, H, nfft).square().sum(), [xg])
    assert relerr(y1, y0) < 1e-6 and relerr(g1, g0) < 1e-6


@pytest.mark.parametrize("db", [0.0, 30.0])
def test_cascade_times_matrix_float_evaluation(gpu, db):
    """the float evaluation of the cascade (1 -+ w basis, two sections per packed instruction) in the Matrix-then-cascade
    forward kernel against the same kernel's double evaluation and against the float64 oracle response"""
    from flamo_amd import _lib, ops
    from flamo_amd.processor import dsp
    from oracle import hotpath as O
    nfft, N = 96000, 8
    torch.manual_seed(21)
    geq = dsp.GEQ(size=(N, N), nfft=nfft, alias_decay_db=db, device=gpu, dtype=torch.float32)
    W = torch.randn(N, N, device=gpu)
    spec = geq._cascade_spec(geq.param)
    Hs = []
    for fast in (1, 0):
        _lib.lib().fl_debug_set_rc_fast(fast)
        try:
            Hs.append(ops.geq_cascade_rc(spec[1], spec[2], W, geq._gamma_f, nfft))
        finally:
            _lib.lib().fl_debug_set_rc_fast(1)
    cc("Hs_0", Hs[0], Hs[1], 1e-6, max_tol=float("inf"))
    gamma = O.gamma_of(db, nfft, torch.float64)
    Href = O.geq_response(geq.param.detach().cpu().double(), nfft, gamma) @ W.cpu().double().to(torch.complex128)
    assert relerr(Hs[0].cpu(), Href) < 2e-6 and relerr(Hs[1].cpu(), Href) < 2e-6
    # per entry as well (largest deviation against the response's scale), for both evaluations
    worst = [((h.cpu() - Href).abs().max() / Href.abs().max()).item() for h in Hs]
    print(f"\ncascade x matrix response, largest deviation / scale: float evaluation {worst[0]:.1e}, double evaluation {worst[1]:.1e}")
    assert worst[0] < 2e-6 and worst[1] < 5e-7, worst          # measured 8.7e-7 / 1.7e-7


@pytest.mark.parametrize("kind", ["geq", "biquad_lowpass", "biquad_bandpass", "svf", "peq"])
def test_cascade_float_evaluation_plain_response(gpu, kind):
    """fl_sos_response_f32eval_c64 (1 -+ w basis, section pairs) against the double evaluation for the cascade-type
    modules: graphic / parametric equalisers (shelving sections at 31 Hz), biquads whose numerator vanishes at DC or
    Nyquist, state-variable filters.  Forward-only calls take the float evaluation for every kind; with gradients only the
    graphic equaliser does (its gradient must then agree with the double evaluation's), the others stay in double."""
    from flamo_amd import _lib, ops
    from flamo_amd.processor import dsp
    nfft, N = 48000, 4
    torch.manual_seed(3)
    kw = dict(nfft=nfft, alias_decay_db=20.0, device=gpu, dtype=torch.float32, requires_grad=True)
    if kind == "geq":
        mod = dsp.parallelGEQ(size=(N,), **kw)
    elif kind == "peq":
        mod = dsp.PEQ(size=(N, N), n_bands=6, **kw)
    elif kind == "svf":
        mod = dsp.SVF(size=(N, N), n_sections=3, **kw)
    else:
        mod = dsp.Biquad(size=(N, N), n_sections=2, filter_type=kind.split("_")[1], **kw)
    out, names = {}, {}
    for fast in (1, 0):
        _lib.lib().fl_debug_set_rc_fast(fast)
        try:
            with torch.no_grad():
                ops.kernel_timer.reset(True)
                Hn = mod.freq_response(mod.param)
                torch.cuda.synchronize()
                ops.kernel_timer.enabled = False
            mod.param.grad = None
            H = mod.freq_response(mod.param)
            c = torch.randn(H.shape, device=gpu, dtype=H.dtype, generator=torch.Generator(device=gpu).manual_seed(5))
            (H * c.conj()).real.sum().backward()
            out[fast] = (Hn.clone(), H.detach().clone(), mod.param.grad.clone())
        finally:
            _lib.lib().fl_debug_set_rc_fast(1)
    scale = out[0][0].abs().max()
    assert ((out[1][0] - out[0][0]).abs().max() / scale).item() < 2e-6       # forward-only: float against double
    cc("out_1_0", out[1][0], out[0][0], 1e-6, max_tol=float("inf"))
    assert 1e-9 < relerr(out[1][0], out[0][0]), "the float evaluation did not run"
    if kind == "geq":
        # a random cotangent makes the gradient a sum with heavy cancellation: the mixed-precision backward itself is a few
        # 1e-6 from the float64 module with either forward
        m64 = dsp.parallelGEQ(size=(N,), nfft=nfft, alias_decay_db=20.0, device=gpu, dtype=torch.float64, requires_grad=True)
        with torch.no_grad():
            m64.param.copy_(mod.param.double())
        (m64.freq_response(m64.param) * c.to(torch.complex128).conj()).real.sum().backward()
        e_float, e_double = relerr(out[1][2], m64.param.grad), relerr(out[0][2], m64.param.grad)
        print(f"\nparallelGEQ gradient against the float64 module: float forward {e_float:.1e}, double forward {e_double:.1e}")
        assert e_float < 1.5e-5 and e_double < 1.5e-5        # measured 6.6e-6 / 1.4e-6
    else:       # with gradients these modules evaluate in double whatever the hook says
        assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][2], out[0][2])


@pytest.mark.parametrize("kind", ["geq", "biquad", "svf", "peq"])
@pytest.mark.parametrize("B", [1, 2, 3])
def test_cascade_applied_to_few_columns_without_gradient_tensor(gpu, kind, B):
    """A full cascade-type filter applied to a vector signal with few columns: product through ops.*_apply, whose backward
    forms dL/dH = gY (x) conj(X) inside the cascade kernel (fl_sos_response_bwd_outer_c64) -- against the layered route
    (response tensor, per-bin product, (M, N, N) gradient tensor): output, parameter gradient, input gradient.  Three
    columns are above the module's threshold and must take the layered route."""
    from flamo_amd import ops
    from flamo_amd.processor import dsp
    nfft, N = 9600, 8
    torch.manual_seed(11)
    kw = dict(nfft=nfft, alias_decay_db=20.0, device=gpu, dtype=torch.float32, requires_grad=True)
    if kind == "geq":
        mod = dsp.GEQ(size=(N, N), **kw)
    elif kind == "peq":
        mod = dsp.PEQ(size=(N, N), n_bands=5, **kw)
    elif kind == "svf":
        mod = dsp.SVF(size=(N, N), n_sections=3, **kw)
    else:
        mod = dsp.Biquad(size=(N, N), n_sections=2, filter_type="bandpass", **kw)
    X0 = torch.randn(B, nfft // 2 + 1, N, device=gpu, dtype=torch.complex64)
    C = torch.randn(B, nfft // 2 + 1, N, device=gpu, dtype=torch.complex64)
    res = {}
    try:
        for narrow in (True, False):
            dsp.NARROW_APPLY = narrow
            mod.param.grad = None
            X = X0.clone().requires_grad_(True)
            ops.kernel_timer.reset(True)
            Y = mod(X)
            (Y * C.conj()).real.sum().backward()
            torch.cuda.synchronize()
            ops.kernel_timer.enabled = False
            res[narrow] = (Y.detach().clone(), mod.param.grad.clone(), X.grad.clone(), set(ops.kernel_timer.summary()))
    finally:
        dsp.NARROW_APPLY = True
        ops.kernel_timer.enabled = False
    took = not any(n.startswith("mimo_gradh[") for n in res[True][3])
    assert took == (B <= 2), res[True][3]
    assert any(n.startswith("mimo_gradh[") for n in res[False][3])
    cc("res_True_0", res[True][0], res[False][0], 1e-6, max_tol=float("inf"))
    cc("res_True_2", res[True][2], res[False][2], 1e-6, max_tol=float("inf"))
    # (the layered route rounds dL/dH to float32 before the cascade backward reads it; the fused one does not)
    cc("res_True_1", res[True][1], res[False][1], (5e-5 if kind == "peq" else 1e-5), max_tol=float("inf"))


@pytest.mark.parametrize("parallel", [True, False])
def test_geq_sigmoid_map_folded_into_the_design_kernel(gpu, parallel):
    """dsp.db_of_sigmoid (20 log10(sigmoid(x)), the attenuation map of e8_fdn.py:97) passed by name is folded into
    fl_geq_sections (in_kind 3 / 4) like the default map; the same math as a lambda runs as torch ops.  Response and
    parameter gradient must agree, float32 and float64; for the full matrix also through the Matrix-then-cascade operator
    and the few-columns route."""
    from collections import OrderedDict
    from flamo_amd.processor import dsp, system
    nfft, N = 9600, 8
    for dt, tol_h, tol_g in ((torch.float64, 1e-12, 1e-9), (torch.float32, 2e-6, 2e-5)):
        torch.manual_seed(4)
        kw = dict(nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt, requires_grad=True)
        mods = []
        for named in (True, False):
            m = dsp.parallelGEQ(size=(N,), **kw) if parallel else dsp.GEQ(size=(N, N), **kw)
            m.map = dsp.db_of_sigmoid if named else (lambda x: 20 * torch.log10(torch.sigmoid(x)))
            mods.append(m)
        with torch.no_grad():
            mods[0].param.copy_(torch.randn_like(mods[0].param) * 0.5 + 1.5)
            mods[1].param.copy_(mods[0].param)
        cd = torch.complex128 if dt == torch.float64 else torch.complex64
        out = []
        for m in mods:
            H = m.freq_response(m.param)
            c = torch.randn(H.shape, device=gpu, dtype=cd, generator=torch.Generator(device=gpu).manual_seed(9))
            (H * c.conj()).real.sum().backward()
            out.append((H.detach(), m.param.grad.clone()))
        cc("out_0_0", out[0][0], out[1][0], tol_h, max_tol=float("inf"))
        cc("out_0_1", out[0][1], out[1][1], tol_g, max_tol=float("inf"))
        if parallel or dt != torch.float32:
            continue
        # few columns (ops.geq_cascade_apply) and Matrix-then-cascade (ops.geq_cascade_rc) with the named map
        X = torch.randn(1, nfft // 2 + 1, N, device=gpu, dtype=cd)
        res = []
        for m in mods:
            m.param.grad = None
            Y = m(X)
            (Y.abs() ** 2).sum().backward()
            res.append((Y.detach(), m.param.grad.clone()))
        assert relerr(res[0][0], res[1][0]) < 2e-6 and relerr(res[0][1], res[1][1]) < 5e-5
        res = []
        for m in mods:
            m.param.grad = None
            mat = dsp.Matrix(size=(N, N), nfft=nfft, alias_decay_db=30.0, device=gpu, dtype=dt)
            torch.manual_seed(2)
            with torch.no_grad():
                mat.param.copy_(torch.randn(N, N, device=gpu))
            ser = system.Series(OrderedDict(mix=mat, eq=m))
            Xb = torch.randn(6, nfft // 2 + 1, N, device=gpu, dtype=cd, generator=torch.Generator(device=gpu).manual_seed(1))
            Y = ser(Xb)
            (Y.abs() ** 2).sum().backward()
            res.append((Y.detach(), m.param.grad.clone()))
        assert relerr(res[0][0], res[1][0]) < 2e-6 and relerr(res[0][1], res[1][1]) < 5e-5


@pytest.mark.parametrize("dt,nfft,N,B", [(torch.float64, 96000, 8, 3), (torch.float64, 96000, 8, 32), (torch.float64, 96000, 4, 2),
                                         (torch.float64, 24000, 2, 5), (torch.float64, 65536, 8, 3), (torch.float32, 96000, 8, 3),
                                         (torch.float32, 65536, 4, 2), (torch.float64, 192000, 8, 2)])
def test_response_gradient_in_one_launch(gpu, dt, nfft, N, B):
    """The training step's backward (the data takes no gradient, trainer.py:172-191) where the float32 batch-walking kernel does not
    apply: fl_spec_gradh_loop_* walks the batch per (row pair, channel group) instead of writing the gradient's spectrum and
    reading it back.  Against the layered form (spec_mid without a response + mimo_gradh) on the same inputs, with a plain
    cotangent and through the fused objective (device-side factor), and against torch.fft + einsum in float64."""
    from flamo_amd import ops
    cd = torch.complex128 if dt == torch.float64 else torch.complex64
    torch.manual_seed(nfft + N + B)
    M = nfft // 2 + 1
    x = torch.randn(B, nfft, N, device=gpu, dtype=dt)
    H = (torch.randn(M, N, N, device=gpu, dtype=cd) / N ** 0.5).requires_grad_(True)
    c = torch.randn(B, nfft, N, device=gpu, dtype=dt)

    def run(objective):
        ops.kernel_timer.reset(True)
        y = ops.spectral_apply(x, ops.permute_bins(H, nfft), nfft, "backward", "backward", None, 30.0)
        loss = ops.mean_square(y) if objective else (y * c).sum()
        (g,) = torch.autograd.grad(loss, [H])
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return g, used

    keep = ops.GRADH_LOOP_MAX_BATCH
    try:
        for objective in (False, True):
            ops.GRADH_LOOP_MAX_BATCH = keep
            _, used = run(objective)
            walks = dt == torch.float32 and B >= 4
            assert ("spec_gradh_walk" in used) == walks and ("spec_gradh_loop" in used) == (not walks and B <= keep), used
            if walks:
                continue
            ops.GRADH_LOOP_MAX_BATCH = 1 << 30        # the kernel itself at any batch size
            g1, used1 = run(objective)
            assert "spec_gradh_loop" in used1 and not any(k.startswith("mimo_gradh") for k in used1), used1
            ops.GRADH_LOOP = False
            try:
                g0, used0 = run(objective)
            finally:
                ops.GRADH_LOOP = True
            assert "spec_gradh_loop" not in used0 and any(k.startswith("mimo_gradh") for k in used0), used0
            tag = f"gradh_loop/{str(dt)[6:]}_{nfft}_{N}_{B}_{int(objective)}"
            check_close(tag + "/vs_layered", g1, g0, 1e-12 if dt == torch.float64 else 2e-6)
    finally:
        ops.GRADH_LOOP_MAX_BATCH = keep
    if B <= 3:
        xr, Hr = x.cpu().double(), H.detach().cpu().to(torch.complex128).requires_grad_(True)
        t = torch.arange(nfft, dtype=torch.float64)
        yr = torch.fft.irfft(torch.einsum("fmn,bfn->bfm", Hr, torch.fft.rfft(xr, n=nfft, dim=1)), n=nfft, dim=1) \
            * (10.0 ** (30.0 / (20.0 * nfft) * t))[:, None]
        (gr,) = torch.autograd.grad((yr * c.cpu().double()).sum(), [Hr])
        g1, used1 = run(False)
        assert "spec_gradh_loop" in used1
        check_close(f"gradh_loop/{str(dt)[6:]}_{nfft}_{N}_{B}/vs_torch_fft", g1.cpu(), gr, 1e-10 if dt == torch.float64 else 1e-5)


@pytest.mark.parametrize("kind,N,db", [("geq", 8, 0.0), ("geq", 8, 30.0), ("geq", 4, 0.0), ("biquad", 8, 0.0), ("geq-then-gain", 8, 0.0),
                                       ("geq", 16, 0.0)])
def test_launch_pair_response_beside_column_pass(gpu, kind, N, db):
    """csrc/fusedfwd.hip: the Matrix-then-cascade response's launch rides in the input's column pass (one grid, either role per
    workgroup).  Same device functions as the two plain kernels: output and gradients EQUAL the two-launch form bit for bit; the
    grid with both roles is issued exactly where the shape has one (float32, 200-point columns, 4 / 8 channels) and the recorded
    launch goes out alone everywhere else (16 channels; another module behind the pair reads the response first)."""
    from flamo_amd import _lib, ops
    from flamo_amd.processor import dsp, system
    nfft, B = 96000, 3
    torch.manual_seed(21)
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=torch.float32, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw)
    if kind == "biquad":       # raw sections WITHOUT a gradient are evaluated by the float kernel (with one: in double, no pair)
        flt = dsp.Biquad(size=(N, N), n_sections=3, filter_type="bandpass", **{**kw, "requires_grad": False})
    else:
        flt = dsp.GEQ(size=(N, N), **kw)
    mods = OrderedDict(mix=mat, flt=flt)
    if kind == "geq-then-gain":
        mods["tail"] = dsp.parallelGain(size=(N,), **kw)
    fin = dsp.FFTAntiAlias(nfft, alias_decay_db=db, device=gpu) if db else dsp.FFT(nfft)
    fout = dsp.iFFTAntiAlias(nfft, alias_decay_db=db, device=gpu) if db else dsp.iFFT(nfft)
    shell = system.Shell(system.Series(mods), fin, fout)
    params = [m.param for m in mods.values() if m.param.requires_grad]
    x = torch.randn(B, nfft, N, device=gpu)
    L = _lib.lib()

    def run():
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), params)
        return [y.detach()] + [t.detach() for t in g]

    run()      # (the first evaluation at a length fills its twiddle tables between the two launches: the recorded one goes out alone)
    n0 = L.fl_debug_launch_pair_count()
    paired = run()
    n1 = L.fl_debug_launch_pair_count()
    assert not L.fl_launch_pair_pending()
    takes = kind in ("geq", "biquad") and N in (4, 8)
    assert n1 - n0 == (1 if takes else 0), (kind, N, n1 - n0)
    ops.LAUNCH_PAIRS = False
    try:
        plain = run()
    finally:
        ops.LAUNCH_PAIRS = True
    assert L.fl_debug_launch_pair_count() == n1
    for a, b in zip(paired, plain):
        assert torch.equal(a, b), kind
