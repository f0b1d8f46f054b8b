"""Round 4: transform lengths the Stockham kernels do not take (chirp-z route, dsp.py:84-89 hands ANY nfft to torch.fft),
fused-pipeline plans derived from the factorisation, batch-walking kernels beyond the benchmark's shape."""
import os
from collections import OrderedDict

import pytest
import torch

from conftest import cc, check_close, relerr

pytestmark = pytest.mark.gpu
F64 = torch.float64
F32 = torch.float32
TOL = {torch.float64: 1e-10, torch.float32: 1e-5}
CD = {torch.float64: torch.complex128, torch.float32: torch.complex64}


@pytest.fixture(params=[torch.float64, torch.float32], ids=["f64", "f32"])
def dt(request):
    return request.param


# 34 = 2 17, 2176 = 2^7 17 (a prime factor > 13), 95999 (odd, = 17 5647), small odd lengths, a prime, and lengths the
# kernels take directly (32000, 44100) as the control
@pytest.mark.parametrize("nfft", [34, 2176, 95999, 95, 7, 3, 1, 4099, 32000, 44100])
def test_transforms_any_length(gpu, dt, nfft):
    """rfft / irfft at arbitrary nfft against torch.fft in float64 (the reference: dsp.py:84-89, 110-115): outputs and
    gradients, truncated / zero-padded input, the three norms, both envelopes."""
    from flamo_amd import ops
    from oracle import hotpath as O
    torch.manual_seed(nfft)
    M = nfft // 2 + 1
    # (the anti-aliased transforms multiply by an nfft-long envelope: like the reference's they take T == nfft only)
    cases = [(nfft, "backward", 0.0), (M, "ortho", 0.0), (nfft + 7, "forward", 0.0), (nfft, "ortho", 30.0)] if nfft < 50000 \
        else [(nfft, "backward", 30.0)]
    for T, norm, db in cases:
        x = torch.randn(2, T, 3, dtype=dt, device=gpu).requires_grad_(True)
        C = torch.randn(2, M, 3, dtype=CD[dt], device=gpu)
        X = ops.rfft(x, nfft, norm, db)
        xr = x.detach().cpu().double().requires_grad_(True)
        Xr = O.rfft(xr, nfft, norm, db or None)
        assert X.shape == Xr.shape
        check_close(f"any_len/{nfft}/{str(dt)[6:]}/{T}{norm}{db}/X", X.detach().cpu(), Xr.detach(), TOL[dt])
        (gx,) = torch.autograd.grad(torch.sum(torch.real(X * torch.conj(C))), [x])
        (gxr,) = torch.autograd.grad(torch.sum(torch.real(Xr * torch.conj(C.cpu().to(torch.complex128)))), [xr])
        check_close(f"any_len/{nfft}/{str(dt)[6:]}/{T}{norm}{db}/gx", gx.cpu(), gxr, TOL[dt])
        Z = torch.randn(2, M, 3, dtype=CD[dt], device=gpu).requires_grad_(True)
        c = torch.randn(2, nfft, 3, dtype=dt, device=gpu)
        y = ops.irfft(Z, nfft, norm, db)
        Zr = Z.detach().cpu().to(torch.complex128).requires_grad_(True)
        yr = O.irfft(Zr, nfft, norm, db or None)
        assert y.shape == yr.shape
        check_close(f"any_len/{nfft}/{str(dt)[6:]}/{T}{norm}{db}/y", y.detach().cpu(), yr.detach(), TOL[dt])
        (gZ,) = torch.autograd.grad(torch.sum(y * c), [Z])
        (gZr,) = torch.autograd.grad(torch.sum(yr * c.cpu().double()), [Zr])
        check_close(f"any_len/{nfft}/{str(dt)[6:]}/{T}{norm}{db}/gZ", gZ.cpu(), gZr, TOL[dt])


@pytest.mark.parametrize("nfft", [95, 2176])
def test_shell_at_a_length_without_a_plan(gpu, dt, nfft):
    """A whole model at an nfft the kernels do not plan (odd; prime factor 17): Shell(FFTAntiAlias -> Series(Matrix, parallelDelay,
    GEQ) -> iFFTAntiAlias), output and parameter gradients against the oracle -- the response generators and the per-bin
    product only ever see M = nfft // 2 + 1 bins and the twiddle table of that length."""
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    N, B, db = 4, 2, 20.0
    torch.manual_seed(nfft)
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=dt)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    dl = dsp.parallelDelay(size=(N,), max_len=20, isint=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    W = torch.randn(N, N).double()
    G = torch.empty(12, N, N).uniform_(10 ** (-6 / 20), 10 ** (6 / 20)).float().double()
    m = torch.tensor([3.0, 7.0, 11.0, 13.0]).double()
    mat.assign_value(W.to(gpu, dt))
    geq.assign_value(G.to(gpu, dt))
    dl.assign_value(dl.sample2s(m.to(gpu, dt)))
    model = system.Shell(system.Series(OrderedDict(mix=mat, d=dl, eq=geq)), dsp.FFTAntiAlias(nfft, alias_decay_db=db, device=gpu, dtype=dt),
                         dsp.iFFTAntiAlias(nfft, alias_decay_db=db, device=gpu, dtype=dt))
    x = torch.randn(B, nfft, N, dtype=dt, device=gpu)
    y = model(x)
    gW, gG = torch.autograd.grad((y ** 2).mean(), [mat.param, geq.param])
    Wl, Gl = W.clone().requires_grad_(True), G.clone().requires_grad_(True)
    gamma = O.gamma_of(db, nfft, F64)
    X = O.rfft(x.cpu().double(), nfft, alias_decay_db=db)
    X = O.mimo_const(O.to_complex(Wl), X)
    X = O.mimo_diag(O.delay_response(m, nfft, gamma), X)
    X = O.mimo_full(O.geq_response(Gl, nfft, gamma), X)
    yr = O.irfft(X, nfft, alias_decay_db=db)
    gWr, gGr = torch.autograd.grad((yr ** 2).mean(), [Wl, Gl])
    tol = 1e-9 if dt == F64 else 1e-5
    check_close(f"shell_noplan/{nfft}/{str(dt)[6:]}/y", y.detach().cpu(), yr.detach(), max(tol, 2e-6))
    check_close(f"shell_noplan/{nfft}/{str(dt)[6:]}/gW", gW.cpu(), gWr, max(tol, 2e-6) * 5)
    check_close(f"shell_noplan/{nfft}/{str(dt)[6:]}/gG", gG.cpu(), gGr, 1e-3)       # float32 section buffers (dsp.py:2573-2585)


@pytest.mark.parametrize("nfft,N,B", [(96000, 8, 5), (96000, 8, 2), (4096, 4, 3), (192000, 16, 2)])
def test_objective_rides_in_the_pipeline(gpu, dt, nfft, N, B):
    """ops.mean_square(model(x)) as one node over (x, H) (value from the inverse column pass's partial sums, gradient by running
    the pipeline's backward on y itself) against the two streaming passes: loss, parameter gradients and input gradient; and y
    used a second time (its own node keeps working beside the fused one).  Walking kernels (batch 5), the one-item row kernel
    (batch 2), a small plan, 16 channels."""
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    torch.manual_seed(nfft + N + B)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=dt)
    mat = dsp.Matrix(size=(N, N), requires_grad=True, **kw)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
    x = torch.randn(B, nfft, N, device=gpu, dtype=dt).requires_grad_(True)
    w = torch.randn(B, nfft, N, device=gpu, dtype=dt)
    plist = [x, mat.param, geq.param]

    def run(fuse, second_use):
        ops.FUSE_OBJECTIVE = fuse
        try:
            y = model(x)
            if getattr(y, "_flamo_sa", None) is None:               # not the three-launch route at this shape / precision
                pytest.skip("layered route: nothing to fuse")
            loss = 3.0 * ops.mean_square(y)
            if second_use:
                loss = loss + (y * w).sum() * 1e-6
            return loss.detach(), torch.autograd.grad(loss, plist)
        finally:
            ops.FUSE_OBJECTIVE = True
    tol = 1e-12 if dt == F64 else 2e-6
    for second_use in (False, True):
        l0, g0 = run(False, second_use)
        l1, g1 = run(True, second_use)
        cc("l1", l1, l0, tol)
        for a, b, k in zip(g1, g0, ("gx", "gW", "gG")):
            cc("a", a, b, 10 * tol)
    with torch.no_grad():                                    # validation: the value alone, still without a pass over y
        yv = model(x)
        ops.kernel_timer.reset(True)
        lv = ops.mean_square(yv)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        assert used == {'mean_square_final'}
        cc("lv", lv, (yv.double() ** 2).mean(), 1e-12 if dt == F64 else 1e-06)
    # the objective of a tensor that was modified after the pipeline produced it: the generic passes
    y = model(x)
    y.mul_(2.0)
    cc("ops_mean_square_y", ops.mean_square(y).detach(), (y.detach() ** 2).mean(), 1e-12 if dt == F64 else 1e-06)


@pytest.mark.parametrize("nfft,NI,NO,B", [(96000, 4, 4, 5), (96000, 2, 2, 7), (96000, 4, 8, 4), (96000, 8, 2, 5), (96000, 2, 4, 6),
                                          (96000, 8, 4, 9), (65536, 8, 8, 5), (131072, 8, 8, 4), (131072, 4, 4, 4), (64000, 8, 8, 6),
                                          (64000, 4, 4, 4), (96000, 8, 8, 9)])
def test_walking_kernels_beyond_the_benchmark_shape(gpu, nfft, NI, NO, B):
    """The batch-walking row kernels at other channel counts (2 / 4 / 8 on either side, non-square included) and at the 256-bin
    rows (nfft = 65536, 131072, 64000): output, input gradient and response gradient against the one-item row kernel
    (fl_debug_set_walk(0)) and against torch.fft + einsum in float64 (dsp.py:922-924); odd batch sizes, ranges that cross a row
    pair, the adjoint product with the channel counts swapped."""
    from flamo_amd import _lib, ops
    L = _lib.lib()
    assert L.fl_spec_walk_supports(nfft, NI, NO) == 1 and ops._walk_applies(nfft, B, NI, NO)
    torch.manual_seed(nfft + 10 * NI + NO)
    M = nfft // 2 + 1
    x = torch.randn(B, nfft, NI, device=gpu, requires_grad=True)
    H = (torch.randn(M, NO, NI, device=gpu, dtype=torch.complex64) / NI ** 0.5).requires_grad_(True)
    c = torch.randn(B, nfft, NO, device=gpu)

    def run():
        y = ops.spectral_apply(x, ops.permute_bins(H, nfft), nfft)
        return (y.detach(),) + torch.autograd.grad((y * c).sum(), [x, H])
    yw, gxw, gHw = run()
    L.fl_debug_set_walk(0, 0, 0, None)
    try:
        assert not ops._walk_applies(nfft, B, NI, NO)
        ym, gxm, gHm = run()
    finally:
        L.fl_debug_set_walk(1, 0, 0, None)
    for a, b, k in ((yw, ym, "y"), (gxw, gxm, "gx"), (gHw, gHm, "gH")):
        cc("a", a, b, 2e-06)
    xr = x.detach().cpu().double().requires_grad_(True)
    Hr = H.detach().cpu().to(torch.complex128).requires_grad_(True)
    yr = torch.fft.irfft(torch.einsum("fmn,bfn->bfm", Hr, torch.fft.rfft(xr, n=nfft, dim=1)), n=nfft, dim=1)
    gxr, gHr = torch.autograd.grad((yr * c.cpu().double()).sum(), [xr, Hr])
    tag = f"walk_shapes/{nfft}/{NI}to{NO}/b{B}"
    check_close(tag + "/y", yw.cpu(), yr.detach(), 1e-5)
    check_close(tag + "/gx", gxw.cpu(), gxr, 1e-5)
    check_close(tag + "/gH", gHw.cpu(), gHr, 1e-5)


@pytest.mark.parametrize("nfft,G,B,t_out", [(96000, 8, 3, 96000), (96000, 16, 2, 50000), (32000, 4, 2, 32000), (4096, 2, 5, 4096), (384000, 8, 1, 384000)])
def test_inverse_column_pass_leaves_the_sum_of_squares(gpu, dt, nfft, G, B, t_out):
    """fl_spec_cols_inv_sumsq_*: the same y as fl_spec_cols_inv_* (bit for bit) plus per-workgroup partial sums whose total is
    sum(y^2) over the samples it stored (truncated output, envelope and scale included); fl_mean_square_final_* reduces them."""
    from flamo_amd import _lib, ops
    if not ops.spectral_supported(nfft, G, G, dt):
        pytest.skip("no fused plan at this shape / precision")
    torch.manual_seed(nfft + G)
    L = nfft // 2
    S2 = torch.randn(B * L * G, dtype=CD[dt], device=gpu)
    # (a full-length output is written over the scratch it is transformed from: ops.INVERSE_IN_PLACE -- each call gets its own)
    y0 = ops._spec_cols_inv(S2.clone(), B, nfft, t_out, G, nfft, 1.0 / nfft, ops.env_log2_of(20.0, nfft))
    y1, parts = ops._spec_cols_inv(S2.clone(), B, nfft, t_out, G, nfft, 1.0 / nfft, ops.env_log2_of(20.0, nfft), want_sumsq=True)
    assert torch.equal(y0, y1)
    assert parts.dtype == torch.float64 and parts.numel() == int(ops._spec_fn("fl_spec_cols_blocks", dt)(nfft, B, G))
    want = (y1.double() ** 2).sum()
    assert abs(parts.sum().item() / want.item() - 1.0) < (1e-12 if dt == F64 else 2e-6)
    assert y1[:, t_out:].abs().max().item() == 0.0 if t_out < nfft else True
    loss = torch.empty((), dtype=dt, device=gpu)
    fn = _lib.lib().fl_mean_square_final_f32 if dt == torch.float32 else _lib.lib().fl_mean_square_final_f64
    _lib.check(fn(parts.data_ptr(), parts.numel(), 1.0 / y1.numel(), loss.data_ptr(), ops._stream()), "mean_square_final")
    assert abs(loss.item() / (want.item() / y1.numel()) - 1.0) < (1e-12 if dt == F64 else 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nfft,N", [("geq", 96000, 8), ("biquad", 48000, 4), ("svf", 4800, 2), ("geq", 2048, 16)])
def test_matrix_cascade_operator_float64(gpu, kind, nfft, N):
    """Series(Matrix, cascade filter) built with dtype=float64 (examples/e7_biquad.py:237's default): the fused pair
    operator in complex128 (fl_sos_response_rc_c128 / fl_geq_response_rc_c128 forward, fl_sos_response_bwd_rc_c128
    backward with the constant factor's partials in double) against the generic composition of the two responses --
    output and both parameter gradients -- and its response against the float64 oracle's."""
    from collections import OrderedDict
    from flamo_amd import ops
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    torch.manual_seed(7 + N)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=F64, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw)
    if kind == "geq":
        flt = dsp.GEQ(size=(N, N), **kw)
    elif kind == "biquad":
        flt = dsp.Biquad(size=(N, N), n_sections=3, filter_type="bandpass", **kw)
    else:
        flt = dsp.SVF(size=(N, N), n_sections=2, **kw)
    shell = system.Shell(system.Series(OrderedDict(mix=mat, flt=flt)), dsp.FFT(nfft, dtype=F64), dsp.iFFT(nfft, dtype=F64))
    params = [mat.param, flt.param]
    x = torch.randn(2, nfft, N, device=gpu, dtype=F64)

    def run():
        ops.kernel_timer.reset(True)
        y = shell(x)
        g = torch.autograd.grad(ops.mean_square(y), params)
        torch.cuda.synchronize()
        used = set(ops.kernel_timer.records)
        ops.kernel_timer.enabled = False
        return y.detach(), g, used

    y1, g1, used1 = run()
    assert "sos_response_rc" in used1 and "sos_response_bwd_rc" in used1, used1
    system.FUSE_MATRIX_CASCADE = False
    try:
        y2, g2, used2 = run()
    finally:
        system.FUSE_MATRIX_CASCADE = True
    assert "sos_response_bwd_rc" not in used2
    cc("y1", y1, y2, 1e-12)
    for a, b in zip(g1, g2):
        assert a.dtype == F64
        cc("a", a, b, 1e-10)
    if kind == "geq":
        spec = flt._cascade_spec(flt.param)
        W = mat.map(mat.param.detach())
        H = ops.geq_cascade_rc(spec[1], spec[2], W, flt._gamma_f, nfft, dtype=F64)
        assert H.dtype == torch.complex128
        Href = O.geq_response(flt.param.detach().cpu().double(), nfft, O.gamma_of(0.0, nfft, F64)) @ W.cpu().to(torch.complex128)
        cc("H", H.cpu(), Href, 1e-12, max_tol=1e-10)      # (max-norm: the response peaks at 2 while single bins sit near 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [F32, F64])
def test_graphed_step_gradient_buckets(gpu, dt):
    """GraphedStep(grad_buckets=True): the captured graph packs its parameter gradients into two flat buckets that alternate
    from replay to replay (fl_pack_toggle: the bucket is chosen from a counter kept on the device) -- after replay i bucket
    i & 1 holds that replay's gradients, the other one still the previous replay's; a parameter the step does not use has
    no view; the bucket index follows the replays however they are issued (replay() / __call__)."""
    from flamo_amd import ops
    from flamo_amd.graph import GraphedStep
    from flamo_amd.processor import dsp, system
    nfft, N = 4800, 4
    torch.manual_seed(3)
    kw = dict(nfft=nfft, alias_decay_db=0.0, device=gpu, dtype=dt, requires_grad=True)
    mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw)
    geq = dsp.GEQ(size=(N, N), **kw)
    model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
    unused = torch.nn.Parameter(torch.zeros(5, device=gpu, dtype=dt))
    params = [mat.param, unused, geq.param]
    xs = [torch.randn(2, nfft, N, device=gpu, dtype=dt) for _ in range(5)]
    step = GraphedStep(lambda xx: ops.mean_square(model(xx)), (xs[0],), params, grad_buckets=True)
    assert [p is unused for p in step.params] == [False, True, False]
    with pytest.raises(RuntimeError):
        step.bucket
    prev = None
    for i, x in enumerate(xs):
        loss = step(x) if i % 2 else (step.static_inputs[0].copy_(x), step.replay())[1]
        torch.cuda.synchronize()
        want = torch.autograd.grad(ops.mean_square(model(x)), [mat.param, geq.param])
        b = step.bucket
        assert b == i & 1
        v = step.bucket_views[b]
        assert v[1] is None and v[0].shape == mat.param.shape and v[2].shape == geq.param.shape
        assert torch.equal(v[0], mat.param.grad) and torch.equal(v[2], geq.param.grad)      # a copy of what the replay left
        cc("v_0", v[0], want[0], 1e-10 if dt == F64 else 0.0001)
        cc("v_2", v[2], want[1], 1e-10 if dt == F64 else 0.0001)
        cc("loss", loss, ops.mean_square(model(x)).detach(), 1e-05)
        if prev is not None:
            o = step.bucket_views[1 - b]
            assert torch.equal(o[0], prev[0]) and torch.equal(o[2], prev[1])                  # untouched by this replay
        prev = (v[0].clone(), v[2].clone())
    with pytest.raises(ValueError):
        GraphedStep(lambda xx: ops.mean_square(model(xx)), (xs[0],), [], grad_buckets=True)


@pytest.mark.gpu
def test_bucket_reducer_single_rank_over_rccl(gpu):
    """dist.BucketReducer with backend "nccl" (= RCCL) and one rank on the test box: replay, in-place asynchronous all-reduce of
    the bucket the replay filled, the next replay ordered behind the collective that last used ITS bucket; the sums of a
    one-rank job are the gradients themselves (tests/test_dist_rccl.py runs bench.py's multi-rank line through it on >= 2 GPUs)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import flamo_amd
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from collections import OrderedDict
from flamo_amd import dist as fd, ops
from flamo_amd.graph import GraphedStep
from flamo_amd.processor import dsp, system
nfft, N = 4800, 4
kw = dict(nfft=nfft, alias_decay_db=0.0, device=dev, dtype=torch.float32, requires_grad=True)
mat = dsp.Matrix(size=(N, N), matrix_type="random", **kw); geq = dsp.GEQ(size=(N, N), **kw)
model = system.Shell(system.Series(OrderedDict(mix=mat, eq=geq)), dsp.FFT(nfft), dsp.iFFT(nfft))
params = [mat.param, geq.param]
xs = [torch.randn(2, nfft, N, device=dev) for _ in range(6)]
step = GraphedStep(lambda xx: ops.mean_square(model(xx)), (xs[0],), params, grad_buckets=True)
red = fd.BucketReducer(step)
for i, x in enumerate(xs):
    loss = red.replay(x)
    if i %% 2 == 0:                  # every other step reads its sums; the others leave the collective in flight
        sums = red.reduced()
        torch.cuda.synchronize()
        want = torch.autograd.grad(ops.mean_square(model(x)), params)
        for s_, w_ in zip(sums, want):
            assert torch.allclose(s_, w_, rtol=1e-4, atol=1e-9), (i, (s_ - w_).abs().max())
red.finish()
torch.cuda.synchronize()
dist.destroy_process_group()
print("BUCKETS-OK")
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert "BUCKETS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("dt", [torch.float32, F64])
def test_lambda_gain_map_takes_the_folded_route(gpu, dt):
    """examples/e8_fdn.py:97 passes `map=lambda x: 20*torch.log10(torch.sigmoid(x))`: recognised by probe (dsp._gain_map_kind), the
    equaliser then runs the same kernels as with the named map -- bit-identical response and gradient -- and differs from the
    torch-op route (an unrecognised callable computing the same function) by rounding only."""
    from flamo_amd.processor import dsp
    nfft, N = 4800, 6
    outs = []
    for m in (dsp.db_of_sigmoid, lambda x: 20 * torch.log10(torch.sigmoid(x)), lambda x: 20 * torch.log10(torch.sigmoid(x)) + 0.0 * x):
        torch.manual_seed(7)
        att = dsp.parallelGEQ(size=(N,), nfft=nfft, alias_decay_db=30.0, requires_grad=True, device=gpu, dtype=dt)
        att.map = m
        att.assign_value(torch.randn(12, N, device=gpu, dtype=dt) * 0.3 + 2)
        H = att.freq_response(att.param)
        w = torch.randn(H.shape, device=gpu, dtype=dt, generator=torch.Generator(device=gpu).manual_seed(3))
        (g,) = torch.autograd.grad((H.real * w).sum() + (H.imag * w.flip(0)).sum(), [att.param])
        outs.append((H.detach(), g))
    assert dsp._gain_map_kind(dsp.db_of_sigmoid) == "sigmoid"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    tol = 1e-12 if dt == F64 else 2e-6
    cc("H_torch_route", outs[2][0], outs[0][0], tol)
    cc("g_torch_route", outs[2][1], outs[0][1], 1e-9 if dt == F64 else 2e-4)
