"""Round 3: the three full-size checks the round-2 review asked for (-m gpu), against the float64 oracle:
  * BASELINE configs[2] WITH the attenuation equaliser (parallelGEQ under 20 log10(sigmoid(x)), the folded fl_solve_fdn route) at
    nfft = 192000: output and every gradient;
  * one BASELINE configs[3] training step at nfft = 192000: the DatasetColorless batch (an impulse of M = 96001 samples), the
    |.| output layer, both criteria and every gradient;
  * BASELINE configs[4]'s core at nfft = 384000 on ~2000 sampled bins (the bins are independent; the oracle evaluates
    (I - F B)^-1 F G on the sample): core output and the gradients of an objective restricted to those bins."""
import math
from collections import OrderedDict

import pytest
import torch

from conftest import cc, check_close, check_closer, relerr

gpu_only = pytest.mark.gpu
F64 = torch.float64
PRIMES16 = [503, 593, 701, 811, 919, 1031, 1151, 1259, 1381, 1493, 1613, 1741, 1873, 2003, 2381, 2713]


@gpu_only
def test_fdn16_with_attenuation_full_size_all_gradients(gpu):
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    N, nfft, db = 16, 192000, 30.0
    torch.manual_seed(316)
    a = dict(in_gain=torch.randn(N, 1), out_gain=torch.randn(1, N), U_param=torch.randn(N, N), attn_param=torch.randn(12, N) * 0.3 + 2)
    a = {k: v.double() for k, v in a.items()}           # float32-representable values, held in float64
    delays_s = (torch.tensor(PRIMES16).double() / 48000 * 100).float().double()
    x = torch.zeros(1, nfft, 1, dtype=F64)
    x[:, 0] = 1
    c = torch.randn(1, nfft, 1).double()
    keys = ["in_gain", "out_gain", "U_param", "attn_param"]
    lv = {k: a[k].clone().requires_grad_(True) for k in keys}
    yref = O.fdn_forward(x, lv["in_gain"], lv["out_gain"], lv["U_param"], delays_s, nfft, db, attn_param=lv["attn_param"],
                         attn_map=lambda p: 20 * torch.log10(torch.sigmoid(p)))
    gref = torch.autograd.grad(torch.sum(yref * c), [lv[k] for k in keys])
    # the float64 backward of the same function (float32 section values, no float32 graph behind them): the yardstick for the
    # equaliser gains, whose gradient in the reference's arithmetic carries ~1e-4 of float32 noise
    lt = {k: a[k].clone().requires_grad_(True) for k in keys}
    yt = O.fdn_forward(x, lt["in_gain"], lt["out_gain"], lt["U_param"], delays_s, nfft, db, attn_param=lt["attn_param"],
                       attn_map=lambda p: 20 * torch.log10(torch.sigmoid(p)), geq_exact=True)
    (g_attn_true,) = torch.autograd.grad(torch.sum(yt * c), [lt["attn_param"]])
    dt = torch.float32
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=dt)
    ig, og = dsp.Gain(size=(N, 1), requires_grad=True, **kw), dsp.Gain(size=(1, N), requires_grad=True, **kw)
    dl = dsp.parallelDelay(size=(N,), max_len=max(PRIMES16), isint=True, **kw)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
    att.map = dsp.db_of_sigmoid                          # by name: folded into the design kernels, the route the benchmark times
    for mod, key in ((ig, "in_gain"), (og, "out_gain"), (mix, "U_param"), (att, "attn_param")):
        mod.assign_value(a[key].to(gpu, dt))
    dl.assign_value(delays_s.to(gpu, dt))
    fb = system.Series(OrderedDict(mixing_matrix=mix, attenuation=att))
    core = system.Series(OrderedDict(input_gain=ig, feedback_loop=system.Recursion(fF=dl, fB=fb), output_gain=og))
    model = system.Shell(core, dsp.FFT(nfft, dtype=dt), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, device=gpu, dtype=dt))
    y = model(x.to(gpu, dt))
    g = torch.autograd.grad(torch.sum(y * c.to(gpu, dt)), [ig.param, og.param, mix.param, att.param])
    check_close("fdn16_attn_full/y", y.detach().cpu(), yref.detach(), 1e-5)
    for gi, gr, k in zip(g, gref, keys):
        # equaliser gains: float32 section buffers in the reference (and the oracle) -- a flat 1e-3, and the recorded achieved
        # error (tests/golden/achieved_errors.json) times five on top of it
        check_close(f"fdn16_attn_full/g_{k}", gi.cpu(), gr, 1e-3 if k == "attn_param" else 1e-5)
        if k == "attn_param":
            check_closer("fdn16_attn_full/g_attn_param_vs_float64_backward", gi.cpu(), gr, g_attn_true, 1e-5)


@gpu_only
def test_colorless_training_step_full_size(gpu):
    from tools import train_colorless_fdn as T
    from flamo_amd import ops
    from oracle import hotpath as O
    N, nfft, db = 16, 192000, 30.0
    M = nfft // 2 + 1
    torch.manual_seed(416)
    a = dict(in_gain=torch.randn(N, 1).double(), out_gain=torch.randn(1, N).double(), U_param=torch.randn(N, N).double())
    delays_s = (torch.tensor(PRIMES16).double() / 48000 * 100).float().double()
    x = torch.zeros(1, M, 1, dtype=F64)                  # DatasetColorless: an impulse of M samples, target ones
    x[:, 0] = 1
    target = torch.ones(1, M, 1, dtype=F64)
    ps = [a[k].clone().requires_grad_(True) for k in ("in_gain", "out_gain", "U_param")]
    est_ref = O.fdn_forward(x, ps[0], ps[1], ps[2], delays_s, nfft, db, output="abs")
    mse_ref = torch.mean((est_ref.sum(-1) - target.squeeze(-1)) ** 2)
    sp_ref = -(torch.sum(torch.abs(O.orthogonal(ps[2]))) - N * math.sqrt(N)) / (N * (math.sqrt(N) - 1))
    gref = torch.autograd.grad(mse_ref + 0.2 * sp_ref, ps)
    dt = torch.float32
    model = T.build(gpu, dt, N, nfft, db, PRIMES16)
    core = model.get_core()
    core.input_gain.assign_value(a["in_gain"].to(gpu, dt))
    core.output_gain.assign_value(a["out_gain"].to(gpu, dt))
    core.feedback_loop.feedback.assign_value(a["U_param"].to(gpu, dt))
    params = [core.input_gain.param, core.output_gain.param, core.feedback_loop.feedback.param]
    with ops.step_scope():
        est = model(x.to(gpu, dt))
        mse, sp = T.mse_criterion(est, target.to(gpu, dt)), T.sparsity_criterion(model)
        g = torch.autograd.grad(mse + 0.2 * sp, params)
    assert est.shape == (1, M, 1)
    cc("est", est.detach().cpu(), est_ref.detach(), 1e-05)
    assert abs(mse.item() - mse_ref.item()) < 1e-5 * abs(mse_ref.item()) + 1e-12
    assert abs(sp.item() - sp_ref.item()) < 1e-5 * abs(sp_ref.item()) + 1e-12
    for gi, gr, k in zip(g, gref, ("in_gain", "out_gain", "U_param")):
        cc("gi", gi.cpu(), gr, 2e-05)


def _config5_params(N, max_len=2000):
    g = torch.Generator().manual_seed(130709)
    m = torch.randint(1, max_len, (N, N), generator=g).double()
    return dict(geq=(torch.rand(12, N, N, generator=g) * (10 ** (6 / 20) - 10 ** (-6 / 20)) + 10 ** (-6 / 20)).float().double(),
                delay_s=(m / 48000 * 100).float().double(),            # seconds * unit, as Delay stores them
                gain=(torch.rand(N, generator=g) * 0.5 / N ** 0.5 + 0.01).float().double(),
                U=torch.randn(N, N, generator=g).float().double())


@gpu_only
def test_config5_core_full_size_on_sampled_bins(gpu):
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    N, nfft, db, nb = 32, 384000, 30.0, 2000
    M = nfft // 2 + 1
    a = _config5_params(N)
    gen = torch.Generator().manual_seed(5384)
    bins = torch.unique(torch.cat([torch.tensor([0, 1, 2, M - 2, M - 1]), torch.randint(0, M, (nb,), generator=gen)]))
    X = torch.randn(1, M, N, generator=gen, dtype=F64) + 1j * torch.randn(1, M, N, generator=gen, dtype=F64)
    X = X.to(torch.complex64).to(torch.complex128)       # float32-representable
    C = torch.randn(1, len(bins), N, generator=gen, dtype=F64) + 1j * torch.randn(1, len(bins), N, generator=gen, dtype=F64)
    # ---- oracle on the sample
    lv = [t.clone().requires_grad_(True) for t in (a["geq"], a["gain"], a["U"])]
    gamma = O.gamma_of(db, nfft, F64)
    G = O.geq_response_at(lv[0], nfft, gamma, bins)                                          # (nb, N, N)
    md = O.delay_samples(a["delay_s"], 48000, 100, True)
    F = O.to_complex(lv[1]).view(1, N, 1) * O.delay_response_at(md, nfft, gamma, bins)      # Series(Delay, parallelGain): diag(g) D
    Bk = O.to_complex(O.orthogonal(lv[2])).unsqueeze(0).expand(len(bins), N, N)
    Xs = O.mimo_full(G, X[:, bins])
    Yref = O.recursion_at(F, Bk, Xs)                                                         # (1, nb, N)
    gref = torch.autograd.grad(torch.sum(torch.real(Yref * torch.conj(C))), lv)
    # (the equaliser gains' yardstick: float64 backward behind the same float32 section values)
    lt = [t.clone().requires_grad_(True) for t in (a["geq"], a["gain"], a["U"])]
    Ft = O.to_complex(lt[1]).view(1, N, 1) * O.delay_response_at(md, nfft, gamma, bins)
    Yt = O.recursion_at(Ft, O.to_complex(O.orthogonal(lt[2])).unsqueeze(0).expand(len(bins), N, N),
                        O.mimo_full(O.geq_response_at(lt[0], nfft, gamma, bins, exact=True), X[:, bins]))
    (g_geq_true,) = torch.autograd.grad(torch.sum(torch.real(Yt * torch.conj(C))), [lt[0]])
    # ---- the HIP path: the whole core at full size, the objective restricted to the sampled bins
    dt = torch.float32
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=dt)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    dly = dsp.Delay(size=(N, N), max_len=2000, isint=True, **kw)
    gain = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    geq.assign_value(a["geq"].to(gpu, dt))
    dly.assign_value(a["delay_s"].to(gpu, dt))
    gain.assign_value(a["gain"].to(gpu, dt))
    mix.assign_value(a["U"].to(gpu, dt))
    core = system.Series(OrderedDict(eq=geq, loop=system.Recursion(fF=system.Series(OrderedDict(d=dly, g=gain)), fB=mix)))
    Y = core(X.to(gpu, torch.complex64))
    assert Y.shape == (1, M, N)
    Ysel = Y[:, bins.to(gpu)]
    g = torch.autograd.grad(torch.sum(torch.real(Ysel * torch.conj(C.to(gpu, torch.complex64)))), [geq.param, gain.param, mix.param])
    check_close("config5_core_full/Ysel", Ysel.detach().cpu(), Yref.detach(), 1e-5)
    for gi, gr, k in zip(g, gref, ("g_geq", "g_gain", "g_U")):
        # equaliser gains: float32 section buffers in the reference (and the oracle); see check_close for the recorded bound
        check_close(f"config5_core_full/{k}", gi.cpu(), gr, 1e-3 if k == "g_geq" else 3e-5)
        if k == "g_geq":
            check_closer("config5_core_full/g_geq_vs_float64_backward", gi.cpu(), gr, g_geq_true, 1e-5)


@gpu_only
@pytest.mark.parametrize("dt,N", [(torch.float32, 80), (torch.float32, 138), (torch.float64, 40), (torch.float64, 97),
                                  (torch.float32, 150), (torch.float64, 110), (torch.float32, 257)])
def test_recursion_beyond_the_register_resident_sizes(gpu, dt, N):
    """Loops larger than a wavefront's lanes (64 channels in float32, 32 in float64) go through the LDS solve kernel, loops larger
    than the LDS (138 / 97) through the workspace form of the same kernel (fl_solve_ws_*): output and
    gradients of Recursion(parallelDelay * parallelGain, orthogonal Matrix) against torch.linalg.solve in float64 (system.py:397-425)."""
    from flamo_amd.processor import dsp, system
    from oracle import hotpath as O
    nfft, db = 240, 30.0
    M = nfft // 2 + 1
    torch.manual_seed(N)
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=dt)
    dl = dsp.parallelDelay(size=(N,), max_len=40, isint=True, **kw)
    gn = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    with torch.no_grad():
        gn.param.copy_(torch.rand(N, device=gpu, dtype=dt) * 0.5 + 0.2)
    rec = system.Recursion(fF=system.Series(OrderedDict(d=dl, g=gn)), fB=mix)
    cd = torch.complex64 if dt == torch.float32 else torch.complex128
    X = torch.randn(2, M, N, device=gpu, dtype=cd)
    C = torch.randn(2, M, N, device=gpu, dtype=cd)
    Y = rec(X)
    g = torch.autograd.grad(torch.sum(torch.real(Y * torch.conj(C))), [gn.param, mix.param])
    # oracle
    lv = [gn.param.detach().cpu().double().requires_grad_(True), mix.param.detach().cpu().double().requires_grad_(True)]
    gamma = O.gamma_of(db, nfft, F64)
    m = O.delay_samples(dl.param.detach().cpu().double(), 48000, 100, True)
    F = torch.diag_embed(O.to_complex(lv[0]).view(1, N) * O.delay_response(m, nfft, gamma))
    Bk = O.to_complex(O.orthogonal(lv[1])).unsqueeze(0).expand(M, N, N)
    Yr = O.recursion(F, Bk, X.cpu().to(torch.complex128))
    gr = torch.autograd.grad(torch.sum(torch.real(Yr * torch.conj(C.cpu().to(torch.complex128)))), lv)
    tol = 1e-9 if dt == torch.float64 else 2e-5
    cc("Y", Y.detach().cpu(), Yr.detach(), tol)
    for gi, gj, k in zip(g, gr, ("g_gain", "g_U")):
        cc("gi", gi.cpu(), gj, 10 * tol)


@gpu_only
@pytest.mark.parametrize("structure", ["fdn", "chain"])
def test_recursion_external_parameters_take_the_fused_loop_routes(gpu, structure):
    """`ext_param` routing of Recursion (system.py:409-415: keys containing "feedback" / "feedforward") through the fused loop forms:
    the result and the gradients w.r.t. the external tensors equal those of a twin model that owns the same values as parameters,
    and equal the generic identity-probing route (FUSE_SERIES off)."""
    from flamo_amd.processor import dsp, system
    N, nfft, db = 8, 480, 30.0
    M = nfft // 2 + 1
    dt = torch.float64
    kw = dict(nfft=nfft, alias_decay_db=db, device=gpu, dtype=dt)

    def build():
        torch.manual_seed(7)
        mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
        att = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
        with torch.no_grad():
            att.param.copy_(torch.rand(N, device=gpu, dtype=dt) * 0.4 + 0.3)
        if structure == "fdn":
            dl = dsp.parallelDelay(size=(N,), max_len=60, isint=True, **kw)
            ff = dl
        else:
            dl = dsp.Delay(size=(N, N), max_len=60, isint=True, **kw)
            g = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
            with torch.no_grad():
                g.param.copy_(torch.rand(N, device=gpu, dtype=dt) * 0.2 / N ** 0.5 + 0.01)
            ff = system.Series(OrderedDict(d=dl, g=g))
        return system.Recursion(fF=ff, fB=system.Series(OrderedDict(mixing_matrix=mix, attenuation=att))), mix, att

    torch.manual_seed(11)
    U_ext = torch.randn(N, N, device=gpu, dtype=dt, requires_grad=True)
    a_ext = (torch.rand(N, device=gpu, dtype=dt) * 0.4 + 0.3).requires_grad_(True)
    X = torch.randn(2, M, N, device=gpu, dtype=torch.complex128)
    C = torch.randn(2, M, N, device=gpu, dtype=torch.complex128)
    obj = lambda Y: torch.sum(torch.real(Y * torch.conj(C)))      # noqa: E731
    # twin: owns the values
    twin, mix_t, att_t = build()
    mix_t.assign_value(U_ext.detach())
    att_t.assign_value(a_ext.detach())
    Yt = twin(X)
    gt = torch.autograd.grad(obj(Yt), [mix_t.param, att_t.param])
    # external parameters, fused routes
    rec, _, _ = build()
    ext = {"feedback": {"mixing_matrix": U_ext, "attenuation": a_ext}}
    Y = rec(X, ext)
    g = torch.autograd.grad(obj(Y), [U_ext, a_ext])
    cc("Y", Y.detach(), Yt.detach(), 1e-12)
    for gi, gj in zip(g, gt):
        cc("gi", gi, gj, 1e-10)
    # and the generic route
    system.FUSE_SERIES = False
    try:
        rec2, _, _ = build()
        Yg = rec2(X, ext)
        gg = torch.autograd.grad(obj(Yg), [U_ext, a_ext])
    finally:
        system.FUSE_SERIES = True
    cc("Y", Y.detach(), Yg.detach(), 1e-10)
    for gi, gj in zip(g, gg):
        cc("gi", gi, gj, 1e-09)
