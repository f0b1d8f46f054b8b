"""Generate golden vectors by running the REFERENCE (float64, CPU) in this container.

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference never travels to the GPU box; these small fixtures do.  Each .npz holds the
inputs, parameters (as assigned), reference outputs and reference gradients of one case,
plus a JSON ``meta`` string describing how to rebuild the same module with the drop-in API.
Gradients are those of  L = sum(Re(Y * conj(C)))  (complex Y) or  L = sum(y * c)  (real y) for
a stored random cotangent C, i.e. autograd's vector-Jacobian product with cotangent C.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
F64 = torch.float64
C128 = torch.complex128


def npy(t):
    return t.detach().cpu().numpy()


def save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: (npy(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    if name.startswith("r2_"):     # lossless: values that ARE float32 numbers are stored as float32 (tests upcast)
        for k, v in list(arrays.items()):
            if v.dtype == np.float64 and np.array_equal(v.astype(np.float32).astype(np.float64), v):
                arrays[k] = v.astype(np.float32)
            elif v.dtype == np.complex128 and np.array_equal(v.astype(np.complex64).astype(np.complex128), v):
                arrays[k] = v.astype(np.complex64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **arrays)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"{name:40s} {sz/1024:8.1f} KiB")


def crandn(*shape):
    return torch.complex(torch.randn(*shape, dtype=F64), torch.randn(*shape, dtype=F64))


def vjp_complex(Y, C, wrt):
    L = torch.sum(torch.real(Y * torch.conj(C)))
    return torch.autograd.grad(L, wrt, allow_unused=True)


# ----------------------------------------------------------------------------- transforms
def gen_transforms(dsp):
    cases = [
        # nfft, T, norm, alias_db, channels-shape
        (96, 96, "backward", None, (3,)),
        (96, 49, "ortho", None, (3,)),
        (96, 96, "forward", 30.0, (3,)),
        (96, 103, "forward", None, (3,)),
        (250, 250, "backward", 30.0, (2,)),
        (250, 126, "backward", None, (2,)),
        (960, 960, "backward", None, (1,)),
        (960, 960, "backward", 30.0, (1,)),
        (1500, 1500, "ortho", 30.0, (1,)),
        (64, 64, "backward", None, (2, 2)),   # trailing dims (identity-probe shape)
        (2048, 2048, "backward", 60.0, (1,)),
    ]
    for i, (nfft, T, norm, db, ch) in enumerate(cases):
        torch.manual_seed(1000 + i)
        x = torch.randn(2 if nfft < 900 else 1, T, *ch, dtype=F64, requires_grad=True)
        if db is None:
            f, g = dsp.FFT(nfft, norm=norm, dtype=F64), dsp.iFFT(nfft, norm=norm, dtype=F64)
        else:
            f = dsp.FFTAntiAlias(nfft, norm=norm, alias_decay_db=db, dtype=F64)
            g = dsp.iFFTAntiAlias(nfft, norm=norm, alias_decay_db=db, dtype=F64)
        X = f(x)
        C = crandn(*X.shape)
        (gx,) = vjp_complex(X, C, [x])
        # inverse on an arbitrary (non-Hermitian-consistent) spectrum, like a processed one
        Z = crandn(*X.shape).requires_grad_(True)
        y = g(Z)
        c = torch.randn(*y.shape, dtype=F64)
        (gZ,) = torch.autograd.grad(torch.sum(y * c), [Z])
        save(f"fft_{i:02d}", dict(kind="transform", nfft=nfft, T=T, norm=norm, alias_decay_db=db),
             x=x, X=X, C=C, gx=gx, Z=Z, y=y, c=c, gZ=gZ)


# ----------------------------------------------------------------------------- single modules
def module_case(name, mod, meta, nin, nfft, extra_identity=True):
    """Run a module on a vector signal (2,M,nin) and a matrix signal (1,M,nin,nin)."""
    M = nfft // 2 + 1
    arrays = dict(param=mod.param.detach().clone())
    try:
        Hresp = mod.freq_response(mod.param)
        if isinstance(Hresp, torch.Tensor):
            arrays["freq_response"] = Hresp
    except Exception:
        pass
    X = crandn(2, M, nin).requires_grad_(True)
    Y = mod(X)
    C = crandn(*Y.shape)
    wrt = [X] + ([mod.param] if mod.param.requires_grad else [])
    g = vjp_complex(Y, C, wrt)
    arrays.update(X=X, Y=Y, C=C, gX=g[0])
    if mod.param.requires_grad and g[1] is not None:
        arrays["gparam"] = g[1]
    if extra_identity:
        X4 = crandn(1, M, nin, nin)
        arrays.update(X4=X4, Y4=mod(X4))
    save(name, meta, **arrays)


def gen_modules(dsp):
    nfft = 96
    for db in (0.0, 30.0):
        tag = f"db{int(db)}"
        kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
        torch.manual_seed(2000 + int(db))
        module_case(f"gain_{tag}", dsp.Gain(size=(3, 2), requires_grad=True, **kw),
                    dict(cls="Gain", kwargs=dict(size=[3, 2], requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"pgain_{tag}", dsp.parallelGain(size=(3,), requires_grad=True, **kw),
                    dict(cls="parallelGain", kwargs=dict(size=[3], requires_grad=True), nfft=nfft, alias_decay_db=db), 3, nfft)
        module_case(f"matrix_orth_{tag}", dsp.Matrix(size=(4, 4), matrix_type="orthogonal", requires_grad=True, **kw),
                    dict(cls="Matrix", kwargs=dict(size=[4, 4], matrix_type="orthogonal", requires_grad=True), nfft=nfft, alias_decay_db=db), 4, nfft)
        module_case(f"filter_{tag}", dsp.Filter(size=(5, 3, 2), requires_grad=True, **kw),
                    dict(cls="Filter", kwargs=dict(size=[5, 3, 2], requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"pfilter_{tag}", dsp.parallelFilter(size=(4, 3), requires_grad=True, **kw),
                    dict(cls="parallelFilter", kwargs=dict(size=[4, 3], requires_grad=True), nfft=nfft, alias_decay_db=db), 3, nfft)
        for ft in ("lowpass", "highpass", "bandpass"):
            module_case(f"biquad_{ft}_{tag}",
                        dsp.Biquad(size=(3, 2), n_sections=2, filter_type=ft, requires_grad=True, **kw),
                        dict(cls="Biquad", kwargs=dict(size=[3, 2], n_sections=2, filter_type=ft, requires_grad=True),
                             nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"pbiquad_{tag}",
                    dsp.parallelBiquad(size=(3,), n_sections=2, filter_type="highpass", requires_grad=True, **kw),
                    dict(cls="parallelBiquad", kwargs=dict(size=[3], n_sections=2, filter_type="highpass", requires_grad=True),
                         nfft=nfft, alias_decay_db=db), 3, nfft)
        module_case(f"geq_{tag}", dsp.GEQ(size=(2, 2), requires_grad=True, **kw),
                    dict(cls="GEQ", kwargs=dict(size=[2, 2], requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"pgeq_{tag}", dsp.parallelGEQ(size=(3,), requires_grad=True, **kw),
                    dict(cls="parallelGEQ", kwargs=dict(size=[3], requires_grad=True), nfft=nfft, alias_decay_db=db), 3, nfft)
        for isint in (True, False):
            it = "int" if isint else "frac"
            module_case(f"delay_{it}_{tag}", dsp.Delay(size=(3, 2), max_len=40, isint=isint, **kw),
                        dict(cls="Delay", kwargs=dict(size=[3, 2], max_len=40, isint=isint), nfft=nfft, alias_decay_db=db), 2, nfft)
            module_case(f"pdelay_{it}_{tag}", dsp.parallelDelay(size=(3,), max_len=40, isint=isint, **kw),
                        dict(cls="parallelDelay", kwargs=dict(size=[3], max_len=40, isint=isint), nfft=nfft, alias_decay_db=db), 3, nfft)
        # learnable fractional delay (softplus map, dsp.py:3418-3419)
        module_case(f"delay_learn_{tag}", dsp.Delay(size=(2, 2), max_len=30, isint=False, requires_grad=True, **kw),
                    dict(cls="Delay", kwargs=dict(size=[2, 2], max_len=30, isint=False, requires_grad=True),
                         nfft=nfft, alias_decay_db=db), 2, nfft)

    # float32-mode GEQ coefficients quirk (F8): record the float64-mode SOS arrays themselves
    torch.manual_seed(2100)
    geq = dsp.GEQ(size=(2, 2), nfft=nfft, alias_decay_db=30.0, dtype=F64)
    H, Bf, Af = geq.get_poly_coeff(geq.map(geq.param))
    save("geq_quirk", dict(kind="geq_quirk", nfft=nfft, alias_decay_db=30.0),
         param=geq.param.detach(), H=H, B=Bf, A=Af, center_freq=geq.center_freq, shelving=geq.shelving_crossover)


def gen_modules_more(dsp):
    """SOS-coefficient, state-variable and parametric-EQ filters, gain+delay."""
    nfft = 96
    for db in (0.0, 30.0):
        tag = f"db{int(db)}"
        kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
        torch.manual_seed(2500 + int(db))
        sos = dsp.SOSFilter(size=(3, 2), n_sections=2, **kw)
        sos.assign_value(torch.randn(2, 6, 3, 2, dtype=F64) * 0.3 + torch.tensor([1.0, 0, 0, 1.5, 0, 0], dtype=F64).view(1, 6, 1, 1))
        module_case(f"sosfilter_{tag}", sos, dict(cls="SOSFilter", kwargs=dict(size=[3, 2], n_sections=2), nfft=nfft, alias_decay_db=db), 2, nfft)
        psos = dsp.parallelSOSFilter(size=(3,), n_sections=2, normalize_a0=False, **kw)
        psos.assign_value(torch.randn(2, 6, 3, dtype=F64) * 0.3 + torch.tensor([1.0, 0, 0, 1.5, 0, 0], dtype=F64).view(1, 6, 1))
        module_case(f"psosfilter_{tag}", psos, dict(cls="parallelSOSFilter", kwargs=dict(size=[3], n_sections=2, normalize_a0=False), nfft=nfft, alias_decay_db=db), 3, nfft)
        for ft in ("lowpass", "lowshelf", "peaking", None):
            module_case(f"svf_{ft}_{tag}", dsp.SVF(size=(3, 2), n_sections=2, filter_type=ft, requires_grad=True, **kw),
                        dict(cls="SVF", kwargs=dict(size=[3, 2], n_sections=2, filter_type=ft, requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"psvf_{tag}", dsp.parallelSVF(size=(3,), n_sections=2, filter_type="highshelf", requires_grad=True, **kw),
                    dict(cls="parallelSVF", kwargs=dict(size=[3], n_sections=2, filter_type="highshelf", requires_grad=True), nfft=nfft, alias_decay_db=db), 3, nfft)
        for design in ("biquad", "svf"):
            module_case(f"peq_{design}_{tag}", dsp.PEQ(size=(2, 2), n_bands=5, design=design, requires_grad=True, **kw),
                        dict(cls="PEQ", kwargs=dict(size=[2, 2], n_bands=5, design=design, requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        for isint in (True, False):
            it = "int" if isint else "frac"
            module_case(f"gaindelay_{it}_{tag}", dsp.GainDelay(size=(3, 2), max_len=40, isint=isint, requires_grad=True, **kw),
                        dict(cls="GainDelay", kwargs=dict(size=[3, 2], max_len=40, isint=isint, requires_grad=True), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"pgaindelay_{tag}", dsp.parallelGainDelay(size=(3,), max_len=40, isint=True, **kw),
                    dict(cls="parallelGainDelay", kwargs=dict(size=[3], max_len=40, isint=True), nfft=nfft, alias_decay_db=db), 3, nfft)


# ----------------------------------------------------------------------------- config-2 miniature
def gen_config2(dsp, system):
    for db, nfft, B in ((0.0, 240, 2), (30.0, 250, 2)):
        torch.manual_seed(3000 + int(db))
        N = 8
        kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
        mat = dsp.Matrix(size=(N, N), matrix_type="random", requires_grad=True, **kw)
        geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
        core = system.Series(OrderedDict({"mix": mat, "eq": geq}))
        model = system.Shell(core=core, input_layer=dsp.FFT(nfft, dtype=F64), output_layer=dsp.iFFT(nfft, dtype=F64))
        x = torch.randn(B, nfft, N, dtype=F64, requires_grad=True)
        y = model(x)
        loss = (y ** 2).mean()
        gx, gW, gG = torch.autograd.grad(loss, [x, mat.param, geq.param])
        keys = list(model.state_dict().keys())
        save(f"config2_db{int(db)}", dict(kind="config2", nfft=nfft, alias_decay_db=db, N=N, B=B, state_keys=keys),
             x=x, W=mat.param, geq_param=geq.param, y=y, loss=loss, gx=gx, gW=gW, gG=gG)


# ----------------------------------------------------------------------------- FDN / Recursion
def build_fdn(dsp, system, N, nfft, db, delays, with_attn, dtype=F64, out_layer="ifft_aa"):
    kw = dict(nfft=nfft, alias_decay_db=db, dtype=dtype)
    input_gain = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
    output_gain = dsp.Gain(size=(1, N), requires_grad=True, **kw)
    dl = dsp.parallelDelay(size=(N,), max_len=int(max(delays)), isint=True, requires_grad=False, **kw)
    dl.assign_value(dl.sample2s(torch.tensor(delays, dtype=dtype)))
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    if with_attn:
        att = dsp.parallelGEQ(size=(N,), octave_interval=1, fs=48000, requires_grad=True, **kw)
        att.map = lambda x: 20 * torch.log10(torch.sigmoid(x))
        fb = system.Series(OrderedDict({"mixing_matrix": mix, "attenuation": att}))
    else:
        att = None
        fb = mix
    rec = system.Recursion(fF=dl, fB=fb)
    core = system.Series(OrderedDict({"input_gain": input_gain, "feedback_loop": rec, "output_gain": output_gain}))
    if out_layer == "ifft_aa":
        ol = dsp.iFFTAntiAlias(nfft=nfft, alias_decay_db=db, dtype=dtype)
    else:
        ol = dsp.Transform(lambda x: torch.abs(x), dtype=dtype)
    model = system.Shell(core=core, input_layer=dsp.FFT(nfft, dtype=dtype), output_layer=ol)
    return model, dict(input_gain=input_gain, output_gain=output_gain, delays=dl, mix=mix, att=att, rec=rec)


def gen_fdn(dsp, system):
    specs = [
        # name, N, nfft, db, delays, attenuation, B, signal
        ("fdn4", 4, 512, 30.0, [101, 157, 211, 263], False, 1, "impulse"),
        ("fdn6", 6, 480, 30.0, [59, 97, 131, 151, 163, 169], True, 2, "wgn"),
        ("fdn6_db0", 6, 480, 0.0, [59, 97, 131, 151, 163, 169], True, 1, "impulse"),
        ("fdn16", 16, 1500, 30.0, [53, 61, 71, 79, 89, 97, 103, 109, 127, 137, 149, 157, 167, 179, 191, 199], True, 1, "impulse"),
    ]
    for name, N, nfft, db, delays, attn, B, sig in specs:
        torch.manual_seed(4000 + N + int(db))
        model, p = build_fdn(dsp, system, N, nfft, db, delays, attn)
        with torch.no_grad():  # keep the loop well inside the unit circle
            if attn:
                p["att"].param.copy_(torch.randn_like(p["att"].param) * 0.3 + 2.0)
        if sig == "impulse":
            x = torch.zeros(B, nfft, 1, dtype=F64)
            x[:, 0, :] = 1
        else:
            x = torch.randn(B, nfft, 1, dtype=F64)
        x.requires_grad_(True)
        y = model(x)
        c = torch.randn(*y.shape, dtype=F64)
        plist = [p["input_gain"].param, p["output_gain"].param, p["mix"].param] + ([p["att"].param] if attn else [])
        grads = torch.autograd.grad(torch.sum(y * c), [x] + plist)
        arrays = dict(x=x, y=y, c=c, gx=grads[0], in_gain=plist[0], out_gain=plist[1], U_param=plist[2],
                      g_in_gain=grads[1], g_out_gain=grads[2], g_U_param=grads[3],
                      delays_s=p["delays"].param.detach())
        if attn:
            arrays.update(attn_param=plist[3], g_attn_param=grads[4])
        # frequency-domain core output for a complex spectrum (vector RHS) and the closed-loop
        # matrices of a few bins (A as the reference builds it)
        core = model.get_core()
        M = nfft // 2 + 1
        with torch.no_grad():
            Xf = crandn(B, M, 1)
            arrays.update(Xf=Xf, Yf=core(Xf))
            rec = p["rec"]
            I = rec.I.unsqueeze(0)
            A = I - rec.feedforward(rec.feedback(I))
            sel = [0, 1, M // 3, M - 1]
            arrays.update(A_bins=np.array(sel), A_sel=A[0, sel])
            # matrix RHS (identity path) through the recursion alone
            Xm = crandn(1, M, N, N)
            if N <= 4:
                arrays.update(Xm=Xm, Ym=rec(Xm))
            # responses
            arrays.update(ir=model.get_time_response(identity=False), fr=model.get_freq_response(identity=False))
            # analytic probe on 16 bins (examples/e10_probe.py) -- independent of the FFT
            ks = np.linspace(0, M - 1, 16).astype(int)
            pr = []
            for k in ks:
                z = torch.tensor(np.exp(2j * np.pi * k / nfft), dtype=C128)
                pr.append(model.probe(z).reshape(-1))
            arrays.update(probe_bins=ks, probe=torch.stack(pr))
        keys = list(model.state_dict().keys())
        save(name, dict(kind="fdn", N=N, nfft=nfft, alias_decay_db=db, delays=delays, attn=attn, B=B, state_keys=keys),
             **arrays)

    # identity responses of a 2-in/2-out recursion-bearing system (get_*_response(identity=True))
    torch.manual_seed(4500)
    nfft, N, db = 480, 3, 30.0
    kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
    dl = dsp.parallelDelay(size=(N,), max_len=60, isint=True, **kw)
    dl.assign_value(dl.sample2s(torch.tensor([23.0, 41.0, 59.0], dtype=F64)))
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", **kw)
    att = dsp.parallelGain(size=(N,), **kw)
    att.assign_value(torch.tensor([0.9, 0.85, 0.8], dtype=F64))
    rec = system.Recursion(fF=dl, fB=system.Series(OrderedDict({"mix": mix, "att": att})))
    model = system.Shell(core=rec)
    with torch.no_grad():
        save("rec3_identity", dict(kind="rec_identity", N=N, nfft=nfft, alias_decay_db=db, delays=[23, 41, 59]),
             U_param=mix.param, att=att.param,
             ir=model.get_time_response(identity=True), fr=model.get_freq_response(identity=True),
             ir_vec=model.get_time_response(identity=False))


# ----------------------------------------------------------------------------- AccurateGEQ
def gen_accurate_geq(dsp):
    """AccurateGEQ / parallelAccurateGEQ (dsp.py:3002-3220): target gains fitted by L-BFGS, then the SOS tail."""
    nfft = 96
    for db in (0.0, 30.0):
        tag = f"db{int(db)}"
        kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
        torch.manual_seed(4500 + int(db))
        module_case(f"accgeq_{tag}", dsp.AccurateGEQ(size=(2, 2), **kw),
                    dict(cls="AccurateGEQ", kwargs=dict(size=[2, 2]), nfft=nfft, alias_decay_db=db), 2, nfft)
        module_case(f"paccgeq_{tag}", dsp.parallelAccurateGEQ(size=(3,), **kw),
                    dict(cls="parallelAccurateGEQ", kwargs=dict(size=[3]), nfft=nfft, alias_decay_db=db), 3, nfft)


# ----------------------------------------------------------------------------- Parallel
def gen_parallel(dsp, system):
    """system.Parallel (system.py:570-772): two branches on one input, summed and concatenated."""
    nfft, B = 128, 2
    M = nfft // 2 + 1
    for db in (0.0, 30.0):
        for sum_output in (True, False):
            torch.manual_seed(4000 + int(db) + int(sum_output))
            kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64, requires_grad=True)
            g = dsp.Gain(size=(3, 2), **kw)
            pg = dsp.parallelGain(size=(3,), **kw)
            fir = dsp.Filter(size=(5, 3, 2), **kw)
            par = system.Parallel(brA=OrderedDict(g=g, pg=pg), brB=fir, sum_output=sum_output)
            X = crandn(B, M, 2).requires_grad_(True)
            Y = par(X)
            C = crandn(*Y.shape)
            gr = vjp_complex(Y, C, [X, g.param, pg.param, fir.param])
            save(f"parallel_{'sum' if sum_output else 'cat'}_db{int(db)}",
                 dict(kind="parallel", nfft=nfft, alias_decay_db=db, sum_output=sum_output,
                      input_channels=par.input_channels, output_channels=par.output_channels),
                 X=X, Y=Y, C=C, g=g.param, pg=pg.param, fir=fir.param, gX=gr[0], gg=gr[1], gpg=gr[2], gfir=gr[3])


# ----------------------------------------------------------------------------- config 4: colorless FDN training
def gen_colorless(dsp, system):
    """examples/e8_colorless_fdn.py in miniature: FDN with an orthogonal feedback matrix, |.| output layer,
    DatasetColorless batches (impulse of M samples -> flat magnitude), criteria mse_loss + 0.2 sparsity_loss
    (flamo/optimize/loss.py:12-103), Adam steps as Trainer.train_step does them (trainer.py:70, 162-191)."""
    from flamo.optimize.loss import mse_loss, sparsity_loss
    from flamo.optimize.dataset import DatasetColorless
    specs = [("colorless6", 6, 480, [59, 97, 131, 151, 163, 169], 2, 1e-2, 8),
             ("colorless16", 16, 1500, [53, 61, 71, 79, 89, 97, 103, 109, 127, 137, 149, 157, 167, 179, 191, 199], 1, 1e-3, 5)]
    for name, N, nfft, delays, B, lr, steps in specs:
        torch.manual_seed(4400 + N)
        db = 30.0
        model, p = build_fdn(dsp, system, N, nfft, db, delays, False, out_layer="abs")
        M = nfft // 2 + 1
        ds = DatasetColorless(input_shape=(1, M, 1), target_shape=(1, M, 1), expand=B, device="cpu", dtype=F64)
        x = torch.stack([ds[i][0] for i in range(B)])
        tgt = torch.stack([ds[i][1] for i in range(B)])
        plist = [p["input_gain"].param, p["output_gain"].param, p["mix"].param]
        init = [q.detach().clone() for q in plist]
        crit = [(1.0, mse_loss(nfft=nfft), False), (0.2, sparsity_loss(), True)]
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        log, g0 = [], None
        for it in range(steps):
            opt.zero_grad()
            est = model(x)
            parts = [c(est, tgt, model) if rm else c(est, tgt) for _, c, rm in crit]
            loss = sum(a * t for (a, _, _), t in zip(crit, parts))
            loss.backward()
            if it == 0:
                g0 = [q.grad.detach().clone() for q in plist]
                est0 = est.detach().clone()
            opt.step()
            log.append([float(parts[0]), float(parts[1]), float(loss)])
        save(name, dict(kind="colorless", N=N, nfft=nfft, alias_decay_db=db, delays=delays, B=B, lr=lr, steps=steps,
                        state_keys=list(model.state_dict().keys())),
             x=x, target=tgt, est0=est0, losses=np.array(log),
             in_gain0=init[0], out_gain0=init[1], U_param0=init[2],
             g_in_gain0=g0[0], g_out_gain0=g0[1], g_U_param0=g0[2],
             in_gain=plist[0], out_gain=plist[1], U_param=plist[2], delays_s=p["delays"].param.detach())


# ----------------------------------------------------------------------------- config 1: e7_biquad training
def gen_biquad_training(dsp, system):
    """BASELINE configs[0], examples/e7_biquad.py at its own size: Shell(FFT(96000) -> Biquad((2,1), 2 sections,
    highpass, alias 30 dB) -> |.|), impulse input, target = |prod B / prod A| of random highpass sections
    (e7_biquad.py:23-61), nn.MSELoss, Adam steps as Trainer.train_step does them.  The estimates are stored
    decimated (every 97th bin).  The target is stored whole: highpass_filter returns `a` in float32, so the reference's
    target goes through a single-precision FFT whose rounding (1e-7) differs from host to host."""
    from flamo.functional import highpass_filter, signal_gallery
    torch.manual_seed(130709)
    nfft, fs, in_ch, out_ch, n_sections, lr, steps = 96000, 48000, 1, 2, 2, 1e-3, 6
    b, a = highpass_filter(fc=torch.tensor(fs // 2, dtype=F64) * torch.rand(size=(n_sections, out_ch, in_ch), dtype=F64),
                           gain=torch.tensor(-1) + torch.tensor(2) * torch.rand(size=(n_sections, out_ch, in_ch), dtype=F64), fs=fs)
    target_filter = torch.prod(torch.fft.rfft(b, nfft, dim=0), dim=1) / torch.prod(torch.fft.rfft(a, nfft, dim=0), dim=1)
    filt = dsp.Biquad(size=(out_ch, in_ch), n_sections=n_sections, filter_type="highpass", nfft=nfft, fs=fs,
                      requires_grad=True, alias_decay_db=30, dtype=F64)
    model = system.Shell(core=filt, input_layer=dsp.FFT(nfft, dtype=F64), output_layer=dsp.Transform(lambda x: torch.abs(x), dtype=F64))
    x = signal_gallery(1, n_samples=nfft, n=in_ch, signal_type="impulse", fs=fs, dtype=F64)
    target = torch.abs(torch.einsum("...ji,...i->...j", target_filter, model.get_inputLayer()(x)))
    param0 = filt.param.detach().clone()
    with torch.no_grad():
        fr0 = model.get_freq_response()
    crit = torch.nn.MSELoss()
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses, est0, g0 = [], None, None
    for it in range(steps):
        opt.zero_grad()
        est = model(x)
        loss = crit(est, target)
        loss.backward()
        if it == 0:
            est0, g0 = est.detach().clone(), filt.param.grad.detach().clone()
        opt.step()
        losses.append(float(loss))
    dec = slice(None, None, 97)
    save("e7_biquad", dict(kind="e7_biquad", nfft=nfft, fs=fs, alias_decay_db=30.0, n_sections=n_sections, lr=lr, steps=steps,
                           decimation=97, state_keys=list(model.state_dict().keys())),
         b=b, a=a, param0=param0, param=filt.param.detach(), g_param0=g0, losses=np.array(losses),
         est0_dec=est0[:, dec], target=target, fr0_dec=fr0[:, dec])


# ----------------------------------------------------------------------------- round 2: reference float32 beside float64, API vectors
def _f32vals(t):
    """float32-representable values held in float64 (both precisions of the reference then see the SAME numbers)"""
    return t.float().double()


def gen_round2(dsp, system):
    F32 = torch.float32
    # (a) config-2 miniature at the size SURVEY 8-c4 names (nfft=960, 8x8): float64 AND float32 runs of the reference
    for db in (0.0, 30.0):
        torch.manual_seed(7000 + int(db))
        N, nfft, B = 8, 960, 3       # (the size SURVEY 8-c4 names)
        W = _f32vals(torch.randn(N, N))
        Gp = _f32vals(torch.empty(12, N, N).uniform_(10 ** (-6 / 20), 10 ** (6 / 20)))
        x = _f32vals(torch.randn(B, nfft, N))
        res = {}
        for tag, dt in (("64", F64), ("32", F32)):
            kw = dict(nfft=nfft, alias_decay_db=db, dtype=dt)
            mat = dsp.Matrix(size=(N, N), matrix_type="random", requires_grad=True, **kw)
            geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
            mat.assign_value(W.to(dt))
            geq.assign_value(Gp.to(dt))
            core = system.Series(OrderedDict({"mix": mat, "eq": geq}))
            if db:
                model = system.Shell(core, dsp.FFTAntiAlias(nfft, alias_decay_db=db, dtype=dt), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, dtype=dt))
            else:
                model = system.Shell(core, dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt))
            xx = x.to(dt).requires_grad_(True)
            y = model(xx)
            g = torch.autograd.grad((y ** 2).mean(), [xx, mat.param, geq.param])
            res[tag] = (y, *g)
        save(f"r2_config2_960_db{int(db)}", dict(kind="config2", nfft=nfft, alias_decay_db=db, N=N, B=B, anti_alias_layers=bool(db)),
             x=x, W=W, geq_param=Gp, y=res["64"][0], gx=res["64"][1], gW=res["64"][2], gG=res["64"][3],
             y32=res["32"][0], gx32=res["32"][1], gW32=res["32"][2], gG32=res["32"][3])
    # (b) FDNs: float32 run of the reference beside the float64 one (damped and undamped loop), and an e10_probe-sized one
    specs = [("r2_fdn6_db30", 6, 480, 30.0, [59, 97, 131, 151, 163, 169], True, "wgn"),
             ("r2_fdn6_db0", 6, 480, 0.0, [59, 97, 131, 151, 163, 169], True, "impulse"),
             ("r2_fdn16_db30", 16, 1500, 30.0, [53, 61, 71, 79, 89, 97, 103, 109, 127, 137, 149, 157, 167, 179, 191, 199], True, "impulse"),
             ("r2_fdn4_4096", 4, 4096, 30.0, [887, 911, 941, 1699], False, "impulse")]
    for name, N, nfft, db, delays, attn, sig in specs:
        torch.manual_seed(7100 + N + int(db))
        pv = dict(in_gain=_f32vals(torch.randn(N, 1)), out_gain=_f32vals(torch.randn(1, N)), U_param=_f32vals(torch.randn(N, N)),
                  attn_param=_f32vals(torch.randn(12 if attn else 1, N) * 0.3 + 2.0))
        x = torch.zeros(1, nfft, 1, dtype=F64)
        if sig == "impulse":
            x[:, 0] = 1
        else:
            x = _f32vals(torch.randn(1, nfft, 1))
        c = _f32vals(torch.randn(1, nfft, 1))
        out = {}
        for tag, dt in (("64", F64), ("32", F32)):
            model, p = build_fdn(dsp, system, N, nfft, db, delays, attn, dtype=dt)
            p["input_gain"].assign_value(pv["in_gain"].to(dt))
            p["output_gain"].assign_value(pv["out_gain"].to(dt))
            p["mix"].assign_value(pv["U_param"].to(dt))
            if attn:
                p["att"].assign_value(pv["attn_param"].to(dt))
            xx = x.to(dt).requires_grad_(True)
            y = model(xx)
            plist = [p["input_gain"].param, p["output_gain"].param, p["mix"].param] + ([p["att"].param] if attn else [])
            g = torch.autograd.grad(torch.sum(y * c.to(dt)), [xx] + plist)
            out[tag] = (y, g)
            if tag == "64":
                dsec = p["delays"].param.detach().clone()
                keys = list(model.state_dict().keys())
        arrays = dict(x=x, c=c, delays_s=dsec, y=out["64"][0], y32=out["32"][0], gx=out["64"][1][0], gx32=out["32"][1][0],
                      g_in_gain=out["64"][1][1], g_out_gain=out["64"][1][2], g_U_param=out["64"][1][3],
                      g_U_param32=out["32"][1][3], **{k: v for k, v in pv.items() if attn or k != "attn_param"})
        if attn:
            arrays.update(g_attn_param=out["64"][1][4], g_attn_param32=out["32"][1][4])
        save(name, dict(kind="fdn", N=N, nfft=nfft, alias_decay_db=db, delays=delays, attn=attn, state_keys=keys), **arrays)
    # (c) transforms: float32 run beside float64
    for i, (nfft, db) in enumerate(((960, None), (4096, 30.0))):
        torch.manual_seed(7200 + i)
        x = _f32vals(torch.randn(1, nfft, 2))
        Z = _f32vals(torch.randn(1, nfft // 2 + 1, 2)) + 1j * _f32vals(torch.randn(1, nfft // 2 + 1, 2))
        r = {}
        for tag, dt in (("64", F64), ("32", F32)):
            if db is None:
                f, g = dsp.FFT(nfft, dtype=dt), dsp.iFFT(nfft, dtype=dt)
            else:
                f, g = dsp.FFTAntiAlias(nfft, alias_decay_db=db, dtype=dt), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, dtype=dt)
            r[tag] = (f(x.to(dt)), g(Z.to(torch.complex128 if dt == F64 else torch.complex64)))
        save(f"r2_fft_{nfft}", dict(kind="transform", nfft=nfft, T=nfft, norm="backward", alias_decay_db=db),
             x=x, Z=Z, X=r["64"][0], y=r["64"][1], X32=r["32"][0], y32=r["32"][1])
    # (d) config-5 structure in miniature (N=8, nfft=960): outputs and ALL gradients from the reference
    torch.manual_seed(7300)
    N, nfft, db = 8, 960, 30.0
    kw = dict(nfft=nfft, alias_decay_db=db, dtype=F64)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    dly = dsp.Delay(size=(N, N), max_len=200, isint=True, **kw)
    gain = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    geq.assign_value(_f32vals(torch.empty(12, N, N).uniform_(10 ** (-6 / 20), 10 ** (6 / 20))))
    gain.assign_value(_f32vals(torch.rand(N) * 0.5 / N ** 0.5 + 0.01))
    mix.assign_value(_f32vals(torch.randn(N, N)))
    dly.assign_value(dly.sample2s(torch.randint(1, 200, (N, N)).double()))
    core = system.Series(OrderedDict(eq=geq, loop=system.Recursion(fF=system.Series(OrderedDict(d=dly, g=gain)), fB=mix)))
    model = system.Shell(core, dsp.FFTAntiAlias(nfft, alias_decay_db=db, dtype=F64), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, dtype=F64))
    x = _f32vals(torch.randn(1, nfft, N) * 0.1).requires_grad_(True)
    c = _f32vals(torch.randn(1, nfft, N))
    y = model(x)
    g = torch.autograd.grad(torch.sum(y * c), [x, geq.param, gain.param, mix.param])
    save("r2_config5_mini", dict(kind="config5", N=N, nfft=nfft, alias_decay_db=db, max_len=200, state_keys=list(model.state_dict().keys())),
         x=x, c=c, geq=geq.param, delay_s=dly.param, gain=gain.param, U=mix.param, y=y, gx=g[0], g_geq=g[1], g_gain=g[2], g_U=g[3])
    # (e) Parallel.probe / probe_w and PEQ.compute_biquad_coeff (API vectors)
    torch.manual_seed(7400)
    kw = dict(nfft=96, alias_decay_db=0.0, dtype=F64)
    ga = dsp.Gain(size=(3, 2), **kw)
    gb = dsp.Gain(size=(3, 2), **kw)
    pg = dsp.parallelGain(size=(3,), **kw)
    z = torch.tensor(0.3 + 0.8j, dtype=C128)
    arrays = dict(ga=ga.param, gb=gb.param, pg=pg.param, z=z)
    for so in (True, False):
        par = system.Parallel(brA=OrderedDict(g=ga, pg=pg), brB=gb, sum_output=so)
        arrays[f"probe_{int(so)}"] = par.probe(z)
        arrays[f"probe_w_{int(so)}"] = par.probe_w(1 / z)
    save("r2_parallel_probe", dict(kind="parallel_probe", nfft=96), **arrays)
    arrays = {}
    f = _f32vals(torch.rand(4, 2, 2) * 1.5 + 0.05)
    R = _f32vals(torch.rand(4, 2, 2) * 0.7 + 0.2)
    G = _f32vals(torch.randn(4, 2, 2) * 6)
    arrays.update(f=f, R=R, G=G)
    for design in ("biquad", "svf"):
        peq = dsp.PEQ(size=(2, 2), n_bands=4, design=design, nfft=96, dtype=F32)
        for kind in ("peaking", "lowshelf", "highshelf"):
            a_, b_ = peq.compute_biquad_coeff(f.float(), R.float(), G.float(), type=kind)
            arrays[f"a_{design}_{kind}"] = a_
            arrays[f"b_{design}_{kind}"] = b_
    save("r2_peq_coeff", dict(kind="peq_coeff"), **arrays)


# ----------------------------------------------------------------------------- sparsity of a Householder mixing matrix
def gen_householder_sparsity(dsp, system):
    """flamo/optimize/loss.py:51-53: the criterion of a HouseholderMatrix feedback is that of I - 2 u u^T.  A 6-channel FDN core
    whose feedback is the Householder module itself (the first place the criterion looks); value and gradient w.r.t. u."""
    from flamo.optimize.loss import sparsity_loss
    torch.manual_seed(20260930)
    N, nfft = 6, 256
    kw = dict(nfft=nfft, dtype=F64)
    delays = dsp.parallelDelay(size=(N,), max_len=40, isint=True, **kw)
    mix = dsp.HouseholderMatrix(size=(N, N), requires_grad=True, **kw)
    core = system.Series(OrderedDict(input_gain=dsp.Gain(size=(N, 1), **kw),
                                     feedback_loop=system.Recursion(fF=delays, fB=mix),
                                     output_gain=dsp.Gain(size=(1, N), **kw)))
    model = system.Shell(core, dsp.FFT(nfft, dtype=F64), dsp.iFFT(nfft, dtype=F64))
    loss = sparsity_loss()(None, None, model)
    (g,) = torch.autograd.grad(loss, [mix.param])
    save("householder_sparsity", dict(kind="householder_sparsity", N=N, nfft=nfft), u=mix.param, loss=loss, g_u=g)


def main():
    torch.set_default_dtype(torch.float32)
    dsp, system = refimport.load()
    import warnings

    warnings.filterwarnings("ignore")
    if "--round2-only" in sys.argv:
        gen_round2(dsp, system)
        return
    if "--more-only" in sys.argv:
        gen_modules_more(dsp)
        return
    if "--parallel-only" in sys.argv:
        gen_parallel(dsp, system)
        return
    if "--accgeq-only" in sys.argv:
        gen_accurate_geq(dsp)
        return
    if "--colorless-only" in sys.argv:
        gen_colorless(dsp, system)
        return
    if "--householder-only" in sys.argv:
        gen_householder_sparsity(dsp, system)
        return
    if "--biquad-only" in sys.argv:
        gen_biquad_training(dsp, system)
        return
    gen_transforms(dsp)
    gen_modules(dsp)
    gen_modules_more(dsp)
    gen_config2(dsp, system)
    gen_fdn(dsp, system)
    gen_parallel(dsp, system)
    gen_accurate_geq(dsp)
    gen_colorless(dsp, system)
    gen_biquad_training(dsp, system)
    gen_householder_sparsity(dsp, system)
    gen_round2(dsp, system)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print(f"total {total/1024:.1f} KiB")


if __name__ == "__main__":
    main()
