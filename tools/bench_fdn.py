#!/usr/bin/env python
"""Secondary benchmark: BASELINE configs[2] -- 16-channel FDN (e8_fdn.py structure), nfft=192000,
Gain(16,1) -> Recursion(fF=parallelDelay(isint), fB=Series(Matrix orthogonal, parallelGEQ)) -> Gain(1,16),
FFT / iFFTAntiAlias(30 dB); forward + backward of (y*c).sum() with all parameter gradients.
Prints bin-solves/s (B*M / time) for float32 and float64.   python tools/bench_fdn.py [--batch B]

--workload config5: BASELINE configs[4] on one GPU -- the active-acoustics structure:
FFTAntiAlias(384000, 30 dB) -> Series(GEQ((32,32)), Recursion(fF=Series(Delay((32,32), isint), parallelGain(32)),
fB=Matrix(32,32, orthogonal))) -> iFFTAntiAlias, impulse-like input (1, 384000, 32), gradients for GEQ, gain, matrix."""
import os as _os_env
_os_env.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see flamo_amd/__init__.py: must precede HIP runtime init

import argparse
import json
import os
import sys
import time
import warnings
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(dev, dtype, N=16, nfft=192000, db=30.0):
    from flamo_amd.processor import dsp, system
    kw = dict(nfft=nfft, alias_decay_db=db, device=dev, dtype=dtype)
    delays = torch.tensor([503, 593, 701, 811, 919, 1031, 1151, 1259, 1381, 1493, 1613, 1741, 1873, 2003, 2381, 2713][:N])
    ig = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
    og = dsp.Gain(size=(1, N), requires_grad=True, **kw)
    dl = dsp.parallelDelay(size=(N,), max_len=3000, isint=True, **kw)
    dl.assign_value(dl.sample2s(delays.to(dev, dtype)))
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
    att.map = dsp.db_of_sigmoid        # 20 log10(sigmoid(x)) as in e8_fdn.py:97, by name: folded into the design kernel
    with torch.no_grad():
        att.param.copy_(torch.randn_like(att.param) * 0.3 + 2.0)
    fb = system.Series(OrderedDict(mixing_matrix=mix, attenuation=att))
    core = system.Series(OrderedDict(input_gain=ig, feedback_loop=system.Recursion(fF=dl, fB=fb), output_gain=og))
    model = system.Shell(core, dsp.FFT(nfft, dtype=dtype), dsp.iFFTAntiAlias(nfft, alias_decay_db=db, device=dev, dtype=dtype))
    return model, [ig.param, og.param, mix.param, att.param]


def build_config5(dev, dtype, N=32, nfft=384000, db=30.0):
    from flamo_amd.processor import dsp, system
    kw = dict(nfft=nfft, alias_decay_db=db, device=dev, dtype=dtype)
    geq = dsp.GEQ(size=(N, N), requires_grad=True, **kw)
    dly = dsp.Delay(size=(N, N), max_len=2000, isint=True, **kw)
    gain = dsp.parallelGain(size=(N,), requires_grad=True, **kw)
    with torch.no_grad():
        gain.param.copy_(torch.rand_like(gain.param) * 0.5 / N ** 0.5 + 0.01)
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    core = system.Series(OrderedDict(eq=geq, loop=system.Recursion(fF=system.Series(OrderedDict(d=dly, g=gain)), fB=mix)))
    model = system.Shell(core, dsp.FFTAntiAlias(nfft, alias_decay_db=db, dtype=dtype),
                         dsp.iFFTAntiAlias(nfft, alias_decay_db=db, dtype=dtype))
    return model, [geq.param, gain.param, mix.param]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="fdn16", choices=["fdn16", "config5"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="both")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    dev = torch.device("cuda:0")
    nfft, N = (192000, 16) if args.workload == "fdn16" else (384000, 32)
    chan = 1 if args.workload == "fdn16" else N
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        if args.dtype not in ("both", name):
            continue
        torch.manual_seed(130709)
        model, params = build(dev, dt, N, nfft) if args.workload == "fdn16" else build_config5(dev, dt, N, nfft)
        x = torch.randn(args.batch, nfft, chan, device=dev, dtype=dt)
        c = torch.randn(args.batch, nfft, chan, device=dev, dtype=dt)

        def step():
            for p in params:
                p.grad = None
            y = model(x)
            (y * c).sum().backward()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        out[name] = {"ms_per_step": ms, "bin_solves_per_s": args.batch * (nfft // 2 + 1) / (ms * 1e-3)}
        # the same step replayed from a HIP graph (launch overhead removed)
        from flamo_amd.graph import GraphedStep
        step()
        ref = [p.grad.clone() for p in params]           # eager gradients, before the graph takes over p.grad
        gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params)
        gs(x)
        torch.cuda.synchronize()
        err = max(((a - b).norm() / b.norm()).item() for a, b in zip(gs.grads, ref))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gs(x)
        torch.cuda.synchronize()
        msg = (time.perf_counter() - t0) / args.steps * 1e3
        out[name].update(graph_ms_per_step=msg, graph_bin_solves_per_s=args.batch * (nfft // 2 + 1) / (msg * 1e-3),
                         graph_vs_eager_grad_relerr=err)
    what = f"16-ch FDN, nfft={nfft}" if args.workload == "fdn16" else f"config 5 chain {N}x{N}, nfft={nfft}"
    print(json.dumps({"workload": f"{what}, batch {args.batch}, fwd+bwd", **out}))


if __name__ == "__main__":
    main()
