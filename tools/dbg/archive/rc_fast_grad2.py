"""GEQ gradient, float vs double forward evaluation: plain response with a random cotangent, and the Matrix-then-GEQ
operator in a Shell with a mean-square loss; several anti-alias decays."""
import os, sys, warnings
from collections import OrderedDict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops
from flamo_amd.processor import dsp, system
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
N = 8
for nfft in (48000, 96000):
    for db in (0.0, 20.0, 30.0):
        for par in (False, True):
            torch.manual_seed(1)
            kw = dict(nfft=nfft, alias_decay_db=db, device=dev, dtype=torch.float32, requires_grad=True)
            mod = dsp.parallelGEQ(size=(N,), **kw) if par else dsp.GEQ(size=(N, N), **kw)
            g = {}
            for fast in (1, 0):
                _lib.lib().fl_debug_set_rc_fast(fast)
                mod.param.grad = None
                H = mod.freq_response(mod.param)
                c = torch.randn(H.shape, device=dev, dtype=H.dtype, generator=torch.Generator(device=dev).manual_seed(5))
                (H * c.conj()).real.sum().backward()
                g[fast] = mod.param.grad.clone()
            _lib.lib().fl_debug_set_rc_fast(1)
            # float64 module as the truth
            m64 = (dsp.parallelGEQ if par else dsp.GEQ)(size=(N,) if par else (N, N), nfft=nfft, alias_decay_db=db, device=dev, dtype=torch.float64, requires_grad=True)
            with torch.no_grad():
                m64.param.copy_(mod.param.double())
            H64 = m64.freq_response(m64.param)
            (H64 * c.to(H64.dtype).conj()).real.sum().backward()
            print(f"nfft={nfft} db={db} parallel={par}: float-vs-double fwd {rel(g[1], g[0]):.1e}; vs f64: float fwd {rel(g[1], m64.param.grad):.1e}, double fwd {rel(g[0], m64.param.grad):.1e}")
