import sys, torch, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.simplefilter("ignore")
from conftest import load_golden, relerr
from flamo_amd.processor import dsp
from flamo_amd import functional as F
from oracle import hotpath as O
dev = torch.device("cuda:0")
meta, a = load_golden("geq_db0")
for dt in (torch.float64,):
    g = dsp.GEQ(size=(2,2), nfft=96, alias_decay_db=0.0, device=dev, dtype=dt)
    g.assign_value(a["param"].to(dev, dt))
    gdb = g.map(g.param)
    b, aa = g._sos_coeffs(gdb)
    bc, ac = g._design.sections(g.map(g.param.detach().cpu()))
    print("coef diff gpu vs cpu", (b.cpu()-bc).abs().max().item(), (aa.cpu()-ac).abs().max().item())
    print("gdb diff", (gdb.cpu() - 20*torch.log10(torch.abs(a["param"]))).abs().max().item())
    H = g.freq_response(g.param)
    print("H relerr", relerr(H.detach().cpu(), a["freq_response"]))
    Hc = O.sos_response(bc, ac, 96, O.gamma_of(0.0, 96))
    print("oracle(H from cpu coefs) vs golden", relerr(Hc, a["freq_response"]))
    from flamo_amd import ops
    H2 = ops.sos_response(bc.double().to(dev), ac.double().to(dev), 1.0, 96)
    print("kernel on cpu coefs", relerr(H2.cpu(), a["freq_response"]), H2.shape, H2.stride())
    d = (H.detach().cpu() - a["freq_response"]).abs()
    print(d.amax(dim=(1,2))[:6], a["freq_response"].abs().amax(dim=(1,2))[:6])
