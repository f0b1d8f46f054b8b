import sys, torch, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.simplefilter("ignore")
from conftest import relerr
from flamo_amd.processor import dsp
from flamo_amd import ops
from oracle import hotpath as O
dev = torch.device("cuda:0")
torch.manual_seed(0)
nfft, N, B = 960, 4, 2
W = torch.randn(N, N, dtype=torch.float64)
G = torch.empty(12, N, N, dtype=torch.float64).uniform_(10 ** (-6 / 20), 10 ** (6 / 20))
x = torch.randn(B, nfft, N, dtype=torch.float64)
g1 = O.gamma_of(0.0, nfft)
Xo = O.rfft(x, nfft); Yo1 = O.mimo_const(O.to_complex(W), Xo); Ho = O.geq_response(G, nfft, g1); Yo2 = O.mimo_full(Ho.to(torch.complex128), Yo1); yo = O.irfft(Yo2, nfft)
for dt in (torch.float32, torch.float64):
    cd = torch.complex64 if dt == torch.float32 else torch.complex128
    mat = dsp.Matrix(size=(N, N), nfft=nfft, device=dev, dtype=dt); mat.assign_value(W.to(dev, dt))
    geq = dsp.GEQ(size=(N, N), nfft=nfft, device=dev, dtype=dt); geq.assign_value(G.to(dev, dt))
    X = ops.rfft(x.to(dev, dt), nfft)
    print(dt, "rfft", relerr(X.cpu(), Xo))
    Y1 = mat(X); print(" matrix", relerr(Y1.cpu(), Yo1))
    H = geq.freq_response(geq.param); print(" H", relerr(H.cpu(), Ho), (H.cpu()-Ho).abs().max().item())
    Y2 = geq(Y1); print(" geq out", relerr(Y2.cpu(), Yo2))
    Y2b = ops.mimo(Ho.to(dev, cd), Yo1.to(dev, cd)); print(" mimo(oracle H, oracle X)", relerr(Y2b.cpu(), Yo2))
    Y2c = ops.mimo(H, Yo1.to(dev, cd)); print(" mimo(my H, oracle X)", relerr(Y2c.cpu(), Yo2))
    y = ops.irfft(Y2, nfft); print(" y", relerr(y.cpu(), yo))
    yb = ops.irfft(Yo2.to(dev, cd), nfft); print(" irfft(oracle Y)", relerr(yb.cpu(), yo))
