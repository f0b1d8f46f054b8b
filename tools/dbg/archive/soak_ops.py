"""Which of our operators, placed in a captured step in front of a torch reduction, makes that reduction's replayed
result change after a tiny eager launch between replays (tools/dbg/soak_fdn*.py)?"""
import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
warnings.simplefilter("ignore")
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
mode = sys.argv[1]
torch.manual_seed(1)
nfft, N = 192000, 16
x = torch.randn(1, nfft, 1, device=dev); c = torch.randn(1, nfft, 1, device=dev)
w = torch.nn.Parameter(torch.randn(1, device=dev))
Wi = torch.nn.Parameter(torch.randn(N, 1, device=dev)); Wo = torch.nn.Parameter(torch.randn(1, N, device=dev))
def fn(xx):
    if mode == "fft_only":
        y = ops.irfft(ops.rfft(xx * w, nfft), nfft)
        return (y * c).sum()
    X = ops.rfft(xx, nfft)
    if mode == "mimo_real":
        Y = ops.mimo(Wo, ops.mimo(Wi, X))
    elif mode == "mimo_cplx":
        Y = ops.mimo(Wo.to(torch.complex64), ops.mimo(Wi.to(torch.complex64), X))
    elif mode == "mimo_in_only":
        Y = ops.mimo(Wi, X)[:, :, :1] * w
    y = ops.irfft(Y, nfft)
    return (y * c).sum()
params = [w] if mode == "fft_only" else ([Wi, w] if mode == "mimo_in_only" else [Wi, Wo])
gs = GraphedStep(fn, (x,), params, warmup=2)
out0 = gs.replay().clone(); torch.cuda.synchronize()
vals = []
for i in range(3):
    out = gs.replay(); torch.cuda.synchronize()
    j = torch.full((1,), 5.0, device=dev); del j
    vals.append(out.clone())
print(mode, [v.item() for v in vals], "first", out0.item())
