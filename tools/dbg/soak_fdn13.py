import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
mode, between = sys.argv[1], sys.argv[2]
torch.manual_seed(1)
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
model, params = bench_fdn.build(dev, torch.float32, 16, 192000)
keep = {}
def fn(xx):
    y = model(xx)
    if mode == "contig": return (y.contiguous() * c).sum()
    if mode == "hold_yc":
        yc = y * c
        if torch.cuda.is_current_stream_capturing(): keep["yc"] = yc
        return yc.sum()
    if mode == "hold_y":
        if torch.cuda.is_current_stream_capturing(): keep["y"] = y
        return (y * c).sum()
    if mode == "dot": return torch.dot(y.reshape(-1), c.reshape(-1))
    return (y * c).sum()
gs = GraphedStep(fn, (x,), params, warmup=2)
out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
torch.cuda.synchronize()
vals = []
for i in range(3):
    out = gs.replay()
    torch.cuda.synchronize()
    if between == "equal":
        for a, b in zip([out] + [p.grad for p in params], [out0] + g0):
            torch.equal(a, b)
    elif between == "equal_out":
        torch.equal(out, out0)
    elif between == "eq_noitem":
        r = (out == out0).all()
    elif between == "junk":
        j = [torch.full((n,), 5.0, device=dev) for n in (1, 8, 512, 4096) for _ in range(8)]; del j
    vals.append(out.clone())
print(mode, between, [v.item() for v in vals], "first", out0.item())
