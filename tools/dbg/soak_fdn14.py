import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
warnings.simplefilter("ignore")
from flamo_amd.graph import GraphedStep
from flamo_amd import ops
dev = torch.device('cuda:0')
mode = sys.argv[1]
torch.manual_seed(1)
x = torch.randn(1, 192000, 1, device=dev)
c = torch.randn(1, 192000, 1, device=dev)
w = torch.nn.Parameter(torch.randn(1, device=dev))
side = torch.cuda.Stream()
def fn(xx):
    if mode == "side":
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            t = xx * w
            ev2 = torch.cuda.Event(); ev2.record()
        torch.cuda.current_stream().wait_event(ev2)
        return (t * c).sum()
    if mode == "rfft":
        X = ops.rfft(xx * w, 192000)
        y = ops.irfft(X, 192000)
        return (y * c).sum()
    if mode == "zeros":
        buf = torch.zeros(1, 1, 192064, device=dev)
        buf[..., :192000] = (xx * w).movedim(1, -1)
        y = buf[..., :192000].movedim(-1, 1)
        return (y * c).sum()
    return ((xx * w) * c).sum()
gs = GraphedStep(fn, (x,), [w], warmup=2)
out0 = gs.replay().clone()
torch.cuda.synchronize()
vals = []
for i in range(3):
    out = gs.replay(); torch.cuda.synchronize()
    j = [torch.full((n,), 5.0, device=dev) for n in (1, 8, 512, 4096) for _ in range(8)]; del j
    vals.append(out.clone())
print(mode, [v.item() for v in vals], "first", out0.item())
