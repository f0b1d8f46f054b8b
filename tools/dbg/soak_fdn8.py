import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd import ops
dev = torch.device('cuda:0')

def run(side_warmup, use_seed, workload="fdn16"):
    torch.manual_seed(1)
    if workload == "fdn16":
        model, params = bench_fdn.build(dev, torch.float32, 16, 192000); nfft, ch = 192000, 1
    x = torch.randn(1, nfft, ch, device=dev); c = torch.randn(1, nfft, ch, device=dev)
    fn = lambda xx: (model(xx) * c).sum()
    seed = None
    def eager():
        nonlocal seed
        for p in params: p.grad = None
        with ops.step_scope():
            out = fn(x); out.backward()
        if use_seed: seed = torch.ones_like(out)
    if side_warmup:
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): eager()
        torch.cuda.current_stream().wait_stream(side)
    else:
        for _ in range(2): eager()
    for p in params: p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        with ops.step_scope():
            out = fn(x)
            grads = torch.autograd.grad(out, params, grad_outputs=seed)
        static_out = out.detach()
    g.replay(); torch.cuda.synchronize()
    v0 = static_out.item(); g0 = [t.clone() for t in grads]
    junk = [torch.full((sz,), 3.0, device=dev) for sz in (1, 64, 1024, 16384, 1 << 18, 1 << 20) for _ in range(6)]
    eq = [torch.equal(a, b) for a, b in zip(grads, g0)]
    torch.cuda.synchronize()
    vals = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize(); vals.append(static_out.item())
    print(f"side_warmup={side_warmup} use_seed={use_seed}: first {v0}, later {vals}")

for sw in (True, False):
    for us in (True, False):
        run(sw, us)
