"""Stability check for the FDN paths: many replays of the captured 16-channel FDN step and of the config-5 chain --
bit-identical output / gradients throughout, no growth of device memory."""
import sys, os, torch, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter("ignore")
import bench_fdn
from flamo_amd.graph import GraphedStep
dev = torch.device('cuda:0')
for name, builder, N, nfft, chan, reps in (("fdn16", bench_fdn.build, 16, 192000, 1, 3000), ("config5", bench_fdn.build_config5, 32, 384000, 32, 300)):
    torch.manual_seed(1)
    model, params = builder(dev, torch.float32, N, nfft)
    x = torch.randn(1, nfft, chan, device=dev)
    c = torch.randn(1, nfft, chan, device=dev)
    gs = GraphedStep(lambda xx: (model(xx) * c).sum(), (x,), params, warmup=2)
    out0 = gs.replay().clone(); g0 = [p.grad.clone() for p in params]
    torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
    bad = 0
    for i in range(reps):
        out = gs.replay()
        if i % (reps // 5) == reps // 5 - 1:
            torch.cuda.synchronize()
            same = torch.equal(out, out0) and all(torch.equal(p.grad, g) for p, g in zip(params, g0))
            bad += (not same)
            print(name, i + 1, "replays: identical =", same, "mem delta", torch.cuda.memory_allocated() - m0)
    print(name, "OK" if bad == 0 else "MISMATCH")
    del gs, model, params
    torch.cuda.empty_cache()
