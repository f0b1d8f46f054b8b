"""The configs[4] structure (32 x 32 chain, nfft = 384000) for a few eager steps -- for rocprofv3 --kernel-trace --stats.
    python tools/dbg/c5_step.py [f32|f64] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flamo_amd  # noqa: E402,F401
import bench_fdn  # noqa: E402

dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(130709)
model, params = bench_fdn.build_config5(dev, dt, 32, 384000)
x = torch.randn(1, 384000, 32, device=dev, dtype=dt)
c = torch.randn(1, 384000, 32, device=dev, dtype=dt)
for i in range(steps + 1):
    if i == 1:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    for p in params:
        p.grad = None
    (model(x) * c).sum().backward()
e1.record()
torch.cuda.synchronize()
print(f"config-5 structure, {dt}: {e0.elapsed_time(e1) / steps:.3f} ms per eager step")
