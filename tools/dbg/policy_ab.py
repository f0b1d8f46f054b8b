"""A few stream-policy masks on the replayed config-2 step, interleaved (one capture per mask and round): step time and the
dominant kernel's slot.  python tools/dbg/policy_ab.py 0x0,0x100,0x1000 [rounds]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402

masks = [int(m, 0) for m in (sys.argv[1] if len(sys.argv) > 1 else "0x0,0x100,0x1000,0x1100,0x1030").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
res = {m: [] for m in masks}
for r in range(rounds):
    for m in masks:
        L.fl_set_stream_policy(m, 0)
        gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
        bench.settle_device(gs.replay, max_steps=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            gs.replay()
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 300 * 1e6)
        del gs
for m in masks:
    v = sorted(res[m])
    print(f"mask {m:#07x}: median {v[len(v) // 2]:.1f} us, min {v[0]:.1f}, all {[round(t, 1) for t in res[m]]}")
