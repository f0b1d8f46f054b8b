"""How long does the captured config-2 step take on its first replays?  (bench.py's timed region starts W = 5 replays after the
capture in the driver's run.)  Prints the time of blocks of 5 replays from the first one on.
    python tools/dbg/replay_transient.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from flamo_amd import ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))
out = []
for blk in range(24):
    t0 = time.perf_counter()
    for _ in range(5):
        gs.replay()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 5 * 1e3)
print("ms per replay, blocks of 5:", " ".join(f"{v:.3f}" for v in out))
