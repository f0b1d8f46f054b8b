"""Cache policy of the pipeline's streams (fl_set_stream_policy, csrc/common.h: StreamPolicy) on the replayed config-2 step.

  python tools/dbg/policy_sweep.py [--steps 200] [--masks 0x0,0x1,...] [--greedy]

For every mask: the step is captured again (kernel arguments are baked into the graph), settled, and `steps` replays are timed;
--greedy walks the bits one at a time keeping each that helps (two rounds)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flamo_amd import _lib, ops  # noqa: E402
from flamo_amd.graph import GraphedStep  # noqa: E402

BITS = {0: "cols_fwd(x) ld nt", 1: "cols_fwd(x) st nt", 2: "cols_inv ld nt", 3: "cols_inv st nt", 4: "mid_walk S dma nt",
        5: "mid_walk S2 st plain", 6: "mid_walk Xp st plain", 7: "gradh Sg dma nt", 8: "gradh Xp dma nt", 9: "gradh dH st nt",
        10: "lanes G ld nt", 11: "lanes gH ld nt", 12: "rc_ba G st nt", 16: "cols_fwd(y) ld nt", 17: "cols_fwd(y) st nt"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--masks", default="")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--base", default="0")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    torch.manual_seed(130709)
    model, params = bench.build_model(dev, torch.float32)
    x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)

    def measure(mask, reps=2):
        L.fl_set_stream_policy(mask, 0)
        gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
        bench.settle_device(gs.replay, max_steps=100)
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                gs.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.steps * 1e6)
        del gs
        return best

    base = int(a.base, 0)
    t0 = measure(base)
    print(f"base {base:#x}: {t0:.1f} us")
    if a.masks:
        for m in a.masks.split(","):
            m = int(m, 0)
            print(f"mask {m:#08x}: {measure(m):.1f} us")
    singles = {}
    for b, name in BITS.items():
        t = measure(base ^ (1 << b))
        singles[b] = t
        print(f"bit {b:2d} {name:24s}: {t:.1f} us ({t - t0:+.1f})")
    if a.greedy:
        cur, tcur = base, t0
        for rnd in range(2):
            for b in sorted(BITS, key=lambda k: singles[k]):
                t = measure(cur ^ (1 << b))
                keep = t < tcur - 0.3
                print(f"  round {rnd} toggle bit {b:2d} ({BITS[b]}): {t:.1f} us {'KEEP' if keep else ''}")
                if keep:
                    cur, tcur = cur ^ (1 << b), t
        print(f"greedy best mask {cur:#x}: {tcur:.1f} us (re-measured {measure(cur):.1f}; base re-measured {measure(base):.1f})")


if __name__ == "__main__":
    main()
