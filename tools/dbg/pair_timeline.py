"""Timeline of the launch pair (csrc/fusedfwd.hip) at BASELINE configs[1]: per-workgroup start / end by the device clock, role and
CU -- how long a workgroup of either role lives, when the last of each role ends, how the two fill the CUs.
    python tools/dbg/pair_timeline.py [--reps 5]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flamo_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
L = _lib.lib()
khz = L.fl_wall_clock_khz()


def step():
    y = model(x)
    torch.autograd.grad(ops.mean_square(y), params)


for _ in range(30):
    step()
torch.cuda.synchronize()
NW = 8192
buf = torch.zeros(4 * NW + 8 * 8 * 4096, dtype=torch.int64, device=dev)
for rep in range(args.reps):
    buf.zero_()
    for _ in range(3):
        step()
    L.fl_debug_set_pair_stamps(buf.data_ptr())
    n0 = L.fl_debug_launch_pair_count()
    y = model(x)
    torch.cuda.synchronize()
    L.fl_debug_set_pair_stamps(None)
    assert L.fl_debug_launch_pair_count() == n0 + 1
    ph = buf[4 * NW:].view(8 * 4096, 8).cpu()
    t = buf[:4 * NW].view(NW, 4).cpu()
    t = t[t[:, 0] > 0]
    us = 1e3 / khz
    t0 = t[:, 0].min().item()
    st = (t[:, 0] - t0).double() * us
    en = (t[:, 1] - t0).double() * us
    role = t[:, 2]
    hw = t[:, 3] & 0xffffffff
    xcc = (t[:, 3] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)
    print(f"--- rep {rep}: {len(t)} workgroups, span {en.max().item():.1f} us; distinct CUs {len(torch.unique(cu))}")
    for r, name in ((0, "column pass"), (1, "response")):
        m = role == r
        life = (en - st)[m]
        print(f"  {name:12s} n={int(m.sum()):5d} life mean {life.mean():6.2f} min {life.min():6.2f} max {life.max():6.2f} us; "
              f"first start {st[m].min():6.2f} last start {st[m].max():6.2f} last end {en[m].max():6.2f}")
    # phases of the response role: workgroup i of the grid with role 1 is response block k (k-th of them), (bx, m) = (k % gx, k // gx)
    idx = torch.nonzero(role == 1).flatten()
    gx = 94
    k = torch.arange(len(idx))
    p = ph[(k // gx) * 4096 + (k % gx)]
    s_r, e_r = t[idx, 0], t[idx, 1]
    d = torch.stack([p[:, 0] - s_r, p[:, 1] - p[:, 0], p[:, 2] - p[:, 1], e_r - p[:, 2]], 1).double() * us
    la = (p[:, 3] - s_r).double() * us
    print(f"  response: operands arrived {la.mean().item():.2f} / {la.max().item():.2f} us after the start")
    print("  response phases (us, mean / max): design %.2f / %.2f, tables %.2f / %.2f, cascades %.2f / %.2f, store %.2f / %.2f" % tuple(
        v for j in range(4) for v in (d[:, j].mean().item(), d[:, j].max().item())))
    # residency over time (workgroups of each role alive, whole device), every 5 us
    for tt in range(0, int(en.max().item()) + 5, 5):
        alive = (st <= tt) & (en > tt)
        print(f"    t={tt:3d} us: column {int((alive & (role == 0)).sum()):5d}  response {int((alive & (role == 1)).sum()):4d}")
    # per CU: time of its last column-pass end and its last response end
    if rep == args.reps - 1:
        ends_c, ends_r = [], []
        for c in torch.unique(cu).tolist():
            m = cu == c
            ends_c.append(en[m & (role == 0)].max().item() if (m & (role == 0)).any() else 0.0)
            ends_r.append(en[m & (role == 1)].max().item() if (m & (role == 1)).any() else 0.0)
        ec, er = torch.tensor(ends_c), torch.tensor(ends_r)
        print(f"  per CU: last column end mean {ec.mean():.1f} (max {ec.max():.1f}); last response end mean {er.mean():.1f} (max {er.max():.1f}); "
              f"CUs whose response ends last: {int((er > ec).sum())}")
        nr = torch.tensor([int(((cu == c) & (role == 1)).sum()) for c in torch.unique(cu).tolist()])
        print(f"  response workgroups per CU: min {nr.min().item()} mean {nr.float().mean():.2f} max {nr.max().item()}")

# the response's plain launch: phases of its workgroups
ops.LAUNCH_PAIRS = False
for _ in range(3):
    step()
buf.zero_()
L.fl_debug_set_pair_stamps(buf.data_ptr())
y = model(x)
torch.cuda.synchronize()
L.fl_debug_set_pair_stamps(None)
ph = buf[4 * NW:].view(8 * 4096, 8).cpu()
ph = ph[ph[:, 4] > 0]
us = 1e3 / khz
t0 = ph[:, 4].min()
d = torch.stack([ph[:, 3] - ph[:, 4], ph[:, 0] - ph[:, 4], ph[:, 1] - ph[:, 0], ph[:, 2] - ph[:, 1], ph[:, 5] - ph[:, 2]], 1).double() * us
print(f"plain response launch: {len(ph)} workgroups, span {(ph[:, 5].max() - t0).item() * us:.1f} us, last start {(ph[:, 4].max() - t0).item() * us:.1f}; "
      "life %.2f us; operands %.2f, design %.2f, tables %.2f, cascades %.2f, store %.2f (means)" % (
          ((ph[:, 5] - ph[:, 4]).double() * us).mean().item(), *[d[:, j].mean().item() for j in range(5)]))
# the two alone, by events
for _ in range(5):
    step()
torch.cuda.synchronize()
ops.kernel_timer.reset(True)
for _ in range(20):
    step()
torch.cuda.synchronize()
ops.kernel_timer.enabled = False
print("two launches:", {k: round(v[1] * 1e3, 1) for k, v in ops.kernel_timer.summary().items()})
ops.LAUNCH_PAIRS = True
ops.kernel_timer.reset(True)
for _ in range(20):
    step()
torch.cuda.synchronize()
ops.kernel_timer.enabled = False
print("pair:", {k: round(v[1] * 1e3, 1) for k, v in ops.kernel_timer.summary().items()})
