"""Batch-walking row kernels (csrc/specwalk.hip) against spec_mid + mimo_gradh: equality and launch times.
    python tools/dbg/walk_bench.py [--batch 32] [--wgs 0,256,248] [--slices 0,2,3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nfft", type=int, default=96000)
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--wgs", default="0")
ap.add_argument("--slices", default="0")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
nfft, N, B = args.nfft, args.n, args.batch
M = nfft // 2 + 1
L = _lib.lib()
torch.manual_seed(0)
x = torch.randn(B, nfft, N, device=dev)
g = torch.randn(B, nfft, N, device=dev)
H = ops.permute_bins(torch.randn(M, N, N, device=dev, dtype=torch.complex64) / N ** 0.5, nfft)
Hp = ops._h_planar(H, True)
hp = ops._lead_pitch(Hp.movedim(0, -1))
W = ops.twiddles(nfft, torch.float32, dev)
S = ops._spec_cols_fwd(x, nfft, 0.0)
Sg = ops._spec_cols_fwd(g, nfft, 0.0)
P = ops._pitch(M)
bnd = ops._walk_partition(nfft, B, dev)
BND = None if os.environ.get("WALK_EQUAL_COUNTS") == "1" else bnd.data_ptr()
bl = bnd.cpu().tolist()
print("partition: units per workgroup min", min(b - a for a, b in zip(bl, bl[1:])), "max", max(b - a for a, b in zip(bl, bl[1:])))
rel = lambda a, b: (torch.linalg.vector_norm((a - b).float()) / torch.linalg.vector_norm(b.float())).item()  # noqa: E731


def timeit(fn, reps=args.reps, flush=True):
    junk = torch.empty(64 * 1024 * 1024, device=dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if flush:
            junk.add_(1.0)          # 256 MB touched: the infinity cache holds nothing of the operands
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


# ---------------------------------------------------------------- forward
for conj_t, scale, i2, ph in ((False, 1.0, 0, 0), (True, 1.0 / nfft, 1, 1)):
    L.fl_debug_set_walk(0, 0, 0, None)
    S2_ref, _ = ops._spec_mid(S.clone(), B, N, N, nfft, Hp, conj_t, False, True, scale, i2, ph)
    hs_m, hs_n = (hp, N * hp) if conj_t else (N * hp, hp)
    for wgs in [int(v) for v in args.wgs.split(",")]:
        L.fl_debug_set_walk(1, wgs, 0, None)
        S2 = torch.full_like(S, float("nan"))
        _lib.check(L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), None, Hp.data_ptr(), hs_m, hs_n, int(conj_t), W.data_ptr(), nfft, B, N, N,
                                          scale, i2, ph, BND, ops._stream()), "walk")
        torch.cuda.synchronize()
        print(f"forward conj_t={conj_t} wgs={wgs}: rel err vs spec_mid {rel(torch.view_as_real(S2), torch.view_as_real(S2_ref)):.3e}"
              f"  nan={torch.isnan(torch.view_as_real(S2)).any().item()}")

S2 = torch.empty_like(S)
Xp = torch.zeros(L.fl_spec_walk_spectrum_elems(nfft, B, N), dtype=torch.complex64, device=dev)
for wgs in [int(v) for v in args.wgs.split(",")]:
    L.fl_debug_set_walk(1, wgs, 0, None)
    t = timeit(lambda: L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), None, Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N, 1.0, 0, 0,
                                              BND, ops._stream()))
    print(f"spec_mid_walk wgs={wgs}: median {t[0]:.1f} us, min {t[1]:.1f} us")
    t = timeit(lambda: L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), Xp.data_ptr(), Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N, 1.0, 0, 0,
                                              BND, ops._stream()))
    print(f"spec_mid_walk wgs={wgs} + pair-major spectrum: median {t[0]:.1f} us, min {t[1]:.1f} us")
Xs = ops._empty_rows((B, N), M, torch.complex64, dev)
t = timeit(lambda: L.fl_spec_mid_f32(S.data_ptr(), S2.data_ptr(), Xs.data_ptr(), N * P, P, Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N,
                                     1.0, 0, 0, ops._stream()))
print(f"spec_mid (spectrum stored, separate output): median {t[0]:.1f} us, min {t[1]:.1f} us")
t = timeit(lambda: L.fl_spec_mid_f32(S.data_ptr(), S2.data_ptr(), None, 0, 0, Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N,
                                     1.0, 0, 0, ops._stream()))
print(f"spec_mid (no spectrum): median {t[0]:.1f} us, min {t[1]:.1f} us")

# phase picture of the walking kernel: span per workgroup
cus = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(cus * 8, dtype=torch.int64, device=dev)
buf2 = torch.zeros_like(buf)
L.fl_debug_set_walk(1, 0, 0, buf2.data_ptr())
L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), Xp.data_ptr(), Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N, 1.0, 7, 0, BND, ops._stream())
torch.cuda.synchronize()
L.fl_debug_set_walk(1, 0, 0, None)
L.fl_debug_set_walk(1, 0, 0, buf.data_ptr())
L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), Xp.data_ptr(), Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N, 1.0, 0, 0, BND, ops._stream())
torch.cuda.synchronize()
L.fl_debug_set_walk(1, 0, 0, None)
w = buf2.view(-1, 8).cpu()
w = w[w[:, 1] > 0][:, 6]
print(f"   P5 of a group-0 wavefront: reads {((w & 0x1FFFFF) * 16).double().mean():8.0f}  butterflies {(((w >> 21) & 0x1FFFFF) * 16).double().mean():8.0f}"
      f"  twiddles + stores {(((w >> 42) & 0x1FFFFF) * 16).double().mean():8.0f} cycles per workgroup")
raw = buf.view(-1, 8).cpu()
raw = raw[raw[:, 1] > 0]
tt = raw.double()
if len(tt):
    for g, nm in ((0, "group 0 (P4, P5 + stores)"), (1, "group 1 (P1, fetch + P2)")):
        w = raw[:, 6 + g]
        print(f"   own work of a wavefront of {nm}: A {((w & 0x1FFFFF) * 16).double().mean():8.0f}  B {(((w >> 21) & 0x1FFFFF) * 16).double().mean():8.0f}"
              f"  C {(((w >> 42) & 0x1FFFFF) * 16).double().mean():8.0f} cycles per workgroup")
if len(tt):
    d = tt[:, 1] - tt[:, 0]
    print(f"walk: {len(tt)} workgroups, body cycles mean {d.mean():.0f} min {d.min():.0f} max {d.max():.0f}; span {(tt[:, 1].max() - tt[:, 0].min()):.0f}")
    names = ["row-pair fill (H, tables, P1, P2)", "step A: P3 split/product/pre", "step B: P4 | P1(next)", "step C: P5 + store | P2(next)"]
    for i, nm in enumerate(names):
        print(f"   {nm:34s} {tt[:, 2 + i].mean():9.0f} cycles per workgroup ({100 * tt[:, 2 + i].mean() / d.mean():4.1f} %)")

# ---------------------------------------------------------------- backward
L.fl_debug_set_walk(0, 0, 0, None)
_, Xs_ref = ops._spec_mid(S.clone(), B, N, N, nfft, None, False, True, False, 1.0, 0, 0)
_, gY_ref = ops._spec_mid(Sg.clone(), B, N, N, nfft, None, False, True, False, 1.0 / nfft, 1, 0)
gH_ref = ops._gradh_launch(gY_ref.movedim(-1, 1), Xs_ref.movedim(-1, 1), False).movedim(-1, 0)
gH_ref_p = ops._h_planar(gH_ref, True)
# the pair-major spectrum against the stored one: unit (r, b), e = 0 -> bins r*L2 + p (row-major order)
L1, L2 = nfft // 2 // 240, 240
Xpv = Xp.view(L1 // 2 + 1, B, 2, N // 2, L2, 2).permute(0, 1, 2, 3, 5, 4).reshape(L1 // 2 + 1, B, 2, N, L2)     # [n/2][pair][n%2] -> [n][pair]
Xs_rm = Xs_ref            # (B, N, M) row-major bin order
for r in (1, 7, 57, 99):
    want = Xs_rm[:, :, r * L2:(r + 1) * L2]
    got = Xpv[r, :, 0].reshape(B, N, L2)
    wantm = Xs_rm[:, :, (L1 - r) * L2:(L1 - r + 1) * L2].flip(-1)
    gotm = Xpv[r, :, 1].reshape(B, N, L2)
    print(f"pair-major spectrum row pair {r}: k {rel(torch.view_as_real(got), torch.view_as_real(want)):.2e}  L-k {rel(torch.view_as_real(gotm), torch.view_as_real(wantm)):.2e}")
for ns in [int(v) for v in args.slices.split(",")]:
    # "4": four output-channel groups per row pair; "17": doubled staging regions, transfers issued by the wavefronts without an FFT stage (tuning variant)
    # "15": ALL output channels in one workgroup (every wavefront has an FFT stage in step 2), the batch in two slices
    L.fl_debug_set_walk(14 if ns == 4 else 17 if ns == 17 else 15 if ns == 15 else 1, 0, 0 if ns in (4, 15, 17) else ns, None)
    nsl = L.fl_spec_gradh_slices(nfft, B)
    parts = torch.full((nsl, N, N, P), float("nan"), dtype=torch.complex64, device=dev)
    out = torch.empty((N, N, P), dtype=torch.complex64, device=dev)

    def run():
        _lib.check(L.fl_spec_gradh_walk_f32(Sg.data_ptr(), Xp.data_ptr(), parts.data_ptr(), N * N * P, N * P, P, nsl, W.data_ptr(), nfft, B, N, N,
                                            1.0 / nfft, 1, ops._stream()), "gradh_walk")

    def run_sum():
        _lib.check(L.fl_sum_parts_c64(parts.data_ptr(), N * N * P, nsl, out.data_ptr(), N * N * P, ops._stream()), "sum_parts")

    run()
    run_sum()
    torch.cuda.synchronize()
    got = out[..., :M]
    ref = gH_ref_p.movedim(0, -1)
    print(f"gradh_walk slices={nsl}: rel err vs spec_mid+mimo_gradh {rel(torch.view_as_real(got), torch.view_as_real(ref)):.3e}"
          f"  nan={torch.isnan(torch.view_as_real(got)).any().item()}")
    t = timeit(run)
    t2 = timeit(run_sum)
    print(f"   gradh_walk: median {t[0]:.1f} us, min {t[1]:.1f} us; sum_parts: median {t2[0]:.1f} us")
gYs = ops._empty_rows((B, N), M, torch.complex64, dev)
t = timeit(lambda: L.fl_spec_mid_f32(Sg.data_ptr(), None, gYs.data_ptr(), N * P, P, None, 0, 0, 0, W.data_ptr(), nfft, B, N, N, 1.0 / nfft, 1, 0,
                                     ops._stream()))
t2 = timeit(lambda: ops._gradh_launch(gY_ref.movedim(-1, 1), Xs_ref.movedim(-1, 1), False))
print(f"spec_mid (spectrum only): median {t[0]:.1f} us; mimo_gradh: median {t2[0]:.1f} us")

# phase picture of the backward kernel
nblk = 8 * ((L1 // 2 + 1 + 7) // 8) * 2 * 4
bufg = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
gmode = int(os.environ.get("WALK_GRADH_MODE", "1"))       # 15: all output channels per workgroup, two batch slices
L.fl_debug_set_walk(gmode, 0, 0, bufg.data_ptr())
nsl = L.fl_spec_gradh_slices(nfft, B)
parts = torch.empty((nsl, N, N, P), dtype=torch.complex64, device=dev)
_lib.check(L.fl_spec_gradh_walk_f32(Sg.data_ptr(), Xp.data_ptr(), parts.data_ptr(), N * N * P, N * P, P, nsl, W.data_ptr(), nfft, B, N, N, 1.0 / nfft, 1, ops._stream()), "gradh_walk")
torch.cuda.synchronize()
L.fl_debug_set_walk(1, 0, 0, None)
tg = bufg.view(-1, 8).cpu().double()
tg = tg[tg[:, 1] > 0]
if len(tg):
    d = tg[:, 1] - tg[:, 0]
    print(f"gradh_walk: {len(tg)} workgroups, cycles mean {d.mean():.0f} min {d.min():.0f} max {d.max():.0f}; step 1 {tg[:, 2].mean():.0f}, step 2 {tg[:, 3].mean():.0f}; "
          f"own work in step 2: wavefront 0 {tg[:, 4].mean():.0f}, 1 {tg[:, 5].mean():.0f}, 4 {tg[:, 6].mean():.0f}, 7 {tg[:, 7].mean():.0f}")

# forward kernel: per-workgroup cycles against the partition's cost model (units, row-pair entries)
if len(tt) and BND is not None:
    G = len(bl) - 1
    full = buf.view(-1, 8).cpu().double()
    rows = []
    for blk in range(G):
        w = (blk & 7) * (G >> 3) + (blk >> 3) if G % 8 == 0 else blk
        lo, hi = bl[w], bl[w + 1]
        if hi <= lo:
            continue
        fills = (hi - 1) // B - lo // B + 1
        halves = sum(1 for u in range(lo, hi) if (u // B) in (0, (nfft // 2 // 240) // 2))
        rows.append((full[blk, 1] - full[blk, 0], hi - lo, fills, halves, blk & 7))
    rows.sort(key=lambda r: -r[0])
    print("slowest workgroups (cycles, units, row-pair entries, self-mirrored units, XCD):", [(int(c), n, f, h, x) for c, n, f, h, x in rows[:8]])
    print("fastest:", [(int(c), n, f, h, x) for c, n, f, h, x in rows[-4:]])
    import collections
    by = collections.defaultdict(list)
    for c, n, f, h, x in rows:
        by[(n, f, h)].append(c.item())
    for k in sorted(by):
        v = by[k]
        print(f"   units {k[0]:2d} entries {k[1]} self-mirrored {k[2]:2d}: {len(v):3d} workgroups, cycles mean {sum(v) / len(v):8.0f} min {min(v):8.0f} max {max(v):8.0f}")
    byx = collections.defaultdict(list)
    for c, n, f, h, x in rows:
        byx[x].append(c.item())
    print("   mean cycles per XCD:", {x: int(sum(v) / len(v)) for x, v in sorted(byx.items())})
