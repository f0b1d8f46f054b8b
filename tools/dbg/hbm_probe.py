"""What the box's memory system sustains on hand-written streaming kernels (fl_hbm_probe, csrc/probe.hip).

  python tools/dbg/hbm_probe.py [--json out.json]

Sweeps access mix (read / write / copy / 8:1 read-mostly) x buffer size (98 MB = the fused pipeline's scratch, resident in
the 256 MiB Infinity Cache; 1 GiB = HBM) x cache policy (plain / non-temporal) x persistent grid (k x 256 workgroups), and the
producer -> consumer hand-over of a 98 MB buffer: a write pass followed by a read pass in the same or the opposite address
order, each pair behind a 1 GiB flush."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib  # noqa: E402

KINDS = {0: "read", 1: "write", 2: "copy", 3: "read8_write1"}


def moved_bytes(kind, nbytes):
    return {0: nbytes, 1: nbytes, 2: 2 * nbytes, 3: nbytes + nbytes // 8}[kind]


def launch(L, kind, src, dst, nbytes, wgs, flags, partial):
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.fl_hbm_probe(kind, src.data_ptr(), dst.data_ptr(), nbytes, wgs, flags, partial.data_ptr(), st), "hbm_probe")


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    GiB = 1 << 30
    small = 98304000 // 32768 * 32768          # 98.3 MB: the scratch between two launches of the fused pipeline
    bufs = {"98MB": small, "1GiB": GiB}
    src = torch.empty(GiB // 4, device=dev).normal_()
    dst = torch.empty(GiB // 4, device=dev)
    flush = torch.empty(GiB // 4, device=dev)
    partial = torch.empty(65536, device=dev)
    out = {"sweep": [], "handover": []}
    print(f"{'mix':14s} {'size':6s} {'policy':8s} " + " ".join(f"k={k:<5d}" for k in (1, 2, 4, 8, 16)))
    for kind, name in KINDS.items():
        for label, nbytes in bufs.items():
            for flags, pol in ((0, "plain"), (3, "nt")):
                row = []
                for k in (1, 2, 4, 8, 16):
                    t = timed(lambda: launch(L, kind, src, dst, nbytes, 256 * k, flags, partial), 20)
                    gbs = moved_bytes(kind, nbytes) / t / 1e9
                    row.append(gbs)
                    out["sweep"].append(dict(mix=name, size=label, policy=pol, wgs=256 * k, GBs=round(gbs, 1)))
                print(f"{name:14s} {label:6s} {pol:8s} " + " ".join(f"{g:7.0f}" for g in row))
    # flushed single launches (what a pass sees when its operands are NOT cache-resident), best grid
    print("\nflushed (1 GiB write in front of every timed launch), k = 8:")
    for kind, name in KINDS.items():
        for flags, pol in ((0, "plain"), (3, "nt")):
            ts = []
            for _ in range(8):
                launch(L, 1, src, flush, GiB, 2048, 0, partial)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(L, kind, src, dst, small, 2048, flags, partial)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3)
            t = sorted(ts)[len(ts) // 2]
            gbs = moved_bytes(kind, small) / t / 1e9
            out["sweep"].append(dict(mix=name, size="98MB_flushed", policy=pol, wgs=2048, GBs=round(gbs, 1)))
            print(f"  {name:14s} {pol:6s} {gbs:7.0f} GB/s  ({t * 1e6:.1f} us)")
    # producer -> consumer: write 98 MB, then read it in the same / the opposite order
    print("\nhand-over of a 98 MB buffer (flush, write pass, timed read pass):")
    for wflags, wpol in ((0, "plain"), (2, "nt-store")):
        for rev in (0, 1):
            for rflags, rpol in ((0, "plain"), (1, "nt-load")):
                ts, tw = [], []
                for _ in range(8):
                    launch(L, 1, src, flush, GiB, 2048, 0, partial)
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    e0.record()
                    launch(L, 1, src, dst, small, 2048, wflags, partial)
                    e1.record()
                    launch(L, 0, dst, dst, small, 2048, rflags | (rev << 2), partial)
                    e2.record()
                    torch.cuda.synchronize()
                    tw.append(e0.elapsed_time(e1) * 1e-3)
                    ts.append(e1.elapsed_time(e2) * 1e-3)
                t, w = sorted(ts)[len(ts) // 2], sorted(tw)[len(tw) // 2]
                rec = dict(store=wpol, load=rpol, reader_order="reverse" if rev else "same", write_GBs=round(small / w / 1e9, 1),
                           read_GBs=round(small / t / 1e9, 1))
                out["handover"].append(rec)
                print("  ", rec)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
