#!/usr/bin/env python
"""Per-bin eigenvalues: ops.eigvals on (F, N, N) complex matrices, forward and
forward + backward of an |eig| loss, against torch.linalg.eigvals on the same device.
    python tools/bench_eig.py [--n 32] [--bins 12000]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(f, n=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--bins", type=int, default=12000)
    args = ap.parse_args()
    from flamo_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = {"N": args.n, "bins": args.bins}
    for name, cd in (("c64", torch.complex64), ("c128", torch.complex128)):
        A = (torch.randn(args.bins, args.n, args.n, dtype=cd, device=dev) / args.n ** 0.5).requires_grad_(True)

        def fwd():
            with torch.no_grad():
                return ops.eigvals(A)

        def fwd_bwd():
            l = ops.eigvals(A)
            ((l.abs() - 0.7) ** 2).mean().backward()
            A.grad = None

        def ref():
            with torch.no_grad():
                return torch.linalg.eigvals(A)

        r = {"fwd_ms": timeit(fwd), "fwd_bwd_ms": timeit(fwd_bwd)}
        try:
            r["torch_eigvals_fwd_ms"] = timeit(ref, n=2)
        except Exception as e:   # noqa: BLE001
            r["torch_eigvals_fwd_ms"] = f"failed: {type(e).__name__}"
        r["matrices_per_s_fwd"] = args.bins / (r["fwd_ms"] * 1e-3)
        out[name] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
