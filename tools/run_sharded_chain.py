#!/usr/bin/env python
"""BASELINE configs[4] with the frequency bins sharded over ranks: the active-acoustics structure
  FFTAntiAlias(nfft, 30 dB) -> Series(GEQ((N,N)), Recursion(fF=Series(Delay((N,N), isint), parallelGain(N)),
  fB=Matrix(N,N, orthogonal))) -> iFFTAntiAlias
forward + backward of (y * c).sum() with gradients for the GEQ gains, the loop gains and the mixing matrix.

One GPU:  python tools/run_sharded_chain.py [--N 32 --nfft 384000 --steps 10]
N GPUs:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
              tools/run_sharded_chain.py --gpus N
          every rank transforms the (small) input, generates responses, composes the loop matrix and solves for ITS
          bins only (flamo_amd.dist.sharded_forward); ONE all-gather reassembles the (B, M, N) spectrum in front of the
          inverse transform (49 MB at nfft = 384000, N = 32); the backward of the gather is a local slice and one
          flat all-reduce sums the replicated parameters' gradients.  Strong scaling: M/world bins per rank.
Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--N", type=int, default=32)
    ap.add_argument("--nfft", type=int, default=384000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) or gloo (ranks sharing one GPU, staged through host)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (test rigs with one GPU)")
    ap.add_argument("--dump", default=None, help="rank 0 saves {y, grads} here (torch.save)")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    import torch.distributed as dist
    from bench_fdn import build_config5
    from flamo_amd import dist as fd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    dtype = getattr(torch, args.dtype)
    torch.manual_seed(130709)                       # replicated parameters and data: the same draw on every rank
    model, params = build_config5(dev, dtype, args.N, args.nfft)
    x = torch.randn(args.batch, args.nfft, args.N, device=dev, dtype=dtype)
    c = torch.randn(args.batch, args.nfft, args.N, device=dev, dtype=dtype)

    def step():
        for p in params:
            p.grad = None
        y = fd.sharded_forward(model, x) if world > 1 else model(x)
        (y * c).sum().backward()
        if world > 1:
            fd.all_reduce_grads(params)
        return y

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    M = args.nfft // 2 + 1
    if rank == 0:
        print(json.dumps({"metric": "chain_bin_solves_per_s", "value": args.batch * M * args.steps / dt, "unit": "bin-solves/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "scaling": "strong", "dtype": args.dtype,
                          "config": {"workload": "configs[4] active-acoustics chain", "N": args.N, "nfft": args.nfft,
                                     "batch": args.batch, "sharding": "bins" if world > 1 else "none"}}))
        if args.dump:
            torch.save({"y": y.detach().cpu(), "grads": [p.grad.detach().cpu() for p in params]}, args.dump)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
