#!/usr/bin/env python
"""BASELINE configs[3]: colorless FDN training (examples/e8_colorless_fdn.py) on the HIP path.

  Shell( FFT(nfft) -> Series(input_gain Gain(N,1), feedback_loop Recursion(fF=parallelDelay(isint), fB=Matrix(N,N,
  orthogonal)), output_gain Gain(1,N)) -> Transform(abs) ),  alias_decay_db = 30
  data      DatasetColorless: an impulse of M = nfft//2+1 samples -> a flat magnitude of ones (optimize/dataset.py:54-85)
  criteria  1.0 * mse_loss + 0.2 * sparsity_loss (optimize/loss.py:12-103), Adam(lr) as Trainer.train_step
            (optimize/trainer.py:70, 162-191)

One GPU:   python tools/train_colorless_fdn.py [--N 16 --nfft 192000 --steps 50]
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
               tools/train_colorless_fdn.py --gpus N
           frequency bins are sharded over the ranks (flamo_amd.dist.sharded_forward): every rank builds the loop
           matrices and solves for its own bins, ONE all-gather reassembles the (B, M, 1) spectrum in front of the
           output layer / criteria, and one small all-reduce sums the replicated parameters' gradients.
Rank 0 prints one JSON line (bin-solves/s = B*M*steps / time, the loss trajectory's ends)."""
import os as _os_env
_os_env.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see flamo_amd/__init__.py: must precede HIP runtime init

import argparse
import json
import math
import os
import sys
import time
import warnings
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PRIMES16 = [503, 593, 701, 811, 919, 1031, 1151, 1259, 1381, 1493, 1613, 1741, 1873, 2003, 2381, 2713]


def build(dev, dtype, N=16, nfft=192000, db=30.0, delays=None):
    from flamo_amd.processor import dsp, system
    kw = dict(nfft=nfft, alias_decay_db=db, device=dev, dtype=dtype)
    delays = list(delays if delays is not None else PRIMES16[:N])
    ig = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
    og = dsp.Gain(size=(1, N), requires_grad=True, **kw)
    dl = dsp.parallelDelay(size=(N,), max_len=max(delays), isint=True, requires_grad=False, **kw)
    dl.assign_value(dl.sample2s(torch.tensor(delays, device=dev, dtype=dtype)))
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    core = system.Series(OrderedDict(input_gain=ig, feedback_loop=system.Recursion(fF=dl, fB=mix), output_gain=og))
    model = system.Shell(core, dsp.FFT(nfft, dtype=dtype), dsp.Transform(lambda z: torch.abs(z), device=dev, dtype=dtype))
    return model


def colorless_batch(B, nfft, dev, dtype):
    """DatasetColorless items stacked into a batch: impulse (B, M, 1), target ones (B, M, 1)."""
    M = nfft // 2 + 1
    x = torch.zeros(B, M, 1, device=dev, dtype=dtype)
    x[:, 0, :] = 1
    return x, torch.ones(B, M, 1, device=dev, dtype=dtype)


_MSE = {}


def mse_criterion(est, target):
    """mse_loss.forward (optimize/loss.py:102-103): channels summed, then the mean squared error -- the drop-in class
    flamo_amd.optimize.mse_loss, as examples/e8_colorless_fdn.py:137 instantiates the reference's (one pass each way on device
    tensors instead of torch's sum / sub / pow / mean kernels and their backward; FLAMO_TORCH_CRITERIA=1: the torch lines)."""
    if os.environ.get("FLAMO_TORCH_CRITERIA", "0") == "1":
        return torch.mean((est.sum(dim=-1) - target.squeeze(-1)) ** 2)
    if "fn" not in _MSE:
        from flamo_amd.optimize import mse_loss
        _MSE["fn"] = mse_loss()
    return _MSE["fn"](est, target)


def sparsity_criterion(model):
    """sparsity_loss.forward for a plain mixing matrix (optimize/loss.py:36-63): the drop-in class flamo_amd.optimize.sparsity_loss,
    as examples/e8_colorless_fdn.py:138 instantiates the reference's (FLAMO_TORCH_CRITERIA=1: the torch lines)."""
    if os.environ.get("FLAMO_TORCH_CRITERIA", "0") != "1":
        if "sp" not in _MSE:
            from flamo_amd.optimize import sparsity_loss
            _MSE["sp"] = sparsity_loss()
        return _MSE["sp"](None, None, model)
    mix = model.get_core().feedback_loop.feedback
    A = mix.map(mix.param)
    N = A.shape[-1]
    return -(torch.sum(torch.abs(A)) - N * math.sqrt(N)) / (N * (math.sqrt(N) - 1))


def train(model, x, target, steps, lr, group=None, log=None, graphed=False, on_ready=None, fused_adam=False, sharded=None):
    """`steps` iterations of Trainer.train_step.  With a process group the core runs on this rank's bins.
    graphed (one GPU): forward + criteria + backward replayed from a HIP graph, the Adam update launched behind it.
    sharded: None = whenever a process group with more than one rank exists; False = this rank alone, whatever groups the
    process has joined (a leg that ONE rank of a multi-rank job runs on its own must not enter a collective)."""
    import torch.distributed as dist
    from flamo_amd import dist as fd
    if sharded is None:
        sharded = group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    rank = dist.get_rank(group) if sharded else 0
    params = [p for p in model.parameters() if p.requires_grad]
    # fused_adam: the same update in one launch for all parameters (torch's foreach default is ~12 launches)
    opt = torch.optim.Adam(params, lr=lr, fused=True) if fused_adam else torch.optim.Adam(params, lr=lr)
    if graphed and steps > 0:
        assert not sharded, "graph replay is wired for the single-GPU run"
        from flamo_amd.graph import GraphedStep
        parts = {}

        def criteria(xx):
            est = model(xx)
            mse, sp = mse_criterion(est, target), sparsity_criterion(model)
            loss = mse + 0.2 * sp
            parts["v"] = torch.stack([mse, sp, loss]).detach()      # static: rewritten by every replay
            return loss

        step = GraphedStep(criteria, (x,), params)
        if on_ready is not None:
            on_ready()                      # the clock starts behind the capture
        for _ in range(steps):
            step.replay()
            opt.step()
            if log is not None:
                log.append(parts["v"].clone())
        return log
    if on_ready is not None:
        on_ready()
    from flamo_amd import ops
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        with ops.step_scope():      # the model and the sparsity criterion share one evaluation of the orthogonal map
            est = fd.sharded_forward(model, x, group) if sharded else model(x)
            mse = mse_criterion(est, target)
            sp = sparsity_criterion(model)
            loss = mse + 0.2 * sp
            # every rank holds the whole gathered spectrum and evaluates the same criteria; its backward reaches only
            # its own bins, so the data term's parameter gradients are partial sums -- the sparsity term acts on the
            # replicated matrix directly and is counted once
            (mse + (0.2 * sp if rank == 0 else 0.0 * sp)).backward()
        if sharded:
            fd.all_reduce_grads(params, group)
        opt.step()
        if log is not None:
            log.append(torch.stack([mse.detach(), sp.detach(), loss.detach()]))      # no host sync inside the loop
    return log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--N", type=int, default=16)
    ap.add_argument("--nfft", type=int, default=192000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) or gloo (ranks sharing one GPU, staged through host)")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward from a HIP graph (one GPU)")
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one launch per update")
    ap.add_argument("--dump", default=None, help="rank 0 saves {losses, state_dict} here (torch.save)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (test rigs with one GPU)")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    dtype = getattr(torch, args.dtype)
    torch.manual_seed(130709)                       # replicated parameters: the same draw on every rank
    model = build(dev, dtype, args.N, args.nfft)
    x, target = colorless_batch(args.batch, args.nfft, dev, dtype)
    train(model, x, target, args.warmup, args.lr)
    log, clock = [], {}

    def start_clock():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        clock["t0"] = time.perf_counter()

    train(model, x, target, args.steps, args.lr, log=log, graphed=args.graph, on_ready=start_clock, fused_adam=args.fused_adam)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - clock["t0"]], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    M = args.nfft // 2 + 1
    log = torch.stack(log).double().cpu().tolist()
    if rank == 0:
        print(json.dumps({"metric": "colorless_fdn_train_bin_solves_per_s", "value": args.batch * M * args.steps / dt,
                          "unit": "bin-solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "scaling": "strong", "dtype": args.dtype,
                          "config": {"workload": "configs[3] colorless FDN training", "N": args.N, "nfft": args.nfft,
                                     "batch": args.batch, "sharding": "bins" if world > 1 else "none", "lr": args.lr,
                                     "graph_replay": bool(args.graph), "fused_adam": bool(args.fused_adam)},
                          "loss_first": log[0], "loss_last": log[-1]}))
        if args.dump:
            torch.save({"losses": log, "state": {k: v.detach().cpu() for k, v in model.state_dict().items()}}, args.dump)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
