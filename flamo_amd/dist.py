"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

Two ways the path shards:

* **bins** -- every module between FFT and iFFT acts on each frequency bin independently, so the
  M = nfft//2+1 bins are split into contiguous ranges, one per rank.  Parameters are replicated
  (a few KB); each rank generates responses for its own bins only (``ops.set_bin_shard``), runs
  the per-bin products / the Recursion solve on (B, M_local, N) tensors, and ONE all-gather of
  the core output reassembles the spectrum before the inverse FFT.  M is odd for every even
  nfft, so shards are uneven: ranks pad to the largest shard for the collective and trim.
  In backward the gathered gradient is sliced (no collective); replicated-parameter gradients
  are summed with one small all-reduce.
* **batch** -- for large-batch Series-only workloads (BASELINE config 2) plain data parallelism:
  no data-path collective at all, only the parameter-gradient all-reduce (bench.py uses this).

Collectives are issued on the current stream through ``torch.distributed`` (backend "nccl" is
RCCL on ROCm; the same code runs over "gloo" on CPU tensors, which is how tests/ cover it).
Payloads are small (config 5: 49 MB gathered, 6 MB per rank), so one collective per step is used
rather than bucketing.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


def shard_bins(M: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous bin range (bin0, m_local) of `rank`: ceil(M/world) bins per rank, the last
    rank(s) short (possibly empty when world > M)."""
    per = -(-M // world)
    bin0 = min(rank * per, M)
    return bin0, max(0, min(per, M - bin0))


@contextmanager
def bin_shard(nfft: int, rank: Optional[int] = None, world: Optional[int] = None):
    """Within the context, response generators produce only this rank's bins."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    bin0, m_local = shard_bins(nfft // 2 + 1, rank, world)
    ops.set_bin_shard(bin0, m_local)
    try:
        yield bin0, m_local
    finally:
        ops.set_bin_shard(0, None)


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo moves host memory: device tensors are staged through the host (ranks sharing one GPU on a
    test rig; RCCL refuses two ranks on one device).  Transport only -- no arithmetic moves to the CPU."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _AllGatherBins(torch.autograd.Function):
    """(B, m_local, ...) per rank -> (B, M, ...) on every rank; backward = local slice."""

    @staticmethod
    def forward(ctx, y_local, M, group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        per = -(-M // world)
        bin0, m_local = shard_bins(M, rank, world)
        assert y_local.shape[1] == m_local, f"rank {rank}: expected {m_local} local bins, got {y_local.shape[1]}"
        B, rest = y_local.shape[0], tuple(y_local.shape[2:])
        # the collective moves (world, B, rest..., per) blocks with the bin axis innermost, i.e. the
        # planar layout the kernels produce: no transposition on either side
        send = y_local.movedim(1, -1)
        if m_local < per:
            pad = torch.zeros((B, *rest, per - m_local), dtype=y_local.dtype, device=y_local.device)
            send = torch.cat([send, pad], dim=-1)
        send = send.contiguous()
        cplx = send.is_complex()
        buf = torch.view_as_real(send) if cplx else send
        if _host_staged(buf, group):
            parts = [torch.empty(buf.shape, dtype=buf.dtype) for _ in range(world)]
            dist.all_gather(parts, buf.cpu(), group=group)
            out = torch.stack(parts).to(buf.device)
        else:
            out = torch.empty((world * buf.shape[0], *buf.shape[1:]), dtype=buf.dtype, device=buf.device)
            dist.all_gather_into_tensor(out, buf, group=group)     # concatenated along dim 0
        out = out.view(world, *buf.shape)
        if cplx:
            out = torch.view_as_complex(out)
        # (world, B, rest..., per) -> (B, rest..., world*per) -> trim -> logical (B, M, rest...)
        full = out.movedim(0, -2).reshape(B, *rest, world * per)[..., :M]
        ctx.meta = (bin0, m_local)
        return full.movedim(-1, 1)

    @staticmethod
    def backward(ctx, g):
        bin0, m_local = ctx.meta
        return g[:, bin0:bin0 + m_local], None, None


def all_gather_bins(y_local: torch.Tensor, M: int, group=None) -> torch.Tensor:
    return _AllGatherBins.apply(y_local, M, group)


def take_local_bins(X: torch.Tensor, group=None) -> torch.Tensor:
    """This rank's slice of a replicated spectrum (B, M, ...) (differentiable)."""
    bin0, m_local = shard_bins(X.shape[1], dist.get_rank(group), dist.get_world_size(group))
    return X[:, bin0:bin0 + m_local]


def all_reduce_grads(params: Iterable[torch.nn.Parameter], group=None) -> None:
    """Sum the gradients of replicated parameters over ranks with ONE flat all-reduce."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1).to(torch.float64) for p in ps])
    if _host_staged(flat, group):
        host = flat.cpu()
        dist.all_reduce(host, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, group=group)
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad).to(p.grad.dtype))
        off += n


def sharded_forward(shell, x: torch.Tensor, group=None) -> torch.Tensor:
    """``shell(x)`` with the core evaluated on this rank's bins only:
    input_layer (replicated) -> local bins -> core -> all-gather -> output_layer."""
    X = shell.get_inputLayer()(x)
    M = X.shape[1]
    with bin_shard(shell.nfft, dist.get_rank(group), dist.get_world_size(group)):
        Yl = shell.get_core()(take_local_bins(X, group))
    Y = all_gather_bins(Yl, M, group)
    return shell.get_outputLayer()(Y)
