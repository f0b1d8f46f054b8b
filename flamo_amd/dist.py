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

import os
import warnings
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


# ----------------------------------------------------------------------------- all-gather algorithm
# "rccl":   one all_gather_into_tensor; RCCL picks the algorithm (ring / direct by message size and topology).
# "direct": the full-mesh form of SURVEY 8-e1 -- every rank posts world-1 peer sends and world-1 peer receives at once
#           (torch.distributed.batch_isend_irecv = one RCCL group call), so its slice leaves on all 7 xGMI links
#           concurrently instead of travelling a ring that is bound by one link.  The payloads here are small (config 5:
#           6 MB per rank), where the ring's world-1 hops of latency are what it costs.
# Selected by FLAMO_ALLGATHER or set_all_gather_algorithm(); results are identical (data movement only).
_AG_ALGOS = ("rccl", "direct")
_ag_algo = os.environ.get("FLAMO_ALLGATHER", "rccl").lower()
if _ag_algo not in _AG_ALGOS:
    raise ValueError(f"FLAMO_ALLGATHER={_ag_algo!r}: expected one of {_AG_ALGOS}")


def set_all_gather_algorithm(name: str) -> str:
    """Choose how all_gather_bins moves the shards ("rccl" | "direct"); returns the previous choice."""
    global _ag_algo
    if name not in _AG_ALGOS:
        raise ValueError(f"all-gather algorithm {name!r}: expected one of {_AG_ALGOS}")
    prev, _ag_algo = _ag_algo, name
    return prev


def get_all_gather_algorithm() -> str:
    return _ag_algo


def _gather_blocks(buf: torch.Tensor, world: int, rank: int, group) -> torch.Tensor:
    """(world, *buf.shape): block q = rank q's buf.  Equal shapes on every rank."""
    out = torch.empty((world, *buf.shape), dtype=buf.dtype, device=buf.device)
    if _ag_algo == "direct" and world > 1:
        out[rank].copy_(buf)
        p2p = []
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world
            p2p.append(dist.P2POp(dist.isend, buf, dist.get_global_rank(group, to) if group is not None else to, group))
            p2p.append(dist.P2POp(dist.irecv, out[frm], dist.get_global_rank(group, frm) if group is not None else frm, group))
        for w in dist.batch_isend_irecv(p2p):
            w.wait()
    else:
        dist.all_gather_into_tensor(out.view(world * buf.shape[0], *buf.shape[1:]), buf, group=group)
    return out


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
            out = _gather_blocks(buf.cpu(), world, rank, group).to(buf.device)
        else:
            out = _gather_blocks(buf, world, rank, group)
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


_flat_cache = {}


class BucketReducer:
    """Data-parallel gradient sums of a replayed step without a copy between two replays: the captured graph packs its
    gradients into two alternating flat buckets (graph.GraphedStep(grad_buckets=True), csrc/reduce.hip fl_pack_toggle), and
    after each replay ONE asynchronous all-reduce runs on the bucket that replay filled -- in place, the sums stay there
    (DistributedDataParallel's gradient-as-bucket-view) -- while the next replay already fills the other bucket.

        red = BucketReducer(step)            # step = GraphedStep(..., grad_buckets=True)
        loss = red.replay()                  # orders itself behind the collective that last used the bucket it will fill
        sums = red.reduced()                 # waits (on the stream) for this step's collective: one view per parameter

    What is ordered, and where: the collective of step k starts behind replay k (the communicator's stream waits for the
    launch stream when the collective is issued); replay k + 2 rewrites the bucket collective k worked on, so replay()
    makes the launch stream wait for that collective first -- it finished a whole step earlier, the wait costs nothing;
    reduced() waits for the newest collective.  The host never blocks.
    (Measured with one rank over RCCL on an MI355X, config 2: 0.354-0.357 ms per step, the same as all_reduce_grads(in_buffer=
    True) behind each replay and 17-20 us above the step without a collective -- the copy this class removes was 5 us of it;
    the rest is idle device in front of the next graph launch wherever a cross-stream event sits between two replays
    (tools/dbg/dist_timeline.py).  bench.py therefore stays on all_reduce_grads, the path its two-rank CPU tests cover.)"""

    def __init__(self, step, group=None):
        if getattr(step, "buckets", None) is None:
            raise ValueError("BucketReducer: the step was not captured with grad_buckets=True")
        self.step, self.group = step, group
        self._work = [None, None]
        self._replays_seen = step.replays

    def _finish(self, b):
        w, self._work[b] = self._work[b], None
        if w is not None:
            w.wait()

    def replay(self, *inputs) -> torch.Tensor:
        if self.step.replays != self._replays_seen:
            raise RuntimeError("BucketReducer: the step was replayed outside the reducer (its device-side bucket counter and the "
                               "reducer's bookkeeping no longer agree, a collective may still own the bucket that replay filled); "
                               "replay through the reducer only")
        nxt = self.step.replays & 1                  # the bucket this replay fills
        self._finish(nxt)
        loss = self.step(*inputs) if inputs else self.step.replay()
        self._replays_seen = self.step.replays
        flat = self.step.buckets[nxt]
        if _host_staged(flat, self.group):
            host = flat.cpu()
            dist.all_reduce(host, group=self.group)
            flat.copy_(host)
        else:
            self._work[nxt] = dist.all_reduce(flat, group=self.group, async_op=True)
        return loss

    def reduced(self):
        """the summed gradients of the last replay: one view of its bucket per parameter (None where there is no gradient)"""
        b = self.step.bucket
        self._finish(b)
        return self.step.bucket_views[b]

    def finish(self):
        self._finish(0)
        self._finish(1)


def all_reduce_grads(params: Iterable[torch.nn.Parameter], group=None, async_op: bool = False, in_buffer: bool = False):
    """Sum the gradients of replicated parameters over ranks with ONE flat all-reduce per dtype, through a flat
    buffer that is allocated once per parameter set (no cat / cast / per-parameter temporaries per step): the gradients
    are copied into views of the buffer (one multi-tensor copy), reduced in place, and copied back.
    async_op: returns a callable that waits for the collective and copies the sums back -- call it before the
    gradients are read AND before the next backward pass (or graph replay) rewrites them: a handle that is still
    open when the same parameters are reduced again is waited for and its sums are dropped, with a warning.
    None otherwise.
    in_buffer (with async_op): the sums STAY in the flat buffer -- the handle's ``reduced`` list holds one view per parameter,
    shaped like its gradient, which is where an optimiser reads them (DistributedDataParallel's gradient-as-bucket-view);
    nothing is copied back unless the handle is called with ``copy_back=True``.  The parameters' own ``.grad`` tensors are free
    at once, so the next backward pass (or graph replay) may start while the collective is still in flight: the next call on
    the same parameters orders its refill of the buffer behind the previous collective on the stream (no host wait, no
    warning)."""
    if in_buffer and not async_op:
        raise ValueError("all_reduce_grads: in_buffer leaves the sums in the flat buffer, reachable only through the handle that "
                         "async_op=True returns; a synchronous call would leave every rank's .grad with its local values")
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return (lambda: None) if async_op else None
    by_dtype = {}
    for p in ps:
        by_dtype.setdefault(p.grad.dtype, []).append(p)
    pending = []
    for dt, plist in by_dtype.items():
        key = (dt, plist[0].grad.device, tuple(p.grad.numel() for p in plist))
        ent = _flat_cache.get(key)
        if ent is None:
            flat = torch.empty(sum(key[2]), dtype=dt, device=key[1])
            views, off = [], 0
            for n in key[2]:
                views.append(flat[off:off + n])
                off += n
            ent = _flat_cache[key] = [flat, views, None]
        flat, views, outstanding = ent
        if outstanding is not None and getattr(outstanding, "in_buffer", False):
            outstanding()              # in-buffer handle: its sums were the buffer's business; order the refill behind its collective
            outstanding = None
        if outstanding is not None:
            # an earlier asynchronous call on this buffer has not been finished: its collective may still be reading and
            # writing the buffer this call is about to refill.  Wait for it -- at most one collective per parameter set is
            # ever in flight -- but do NOT copy its sums back: the gradients have been rewritten since (that is why a new
            # reduction is being asked for), and the stale sums would replace them and be reduced again.  The caller broke
            # the contract (finish before the next backward); its earlier handle becomes a no-op.
            warnings.warn("all_reduce_grads: the previous asynchronous reduction of these parameters was never finished; "
                          "its sums are discarded (call the returned handle before the next backward pass)", RuntimeWarning,
                          stacklevel=2)
            outstanding(copy_back=False)
        grads = [p.grad.reshape(-1) for p in plist]
        torch._foreach_copy_(views, grads)
        if _host_staged(flat, group):
            host = flat.cpu()
            dist.all_reduce(host, group=group)
            flat.copy_(host)
            work = None
        else:
            work = dist.all_reduce(flat, group=group, async_op=async_op)
        pending.append((work, views, plist, ent))

    done = [False]

    def finish(copy_back: bool = not in_buffer):
        if done[0]:
            return
        done[0] = True
        for work, views, plist, ent in pending:
            if ent[2] is finish:
                ent[2] = None
            if work is not None:
                work.wait()
            if not copy_back:
                continue
            dst, src = [], []
            for p, v in zip(plist, views):
                if p.grad is None:          # cleared since the call: nothing to hand the sum to
                    continue
                if p.grad.is_contiguous():
                    dst.append(p.grad.view(-1))
                    src.append(v)
                else:
                    p.grad.copy_(v.view_as(p.grad))
            if dst:
                torch._foreach_copy_(dst, src)

    if async_op:
        finish.in_buffer = bool(in_buffer)
        finish.reduced = [v.view_as(p.grad) for _, views, plist, _ in pending for p, v in zip(plist, views)]
        finish.params = [p for _, _, plist, _ in pending for p in plist]
        for _, _, _, ent in pending:
            ent[2] = finish
        return finish
    finish()
    return None


# ----------------------------------------------------------------------------- batch-sharded <-> bin-sharded spectra
def _split_sizes(M: int, world: int):
    return [shard_bins(M, r, world)[1] for r in range(world)]


class _BatchToBins(torch.autograd.Function):
    """(B_local, M, rest...) on every rank (batch-sharded, all bins) -> (B_local * world, m_local, rest...) (all batch
    items, this rank's bins): one all-to-all.  Backward: the opposite exchange."""

    @staticmethod
    def forward(ctx, X, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Bl, M = X.shape[0], X.shape[1]
        rest = tuple(X.shape[2:])
        sizes = _split_sizes(M, world)
        ctx.meta = (Bl, M, rest, sizes, group)
        # send block q: my batch items, rank q's bins, laid out (m_q, Bl, rest...) so that blocks concatenate along dim 0
        send = X.movedim(1, 0).contiguous()                                # (M, Bl, rest...)
        recv = torch.empty((sizes[rank] * world, Bl, *rest), dtype=X.dtype, device=X.device)
        _all_to_all(recv, send, [sizes[rank]] * world, sizes, group)
        # recv: world blocks (m_local, Bl, rest...) -> (world*Bl, m_local, rest...)
        out = recv.view(world, sizes[rank], Bl, *rest).movedim(1, 2).reshape(world * Bl, sizes[rank], *rest)
        return out

    @staticmethod
    def backward(ctx, g):
        Bl, M, rest, sizes, group = ctx.meta
        return _bins_to_batch(g, Bl, M, rest, sizes, group), None


def _bins_to_batch(Y, Bl, M, rest, sizes, group):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ml = sizes[rank]
    send = Y.reshape(world, Bl, ml, *rest).movedim(2, 1).contiguous().view(world * ml, Bl, *rest)   # block q: batch items of rank q
    recv = torch.empty((M, Bl, *rest), dtype=Y.dtype, device=Y.device)
    _all_to_all(recv, send, sizes, [ml] * world, group)
    return recv.movedim(0, 1)                                                # (Bl, M, rest...)


class _BinsToBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, M, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        sizes = _split_sizes(M, world)
        Bl = Y.shape[0] // world
        rest = tuple(Y.shape[2:])
        ctx.meta = group
        return _bins_to_batch(Y, Bl, M, rest, sizes, group)

    @staticmethod
    def backward(ctx, g):
        return _BatchToBins.apply(g, ctx.meta), None, None


def _all_to_all(recv, send, recv_splits, send_splits, group):
    cplx = send.is_complex()
    s = torch.view_as_real(send) if cplx else send
    r = torch.view_as_real(recv) if cplx else recv
    if dist.get_backend(group) == "gloo":
        # gloo has no all_to_all_single on every build: pairwise through all_gather of the split tables is overkill for a
        # test transport -- emulate with point-to-point lists (host memory)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        sh = s.cpu() if s.is_cuda else s
        outs = list(torch.split(torch.empty((sum(recv_splits), *sh.shape[1:]), dtype=sh.dtype), recv_splits))
        ins = list(torch.split(sh, send_splits))
        reqs = []
        for q in range(world):
            if q == rank:
                outs[q].copy_(ins[q])
            else:
                reqs.append(dist.isend(ins[q].contiguous(), q, group=group))
        for q in range(world):
            if q != rank:
                dist.recv(outs[q], q, group=group)
        for w in reqs:
            w.wait()
        r.copy_(torch.cat(outs).to(r.device))
    else:
        dist.all_to_all_single(r, s, recv_splits, send_splits, group=group)


def batch_to_bins(X: torch.Tensor, group=None) -> torch.Tensor:
    """Batch-sharded spectrum (B_local, M, ...) -> bin-sharded (B_local * world, m_local, ...) (differentiable)."""
    return _BatchToBins.apply(X, group)


def bins_to_batch(Y: torch.Tensor, M: int, group=None) -> torch.Tensor:
    """Bin-sharded (B_local * world, m_local, ...) -> batch-sharded (B_local, M, ...) (differentiable)."""
    return _BinsToBatch.apply(Y, M, group)


def bin_exchange_forward(shell, x: torch.Tensor, group=None) -> torch.Tensor:
    """``shell(x)`` for a batch-sharded input with the core evaluated bin-sharded (large-batch form):
    local input transform of this rank's batch items -> all-to-all into bin shards (every rank: all batch items, its
    own bins) -> core on the local bins with locally generated responses -> all-to-all back -> local inverse
    transform.  Two data-path collectives per direction; parameters replicated."""
    X = shell.get_inputLayer()(x)
    M = X.shape[1]
    Xb = batch_to_bins(X, group)
    with bin_shard(shell.nfft, dist.get_rank(group), dist.get_world_size(group)):
        Yb = shell.get_core()(Xb)
    Y = bins_to_batch(Yb, M, group)
    return shell.get_outputLayer()(Y)


def sharded_forward(shell, x: torch.Tensor, group=None) -> torch.Tensor:
    """``shell(x)`` with the core evaluated on this rank's bins only:
    input_layer (replicated) -> local bins -> core -> all-gather -> output_layer."""
    X = shell.get_inputLayer()(x)
    M = X.shape[1]
    with bin_shard(shell.nfft, dist.get_rank(group), dist.get_world_size(group)):
        Yl = shell.get_core()(take_local_bins(X, group))
    Y = all_gather_bins(Yl, M, group)
    return shell.get_outputLayer()(Y)
