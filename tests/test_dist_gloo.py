"""Multi-process (world_size 2, gloo, CPU) tests of the bin-sharding collectives in flamo_amd.dist:
uneven all-gather with autograd, local slicing, gradient all-reduce, shard coverage."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flamo_amd import dist as fd
        torch.manual_seed(0)                                  # same "replicated" tensors on both ranks
        full = torch.randn(3, M, 4, 2, dtype=torch.complex128)
        bin0, m_local = fd.shard_bins(M, rank, world)
        local = fd.take_local_bins(full).clone().requires_grad_(True)
        assert local.shape[1] == m_local
        gathered = fd.all_gather_bins(local, M)
        ok_fwd = torch.equal(gathered, full)                  # data movement only: bit exact
        prev = fd.set_all_gather_algorithm("direct")          # full-mesh peer sends: the same bits
        try:
            ok_fwd = ok_fwd and torch.equal(fd.all_gather_bins(local, M), full)
        finally:
            fd.set_all_gather_algorithm(prev)
        w = torch.randn(3, M, 4, 2, dtype=torch.complex128)
        loss = torch.sum(torch.real(gathered * torch.conj(w)))
        (g,) = torch.autograd.grad(loss, [local])
        ok_bwd = torch.allclose(g, w[:, bin0:bin0 + m_local])
        # replicated-parameter gradient all-reduce
        p = torch.nn.Parameter(torch.zeros(5))
        p.grad = torch.full((5,), float(rank + 1))
        q = torch.nn.Parameter(torch.zeros(2, 2, dtype=torch.float32))
        q.grad = torch.ones(2, 2) * (10 ** rank)
        fd.all_reduce_grads([p, q])
        ok_red = torch.allclose(p.grad, torch.full((5,), float(sum(range(1, world + 1))))) and \
            torch.allclose(q.grad, torch.full((2, 2), float(sum(10 ** r for r in range(world)))))
        # real-valued tensors and the shard context
        with fd.bin_shard(2 * (M - 1)) as (b0, ml):
            from flamo_amd import ops
            ok_ctx = ops.bin_shard(2 * (M - 1)) == (b0, ml) == (bin0, m_local)
        ok_ctx = ok_ctx and ops.bin_shard(2 * (M - 1)) == (0, M)
        rl = fd.all_gather_bins(full.real[:, bin0:bin0 + m_local].contiguous(), M)
        # batch-sharded <-> bin-sharded exchange (all-to-all both ways) with autograd
        Bl = 2
        torch.manual_seed(1)
        glob = torch.randn(world * Bl, M, 3, dtype=torch.complex128)          # the global batch, known to both ranks
        mine = glob[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
        xb = fd.batch_to_bins(mine)
        ok_x = xb.shape == (world * Bl, m_local, 3) and torch.equal(xb, glob[:, bin0:bin0 + m_local])
        back = fd.bins_to_batch(xb * 2.0, M)
        ok_x = ok_x and torch.equal(back, 2.0 * mine)
        wgt = torch.randn(Bl, M, 3, dtype=torch.complex128)
        (gm,) = torch.autograd.grad(torch.sum(torch.real(back * torch.conj(wgt))), [mine])
        ok_x = ok_x and torch.allclose(gm, 2.0 * wgt)
        # asynchronous gradient all-reduce through the cached flat buffer, twice (buffer reuse)
        for rep in range(2):
            p.grad = torch.full((5,), float(rank + 1 + rep))
            fin = fd.all_reduce_grads([p], async_op=True)
            fin()
            ok_red = ok_red and torch.allclose(p.grad, torch.full((5,), float(sum(r + 1 + rep for r in range(world)))))
        # a handle left open across a "backward" is waited for, its stale sums are NOT written over the fresh gradients
        import warnings
        p.grad = torch.full((5,), 100.0)
        fd.all_reduce_grads([p], async_op=True)               # never finished
        p.grad = torch.full((5,), float(rank + 1))            # the next backward pass
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            fd.all_reduce_grads([p])
        ok_red = ok_red and torch.allclose(p.grad, torch.full((5,), float(sum(range(1, world + 1))))) and \
            any("never finished" in str(w.message) for w in wlist)
        # sums left in the flat buffer (DDP's gradient-as-bucket-view): the gradients themselves stay local and free for the next
        # backward pass; a second reduction orders itself behind the first without a warning
        p.grad = torch.full((5,), float(rank + 1))
        h1 = fd.all_reduce_grads([p], async_op=True, in_buffer=True)
        p.grad = torch.full((5,), 10.0 * (rank + 1))          # the next backward pass, while h1 may still be in flight
        with warnings.catch_warnings(record=True) as wlist2:
            warnings.simplefilter("always")
            h2 = fd.all_reduce_grads([p], async_op=True, in_buffer=True)
        h2()
        ok_red = ok_red and not wlist2 and torch.allclose(h2.reduced[0], torch.full((5,), 10.0 * sum(range(1, world + 1)))) and \
            torch.allclose(p.grad, torch.full((5,), 10.0 * (rank + 1))) and h2.params[0] is p
        h2(copy_back=True)                                    # (already finished: a no-op)
        results[rank] = bool(ok_fwd and ok_bwd and ok_red and ok_ctx and ok_x and torch.equal(rl, full.real))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [49, 48001 // 100 + 1, 2])
def test_bin_sharding_collectives_gloo(M):
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), M, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_shard_ranges_cover_all_bins():
    from flamo_amd.dist import shard_bins
    for M in (1, 2, 7, 48001, 96001, 192001):
        for world in (1, 2, 4, 8, 16):
            spans = [shard_bins(M, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(m for _, m in spans) == M
            for (b0, m0), (b1, _) in zip(spans, spans[1:]):
                assert b1 == min(b0 + m0, M) or m0 == 0
            assert max(m for _, m in spans) == -(-M // world)
