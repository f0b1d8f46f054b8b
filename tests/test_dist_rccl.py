"""Multi-GPU tests over RCCL (backend "nccl"), one rank per GPU, launched by the tests themselves.  They need at least two
visible GPUs and skip on a one-GPU box; on an 8-GPU node they also run at world 4 and 8 without edits.  What they pin:
the collectives of flamo_amd.dist on device tensors (uneven all-gather with autograd in both algorithms, the batch <-> bin
all-to-all, the synchronous and asynchronous gradient all-reduce), the two sharded tools against their unsharded runs
(examples/e8_colorless_fdn.py:120-138's training loop; the config-5 chain), and bench.py's multi-rank line."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import cc, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worlds():
    return [w for w in (2, 4, 8) if w <= max(_n_gpus(), 2)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need(world):
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs, {_n_gpus()} visible")


def _worker(rank, world, port, M, results, backend="nccl"):
    """backend "nccl": one rank per GPU over RCCL.  backend "gloo": the SAME body with every rank on GPU 0 and the collectives
    staged through host memory by flamo_amd.dist (RCCL refuses two ranks on one device) -- how a one-GPU box executes it."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    local = rank if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from flamo_amd import dist as fd
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ok = int(ones.item()) == world                               # RCCL carries every rank
        torch.manual_seed(0)                                         # the same "replicated" tensors on every rank
        full = torch.randn(3, M, 4, 2, dtype=torch.complex64, device=dev)
        bin0, m_local = fd.shard_bins(M, rank, world)
        w = torch.randn(3, M, 4, 2, dtype=torch.complex64, device=dev)
        for algo in ("rccl", "direct"):
            prev = fd.set_all_gather_algorithm(algo)
            try:
                local = fd.take_local_bins(full).clone().requires_grad_(True)
                gathered = fd.all_gather_bins(local, M)
                ok = ok and gathered.is_cuda and torch.equal(gathered, full)          # data movement only: bit exact
                (g,) = torch.autograd.grad(torch.sum(torch.real(gathered * torch.conj(w))), [local])
                ok = ok and torch.allclose(g, w[:, bin0:bin0 + m_local])
                rl = fd.all_gather_bins(full.real[:, bin0:bin0 + m_local].contiguous(), M)
                ok = ok and torch.equal(rl, full.real)
            finally:
                fd.set_all_gather_algorithm(prev)
        # batch-sharded <-> bin-sharded exchange (all-to-all both ways) with autograd
        Bl = 2
        torch.manual_seed(1)
        glob = torch.randn(world * Bl, M, 3, dtype=torch.complex64, device=dev)
        mine = glob[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
        xb = fd.batch_to_bins(mine)
        ok = ok and xb.shape == (world * Bl, m_local, 3) and torch.equal(xb, glob[:, bin0:bin0 + m_local])
        back = fd.bins_to_batch(xb * 2.0, M)
        ok = ok and torch.equal(back, 2.0 * mine)
        wgt = torch.randn(Bl, M, 3, dtype=torch.complex64, device=dev)
        (gm,) = torch.autograd.grad(torch.sum(torch.real(back * torch.conj(wgt))), [mine])
        ok = ok and torch.allclose(gm, 2.0 * wgt)
        # replicated-parameter gradients: synchronous, then asynchronous through the cached flat buffer (reused)
        p = torch.nn.Parameter(torch.zeros(5, device=dev))
        q = torch.nn.Parameter(torch.zeros(2, 2, device=dev, dtype=torch.float64))
        p.grad = torch.full((5,), float(rank + 1), device=dev)
        q.grad = torch.ones(2, 2, device=dev, dtype=torch.float64) * (10.0 ** rank)
        fd.all_reduce_grads([p, q])
        ok = ok and torch.allclose(p.grad, torch.full((5,), float(sum(range(1, world + 1))), device=dev)) and \
            torch.allclose(q.grad, torch.full((2, 2), float(sum(10 ** r for r in range(world))), device=dev, dtype=torch.float64))
        for rep in range(3):
            p.grad = torch.full((5,), float(rank + 1 + rep), device=dev)
            fin = fd.all_reduce_grads([p], async_op=True)
            fin()
            ok = ok and torch.allclose(p.grad, torch.full((5,), float(sum(r + 1 + rep for r in range(world))), device=dev))
        torch.cuda.synchronize()
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("M", [49, 4801, 3])
def test_collectives_over_rccl(world, M):
    _need(world)
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), M, results), nprocs=world, join=True)
    assert dict(results) == {r: True for r in range(world)}


@pytest.mark.parametrize("M", [49, 4801, 3])
def test_collectives_two_ranks_on_one_device(M):
    """The body of test_collectives_over_rccl, two ranks sharing GPU 0 over gloo (host-staged): runs on a one-GPU box, so the
    worker's code has been executed on device tensors before a multi-GPU node meets it."""
    _need(1)
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), M, results, "gloo"), nprocs=2, join=True)
    assert dict(results) == {0: True, 1: True}


def _torchrun(world, script, *args, timeout=900, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *args]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run(cmd, check=True, timeout=timeout, cwd=ROOT, capture_output=True, text=True, env=e)


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("algo", ["rccl", "direct"])
def test_colorless_training_bin_sharded_over_rccl(world, algo, tmp_path):
    """examples/e8_colorless_fdn.py:120-138 with the bins sharded over `world` GPUs: losses and final parameters of the
    unsharded trajectory (float64: the collectives only move data and sum a few hundred gradient values)."""
    _need(world)
    tool = os.path.join(ROOT, "tools", "train_colorless_fdn.py")
    common = ["--N", "6", "--nfft", "4800", "--batch", "2", "--steps", "4", "--warmup", "0", "--lr", "1e-2", "--dtype", "float64"]
    one, many = str(tmp_path / "one.pt"), str(tmp_path / "many.pt")
    subprocess.run([sys.executable, tool, *common, "--dump", one], check=True, timeout=600, cwd=ROOT)
    _torchrun(world, tool, *common, "--gpus", str(world), "--backend", "nccl", "--dump", many, env={"FLAMO_ALLGATHER": algo})
    r1, r2 = torch.load(one), torch.load(many)
    cc("losses", torch.tensor(r2["losses"], dtype=torch.float64), torch.tensor(r1["losses"], dtype=torch.float64), 1e-10, max_tol=float("inf"))
    for k, v in r1["state"].items():
        cc("r2_state_k", r2["state"][k], v, 1e-9, max_tol=float("inf"))


@pytest.mark.parametrize("world", [2, 8])
def test_config5_chain_bin_sharded_over_rccl(world, tmp_path):
    """BASELINE configs[4]'s structure with the bins sharded over `world` GPUs: output of the inverse transform behind the
    all-gather and every parameter gradient equal the unsharded run."""
    _need(world)
    tool = os.path.join(ROOT, "tools", "run_sharded_chain.py")
    common = ["--N", "32", "--nfft", "3840", "--steps", "1", "--warmup", "0", "--dtype", "float64"]
    one, many = str(tmp_path / "one.pt"), str(tmp_path / "many.pt")
    subprocess.run([sys.executable, tool, *common, "--dump", one], check=True, timeout=600, cwd=ROOT)
    _torchrun(world, tool, *common, "--gpus", str(world), "--backend", "nccl", "--dump", many)
    r1, r2 = torch.load(one), torch.load(many)
    cc("r2_y", r2["y"], r1["y"], 1e-11, max_tol=float("inf"))
    for g2, g1 in zip(r2["grads"], r1["grads"]):
        cc("g2", g2, g1, 1e-9, max_tol=float("inf"))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_line_over_rccl(world):
    """bench.py launched as the driver launches it: one JSON line from rank 0, whole-job value, the ranks RCCL carried,
    the weak-scaling (batch) figure as `value` and the strong-scaling bin-sharded figures beside it."""
    _need(world)
    out = _torchrun(world, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                    "--no-cpu-baseline", timeout=1200).stdout
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_bench_line(d, world)
    assert d["collective_backend"].startswith("nccl")


def _check_bench_line(d, world):
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["rccl_ranks_seen"] == world
    assert d["config"]["batch_per_gpu"] == 32 and d["value"] > 0
    assert "bin_sharded" in d and "error" not in d["bin_sharded"], d.get("bin_sharded")


def test_bench_line_two_ranks_on_one_device():
    """test_bench_line_over_rccl's launch and checks with two ranks on GPU 0 over gloo (BENCH_ALLOW_SHARED_GPU / BENCH_BACKEND,
    bench.py's test-rig hooks): the multi-rank branch of bench.py -- gradient all-reduce after every replay, max over ranks,
    the bin-sharded legs -- executed end to end on a one-GPU box.  The figures of such a line mean nothing (shared device)."""
    _need(1)
    out = _torchrun(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                    timeout=900, env={"BENCH_ALLOW_SHARED_GPU": "1", "BENCH_BACKEND": "gloo"}).stdout
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_bench_line(d, 2)
    assert "gloo" in d["collective_backend"]
