"""What a collective between two graph replays costs (one rank over RCCL): ms per step of bench.py's config-2 step with
nothing / a tiny eager kernel / a synchronous all-reduce / an asynchronous one between the replays."""
import os, sys, time
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flamo_amd  # noqa: F401
import torch
import torch.distributed as dist
import bench
from flamo_amd import ops
from flamo_amd.graph import GraphedStep
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(130709)
model, params = bench.build_model(dev, torch.float32)
x = torch.randn(bench.BATCH, bench.NFFT, bench.NCH, device=dev)
gs = GraphedStep(lambda xx: ops.mean_square(model(xx)), (x,), params, warmup=2)
flat = torch.zeros(1024, device=dev)


def run(name, between, n=300):
    for _ in range(100):
        gs.replay(); between()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gs.replay(); between()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step", flush=True)


pend = []
def asy():
    if pend:
        pend.pop().wait()
    pend.append(dist.all_reduce(flat, async_op=True))
run("replays only", lambda: None)
run("tiny eager kernel between", lambda: flat.add_(0.0))
run("sync all_reduce between", lambda: dist.all_reduce(flat))
run("async all_reduce + wait between", asy)
run("replays only (again)", lambda: None)
dist.destroy_process_group()
