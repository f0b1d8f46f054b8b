"""Time the graphic equaliser's cascade backward (config 2: 8 x 8, nfft 96000, row-major bins) -- lanes kernel against the
first generation, sweeping the grid sizing of the lanes kernel.  python tools/dbg/cascade2_bench.py [N nfft]"""
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flamo_amd import _lib, ops
from flamo_amd.processor import dsp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nfft = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
dev = torch.device("cuda:0")
torch.manual_seed(1)
L = _lib.lib()
geq = dsp.GEQ(size=(N, N), nfft=nfft, alias_decay_db=0.0, device=dev, dtype=torch.float32)
W = torch.randn(N, N, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def setup():
    x = geq.param.detach().clone().requires_grad_(True)
    Wl = W.clone().requires_grad_(True)
    spec = geq._cascade_spec(x)
    with ops.row_major_bins(nfft):
        H = ops.geq_cascade_rc(spec[1], spec[2], Wl, geq._gamma_f, nfft)
    ct = torch.randn(H.shape, device=dev) + 1j * torch.randn(H.shape, device=dev)
    loss = (H * ct.conj()).real.sum()
    return x, Wl, H, ct, loss


x, Wl, H, ct, loss = setup()
g = ct.to(torch.complex64)


def bwd():
    with ops.row_major_bins(nfft):
        torch.autograd.grad(H, [x, Wl], g, retain_graph=True)


def fwd():
    spec = geq._cascade_spec(geq.param)
    with ops.row_major_bins(nfft), torch.no_grad():
        ops.geq_cascade_rc(spec[1], spec[2], W, geq._gamma_f, nfft)


L.fl_debug_set_cascade_lanes(0, -1, -1)
print(f"first generation: backward (cascade + design) {timed(bwd):.1f} us   forward {timed(fwd):.1f} us")
ref = torch.autograd.grad(H, [x, Wl], g, retain_graph=True)
M = nfft // 2 + 1
for bpc in (1, 2, 3):
    for tb in (0, 20, 24, 30, 40, 60):
        L.fl_debug_set_cascade_lanes(1, bpc, tb)
        with ops.row_major_bins(nfft):
            nbx = L.fl_geq_bwd_lanes_blocks(M, N * N, 12, nfft, -ops.bin_order(nfft)[1], N, N, 1)
        if nbx == 0:
            continue
        t = timed(bwd)
        got = torch.autograd.grad(H, [x, Wl], g, retain_graph=True)
        e = [((a - b).norm() / b.norm()).item() for a, b in zip(got, ref)]
        print(f"lanes bpc {bpc} tile {tb:2d} blocks {nbx:4d}: backward {t:.1f} us   vs gen1: gain {e[0]:.1e} W {e[1]:.1e}")
L.fl_debug_set_cascade_lanes(1, 2, 0)
