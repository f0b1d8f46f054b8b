"""Do two independent branches of a captured HIP graph run concurrently on replay?  The config-2 pair that could overlap: the
cascade-times-matrix response (ALU-bound, ~29 us) beside the input's column pass (HBM-bound, ~41 us), captured on one stream
and on two.
    python tools/dbg/graph_branch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
nfft, N, B, S = 96000, 8, 32, 12
torch.manual_seed(0)
x = torch.randn(B, nfft, N, device=dev)
b = torch.randn(3, S, N, N, device=dev, dtype=torch.float64) * 0.1
a = torch.randn(3, S, N, N, device=dev, dtype=torch.float64) * 0.1
b[0] += 1
a[0] += 1
Wr = torch.randn(N, N, device=dev)


def resp():
    with ops.row_major_bins(nfft):
        return ops.sos_response_rc(b, a, Wr, 1.0, nfft)


def cols():
    return ops._spec_cols_fwd(x, nfft, 0.0)


def timed(g, n=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


resp(); cols()
torch.cuda.synchronize()
res = {}
for label in ("response only", "column pass only", "both, one stream", "both, two streams"):
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if label == "response only":
            keep = resp()
        elif label == "column pass only":
            keep = cols()
        elif label == "both, one stream":
            keep = (resp(), cols())
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                k1 = resp()
            k2 = cols()
            torch.cuda.current_stream().wait_stream(side)
            keep = (k1, k2)
    res[label] = timed(g)
    print(f"{label:20s}: {res[label]:7.1f} us per replay")
print("switches:", {k: os.environ.get(k) for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_FORCE_GRAPH_QUEUES")})
