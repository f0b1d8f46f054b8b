"""Minimal forms of the graph-replay hazard of DESIGN 4.5 (run WITHOUT DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, i.e. with ROCm's
pre-built graph packets): a captured [producer kernel -> torch reduction] pair, a tiny eager launch between two replays,
does the reduction's value change?  Producers: a torch kernel, trivial non-library kernels (plain arguments / a 240-byte
by-value argument / 100 KB of dynamic LDS: tools/dbg/tiny_kernels.hip), and kernels of libflamo_hip.
    env -u DEBUG_CLR_GRAPH_PACKET_CAPTURE python tools/dbg/replay_min.py"""
import ctypes
import os
import sys

os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None) if os.environ.get("REPLAY_MIN_KEEP_ENV") != "1" else None
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
dev = torch.device("cuda:0")
tiny = ctypes.CDLL(os.path.join(HERE, "bin", "libtiny.so"))
for f in ("tiny_plain", "tiny_big", "tiny_lds"):
    getattr(tiny, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
n = 192000
torch.manual_seed(0)
x = torch.randn(n, device=dev)


def stream():
    return torch.cuda.current_stream().cuda_stream


def producers():
    yield "torch mul", lambda out: torch.mul(x, 1.5, out=out)
    yield "tiny plain", lambda out: tiny.tiny_plain(x.data_ptr(), out.data_ptr(), n, stream())
    yield "tiny 240-byte by-value arg", lambda out: tiny.tiny_big(x.data_ptr(), out.data_ptr(), n, stream())
    yield "tiny 100 KB dynamic LDS", lambda out: tiny.tiny_lds(x.data_ptr(), out.data_ptr(), n, stream())
    def rep(f, k):
        def run(out):
            for _ in range(k):
                f(out)
        return run
    yield "tiny plain x12", rep(lambda out: tiny.tiny_plain(x.data_ptr(), out.data_ptr(), n, stream()), 12)
    yield "tiny 240-byte arg x12", rep(lambda out: tiny.tiny_big(x.data_ptr(), out.data_ptr(), n, stream()), 12)
    yield "tiny 100 KB LDS x12", rep(lambda out: tiny.tiny_lds(x.data_ptr(), out.data_ptr(), n, stream()), 12)
    yield "torch mul x12 + complex temporaries", lambda out: out.copy_(torch.view_as_real(torch.fft.rfft(x * 1.5)).reshape(-1)[:n])
    yield "torch.empty temporaries in the capture", lambda out: out.copy_((torch.empty_like(x).copy_(x) * 1.5 + torch.empty_like(x).fill_(0.0)))
    if os.environ.get("REPLAY_MIN_LIB", "1") == "1":
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        # importing flamo_amd sets the switch for *later* processes only; the runtime of this one is already up
        import flamo_amd  # noqa: F401
        from flamo_amd import _lib, ops
        L = _lib.lib()
        xs = torch.randn(4, 24000, 2, device=dev)
        yield "fl_transpose", lambda out: L.fl_transpose(x.data_ptr(), out.data_ptr(), 1, 375, 512, 375, 4, stream())
        yield "fl_mean_square + copy", lambda out: (out.copy_(x), ops.mean_square(out))[0]
        yield "fused Shell pipeline (3 fl_spec_* launches)", lambda out: out.copy_(ops.spectral_apply(
            xs, ops.permute_bins(torch.ones(24001, 2, 2, device=dev, dtype=torch.complex64), 48000), 48000).reshape(-1)[:n])
        yield "fl_rfft layered", lambda out: out.copy_(torch.view_as_real(ops.rfft(x.view(1, n, 1), n)).reshape(-1)[:n])
        yield "fl_transpose x12", rep(lambda out: L.fl_transpose(x.data_ptr(), out.data_ptr(), 1, 375, 512, 375, 4, stream()), 12)
        S = torch.empty(4 * 24000 * 2, dtype=torch.complex64, device=dev)
        W48 = ops.twiddles(48000, torch.float32, dev)
        yield "fl_spec_cols_fwd alone (preallocated)", lambda out: (L.fl_spec_cols_fwd_f32(xs.data_ptr(), 4, 24000 * 2, 2, S.data_ptr(), W48.data_ptr(), 48000, 0.0, stream()), out.copy_(torch.view_as_real(S).reshape(-1)[:n]))[1]
        sc = torch.empty(n, dtype=torch.complex64, device=dev)
        Xo = torch.empty(n // 2 + 32, dtype=torch.complex64, device=dev)
        Wn = ops.twiddles(n, torch.float32, dev)
        def raw_rfft_alloc(pad):
            def run(out):
                n_scr = L.fl_fft_scratch_elems(n, 0, 1)
                scr = torch.empty(max(n_scr, 1) + pad, dtype=torch.complex64, device=dev)
                P = ops._pitch(n // 2 + 1)
                Xn = torch.empty(P + pad, dtype=torch.complex64, device=dev)
                L.fl_rfft_f32(x.data_ptr(), n, n, Xn.data_ptr(), P, scr.data_ptr(), Wn.data_ptr(), 1, n, 1.0, 0.0, 0, stream())
                out.copy_(torch.view_as_real(Xn).reshape(-1)[:n])
            return run
        yield "fl_rfft_f32, buffers allocated in the capture", raw_rfft_alloc(0)
        yield "fl_rfft_f32, in-capture buffers padded by 64K elements", raw_rfft_alloc(65536)
        yield "torch only, MB-sized temporaries in the capture", lambda out: out.copy_((torch.empty(n * 4, device=dev).fill_(1.0)[:n] * x + torch.empty(n * 2, dtype=torch.complex64, device=dev).fill_(0)[:n].real))
        yield "fl_rfft_f32 alone (preallocated)", lambda out: (L.fl_rfft_f32(x.data_ptr(), n, n, Xo.data_ptr(), n // 2 + 32, sc.data_ptr(), Wn.data_ptr(), 1, n, 1.0, 0.0, 0, stream()), out.copy_(torch.view_as_real(Xo).reshape(-1)[:n]))[1]


for name, prod in producers():
    out = torch.zeros(n, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            prod(out)
            r = out.sum()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        prod(out)
        r = out.sum()
        r2 = out.abs().max()
    g.replay()
    torch.cuda.synchronize()
    first = (r.item(), r2.item())
    bad = 0
    for i in range(6):
        junk = torch.full((1,), float(i), device=dev)       # the tiny eager launch
        g.replay()
        torch.cuda.synchronize()
        if (r.item(), r2.item()) != first:
            bad += 1
    print(f"{name:46s} first sum {first[0]: .6e} max {first[1]:.6e}   replays that differ after an eager launch: {bad}/6")
