"""Autograd operators of the hot path, each a thin host wrapper over one C-ABI entry point.

Layout.  Frequency-domain tensors keep flamo's logical shape ``(B, M, N, ...)`` but are stored
*bin-planar*: memory order ``(B, N, ..., M)`` with the bin axis contiguous, exposed through a
``movedim`` view -- the same memory order ``torch.fft.rfft(dim=1)`` itself returns for a
contiguous ``(B, T, N)`` input, so user code written against the reference sees identical
shapes.  Time-domain tensors produced here are signal-planar ``(B, N, ..., T)`` likewise.
Tensors that arrive in another layout are converted once with the LDS transpose kernel.

PyTorch is used for device memory, streams and autograd bookkeeping only; every arithmetic
pass over (B, M, ...) data is one of the HIP kernels.
"""
from __future__ import annotations

import math
import threading
from typing import Optional, Tuple

import torch

from . import _lib

__all__ = ["rfft", "irfft", "spectral_apply", "permute_bins", "mimo", "solve", "delay_response", "sos_response", "geq_sections", "solve_dud", "to_planar", "set_bin_shard",
           "bin_shard"]


# ----------------------------------------------------------------------------- plumbing
def _require_gpu(*ts: torch.Tensor) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "flamo_amd: tensors must live on a ROCm device (got %s); the HIP path has no CPU fallback" % t.device
            )
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise RuntimeError("flamo_amd: tensors on different devices")
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        # the launches go to the CURRENT device's stream: foreign pointers there are a fault or a silent mis-ordering
        raise RuntimeError(f"flamo_amd: tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in `with torch.cuda.device(...)`")
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """The current stream's handle for the C ABI.  torch.cuda.current_stream() builds a Stream object through three Python
    layers (8 us a call, a dozen calls per eager step: a sixth of a batch-1 FDN step's host time); the raw accessor is what
    it wraps."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _rdtype(t: torch.Tensor) -> torch.dtype:
    if t.dtype in (torch.complex64, torch.float32):
        return torch.float32
    if t.dtype in (torch.complex128, torch.float64):
        return torch.float64
    raise TypeError(f"flamo_amd: unsupported dtype {t.dtype} (float32/float64 and their complex types only)")


def _cdtype(real: torch.dtype) -> torch.dtype:
    return torch.complex64 if real == torch.float32 else torch.complex128


def _sfx(real: torch.dtype, cplx: bool) -> str:
    if cplx:
        return "c64" if real == torch.float32 else "c128"
    return "f32" if real == torch.float32 else "f64"


def _prod(xs) -> int:
    p = 1
    for v in xs:
        p *= int(v)
    return p


_twiddles = {}
_twiddles_lock = threading.Lock()


def twiddles(nfft: int, real: torch.dtype, device: torch.device) -> torch.Tensor:
    """Master table W[j] = exp(-2 pi i j / nfft), cached per (nfft, precision, device).  The table is filled on
    whichever stream asks first; a consumer on ANOTHER stream waits for the fill's event (first forward pass: the
    input transform fills the table on the main stream while a response generator reads it on the side stream)."""
    key = (int(nfft), real, device.index if device.index is not None else torch.cuda.current_device())
    ent = _twiddles.get(key)
    if ent is None:
        with _twiddles_lock:
            ent = _twiddles.get(key)
            if ent is None:
                L = _lib.lib()
                # tables of the fused pipeline's lengths carry its contiguous copies behind the nfft entries
                aux = int(L.fl_spec_aux_elems(int(nfft)))
                W = torch.empty(nfft + aux, dtype=_cdtype(real), device=device)
                fn = L.fl_twiddle_fill_f32 if real == torch.float32 else L.fl_twiddle_fill_f64
                cur = torch.cuda.current_stream(device)
                _lib.check(fn(W.data_ptr(), nfft, cur.cuda_stream), "twiddle_fill")
                if aux:
                    fill = L.fl_spec_aux_fill_f32 if real == torch.float32 else L.fl_spec_aux_fill_f64
                    _lib.check(fill(W.data_ptr(), nfft, cur.cuda_stream), "spec_aux_fill")
                ev = torch.cuda.Event()
                ev.record(cur)
                if not torch.cuda.is_current_stream_capturing():
                    # The host waits for the fill here, once per (length, precision, device): from now on the table is
                    # plain read-only memory and NO consumer on any stream ever waits for an event of it.  It used to keep
                    # the event, and a consumer on a capturing stream waited for it: that records, inside the graph, a wait
                    # on an event that lives OUTSIDE it -- and with ROCm 7.2's pre-built graph packets the torch nodes that
                    # follow in the graph (the memset + reduction pair of a captured sum() / max()) then return wrong
                    # values after any eager launch between two replays.  tools/dbg/archive/replay_min.py: the hazard of DESIGN
                    # 4.5 reproduced with exactly the producers that went through that wait, and with no others.
                    ev.synchronize()
                    ev = None
                ent = _twiddles[key] = [W, ev, cur.cuda_stream]
    W, ev, filled_on = ent
    if ev is not None:
        # a table first built INSIDE a capture: its fill and this event are nodes of that graph; other streams of the same
        # capture order themselves behind it
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream != filled_on:
            if torch.cuda.is_current_stream_capturing():
                cur.wait_event(ev)
            elif ev.query():
                ent[1] = None
            else:
                cur.wait_event(ev)
    return W


_ALIGN = 32  # bin rows are padded to a multiple of 32 elements (256 B in c64): aligned planes


def _pitch(n: int) -> int:
    return (int(n) + _ALIGN - 1) // _ALIGN * _ALIGN


def _transpose(src: torch.Tensor, nbatch: int, rows: int, cols: int, dst_pitch: Optional[int] = None) -> torch.Tensor:
    """src: contiguous memory (nbatch, rows, cols) -> new memory (nbatch, cols, dst_pitch) holding
    the transposed blocks in [..., :rows] (dst_pitch defaults to rows: a plain contiguous result)."""
    pitch = rows if dst_pitch is None else int(dst_pitch)
    dst = torch.empty(nbatch * cols * pitch, dtype=src.dtype, device=src.device)
    if nbatch and rows and cols:
        _lib.check(_lib.lib().fl_transpose(src.data_ptr(), dst.data_ptr(), nbatch, rows, cols, pitch,
                                           src.element_size(), _stream()), "transpose")
    return dst


def _lead_pitch(mem: torch.Tensor) -> Optional[int]:
    """For memory-order tensor ``mem`` (lead..., A): the pitch P such that row i of the flattened
    leading dims starts at element i*P with its A entries contiguous -- or None if the tensor is
    not laid out that way.  P >= A; P > A means padded rows."""
    A = mem.shape[-1]
    if A > 1 and mem.stride(-1) != 1:
        return None
    P, expect = None, None
    for size, stride in reversed(list(zip(mem.shape[:-1], mem.stride()[:-1]))):
        if size == 1:
            continue
        if P is None:
            if stride < A:
                return None
            P, expect = stride, stride * size
        else:
            if stride != expect:
                return None
            expect *= size
    return A if P is None else P


def _is_planar(x: torch.Tensor) -> bool:
    """memory order (B, rest..., axis1) with axis 1 contiguous (rows possibly padded)"""
    return _lead_pitch(x.movedim(1, -1)) is not None


def to_planar(x: torch.Tensor) -> torch.Tensor:
    """Same logical tensor (B, A, rest...) stored with axis 1 contiguous (memory (B, rest..., pitch))."""
    if x.dim() < 2:
        raise ValueError("expected at least 2 dims")
    if _is_planar(x):
        return x
    B, A = x.shape[0], x.shape[1]
    rest = tuple(x.shape[2:])
    xc = x.contiguous()  # no-op for the usual channel-innermost layout
    P = _pitch(A)
    out = _transpose(xc, B, A, _prod(rest), P)
    return out.view(B, *rest, P)[..., :A].movedim(-1, 1)


def _empty_planar(shape: Tuple[int, ...], dtype, device) -> torch.Tensor:
    B, A = shape[0], shape[1]
    rest = tuple(shape[2:])
    return torch.empty((B, *rest, _pitch(A)), dtype=dtype, device=device)[..., :A].movedim(-1, 1)


def _empty_rows(lead: Tuple[int, ...], A: int, dtype, device) -> torch.Tensor:
    """memory (lead..., pitch) viewed as (lead..., A)"""
    return torch.empty((*lead, _pitch(A)), dtype=dtype, device=device)[..., :A]


def _bnk(x: torch.Tensor):
    """(B, M, N, K) sizes and element strides (s_b, s_n, s_k) of a planar tensor (B, M, N, rest...)."""
    B, M, N = x.shape[0], x.shape[1], x.shape[2]
    K = _prod(x.shape[3:])
    P = _lead_pitch(x.movedim(1, -1))  # rows of (B, N, rest...) are P apart
    assert P is not None
    return B, M, N, K, N * K * P, K * P, P


# ----------------------------------------------------------------------------- kernel timing (bench.py)
class KernelTimer:
    """HIP-event timing of individual kernel launches on the stream they are launched on
    (bench.py's roofline leg).  Disabled by default: zero overhead on the hot path."""

    def __init__(self):
        self.enabled = False
        self.records = {}
        self.prefill_cycles = 0

    def begin_step(self, prefill_ms: float = 2.0):
        """In-step mode: queue ~prefill_ms of streaming copies ONCE, in front of a whole eager step whose launches are
        then timed with prefill_cycles = 0.  The host runs ahead of the GPU for the whole step, so its kernels execute
        back to back on the stream -- each behind its real predecessor, with the caches in the state that predecessor
        left -- as they do in a replayed graph (where events cannot be recorded on ROCm); the event pairs add only
        their own record packets between two kernels."""
        if not self.enabled:
            return
        keep, self.prefill_cycles = self.prefill_cycles, int(prefill_ms * 1e3 / 35.0) * 100_000
        try:
            self._prefill()
        finally:
            self.prefill_cycles = keep

    def reset(self, enabled: bool, prefill_cycles: int = 0):
        """prefill_cycles > 0: about prefill_cycles / 100k streaming copies of 96 MB (~35 us each) are
        enqueued in front of every timed launch, so that the start event, the launch and the stop event
        are all queued before the GPU reaches them -- in eager steps the host is slower than the GPU and
        an event recorded on an idle stream is stamped at once, which would add the host's time between
        the two calls (and the dispatch latency of a cold queue) to the kernel's."""
        self.enabled = enabled
        self.records = {}
        self.prefill_cycles = int(prefill_cycles)

    def _prefill(self):
        """Queue ~0.2 ms of work in front of a timed launch.  Streaming copies rather than a spin loop:
        behind an idle or compute-only stretch the memory clocks have ramped down and a bandwidth-bound
        kernel measures 10-15 % slower than the same kernel inside a replayed step."""
        buf = self.__dict__.get("_prefill_buf")
        if buf is None or buf[0].device != torch.device("cuda", torch.cuda.current_device()):
            n = 24 * 1024 * 1024          # 96 MB of float32 each
            buf = (torch.empty(n, device="cuda"), torch.empty(n, device="cuda"))
            buf[0].zero_()
            self.__dict__["_prefill_buf"] = buf
        for _ in range(max(1, self.prefill_cycles // 100_000)):
            buf[1].copy_(buf[0])

    def span(self, name):
        timer = self

        class _Span:
            def __enter__(self_):
                if timer.enabled:
                    self_.a = torch.cuda.Event(enable_timing=True)
                    self_.b = torch.cuda.Event(enable_timing=True)
                    if timer.prefill_cycles:
                        timer._prefill()
                    self_.a.record(torch.cuda.current_stream())
                return self_

            def __exit__(self_, *exc):
                if timer.enabled:
                    self_.b.record(torch.cuda.current_stream())
                    timer.records.setdefault(name, []).append((self_.a, self_.b))
                return False

        return _Span()

    def drop_last(self, name):
        """forget the newest span of this name (its launch was recorded for a launch pair, not issued: see paired_launch)"""
        if self.enabled and self.records.get(name):
            self.records[name].pop()
            if not self.records[name]:
                del self.records[name]

    def summary(self):
        """name -> (launches, mean milliseconds); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


kernel_timer = KernelTimer()


# ----------------------------------------------------------------------------- response / transform overlap
# Shell.forward marks the point on the current stream BEFORE the input transform is enqueued; the
# parameter-only work of the core (response generators, their composition) is then enqueued on a
# side stream that waits for that point only, so it runs concurrently with the FFT kernels of the
# input instead of after them.  Under hipGraph capture the fork/join become graph edges.
# Memory safety rests on stream order, not on record_stream: every side region starts by waiting
# for an event recorded on the main stream after all earlier main-stream work was enqueued, and the
# main stream waits for the side stream before it touches a result.
class _PerThread(threading.local):
    """State of the forward pass in flight on THIS thread (two models driven from two threads, or autograd's worker
    threads, never see each other's fork point, memo, bin range or side streams)."""

    def __init__(self):
        self.fork = {"event": None, "memo": None}
        self.shard = {"bin0": 0, "m_local": None, "order": None}
        self.side_streams = {}
        self.loop_depth = 0
        self.step_memo = None
        self.capture_constants = None


_tls = _PerThread()


class loop_scope:
    """Marks the forward pass of a Recursion's paths: modules inside keep their responses as tensors (the loop matrix needs
    them), so the apply-without-the-gradient-tensor route of the cascade filters stays off there."""

    def __enter__(self):
        _tls.loop_depth += 1

    def __exit__(self, *exc):
        _tls.loop_depth -= 1
        return False


def in_loop() -> bool:
    return _tls.loop_depth > 0


class step_scope:
    """One training / evaluation step (forward, criteria, backward): inside it a parameter map that is a launch of its own
    -- the orthogonal map exp(skew(x)) -- is evaluated once per parameter version, however many callers ask for it (the
    model and a sparsity criterion on the same mixing matrix, as in the reference's colorless-FDN training, each evaluate
    it).  The memo dies with the scope, so nothing outlives a parameter update or a graph capture
    (flamo_amd.graph.GraphedStep opens one scope per captured call)."""

    def __enter__(self):
        self._prev = _tls.step_memo
        _tls.step_memo = {}
        return self

    def __exit__(self, *exc):
        _tls.step_memo = self._prev
        return False


def step_memo():
    return _tls.step_memo


class capture_scope:
    """Opened by flamo_amd.graph.GraphedStep around its capture: constants of the parameter VALUES that the eager warm-up runs
    left in a cache (the integer-delay response: `round` has a zero gradient) may be taken from it instead of being recorded as
    launches of every replay.  Each such constant is registered here as (weak reference to the parameter, its version counter,
    the tensor): the step keeps the tensor alive for the graph's lifetime and refuses to replay once the parameter has been
    assigned a new value."""

    def __init__(self, sink: list):
        self.sink = sink

    def __enter__(self):
        self._prev = _tls.capture_constants
        _tls.capture_constants = self.sink
        return self

    def __exit__(self, *exc):
        _tls.capture_constants = self._prev
        return False


def capture_constants():
    """the list a capturing GraphedStep collects its constants in, or None"""
    return _tls.capture_constants


def fork_event():
    return _tls.fork["event"]


def forward_memo():
    """dict scoped to the current Shell.forward call (None outside one): per-forward memo of responses"""
    return _tls.fork["memo"]


def side_stream(dev: torch.device) -> "torch.cuda.Stream":
    """The side stream of (this thread, device)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _tls.side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _tls.side_streams[key] = s
    return s


class fork_point:
    """with ops.fork_point(x): ...  -- records the fork event for the enclosed forward pass."""

    def __init__(self, x):
        self.on = torch.is_tensor(x) and x.is_cuda
        self.dev = x.device if self.on else None

    def __enter__(self):
        f = _tls.fork
        self.prev = (f["event"], f["memo"])
        if self.on:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            f["event"] = ev
            f["memo"] = {}
        return self

    def __exit__(self, *exc):
        _tls.fork["event"], _tls.fork["memo"] = self.prev
        return False


# ----------------------------------------------------------------------------- bin sharding (multi-GPU)


def set_bin_shard(bin0: int = 0, m_local: Optional[int] = None) -> None:
    """Restrict the response generators to bins [bin0, bin0+m_local) (flamo_amd.dist sets this
    per rank; default: all nfft//2+1 bins)."""
    _tls.shard["bin0"] = int(bin0)
    _tls.shard["m_local"] = None if m_local is None else int(m_local)


def bin_shard(nfft: int) -> Tuple[int, int]:
    M = nfft // 2 + 1
    if _tls.shard["m_local"] is None:
        return 0, M
    return _tls.shard["bin0"], _tls.shard["m_local"]


class row_major_bins:
    """with ops.row_major_bins(nfft): per-bin responses are generated in the ROW-MAJOR bin order of the fused
    Shell pipeline (element k1*L2 + k2 = bin k1 + L1*k2): natively by the cascade / integer-delay kernels,
    through ``permute_bins`` by everything else (``DSP._response_once``)."""

    def __init__(self, nfft: int):
        import ctypes
        L1, L2 = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().fl_spec_plan(int(nfft), ctypes.byref(L1), ctypes.byref(L2)), "spec_plan")
        self.order = (int(nfft), L1.value, L2.value)

    def __enter__(self):
        self.prev = _tls.shard.get("order")
        _tls.shard["order"] = self.order
        return self

    def __exit__(self, *exc):
        _tls.shard["order"] = self.prev
        return False


def bin_order(nfft: int):
    """(L1, L2) when responses for this nfft are to be generated in row-major bin order, else None"""
    o = _tls.shard.get("order")
    return (o[1], o[2]) if (o is not None and o[0] == int(nfft)) else None


def _bin0_arg(nfft: int) -> Tuple[int, int]:
    """(bin0, m_local) as the response kernels take them: bin0 = -L2 selects row-major order"""
    o = bin_order(nfft)
    if o is not None:
        if _tls.shard["m_local"] is not None:
            raise RuntimeError("row-major bin order and bin sharding are mutually exclusive")
        return -o[1], nfft // 2 + 1
    return bin_shard(nfft)


# ----------------------------------------------------------------------------- transforms
_NORM_FWD = {"backward": lambda n: 1.0, "ortho": lambda n: 1.0 / math.sqrt(n), "forward": lambda n: 1.0 / n}
_NORM_INV = {"backward": lambda n: 1.0 / n, "ortho": lambda n: 1.0 / math.sqrt(n), "forward": lambda n: 1.0}


def _rfft_launch(xp: torch.Tensor, t_in: int, nfft: int, scale: float, env_log2: float, interior_x2: int):
    """xp: signal-planar real, logical (B, T, rest...), memory (B, rest..., pitch).  Returns planar X."""
    real = _rdtype(xp)
    dev = xp.device
    B = xp.shape[0]
    rest = tuple(xp.shape[2:])
    nsig = B * _prod(rest)
    M = nfft // 2 + 1
    x_pitch = _lead_pitch(xp.movedim(1, -1))
    X = _empty_rows((B, *rest), M, _cdtype(real), dev)
    L = _lib.lib()
    f64 = int(real == torch.float64)
    n_scr = L.fl_fft_scratch_elems(nfft, f64, nsig)
    scratch = torch.empty(max(n_scr, 1), dtype=_cdtype(real), device=dev)
    fn = L.fl_rfft_f64 if f64 else L.fl_rfft_f32
    _lib.check(fn(xp.data_ptr(), x_pitch, t_in, X.data_ptr(), _pitch(M),
                  scratch.data_ptr(), twiddles(nfft, real, dev).data_ptr(), nsig, nfft, scale, env_log2, interior_x2,
                  _stream()), "rfft")
    return X.movedim(-1, 1)


# Channel counts up to which the (B,T,N) -> planar conversion is fused into the FFT's first pass
# (fl_rfft_ci_*).  Measured on MI355X at config 2 the fused pass costs 127 us against 66 us (FFT pass)
# + 48 us (LDS transpose): each workgroup touches every cache line of its batch item but uses 1/N of
# it, so the separate transpose stays the default (0 = never fuse); the path is kept and tested.
CI_MAX_CHANNELS = 0


def _rfft_launch_ci(x: torch.Tensor, nfft: int, scale: float, env_log2: float, interior_x2: int):
    """x: contiguous channel-innermost real (B, T, rest...).  Returns planar X (B, M, rest...)."""
    real = _rdtype(x)
    dev = x.device
    B, T = x.shape[0], x.shape[1]
    rest = tuple(x.shape[2:])
    C_ = _prod(rest)
    nsig = B * C_
    M = nfft // 2 + 1
    X = _empty_rows((B, *rest), M, _cdtype(real), dev)
    L = _lib.lib()
    f64 = int(real == torch.float64)
    n_scr = L.fl_fft_scratch_elems(nfft, f64, nsig)
    scratch = torch.empty(max(n_scr, 1), dtype=_cdtype(real), device=dev)
    fn = L.fl_rfft_ci_f64 if f64 else L.fl_rfft_ci_f32
    _lib.check(fn(x.data_ptr(), C_, T, X.data_ptr(), _pitch(M), scratch.data_ptr(),
                  twiddles(nfft, real, dev).data_ptr(), nsig, nfft, scale, env_log2, interior_x2, _stream()), "rfft_ci")
    return X.movedim(-1, 1)


def _rfft_any(x: torch.Tensor, nfft: int, scale: float, env_log2: float, interior_x2: int):
    """rfft of a real (B, T, rest...) tensor in whatever layout it arrives."""
    if not _is_planar(x) and x.is_contiguous() and x.dim() >= 3 and 1 < _prod(x.shape[2:]) <= CI_MAX_CHANNELS \
            and x.shape[1] <= nfft:
        return _rfft_launch_ci(x, nfft, scale, env_log2, interior_x2)
    xp = to_planar(x)
    return _rfft_launch(xp, min(x.shape[1], nfft), nfft, scale, env_log2, interior_x2)


def _irfft_launch(Xp: torch.Tensor, nfft: int, t_out: int, t_alloc: int, scale: float, env_log2: float,
                  interior_half: int):
    """Xp: bin-planar complex (B, M, rest...).  Returns signal-planar real (B, t_alloc, rest...)."""
    real = _rdtype(Xp)
    dev = Xp.device
    B = Xp.shape[0]
    rest = tuple(Xp.shape[2:])
    nsig = B * _prod(rest)
    X_pitch = _lead_pitch(Xp.movedim(1, -1))
    alloc = torch.zeros if t_alloc > t_out else torch.empty
    y = alloc((B, *rest, t_alloc), dtype=real, device=dev)
    L = _lib.lib()
    f64 = int(real == torch.float64)
    n_scr = L.fl_fft_scratch_elems(nfft, f64, nsig)
    scratch = torch.empty(max(n_scr, 1), dtype=_cdtype(real), device=dev)
    fn = L.fl_irfft_f64 if f64 else L.fl_irfft_f32
    _lib.check(fn(Xp.data_ptr(), X_pitch, y.data_ptr(), t_alloc, t_out, scratch.data_ptr(),
                  twiddles(nfft, real, dev).data_ptr(), nsig, nfft, scale, env_log2, interior_half, _stream()), "irfft")
    return y.movedim(-1, 1)


class _Rfft(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nfft, scale, env_log2):
        _require_gpu(x)
        if x.is_complex():
            raise TypeError("rfft expects a real tensor")
        ctx.meta = (nfft, scale, env_log2, x.shape[1])
        return _rfft_any(x, nfft, scale, env_log2, 0)

    @staticmethod
    def backward(ctx, gX):
        nfft, scale, env_log2, T = ctx.meta
        # g_x[t] = scale e(t) Re sum_k g_X[k] exp(+j w_k t): an inverse real FFT with the
        # interior bins halved (undoing the C2R doubling), cropped / zero-extended to T
        g = _irfft_launch(to_planar(gX.resolve_conj()), nfft, min(T, nfft), T, scale, env_log2, 1)
        return g, None, None, None


class _Irfft(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, nfft, scale, env_log2):
        _require_gpu(X)
        if not X.is_complex():
            raise TypeError("irfft expects a complex tensor")
        if X.shape[1] != nfft // 2 + 1:
            raise ValueError(f"irfft: expected {nfft // 2 + 1} bins along dim 1, got {X.shape[1]}")
        ctx.meta = (nfft, scale, env_log2)
        return _irfft_launch(to_planar(X.resolve_conj()), nfft, nfft, nfft, scale, env_log2, 0)

    @staticmethod
    def backward(ctx, gy):
        nfft, scale, env_log2 = ctx.meta
        # g_X[k] = w_k scale sum_t g_y[t] e(t) exp(-j w_k t), w_k = 2 on interior bins
        g = _rfft_any(gy, nfft, scale, env_log2, 1)
        return g, None, None, None


def env_log2_of(alias_decay_db: float, nfft: int) -> float:
    """log2 of the per-sample growth of the envelope gamma^-t, gamma = 10^(-|dB|/(20 nfft))
    (dsp.py:153-160), derived in float64 on the host."""
    return abs(float(alias_decay_db)) / (20.0 * nfft) * math.log2(10.0)


# ----------------------------------------------------------------------------- any transform length (chirp-z route)
# The Stockham kernels take even lengths whose half is 13-smooth (fl_fft_plan).  The reference hands ANY nfft to
# torch.fft.rfft / irfft (dsp.py:84-89, 110-115): odd lengths, or lengths with a large prime factor (34 = 2 17,
# 2176 = 2^7 17, 95999).  Those go through Bluestein's identity  k t = (k^2 + t^2 - (k - t)^2) / 2,
#     X[k] = c[k] sum_t (x[t] c[t]) conj(c[k - t]),      c[t] = exp(-i pi t^2 / n),
# i.e. a circular convolution of length P >= 2n - 1 with the chirp, P a length the kernels DO take: two real transforms
# of length P (real and imaginary part of x c), one real-coefficient mix per bin with the chirp filter's spectrum -- the
# chirp is even in t, so its spectrum p + i q is even too and the product A B splits into the two Hermitian halves
#     Ce = R p - I q,   Co = R q + I p          (R, I = rfft of Re / Im of x c)
# whose inverse real transforms are the real and imaginary part of the convolution -- and two inverse real transforms,
# all on the library's own rfft / irfft kernels; what is left in torch is elementwise.  ~8x the work of a direct transform
# of that length: a route for lengths nobody tunes for, on the layered path only (the fused Shell pipeline plans 13-smooth
# lengths, fl_spec_plan).  Differentiable by construction (linear ops composed of differentiable pieces).
_fft_plan_ok_cache = {}
_bluestein_cache = {}


def fft_plan_ok(nfft: int, real: torch.dtype) -> bool:
    """Do the Stockham kernels take this length directly?"""
    key = (int(nfft), real == torch.float64)
    ok = _fft_plan_ok_cache.get(key)
    if ok is None:
        import ctypes
        l1, l2 = ctypes.c_int(0), ctypes.c_int(0)
        ok = _fft_plan_ok_cache[key] = nfft >= 2 and _lib.lib().fl_fft_plan(int(nfft), int(key[1]), ctypes.byref(l1), ctypes.byref(l2)) == 0
    return ok


def _bluestein_setup(n: int, real: torch.dtype, dev: torch.device):
    key = (n, real, dev.index if dev.index is not None else torch.cuda.current_device())
    ent = _bluestein_cache.get(key)
    if ent is None:
        P = 2 * n - 1 + ((2 * n - 1) & 1)              # even candidates upwards; the filter's spectrum is formed in double
        while not (fft_plan_ok(P, torch.float32) and fft_plan_ok(P, torch.float64)):
            P += 2
        # chirp phases from t^2 mod 2n in integers: exact arguments at any length
        t = torch.arange(n, dtype=torch.int64)
        ang = (t * t % (2 * n)).to(torch.float64) * (math.pi / n)
        c = torch.complex(torch.cos(ang), -torch.sin(ang))                     # exp(-i pi t^2 / n)
        b = torch.zeros(P, dtype=torch.complex128)                              # conj(c) on lags -(n-1) .. n-1, circular
        b[:n] = c.conj()
        if n > 1:
            b[P - n + 1:] = c[1:].flip(0).conj()
        bd = b.to(dev)
        cplx = _cdtype(real)
        # spectrum of the (even) chirp filter through the library's own transform: p + i q, both real
        pq = _Rfft.apply(torch.stack([bd.real, bd.imag], dim=-1).view(1, P, 2), P, 1.0, 0.0)            # (1, P/2+1, 2), float64
        p_, q_ = pq[0, :, 0].real.to(real).contiguous(), pq[0, :, 1].real.to(real).contiguous()
        ent = _bluestein_cache[key] = (P, c.to(dev).to(cplx), p_, q_)
    return ent


def _chirp_convolve(a: torch.Tensor, P: int, p_: torch.Tensor, q_: torch.Tensor, t_out: int, conj_filter: bool) -> torch.Tensor:
    """sum_t a[t] conj(c[k - t]) (conj_filter: c[k - t]) for k < t_out; a complex (B, T, rest...), T <= P."""
    real = _rdtype(a)
    sh = (1, -1) + (1,) * (a.dim() - 2)
    ri = torch.stack([a.real, a.imag], dim=-1)                                 # one launch for both parts
    RI = _Rfft.apply(ri, P, 1.0, 0.0)                                          # (B, P/2+1, rest..., 2)
    R, I_ = RI[..., 0], RI[..., 1]
    pp, qq = p_.view(sh), (-q_ if conj_filter else q_).view(sh)
    C = torch.stack([R * pp - I_ * qq, R * qq + I_ * pp], dim=-1)
    y = _Irfft.apply(C, P, 1.0 / P, 0.0)[:, :t_out]                             # (B, t_out, rest..., 2) real
    return torch.complex(y[..., 0], y[..., 1])


def _envelope(nfft: int, env_log2: float, real: torch.dtype, dev: torch.device, ndim: int) -> torch.Tensor:
    e = torch.exp2(torch.arange(nfft, dtype=torch.float64, device=dev) * env_log2).to(real)
    return e.view((1, -1) + (1,) * (ndim - 2))


def _rfft_chirp(x: torch.Tensor, nfft: int, scale: float, env_log2: float) -> torch.Tensor:
    real, dev, M = _rdtype(x), x.device, nfft // 2 + 1
    if nfft == 1:
        return (x[:, :1] * scale).to(_cdtype(real))
    P, c, p_, q_ = _bluestein_setup(nfft, real, dev)
    x = x[:, :nfft]
    T = x.shape[1]
    if env_log2:
        x = x * _envelope(nfft, env_log2, real, dev, x.dim())[:, :T]
    sh = (1, -1) + (1,) * (x.dim() - 2)
    conv = _chirp_convolve(x * c[:T].view(sh), P, p_, q_, M, False)
    return conv * (c[:M] * scale).view(sh)


def _irfft_chirp(X: torch.Tensor, nfft: int, scale: float, env_log2: float) -> torch.Tensor:
    real, dev, M = _rdtype(X), X.device, nfft // 2 + 1
    if X.shape[1] != M:
        raise ValueError(f"irfft: expected {M} bins along dim 1, got {X.shape[1]}")
    if nfft == 1:
        return X.real * scale
    P, c, p_, q_ = _bluestein_setup(nfft, real, dev)
    sh = (1, -1) + (1,) * (X.dim() - 2)
    # y[t] = scale Re sum_{k < M} w_k X'[k] exp(+2 pi i k t / n): interior bins twice, the imaginary parts of bin 0 (and of
    # the Nyquist bin of an even length) ignored, as torch.fft.irfft does
    w = torch.full((M,), 2.0, dtype=real, device=dev)
    w[0] = 1.0
    edge = [0] + ([M - 1] if nfft % 2 == 0 else [])
    if nfft % 2 == 0:
        w[M - 1] = 1.0
    mask = torch.ones(M, dtype=real, device=dev)
    mask[edge] = 0.0
    U = torch.complex(X.real, X.imag * mask.view(sh)) * w.view(sh)
    conv = _chirp_convolve(U * c[:M].conj().view(sh), P, p_, q_, nfft, True)
    y = (conv * c.conj().view(sh)).real * scale
    if env_log2:
        y = y * _envelope(nfft, env_log2, real, dev, y.dim())
    return y


def rfft(x: torch.Tensor, nfft: int, norm: str = "backward", alias_decay_db: Optional[float] = None) -> torch.Tensor:
    """torch.fft.rfft(x [* gamma^-t], n=nfft, dim=1, norm=norm) on the HIP path."""
    if norm not in _NORM_FWD:
        raise ValueError(f"Invalid normalization mode: {norm}")
    env = 0.0 if not alias_decay_db else env_log2_of(alias_decay_db, nfft)
    _require_gpu(x)      # (before the plan query: that one loads the library)
    if not fft_plan_ok(int(nfft), _rdtype(x)):
        if x.is_complex():
            raise TypeError("rfft expects a real tensor")
        return _rfft_chirp(x, int(nfft), _NORM_FWD[norm](nfft), env)
    return _Rfft.apply(x, int(nfft), _NORM_FWD[norm](nfft), env)


def irfft(X: torch.Tensor, nfft: int, norm: str = "backward", alias_decay_db: Optional[float] = None) -> torch.Tensor:
    """torch.fft.irfft(X, n=nfft, dim=1, norm=norm) [* gamma^-t] on the HIP path."""
    if norm not in _NORM_INV:
        raise ValueError(f"Invalid normalization mode: {norm}")
    env = 0.0 if not alias_decay_db else env_log2_of(alias_decay_db, nfft)
    _require_gpu(X)
    if not fft_plan_ok(int(nfft), _rdtype(X)):
        if not X.is_complex():
            raise TypeError("irfft expects a complex tensor")
        return _irfft_chirp(X.resolve_conj(), int(nfft), _NORM_INV[norm](nfft), env)
    return _Irfft.apply(X, int(nfft), _NORM_INV[norm](nfft), env)


# ----------------------------------------------------------------------------- fused Shell pipeline
# y = irfft(H[f] . rfft(x)) in three launches (csrc/spectral.hip): time-domain tensors stay channel-innermost
# (B, T, G), the spectrum between the transforms and the product never goes through HBM, and what does cross the
# boundary (the spectrum kept for the backward pass, H, dL/dH) is bin-planar in ROW-MAJOR BIN ORDER
# (bin k = k1 + L1 k2 at element k1 L2 + k2).  float32 and float64 (the same kernels compiled for double, one workgroup per
# (row pair, batch item) and equal channel counts only); everything else takes the layered operators above.
def spectral_supported(nfft: int, n_in: int, n_out: int, real: torch.dtype = torch.float32) -> bool:
    L = _lib.lib()
    fn = L.fl_spec_supports if real == torch.float32 else L.fl_spec_supports_f64
    return bool(fn(int(nfft), int(n_in), int(n_out)))


def _spec_fn(name: str, real: torch.dtype):
    return getattr(_lib.lib(), name + ("_f32" if real == torch.float32 else "_f64"))


class _PermuteBins(torch.autograd.Function):
    """(M, rest...) per-bin tensor: natural bin order <-> row-major bin order of the fused pipeline's plan."""

    @staticmethod
    def forward(ctx, H, nfft, inverse):
        _require_gpu(H)
        if H.dtype not in (torch.complex64, torch.complex128):
            raise TypeError("permute_bins expects a complex64 / complex128 tensor")
        M = nfft // 2 + 1
        if H.shape[0] != M:
            raise ValueError(f"permute_bins: expected {M} bins along dim 0, got {H.shape[0]}")
        Hp = _h_planar(H.resolve_conj(), True)
        rest = tuple(H.shape[1:])
        out = _empty_rows(rest, M, H.dtype, H.device)
        fn = _lib.lib().fl_permute_bins_c64 if H.dtype == torch.complex64 else _lib.lib().fl_permute_bins_c128
        _lib.check(fn(Hp.data_ptr(), _lead_pitch(Hp.movedim(0, -1)), out.data_ptr(), _pitch(M),
                      max(_prod(rest), 1), nfft, int(inverse), _stream()), "permute_bins")
        ctx.cfg = (nfft, inverse)
        return out.movedim(-1, 0)

    @staticmethod
    def backward(ctx, g):
        nfft, inverse = ctx.cfg
        return _PermuteBins.apply(g, nfft, not inverse), None, None


def permute_bins(H: torch.Tensor, nfft: int, inverse: bool = False) -> torch.Tensor:
    """Per-bin tensor (M, ...) from natural to row-major bin order (``inverse``: back)."""
    return _PermuteBins.apply(H, int(nfft), bool(inverse))


LAUNCH_PAIRS = True       # the cascade response's launch rides in the input's column pass (csrc/fusedfwd.hip); False: two launches


class paired_launch:
    """Between enter and exit a Matrix-then-cascade response launch of this thread (fl_geq_response_rc_c64 /
    fl_sos_response_rc_c64, second-generation kernel) is RECORDED by the library and issued by the next float32 forward column
    pass in the same grid (csrc/fusedfwd.h): the response's packed arithmetic runs beside the column pass's HBM traffic instead of
    in front of it.  Any other library call in between issues the recorded launch first (flamo_amd/_lib.py: lib()), exit issues
    one that is still recorded; torch operations that READ the response in between must be preceded by ``flush()`` -- the
    only user, Shell's fused forward (processor/system.py), does that."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled and LAUNCH_PAIRS)

    def __enter__(self):
        if self.enabled:
            if getattr(_lib._pair, "stream_of", None) is not None:      # nested: the outer region owns the mode
                self.enabled = False
            else:
                _lib.lib().fl_launch_pair_begin()
                _lib._pair.stream_of = _stream
        return self

    def flush(self):
        if self.enabled and getattr(_lib._pair, "stream_of", None) is not None:
            L = _lib.lib(pair_ok=True)
            if L.fl_launch_pair_pending():
                _lib.check(L.fl_launch_pair_flush(_stream()), "launch pair flush")
                L.fl_launch_pair_begin()

    def __exit__(self, *exc):
        if self.enabled:
            _lib._pair.stream_of = None
            _lib.check(_lib.lib().fl_launch_pair_flush(_stream()), "launch pair flush")
        return False


def _spec_cols_fwd(x, nfft, env_log2, site=0):
    """x: contiguous real (B, T, G) -> scratch (B*L*G,) complex.  site 1: the gradient's transform (its input was written by the
    launch in front of it: another cache policy than for the forward transform's input, fl_set_stream_policy)"""
    B, T, G = x.shape
    S = torch.empty(B * (nfft // 2) * G, dtype=_cdtype(x.dtype), device=x.device)
    W = twiddles(nfft, x.dtype, x.device)
    L = _lib.lib(pair_ok=not site)        # the forward transform's pass carries a recorded response launch (paired_launch)
    if site:
        L.fl_set_stream_policy(0xFFFFFFFF, site)
    try:
        with kernel_timer.span("spec_cols_fwd+response" if (not site and kernel_timer.enabled and L.fl_launch_pair_pending())
                               else "spec_cols_fwd"):
            fn = L.fl_spec_cols_fwd_f32 if x.dtype == torch.float32 else L.fl_spec_cols_fwd_f64
            _lib.check(fn(x.data_ptr(), B, T, G, S.data_ptr(), W.data_ptr(), nfft, env_log2, _stream()), "spec_cols_fwd")
    finally:
        if site:
            L.fl_set_stream_policy(0xFFFFFFFF, 0)
    return S


def _spec_mid(S, B, NI, NO, nfft, Hrm, conj_t, want_spec, want_inverse, spec_scale, interior2, pre_half):
    """-> (S2 or None, spectrum rows (B, NI, M) or None).  Hrm: (M, NO_h, NI_h) row-major-order response view or None."""
    dev = S.device
    M = nfft // 2 + 1
    real = _rdtype(S)
    Xs = _empty_rows((B, NI), M, S.dtype, dev) if want_spec else None
    S2 = None
    if want_inverse:
        S2 = S if NI == NO else torch.empty(B * (nfft // 2) * NO, dtype=S.dtype, device=dev)   # row pairs are private: in place
    hp = hs_m = hs_n = 0
    if Hrm is not None:
        hp = _lead_pitch(Hrm.movedim(0, -1))
        hs_m, hs_n = Hrm.shape[2] * hp, hp
        if conj_t:
            hs_m, hs_n = hs_n, hs_m
    P = _pitch(M)
    tag = f"spec_mid[{NI}->{NO}" + (",H" if Hrm is not None else "") + (",inv" if want_inverse else "") + (",spec" if want_spec else "") + "]"
    with kernel_timer.span(tag):
        _lib.check(_spec_fn("fl_spec_mid", real)(S.data_ptr(), None if S2 is None else S2.data_ptr(),
                                                 None if Xs is None else Xs.data_ptr(), NI * P, P,
                                                 None if Hrm is None else Hrm.data_ptr(), hs_m, hs_n, int(bool(conj_t)),
                                                 twiddles(nfft, real, dev).data_ptr(), nfft, B, NI, NO, spec_scale,
                                                 int(interior2), int(pre_half), _stream()), "spec_mid")
    return S2, Xs


def _spec_cols_inv(S2, B, t_len, t_out, G, nfft, scale, env_log2, want_sumsq=False, dev_scale=None, grad_cols=False, sg_buf=None):
    """-> y (B, t_len, G); with want_sumsq also the per-workgroup partial sums of y^2 (double) the same launch leaves behind;
    dev_scale: a device scalar of the pipeline's real dtype multiplied into `scale` by the kernel; grad_cols (with want_sumsq, plain
    shape): also Sg = _spec_cols_fwd(y), formed by the same launch from its tiles (-> y, parts, Sg)"""
    alloc = torch.zeros if t_len > t_out else torch.empty
    real = _rdtype(S2)
    if INVERSE_IN_PLACE and t_len == t_out == nfft and S2.numel() == B * (nfft // 2) * G and S2.is_contiguous() and \
            _spec_fn("fl_spec_cols_inv_inplace_ok", real)(nfft, G):
        # y over the scratch it is transformed from (a workgroup's tile of samples is the bytes of the tile of column values it
        # has read before its first store): one streaming array less in the step -- the arrays' total, not only the passes'
        # bytes, is what the memory-side cache sees
        # (a tensor of its own over the same storage, not a view of S2: the caller hands it out as a Function's output)
        y = torch.empty(0, dtype=real, device=S2.device).set_(S2.untyped_storage(), 2 * S2.storage_offset(), (B, t_len, G))
    else:
        y = alloc((B, t_len, G), dtype=real, device=S2.device)
    parts = None
    if grad_cols:
        assert want_sumsq and t_len == t_out == nfft and env_log2 == 0.0
        nblk = int(_spec_fn("fl_spec_cols_blocks", real)(nfft, B, G))
        parts = torch.empty(max(nblk, 1), dtype=torch.float64, device=S2.device)
        # (sg_buf: a dead scratch array of the same size -- the forward transform's column pass, consumed by the row kernel --
        # so that the step's streaming arrays stay four, as without the fused pass)
        n_sg = B * (nfft // 2) * G
        Sg = sg_buf if sg_buf is not None and sg_buf.numel() == n_sg and sg_buf.dtype == S2.dtype and sg_buf.data_ptr() != S2.data_ptr() \
            else torch.empty(n_sg, dtype=S2.dtype, device=S2.device)
        with kernel_timer.span("spec_cols_inv+grad_cols"):
            _lib.check(_spec_fn("fl_spec_cols_inv_sumsq_grad", real)(S2.data_ptr(), y.data_ptr(), Sg.data_ptr(), B, G,
                                                                     twiddles(nfft, real, S2.device).data_ptr(), nfft, scale,
                                                                     parts.data_ptr(), _stream()), "spec_cols_inv_sumsq_grad")
        return y, parts, Sg
    with kernel_timer.span("spec_cols_inv"):
        if want_sumsq:
            nblk = int(_spec_fn("fl_spec_cols_blocks", real)(nfft, B, G))
            parts = torch.empty(max(nblk, 1), dtype=torch.float64, device=S2.device)
            _lib.check(_spec_fn("fl_spec_cols_inv_sumsq", real)(S2.data_ptr(), y.data_ptr(), B, t_len, t_out, G,
                                                                twiddles(nfft, real, S2.device).data_ptr(), nfft, scale, env_log2,
                                                                parts.data_ptr(), _stream()), "spec_cols_inv_sumsq")
        elif dev_scale is not None:
            if dev_scale.dtype != real or not dev_scale.is_cuda or dev_scale.numel() != 1:
                raise ValueError("spec_cols_inv: dev_scale must be one device scalar of the signal's dtype")
            _lib.check(_spec_fn("fl_spec_cols_inv_scaled", real)(S2.data_ptr(), y.data_ptr(), B, t_len, t_out, G,
                                                                 twiddles(nfft, real, S2.device).data_ptr(), nfft, scale,
                                                                 dev_scale.data_ptr(), env_log2, _stream()), "spec_cols_inv_scaled")
        else:
            _lib.check(_spec_fn("fl_spec_cols_inv", real)(S2.data_ptr(), y.data_ptr(), B, t_len, t_out, G,
                                                          twiddles(nfft, real, S2.device).data_ptr(), nfft, scale, env_log2,
                                                          _stream()), "spec_cols_inv")
    return (y, parts) if want_sumsq else y


# Batch-walking row kernels (csrc/specwalk.hip): one workgroup per CU keeps the row pair's response slice in registers and
# walks (row pair, batch item) units; the spectrum the backward pass needs is kept pair-major (private to the two kernels) and
# dL/dH is accumulated in registers from it -- no (B, M, N) spectrum, no separate gradient pass over two of them.
_WALK_MIN_BATCH = 4        # below this a workgroup's range is a unit or two: the fill (response into registers) dominates


def _walk_applies(nfft: int, B: int, NI: int, NO: int) -> bool:
    return B >= _WALK_MIN_BATCH and bool(_lib.lib().fl_spec_walk_supports(int(nfft), int(NI), int(NO)))


_walk_bounds = {}


def _walk_partition(nfft: int, B: int, dev: torch.device) -> torch.Tensor:
    """Device copy of the forward walking kernel's work partition (fl_spec_walk_partition), cached per (nfft, batch, device)."""
    L = _lib.lib()
    n_wg = int(L.fl_spec_walk_workgroups(nfft, B))
    key = (int(nfft), int(B), n_wg, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _walk_bounds.get(key)
    if t is None:
        import ctypes
        host = (ctypes.c_int * (n_wg + 1))()
        _lib.check(L.fl_spec_walk_partition(nfft, B, n_wg, host), "spec_walk_partition")
        if torch.cuda.is_current_stream_capturing():
            return None          # no host-to-device copy inside a capture: this call runs on equal counts; the warm-up before it filled the cache
        t = torch.tensor(list(host), dtype=torch.int32).to(dev)
        torch.cuda.current_stream(dev).synchronize()
        _walk_bounds[key] = t
    return t


def _spec_mid_walk(S, B, NI, NO, nfft, Hrm, conj_t, want_spec, spec_scale, interior2, pre_half):
    """-> (S2, pair-major spectrum or None) through fl_spec_mid_walk_f32"""
    dev = S.device
    L = _lib.lib()
    bounds = _walk_partition(nfft, B, dev)
    S2 = torch.empty(B * (nfft // 2) * NO, dtype=torch.complex64, device=dev)
    Xp = torch.empty(int(L.fl_spec_walk_spectrum_elems(nfft, B, NI)), dtype=torch.complex64, device=dev) if want_spec else None
    hp = _lead_pitch(Hrm.movedim(0, -1))
    hs_m, hs_n = Hrm.shape[2] * hp, hp
    if conj_t:
        hs_m, hs_n = hs_n, hs_m
    tag = f"spec_mid_walk[{NI}->{NO}" + (",spec" if want_spec else "") + "]"
    with kernel_timer.span(tag):
        _lib.check(L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), None if Xp is None else Xp.data_ptr(), Hrm.data_ptr(), hs_m, hs_n,
                                          int(bool(conj_t)), twiddles(nfft, torch.float32, dev).data_ptr(), nfft, B, NI, NO, spec_scale,
                                          int(interior2), int(pre_half), None if bounds is None else bounds.data_ptr(), _stream()),
                   "spec_mid_walk")
    return S2, Xp


INVERSE_IN_PLACE = True      # False: the inverse column pass writes a fresh (B, nfft, G) array (see _spec_cols_inv)
GRADH_LOOP = True      # False: the layered backward (spec_mid without a response + mimo_gradh); tests compare the two
# The one-launch form's parallelism is (row pairs x channel groups) -- 202 workgroups at nfft = 96000 -- whatever the batch; the
# layered form's grows with the batch.  Measured on an MI355X (tools/dbg/gradloop_dbg.py): 29.5 against 36.1 us at two items in
# float64, 22.6 against 27.8 in float32, 31.6 against 36.0 (4 x 4, eight items), but 62.4 against 56.4 at nfft = 65536 (130
# workgroups) with eight items and 211 against 206 us at 32 items in float64 (its two FFT stages run on two to six of the
# workgroup's eight wavefronts and are latency-bound).  Up to this many items it is taken:
GRADH_LOOP_MAX_BATCH = 4


def _spec_gradh_loop(Sg, Xs, B, NI, NO, nfft, scale_g, host_factor, out_scale):
    """dL/dH (M, NO, NI) view in row-major bin order from the gradient's column pass Sg and the kept spectrum Xs (B, NI, P)"""
    dev = Sg.device
    M = nfft // 2 + 1
    real = _rdtype(Sg)
    P = _pitch(M)
    dH = _empty_rows((NO, NI), M, Sg.dtype, dev)
    with kernel_timer.span("spec_gradh_loop"):
        _lib.check(_spec_fn("fl_spec_gradh_loop", real)(Sg.data_ptr(), Xs.data_ptr(), NI * P, P, dH.data_ptr(), NI * P, P,
                                                        twiddles(nfft, real, dev).data_ptr(), nfft, B, NI, NO, float(scale_g), 1,
                                                        float(host_factor), None if out_scale is None else out_scale.data_ptr(),
                                                        _stream()), "spec_gradh_loop")
    return dH.movedim(-1, 0)


def _spec_gradh_walk(Sg, Xp, B, NI, NO, nfft, scale_g, out_scale=None):
    """dL/dH (M, NO, NI) view, row-major bin order, from the gradient's scratch rows and the pair-major spectrum;
    out_scale: float32 device scalar multiplied in on the way out"""
    dev = Sg.device
    L = _lib.lib()
    M = nfft // 2 + 1
    P = _pitch(M)
    ns = int(L.fl_spec_gradh_slices(nfft, B))
    parts = torch.empty((ns, NO, NI, P), dtype=torch.complex64, device=dev)
    with kernel_timer.span("spec_gradh_walk"):
        _lib.check(L.fl_spec_gradh_walk_scaled_f32(Sg.data_ptr(), Xp.data_ptr(), parts.data_ptr(), NO * NI * P, NI * P, P, ns,
                                                   twiddles(nfft, torch.float32, dev).data_ptr(), nfft, B, NI, NO, scale_g, 1,
                                                   None if out_scale is None else out_scale.data_ptr(), _stream()),
                   "spec_gradh_walk")
    if ns == 1:
        out = parts[0]
    else:
        out = torch.empty((NO, NI, P), dtype=torch.complex64, device=dev)
        with kernel_timer.span("sum_parts"):
            _lib.check(L.fl_sum_parts_c64(parts.data_ptr(), NO * NI * P, ns, out.data_ptr(), NO * NI * P, _stream()), "sum_parts")
    return out[..., :M].movedim(-1, 0)


class _SpectralApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Hrm, nfft, scale_f, env_f, scale_i, env_i):
        _require_gpu(x, Hrm)
        if x.dtype not in (torch.float32, torch.float64) or x.dim() != 3:
            raise TypeError("spectral_apply expects a real float32 / float64 (B, T, N) signal")
        M = nfft // 2 + 1
        if Hrm.dim() != 3 or Hrm.shape[0] != M or Hrm.dtype != _cdtype(x.dtype):
            raise ValueError(f"spectral_apply: the response must be {_cdtype(x.dtype)} ({M}, N_out, N_in)")
        NO, NI = Hrm.shape[1], Hrm.shape[2]
        if x.shape[2] != NI:
            raise ValueError(f"response expects {NI} input channels, signal has {x.shape[2]}")
        xc = x.contiguous()
        if xc.data_ptr() % (2 * xc.element_size()):
            xc = xc.clone()
        Hp = _h_planar(Hrm.resolve_conj(), True)
        B, T = xc.shape[0], xc.shape[1]
        S = _spec_cols_fwd(xc, nfft, env_f)
        walk = x.dtype == torch.float32 and _walk_applies(nfft, B, NI, NO)
        if walk:
            S2, Xs = _spec_mid_walk(S, B, NI, NO, nfft, Hp, False, ctx.needs_input_grad[1], scale_f, 0, 0)
        else:
            S2, Xs = _spec_mid(S, B, NI, NO, nfft, Hp, False, ctx.needs_input_grad[1], True, scale_f, 0, 0)
        # (the gradient's first pass rides in the inverse pass when this operator's output went into mean_square the last time
        # it ran with this shape -- _grad_cols_key / GRAD_COLS_IN_FORWARD)
        key = _grad_cols_key(x, nfft, NI, NO)
        Sg = None
        if GRAD_COLS_IN_FORWARD and key in _GRAD_COLS_SEEN and env_i == 0.0 and any(ctx.needs_input_grad[:2]) and \
                _spec_fn("fl_spec_cols_inv_grad_supported", x.dtype)(nfft, NO):
            y, parts, Sg = _spec_cols_inv(S2, B, nfft, nfft, NO, nfft, scale_i, env_i, want_sumsq=True, grad_cols=True, sg_buf=S)
        else:
            y, parts = _spec_cols_inv(S2, B, nfft, nfft, NO, nfft, scale_i, env_i, want_sumsq=True)
        ctx.save_for_backward(Hp, *([Xs] if Xs is not None else []))
        ctx.cfg = (nfft, scale_f, env_f, scale_i, env_i, T, NI, NO, walk)
        ctx.speculated = key if Sg is not None else None
        # what an objective computed from y alone can reuse (mean_square): the partial sums the inverse pass left behind, and
        # everything the backward pass needs -- see _SpectralMeanSquare
        y._flamo_sa = _SpectralTag(x, Hrm, Hp, Xs, ctx.cfg, parts, y._version, Sg, key)
        return y

    @staticmethod
    def backward(ctx, gy):
        Hp, *kept = ctx.saved_tensors
        if ctx.speculated is not None:
            _GRAD_COLS_SEEN.discard(ctx.speculated)      # another criterion came: this shape stops forming the column pass ahead
        return _SpectralApply._backward(ctx.cfg, Hp, kept[0] if kept else None, ctx.needs_input_grad[0], ctx.needs_input_grad[1], gy, None) \
            + (None, None, None, None, None)

    @staticmethod
    def _backward(cfg, Hp, Xkept, need_x, need_h, gy, out_scale, host_factor=1.0, Sg=None):
        """Gradients (gx, gH) for the output gradient gy -- or, with out_scale (a device scalar c of gy's dtype) and host_factor
        (a Python float f), for (c f) * gy without forming it: the pipeline is linear, so the factor is applied where the
        results are small (f rides in the walking kernel's transform scale, c in its epilogue: no launch of its own)."""
        kept = [Xkept]
        nfft, scale_f, env_f, scale_i, env_i, T, NI, NO, walk = cfg
        g = gy.contiguous()
        if g.data_ptr() % (2 * g.element_size()):
            g = g.clone()
        B = g.shape[0]
        # irfft' : g_Y[k] = w_k scale_i sum_t g_y[t] e_i(t) exp(-j w_k t) -- a forward transform with doubled interior bins
        # (Sg given: its column pass was formed by the forward pass's inverse launch from the tiles of y -- _SpectralMeanSquare)
        kept_sg = Sg is not None      # (a tensor the autograd node keeps: spec_mid works in place when NI == NO -- it gets a copy)
        if Sg is None:
            Sg = _spec_cols_fwd(g, nfft, env_i, site=1)
        gx = gH = None
        if walk:
            if need_h:
                gH = _spec_gradh_walk(Sg, kept[0], B, NI, NO, nfft, scale_i * host_factor, out_scale)      # (walking kernels: float32)
            if need_x:
                # rfft' : g_x[t] = scale_f e_f(t) Re sum_k g_X[k] exp(+j w_k t), g_X = H^H g_Y
                if _walk_applies(nfft, B, NO, NI):
                    S3, _ = _spec_mid_walk(Sg, B, NO, NI, nfft, Hp, True, False, scale_i, 1, 1)
                else:
                    S3, _ = _spec_mid(Sg.clone() if kept_sg else Sg, B, NO, NI, nfft, Hp, True, False, True, scale_i, 1, 1)
                # (the objective's factor: its host part in the pass's scale, its device part multiplied in by the kernel)
                gx = _spec_cols_inv(S3, B, T, min(T, nfft), NI, nfft, scale_f * (host_factor if out_scale is not None else 1.0), env_f,
                                    dev_scale=out_scale)
            return gx, gH
        if need_h and not need_x and GRADH_LOOP and B <= GRADH_LOOP_MAX_BATCH and _spec_fn("fl_spec_gradh_loop_supports", _rdtype(Sg))(nfft, NI, NO):
            # the response's gradient alone (the training step: the data tensor takes no gradient, trainer.py:172-191): one launch
            # that walks the batch -- the gradient's spectrum is never written and read back (fl_spec_gradh_loop_*)
            return None, _spec_gradh_loop(Sg, kept[0], B, NI, NO, nfft, scale_i, host_factor if out_scale is not None else 1.0, out_scale)
        # rfft' : g_x[t] = scale_f e_f(t) Re sum_k g_X[k] exp(+j w_k t), g_X = H^H g_Y -- an inverse transform with halved interior bins
        S3, gYs = _spec_mid(Sg.clone() if kept_sg and need_x else Sg, B, NO, NI if need_x else NO, nfft, Hp if need_x else None, True, need_h,
                            need_x, scale_i, 1, 1)
        if need_x:
            gx = _spec_cols_inv(S3, B, T, min(T, nfft), NI, nfft, scale_f * (host_factor if out_scale is not None else 1.0), env_f,
                                dev_scale=out_scale)
        if need_h:
            Xs = kept[0]
            gH = _gradh_launch(gYs.movedim(-1, 1), Xs.movedim(-1, 1), False, host_factor if out_scale is not None else 1.0,
                               out_scale).movedim(-1, 0)
        return gx, gH


class _SpectralTag:
    """Rides on the tensor _SpectralApply returns (attribute ``_flamo_sa``): the inputs and kept arrays of that evaluation."""
    __slots__ = ("x", "Hrm", "Hp", "Xs", "cfg", "parts", "version", "Sg", "key")

    def __init__(self, x, Hrm, Hp, Xs, cfg, parts, version, Sg=None, key=None):
        self.x, self.Hrm, self.Hp, self.Xs, self.cfg, self.parts, self.version = x, Hrm, Hp, Xs, cfg, parts, version
        self.Sg, self.key = Sg, key


# The first pass of the gradient's transform inside the forward pass's inverse launch (fl_spec_cols_inv_sumsq_grad_*): worth a
# 98 MB store at BASELINE configs[1] only when the backward pass of mean_square(y) follows -- which the operator cannot know when
# it runs.  It goes by what happened the last time: a shape whose output went into mean_square AND was differentiated is
# remembered (_SpectralMeanSquare.backward) and takes the fused launch from then on; a miss costs that store, never a result,
# and a gradient that arrives through the operator's own node instead (another criterion) makes the shape forget.
GRAD_COLS_IN_FORWARD = True
_GRAD_COLS_SEEN = set()


def _grad_cols_key(x, nfft, NI, NO):
    return (int(nfft), int(x.shape[0]), int(NI), int(NO), x.dtype, x.device.index)


class _SpectralMeanSquare(torch.autograd.Function):
    """mean(y^2) of y = spectral_apply(x, H) as ONE node over (x, H): the value comes from the partial sums the inverse column
    pass left behind (no pass over y), and the backward pass -- g_y = (2 g / N) y, a multiple of y itself -- runs the pipeline's
    backward ON y with the factor applied to its small results: neither the objective's read of y nor the read + write that
    forms g_y happen (3 of the step's 15.5 signal-sized passes at BASELINE configs[1]).  The tensor y stays a normal
    output of its own node: whatever else consumes it differentiates through that node as before."""

    @staticmethod
    def forward(ctx, x, Hrm, y, tag):
        real = y.dtype
        loss = torch.empty((), dtype=real, device=y.device)
        fn = _lib.lib().fl_mean_square_final_f32 if real == torch.float32 else _lib.lib().fl_mean_square_final_f64
        with kernel_timer.span("mean_square_final"):
            _lib.check(fn(tag.parts.data_ptr(), tag.parts.numel(), 1.0 / y.numel(), loss.data_ptr(), _stream()), "mean_square_final")
        ctx.has_xs, ctx.has_sg = tag.Xs is not None, tag.Sg is not None
        ctx.save_for_backward(y, tag.Hp, *([tag.Xs] if ctx.has_xs else []), *([tag.Sg] if ctx.has_sg else []))
        ctx.cfg = tag.cfg
        ctx.key = tag.key
        return loss

    @staticmethod
    def backward(ctx, gloss):
        y, Hp, *kept = ctx.saved_tensors
        Xs = kept.pop(0) if ctx.has_xs else None
        Sg = kept.pop(0) if ctx.has_sg else None
        if ctx.key is not None:
            _GRAD_COLS_SEEN.add(ctx.key)           # the next forward pass of this shape forms the gradient's column pass itself
        c = gloss.to(y.dtype).reshape(())          # device scalar in the pipeline's precision (no launch when it already is)
        gx, gH = _SpectralApply._backward(ctx.cfg, Hp, Xs, ctx.needs_input_grad[0], ctx.needs_input_grad[1], y, c,
                                          2.0 / y.numel(), Sg=Sg)
        return gx, gH, None, None


def spectral_apply(x: torch.Tensor, Hrm: torch.Tensor, nfft: int, norm_f: str = "backward", norm_i: str = "backward",
                   db_f: Optional[float] = None, db_i: Optional[float] = None) -> torch.Tensor:
    """irfft(H[f] . rfft(x [* gamma_f^-t], n=nfft, norm_f), n=nfft, norm_i) [* gamma_i^-t] along dim 1 of a real float32 / float64
    (B, T, N_in) signal, with ``Hrm`` the (M, N_out, N_in) per-bin response in ROW-MAJOR bin order
    (``permute_bins``).  Returns (B, nfft, N_out), contiguous."""
    if norm_f not in _NORM_FWD or norm_i not in _NORM_INV:
        raise ValueError(f"Invalid normalization mode: {norm_f if norm_f not in _NORM_FWD else norm_i}")
    env_f = 0.0 if not db_f else env_log2_of(db_f, nfft)
    env_i = 0.0 if not db_i else env_log2_of(db_i, nfft)
    return _SpectralApply.apply(x, Hrm, int(nfft), _NORM_FWD[norm_f](nfft), env_f, _NORM_INV[norm_i](nfft), env_i)


# ----------------------------------------------------------------------------- per-bin MIMO product
def _h_planar(H: torch.Tensor, per_bin: bool) -> torch.Tensor:
    """Per-bin responses (M, ...) are used with the bin axis contiguous (rows possibly padded)."""
    if not per_bin:
        return H          # frequency-independent: used through its strides, whatever they are
    if _lead_pitch(H.movedim(0, -1)) is not None:
        return H
    M = H.shape[0]
    rest = tuple(H.shape[1:])
    P = _pitch(M)
    out = _transpose(H.contiguous(), 1, M, _prod(rest), P)
    return out.view(*rest, P)[..., :M].movedim(-1, 0)


def _mimo_launch(H, per_bin, diag, conj_t, X):
    """Y = op(H) X.  conj_t: use H^H (swap m/n, conjugate)."""
    real = _rdtype(X)
    B, M, Nx, K, xs_b, xs_n, xs_k = _bnk(X)
    L = _lib.lib()
    hp = _lead_pitch(H.movedim(0, -1)) if per_bin else 0   # pitch of the per-bin response rows
    if diag:
        N = H.shape[-1]
        hs_f, hs_n = (1, hp) if per_bin else (0, H.stride(-1))
        Y = _empty_planar(X.shape, X.dtype, X.device)
        _, _, _, _, ys_b, ys_n, ys_k = _bnk(Y)
        fn = L.fl_mimo_diag_c64 if real == torch.float32 else L.fl_mimo_diag_c128
        _lib.check(fn(H.data_ptr(), hs_f, hs_n, int(conj_t), X.data_ptr(), xs_b, xs_n, xs_k, Y.data_ptr(), ys_b, ys_n,
                      ys_k, B, M, N, K, _stream()), "mimo_diag")
        return Y
    No_h, Ni_h = H.shape[-2], H.shape[-1]
    if per_bin:
        hs_f, hs_m, hs_n = 1, Ni_h * hp, hp
    else:
        hs_f, hs_m, hs_n = 0, H.stride(-2), H.stride(-1)
    if conj_t:
        No, Ni, hs_m, hs_n = Ni_h, No_h, hs_n, hs_m
    else:
        No, Ni = No_h, Ni_h
    Y = _empty_planar((B, M, No, *X.shape[3:]), X.dtype, X.device)
    _, _, _, _, ys_b, ys_m, ys_k = _bnk(Y)
    fn = L.fl_mimo_c64 if real == torch.float32 else L.fl_mimo_c128
    tag = ("mimo_bin" if per_bin else "mimo_const") + ("_adj" if conj_t else "_fwd") + f"[cols={B * K},{No}x{Ni}]"
    with kernel_timer.span(tag):
        _lib.check(fn(H.data_ptr(), hs_f, hs_m, hs_n, int(conj_t), X.data_ptr(), xs_b, xs_n, xs_k, Y.data_ptr(), ys_b,
                      ys_m, ys_k, B, M, No, Ni, K, _stream()), "mimo")
    return Y


def _gradh_launch(G, X, diag, scale=1.0, dev_scale=None):
    """sum over batch/trailing dims of G x conj(X): planar (No, Ni, M) [full] or (N, M) [diag].
    dev_scale (full form): a device scalar of the real dtype multiplied into `scale` by the kernel."""
    real = _rdtype(X)
    B, M, Ni, K, xs_b, xs_n, xs_k = _bnk(X)
    _, _, No, _, gs_b, gs_m, gs_k = _bnk(G)
    L = _lib.lib()
    P = _pitch(M)
    if diag:
        dh = _empty_rows((Ni,), M, X.dtype, X.device)
        fn = L.fl_mimo_gradh_diag_c64 if real == torch.float32 else L.fl_mimo_gradh_diag_c128
        _lib.check(fn(G.data_ptr(), gs_b, gs_m, gs_k, X.data_ptr(), xs_b, xs_n, xs_k, dh.data_ptr(), P, B, M, Ni, K,
                      _stream()), "mimo_gradh_diag")
        return dh if scale == 1.0 else dh * scale
    dH = _empty_rows((No, Ni), M, X.dtype, X.device)
    fn = L.fl_mimo_gradh_c64 if real == torch.float32 else L.fl_mimo_gradh_c128
    with kernel_timer.span(f"mimo_gradh[cols={B * K},{No}x{Ni}]"):
        if dev_scale is not None:
            if dev_scale.dtype != real or not dev_scale.is_cuda or dev_scale.numel() != 1:
                raise ValueError("mimo_gradh: dev_scale must be one device scalar of the signal's real dtype")
            fn = L.fl_mimo_gradh_scaled_c64 if real == torch.float32 else L.fl_mimo_gradh_scaled_c128
            _lib.check(fn(G.data_ptr(), gs_b, gs_m, gs_k, X.data_ptr(), xs_b, xs_n, xs_k, dH.data_ptr(), P, float(scale),
                          dev_scale.data_ptr(), B, M, No, Ni, K, _stream()), "mimo_gradh_scaled")
        else:
            _lib.check(fn(G.data_ptr(), gs_b, gs_m, gs_k, X.data_ptr(), xs_b, xs_n, xs_k, dH.data_ptr(), P, float(scale), B, M,
                          No, Ni, K, _stream()), "mimo_gradh")
    return dH


def _gradw_launch(G, X):
    """sum over batch, trailing dims AND bins of G x conj(X): (No, Ni) -- gradient of a
    frequency-independent matrix, reduced in the kernel."""
    real = _rdtype(X)
    B, M, Ni, K, xs_b, xs_n, xs_k = _bnk(X)
    _, _, No, _, gs_b, gs_m, gs_k = _bnk(G)
    L = _lib.lib()
    nblk = L.fl_mimo_gradw_blocks(M)
    part = torch.empty((nblk, No, Ni), dtype=X.dtype, device=X.device)
    dW = torch.empty((No, Ni), dtype=X.dtype, device=X.device)
    fn = L.fl_mimo_gradw_c64 if real == torch.float32 else L.fl_mimo_gradw_c128
    with kernel_timer.span(f"mimo_gradw[cols={B * K},{No}x{Ni}]"):
        _lib.check(fn(G.data_ptr(), gs_b, gs_m, gs_k, X.data_ptr(), xs_b, xs_n, xs_k, part.data_ptr(), dW.data_ptr(), B, M, No, Ni,
                      K, _stream()), "mimo_gradw")
    return dW


class _Mimo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, X, diag):
        _require_gpu(H, X)
        if not X.is_complex():
            raise TypeError("the per-bin product expects a complex (frequency-domain) signal")
        per_bin = H.dim() == (2 if diag else 3)
        if H.dim() not in ((1, 2) if diag else (2, 3)):
            raise ValueError(f"bad response rank {H.dim()}")
        M = X.shape[1]
        if per_bin and H.shape[0] != M:
            raise ValueError(f"response has {H.shape[0]} bins, signal has {M}")
        if H.shape[-1] != X.shape[2]:
            raise ValueError(f"response expects {H.shape[-1]} input channels, signal has {X.shape[2]}")
        Hp = _h_planar(H.resolve_conj(), per_bin)
        Xp = to_planar(X.resolve_conj())
        ctx.save_for_backward(Hp, Xp)
        ctx.cfg = (per_bin, bool(diag))
        return _mimo_launch(Hp, per_bin, diag, False, Xp)

    @staticmethod
    def backward(ctx, gY):
        Hp, Xp = ctx.saved_tensors
        per_bin, diag = ctx.cfg
        gY = to_planar(gY.resolve_conj())
        gH = gX = None
        if ctx.needs_input_grad[1]:
            gX = _mimo_launch(Hp, per_bin, diag, True, gY)
        if ctx.needs_input_grad[0]:
            if per_bin or diag:
                g = _gradh_launch(gY, Xp, diag)  # planar (.., M)
                gH = g.movedim(-1, 0) if per_bin else g.sum(dim=-1)
            else:
                gH = _gradw_launch(gY, Xp)       # frequency-independent matrix: reduced over bins in-kernel
        return gH, gX, None


# A real frequency-independent matrix (Gain, Matrix) is applied as it is (fl_mimo_* with conj_h bit 1, real gradient from
# fl_mimo_gradw_re_*): the real -> complex cast of the reference (dsp.py:466-468) and its backward are three tiny launches
# per module and step, which at batch 1 is what a feedback delay network's input and output gains cost.  False = cast.
REAL_CONST_MIMO = True


def _mimo_real_launch(W, conj_t, X):
    """Y = W X or W^T X for a real (No, Ni) matrix W of X's precision: einsum("mn,bfn...->bfm...", to_complex(W), X) of
    dsp.py:466-468 without forming to_complex(W)."""
    real = _rdtype(X)
    B, M, Nx, K, xs_b, xs_n, xs_k = _bnk(X)
    hs_m, hs_n = W.stride(0), W.stride(1)
    No, Ni = W.shape
    if conj_t:
        No, Ni, hs_m, hs_n = Ni, No, hs_n, hs_m
    if Ni != Nx:
        raise ValueError(f"response expects {Ni} input channels, signal has {Nx}")
    Y = _empty_planar((B, M, No, *X.shape[3:]), X.dtype, X.device)
    _, _, _, _, ys_b, ys_m, ys_k = _bnk(Y)
    L = _lib.lib()
    fn = L.fl_mimo_c64 if real == torch.float32 else L.fl_mimo_c128
    with kernel_timer.span("mimo_const_real" + ("_adj" if conj_t else "_fwd") + f"[cols={B * K},{No}x{Ni}]"):
        _lib.check(fn(W.data_ptr(), 0, hs_m, hs_n, 2, X.data_ptr(), xs_b, xs_n, xs_k, Y.data_ptr(), ys_b, ys_m, ys_k, B, M, No, Ni,
                      K, _stream()), "mimo")
    return Y


class _MimoRealConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, X):
        _require_gpu(W, X)
        Xp = to_planar(X.resolve_conj())
        ctx.save_for_backward(W, Xp)
        return _mimo_real_launch(W, False, Xp)

    @staticmethod
    def backward(ctx, gY):
        W, Xp = ctx.saved_tensors
        gY = to_planar(gY.resolve_conj())
        gW = gX = None
        if ctx.needs_input_grad[1]:
            gX = _mimo_real_launch(W, True, gY)
        if ctx.needs_input_grad[0]:
            real = _rdtype(Xp)
            B, M, Ni, K, xs_b, xs_n, xs_k = _bnk(Xp)
            _, _, No, _, gs_b, gs_m, gs_k = _bnk(gY)
            L = _lib.lib()
            part = torch.empty((L.fl_mimo_gradw_blocks(M), No, Ni), dtype=Xp.dtype, device=Xp.device)
            gW = torch.empty((No, Ni), dtype=real, device=Xp.device)
            fn = L.fl_mimo_gradw_re_c64 if real == torch.float32 else L.fl_mimo_gradw_re_c128
            with kernel_timer.span(f"mimo_gradw[cols={B * K},{No}x{Ni}]"):
                _lib.check(fn(gY.data_ptr(), gs_b, gs_m, gs_k, Xp.data_ptr(), xs_b, xs_n, xs_k, part.data_ptr(), gW.data_ptr(), B, M,
                              No, Ni, K, _stream()), "mimo_gradw")
        return gW, gX


def _real_const_applies(H, X, diag) -> bool:
    """a real (No, Ni) matrix of the signal's precision, small enough that the lane-per-bin kernel (the one with the real
    path) is the kernel the product would take anyway (the matrix-core kernels start at 16 rows x 8 columns x 8 deep)"""
    if diag or H.is_complex() or H.dim() != 2 or not X.is_complex() or not X.is_cuda or not REAL_CONST_MIMO:
        return False
    if H.dtype != _rdtype(X):
        return False
    cols = X.shape[0] * _prod(X.shape[3:])
    return not (X.dtype == torch.complex64 and H.shape[0] >= 16 and cols >= 8 and H.shape[1] >= 8)


def mimo(H: torch.Tensor, X: torch.Tensor, diag: bool = False) -> torch.Tensor:
    """Per-bin complex product.  ``H``: (M,No,Ni) | (No,Ni) for ``diag=False``; (M,N) | (N,) for
    ``diag=True``.  ``X``: (B, M, Ni, ...).  Implements the four flamo einsum patterns
    "fmn,bfn...->bfm...", "mn,bfn...->bfm...", "fn,bfn...->bfn...", "n,bfn...->bfn...".
    A real frequency-independent ``H`` is accepted as it is (no complex cast is formed for small matrices)."""
    if _real_const_applies(H, X, diag):
        if H.shape[-1] != X.shape[2]:
            raise ValueError(f"response expects {H.shape[-1]} input channels, signal has {X.shape[2]}")
        return _MimoRealConst.apply(H, X)
    if H.dtype != X.dtype:
        H = H.to(X.dtype)  # differentiable cast (real -> complex, or precision)
    return _Mimo.apply(H, X, bool(diag))


# ----------------------------------------------------------------------------- closed-loop solve
def _solve_launch(Pp, one_minus, adjoint, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    nws = int(L.fl_solve_ws_bytes(N, M, int(real == torch.float64)))
    if nws > 0:
        # loops beyond what one workgroup's LDS holds (N > 138 / 97): the factorisation's matrices in a workspace this call owns
        ws = torch.empty(nws, dtype=torch.uint8, device=R.device)
        fn = L.fl_solve_ws_c64 if real == torch.float32 else L.fl_solve_ws_c128
        with kernel_timer.span("solve_adj" if adjoint else "solve"):
            _lib.check(fn(Pp.data_ptr(), _lead_pitch(Pp.movedim(0, -1)), int(one_minus), int(adjoint), R.data_ptr(), rs_b, rs_n,
                          rs_k, OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, ws.data_ptr(), nws, _stream()), "solve_ws")
        return OUT
    fn = L.fl_solve_c64 if real == torch.float32 else L.fl_solve_c128
    with kernel_timer.span("solve_adj" if adjoint else "solve"):
        _lib.check(fn(Pp.data_ptr(), _lead_pitch(Pp.movedim(0, -1)), int(one_minus), int(adjoint), R.data_ptr(), rs_b, rs_n,
                      rs_k, OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, _stream()), "solve")
    return OUT


class _Solve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, R, one_minus):
        _require_gpu(P, R)
        # P: (1, M, N, N) or (M, N, N) logical; stored planar (N, N, M)
        P3 = P[0] if P.dim() == 4 else P
        if P.dim() == 4 and P.shape[0] != 1:
            raise ValueError("solve: the system matrix must not be batched (it is batch-independent)")
        M, N = P3.shape[0], P3.shape[1]
        if P3.shape[2] != N or R.shape[1] != M or R.shape[2] != N:
            raise ValueError(f"solve: incompatible shapes {tuple(P.shape)} and {tuple(R.shape)}")
        Pp = _h_planar(P3.resolve_conj(), True)
        Rp = to_planar(R.resolve_conj())
        OUT = _solve_launch(Pp, one_minus, False, Rp)
        ctx.save_for_backward(Pp, OUT)
        ctx.cfg = (bool(one_minus), P.dim())
        return OUT

    @staticmethod
    def backward(ctx, gOUT):
        Pp, OUT = ctx.saved_tensors
        one_minus, pdim = ctx.cfg
        gR = _solve_launch(Pp, one_minus, True, to_planar(gOUT.resolve_conj()))  # A^-H g
        gP = None
        if ctx.needs_input_grad[0]:
            # dA = -gR out^H ;  dP = -dA when A = I - P
            gP = _gradh_launch(gR, OUT, False, scale=(1.0 if one_minus else -1.0)).movedim(-1, 0)
            if pdim == 4:
                gP = gP.unsqueeze(0)
        return gP, (gR if ctx.needs_input_grad[1] else None), None


def solve(P: torch.Tensor, R: torch.Tensor, one_minus: bool = True) -> torch.Tensor:
    """Per bin: (I - P[f])^-1 R[:, f] (``one_minus``) or P[f]^-1 R[:, f]; R: (B, M, N[, K...])."""
    if P.dtype != R.dtype:
        P = P.to(R.dtype)
    return _Solve.apply(P, R, bool(one_minus))


def _diag_args(d: Optional[torch.Tensor]):
    """(ptr, s_n, s_f) of an optional diagonal factor: per-bin (M, N), constant (N,) or None."""
    if d is None:
        return None, 0, 0
    if d.dim() == 1:
        return d.data_ptr(), d.stride(0), 0
    return d.data_ptr(), d.stride(1), d.stride(0)


def _solve_dud_launch(l, U, r, adjoint, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    fn = L.fl_solve_dud_c64 if real == torch.float32 else L.fl_solve_dud_c128
    lp, l_sn, l_sf = _diag_args(l)
    rp, r_sn, r_sf = _diag_args(r)
    with kernel_timer.span("solve_dud_adj" if adjoint else "solve_dud"):
        _lib.check(fn(lp, l_sn, l_sf, U.data_ptr(), rp, r_sn, r_sf, int(adjoint), R.data_ptr(), rs_b, rs_n, rs_k,
                      OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, _stream()), "solve_dud")
    return OUT


# one-pass backward of the factored solve (fl_solve_dud_grads_*); False = the layered form (tests compare the two)
FUSE_DUD_GRADS = True


def _dud_grads_launch(lp, Uc, rp, gR, OUT, need_l, need_U, need_r):
    """(gl (M, N) | None, gU (N, N) | None, gr (M, N) | None) from gR = A^-H g and OUT = A^-1 R (planar, same strides)."""
    real = _rdtype(OUT)
    B, M, N, K, s_b, s_n, s_k = _bnk(OUT)
    assert _bnk(gR) == (B, M, N, K, s_b, s_n, s_k)
    dev = OUT.device
    L = _lib.lib()
    gl = _empty_rows((N,), M, OUT.dtype, dev) if need_l else None
    gr = _empty_rows((N,), M, OUT.dtype, dev) if need_r else None
    part = gU = None
    if need_U:
        part = torch.empty((L.fl_solve_dud_grads_blocks(M, N), N, N), dtype=OUT.dtype, device=dev)
        gU = torch.empty((N, N), dtype=OUT.dtype, device=dev)
    lptr, l_sn, l_sf = _diag_args(lp)
    rptr, r_sn, r_sf = _diag_args(rp)
    fn = L.fl_solve_dud_grads_c64 if real == torch.float32 else L.fl_solve_dud_grads_c128
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with kernel_timer.span("solve_dud_grads"):
        _lib.check(fn(lptr, l_sn, l_sf, Uc.data_ptr(), rptr, r_sn, r_sf, gR.data_ptr(), OUT.data_ptr(), s_b, s_n, s_k, B, M, N, K,
                      ptr(gl), _pitch(M), ptr(gr), _pitch(M), ptr(part), ptr(gU), _stream()), "solve_dud_grads")
    return (None if gl is None else gl.movedim(-1, 0)), gU, (None if gr is None else gr.movedim(-1, 0))


class _SolveDUD(torch.autograd.Function):
    """OUT = (I - diag(l) U diag(r))^-1 R per bin; l, r: per-bin (M,N) / constant (N,) / None."""

    @staticmethod
    def forward(ctx, l, U, r, R):
        _require_gpu(U, R)
        Rp = to_planar(R.resolve_conj())
        Uc = U.resolve_conj().contiguous()
        lp = None if l is None else (_h_planar(l.resolve_conj(), True) if l.dim() == 2 else l.resolve_conj().contiguous())
        rp = None if r is None else (_h_planar(r.resolve_conj(), True) if r.dim() == 2 else r.resolve_conj().contiguous())
        OUT = _solve_dud_launch(lp, Uc, rp, False, Rp)
        ctx.have = (l is not None, r is not None)
        ctx.save_for_backward(*([t for t in (lp, rp) if t is not None] + [Uc, OUT]))
        return OUT

    @staticmethod
    def backward(ctx, gOUT):
        saved = list(ctx.saved_tensors)
        lp = saved.pop(0) if ctx.have[0] else None
        rp = saved.pop(0) if ctx.have[1] else None
        Uc, OUT = saved
        gR = _solve_dud_launch(lp, Uc, rp, True, to_planar(gOUT.resolve_conj()))          # A^-H g
        gl = gU = gr = None
        need_l, need_U, need_r = ctx.needs_input_grad[0] and lp is not None, ctx.needs_input_grad[1], \
            ctx.needs_input_grad[2] and rp is not None
        fusable = (lp is None or lp.dim() == 2) and (rp is None or rp.dim() == 2)      # per-bin or absent diagonal factors
        if (need_l or need_U or need_r) and fusable and FUSE_DUD_GRADS:
            gl, gU, gr = _dud_grads_launch(lp, Uc, rp, gR, OUT, need_l, need_U, need_r)
        elif need_l or need_U or need_r:
            # dP_ij = sum_b gR_i conj(out_j) with P_ij = l_i U_ij r_j, contracted without forming dP:
            t1 = OUT if rp is None else _mimo_launch(rp, rp.dim() == 2, True, False, OUT)   # r * out
            t2 = gR if lp is None else _mimo_launch(lp, lp.dim() == 2, True, True, gR)      # conj(l) * gR
            if need_U:
                gU = _gradw_launch(t2, t1)                                                  # sum_f,b t2_i conj(t1_j)
            if need_l:
                v = _mimo_launch(Uc, False, False, False, t1)                               # U (r*out)
                g = _gradh_launch(gR, v, True)                                              # (N, M): sum_b gR conj(v)
                gl = g.movedim(-1, 0) if lp.dim() == 2 else g.sum(dim=-1)
            if need_r:
                w = _mimo_launch(Uc, False, False, True, t2)                                # U^H (conj(l)*gR)
                g = _gradh_launch(w, OUT, True)                                             # sum_b w conj(out)
                gr = g.movedim(-1, 0) if rp.dim() == 2 else g.sum(dim=-1)
        return gl, gU, gr, (gR if ctx.needs_input_grad[3] else None)


class _SolveDUD2(torch.autograd.Function):
    """OUT = (I - diag(l . l2) U diag(r))^-1 (l2 . R0) per bin: the loop of a feedback delay network with the feedforward
    path's diagonal l2 (no gradient) applied where the kernels load l and the right-hand side (fl_solve_dud2_*).
    l, r: per-bin (M, N) or None; l2: per-bin (M, N)."""

    @staticmethod
    def forward(ctx, l, l2, U, r, R0):
        _require_gpu(l2, U, R0)
        Rp = to_planar(R0.resolve_conj())
        Uc = U.resolve_conj().contiguous()
        pl = lambda t: None if t is None else _h_planar(t.resolve_conj(), True)  # noqa: E731
        lp, l2p, rp = pl(l), pl(l2), pl(r)
        OUT = _solve_dud2_launch(lp, l2p, True, Uc, rp, False, Rp)
        ctx.have = (l is not None, r is not None)
        ctx.save_for_backward(*([t for t in (lp, rp) if t is not None] + [l2p, Uc, OUT]))
        return OUT

    @staticmethod
    def backward(ctx, gOUT):
        saved = list(ctx.saved_tensors)
        lp = saved.pop(0) if ctx.have[0] else None
        rp = saved.pop(0) if ctx.have[1] else None
        l2p, Uc, OUT = saved
        gR = _solve_dud2_launch(lp, l2p, False, Uc, rp, True, to_planar(gOUT.resolve_conj()))     # A^-H g
        need_l, need_U, need_r, need_R = (ctx.needs_input_grad[0] and lp is not None, ctx.needs_input_grad[2],
                                          ctx.needs_input_grad[3] and rp is not None, ctx.needs_input_grad[4])
        real = _rdtype(OUT)
        B, M, N, K, s_b, s_n, s_k = _bnk(OUT)
        dev = OUT.device
        L = _lib.lib()
        gl = _empty_rows((N,), M, OUT.dtype, dev) if need_l else None
        gr = _empty_rows((N,), M, OUT.dtype, dev) if need_r else None
        gR0 = _empty_planar(OUT.shape, OUT.dtype, dev) if need_R else None
        part = gU = None
        if need_U:
            part = torch.empty((L.fl_solve_dud_grads_blocks(M, N), N, N), dtype=OUT.dtype, device=dev)
            gU = torch.empty((N, N), dtype=OUT.dtype, device=dev)
        if need_l or need_U or need_r or need_R:
            lptr, l_sn, l_sf = _diag_args(lp)
            l2ptr, l2_sn, l2_sf = _diag_args(l2p)
            rptr, r_sn, r_sf = _diag_args(rp)
            fn = L.fl_solve_dud2_grads_c64 if real == torch.float32 else L.fl_solve_dud2_grads_c128
            ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            with kernel_timer.span("solve_dud_grads"):
                _lib.check(fn(lptr, l_sn, l_sf, l2ptr, l2_sn, l2_sf, Uc.data_ptr(), rptr, r_sn, r_sf, gR.data_ptr(), OUT.data_ptr(),
                              s_b, s_n, s_k, B, M, N, K, ptr(gl), _pitch(M), ptr(gr), _pitch(M), ptr(part), ptr(gU), ptr(gR0),
                              None, 0, None, 0, None, _stream()), "solve_dud2_grads")
        return (None if gl is None else gl.movedim(-1, 0)), None, gU, (None if gr is None else gr.movedim(-1, 0)), gR0


def _apply_const(W, transpose, X):
    """W X or W^T X (W^H for a complex W) for a frequency-independent (No, Ni) matrix, real or complex"""
    if not W.is_complex():
        return _mimo_real_launch(W, transpose, X)
    return _mimo_launch(W, False, False, transpose, X)


# ops.fdn_core: build the right-hand side b x and apply the output-gain row inside the solve kernels (fl_solve_fdn_*; the
# in-place kernels: N <= 32 in float32, 16 in float64).  False = separate launches for b x, c OUT and c^H gy.
FDN_GAINS_IN_SOLVE = True


def _fdn_in_solve(real, N) -> bool:
    return FDN_GAINS_IN_SOLVE and N <= (32 if real == torch.float32 else 16)


# The FDN's forward solve CAN keep its LU factors (8 N^2 bytes per bin: 197 MB at 16 channels, nfft = 192000) for the backward
# pass's adjoint system (fl_solve_fdn_keep_* / fl_solve_kept_adjoint_rank1_*; 8 < N <= 16) -- measured and left OFF: at N = 16 the
# two passes over the factors (stores + 39 us on the forward kernel, 43.5 us for the substitution, both at HBM rate) cost more
# than the elimination they replace (66 us): 0.339 against 0.311 ms on the replayed FDN step.  The elimination grows with N^3, the
# factors with N^2: the scaled loop keeps them above 16 channels (KEEP_LU), where it wins 1.8 ms at N = 32.  Tests run both.
KEEP_LU_FDN = False


# A one-output FDN's backward right-hand side is the output-gain row times a scalar per (batch item, bin): its adjoint solution is
# w[f] gy[b][f] with w = A^-H c^H formed by the FORWARD launch from its own factors (fl_solve_fdn_wadj_c64; float32, 4 < N <= 16) --
# the backward pass then runs no solve.  False: the adjoint system is factored and solved by a launch of its own.
FDN_ADJOINT_IN_FORWARD = True
FDN_ADJOINT_IN_FORWARD_F64 = True      # (float64: the forward kernel with w runs at one wavefront per SIMD -- still ahead of a second solve)


def _solve_fdn_launch(l, l2, U, r, adjoint, gain, sig, cw, keep=False, wadj=False):
    """OUT = A^-1 (l2 . (gain sig)) [forward] or A^-H (conj(gain) sig) [adjoint]; with cw (forward) also z = cw . OUT.
    gain, cw: contiguous N-vectors, real or complex; sig: planar one-channel signal (B, M, 1).  -> (OUT, z | None), or with
    keep (forward): (OUT, z, (LU, piv, tile)) -- the factors for _solve_fdn_kept_adjoint_launch."""
    real = _rdtype(sig)
    B, M, _, K, ss_b, _, _ = _bnk(sig)
    assert K == 1
    N = U.shape[0]
    OUT = _empty_planar((B, M, N), sig.dtype, sig.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    z = _empty_planar((B, M, 1), sig.dtype, sig.device) if cw is not None else None
    zs_b = _bnk(z)[4] if z is not None else 0
    L = _lib.lib()
    lp, l_sn, l_sf = _diag_args(l)
    l2p, l2_sn, l2_sf = _diag_args(l2)
    rp, r_sn, r_sf = _diag_args(r)
    if wadj:        # forward: (OUT, z, W) with W (1, M, N) planar = A^-H cw^H
        assert not adjoint and cw is not None
        W = _empty_planar((1, M, N), sig.dtype, sig.device)
        _, _, _, _, _, ws_n, _ = _bnk(W)
        fnw = L.fl_solve_fdn_wadj_c64 if real == torch.float32 else L.fl_solve_fdn_wadj_c128
        with kernel_timer.span("solve_dud"):
            _lib.check(fnw(lp, l_sn, l_sf, l2p, l2_sn, l2_sf, U.data_ptr(), rp, r_sn, r_sf, gain.data_ptr(),
                           int(not gain.is_complex()), sig.data_ptr(), ss_b, cw.data_ptr(), int(not cw.is_complex()),
                           z.data_ptr(), zs_b, OUT.data_ptr(), os_b, os_n, os_k, B, M, N, W.data_ptr(), ws_n,
                           _stream()), "solve_fdn_wadj")
        return OUT, z, W
    if keep:
        assert not adjoint
        tile = L.fl_solve_fdn_keep_tile(N, int(real == torch.float64))
        nt = (M + tile - 1) // tile
        LU = torch.empty(nt * tile * N * N, dtype=sig.dtype, device=sig.device)
        piv = torch.empty(nt * tile * N, dtype=torch.int32, device=sig.device)
        fn = L.fl_solve_fdn_keep_c64 if real == torch.float32 else L.fl_solve_fdn_keep_c128
        with kernel_timer.span("solve_dud"):
            _lib.check(fn(lp, l_sn, l_sf, l2p, l2_sn, l2_sf, U.data_ptr(), rp, r_sn, r_sf, gain.data_ptr(),
                          int(not gain.is_complex()), sig.data_ptr(), ss_b, None if cw is None else cw.data_ptr(),
                          int(cw is not None and not cw.is_complex()), None if z is None else z.data_ptr(), zs_b, OUT.data_ptr(), os_b,
                          os_n, os_k, B, M, N, LU.data_ptr(), piv.data_ptr(), _stream()), "solve_fdn_keep")
        return OUT, z, (LU, piv, tile)
    fn = L.fl_solve_fdn_c64 if real == torch.float32 else L.fl_solve_fdn_c128
    with kernel_timer.span("solve_dud_adj" if adjoint else "solve_dud"):
        _lib.check(fn(lp, l_sn, l_sf, l2p, l2_sn, l2_sf, U.data_ptr(), rp, r_sn, r_sf, int(adjoint), gain.data_ptr(),
                      int(not gain.is_complex()), sig.data_ptr(), ss_b, None if cw is None else cw.data_ptr(),
                      int(cw is not None and not cw.is_complex()), None if z is None else z.data_ptr(), zs_b, OUT.data_ptr(), os_b,
                      os_n, os_k, B, M, N, _stream()), "solve_fdn")
    return OUT, z


def _solve_fdn_kept_adjoint_launch(LU, piv, tile, gain, sig, N):
    """A^-H (conj(gain) sig) from the factors the forward solve kept"""
    real = _rdtype(sig)
    B, M, _, K, ss_b, _, _ = _bnk(sig)
    OUT = _empty_planar((B, M, N), sig.dtype, sig.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    fn = L.fl_solve_kept_adjoint_rank1_c64 if real == torch.float32 else L.fl_solve_kept_adjoint_rank1_c128
    with kernel_timer.span("solve_dud_adj"):
        _lib.check(fn(LU.data_ptr(), piv.data_ptr(), int(tile), gain.data_ptr(), int(not gain.is_complex()), sig.data_ptr(), ss_b,
                      OUT.data_ptr(), os_b, os_n, os_k, B, M, N, _stream()), "solve_kept_adjoint_rank1")
    return OUT


class _FdnCore(torch.autograd.Function):
    """y = c (I - diag(l . l2) U diag(r))^-1 (l2 . (b x)) per bin: a feedback delay network between its input-gain column
    b (N, 1) and output-gain row c (1, N) (Series(Gain(N,1), Recursion, Gain(1,N)), reverb.py:117-199 / e8_fdn.py:60-100).
    Forward: the three launches the modules would make.  Backward: c^H gy, the adjoint solve, and ONE pass that returns the
    gradients of l, U, r, b and c (fl_solve_dud2_grads_* with its side reductions) -- the two gains' gradients are sums over
    bins of products the pass already holds."""

    @staticmethod
    def forward(ctx, b, c, l, l2, U, r, X):
        _require_gpu(b, c, l2, U, X)
        N = U.shape[0]
        if b.shape != (N, 1) or c.shape != (1, N) or X.dim() != 3 or X.shape[2] != 1:
            raise ValueError("fdn_core: expected b (N, 1), c (1, N) and a one-channel spectrum X (B, M, 1)")
        Xp = to_planar(X.resolve_conj())
        Uc = U.resolve_conj().contiguous()
        pl = lambda t: None if t is None else _h_planar(t.resolve_conj(), True)  # noqa: E731
        lp, l2p, rp = pl(l), pl(l2), pl(r)
        bc, cc = b.resolve_conj().contiguous(), c.resolve_conj().contiguous()
        ctx.in_solve = _fdn_in_solve(_rdtype(Xp), N)
        kept = None
        Wadj = None
        if ctx.in_solve and FDN_ADJOINT_IN_FORWARD and any(ctx.needs_input_grad) and \
                (_rdtype(Xp) == torch.float32 or FDN_ADJOINT_IN_FORWARD_F64) and _lib.lib().fl_solve_fdn_wadj_supported(N):
            OUT, y, Wadj = _solve_fdn_launch(lp, l2p, Uc, rp, False, bc, Xp, cc, wadj=True)
        elif ctx.in_solve and KEEP_LU_FDN and any(ctx.needs_input_grad) and \
                _lib.lib().fl_solve_fdn_keep_tile(N, int(_rdtype(Xp) == torch.float64)) > 0:
            OUT, y, kept = _solve_fdn_launch(lp, l2p, Uc, rp, False, bc, Xp, cc, keep=True)
        elif ctx.in_solve:
            OUT, y = _solve_fdn_launch(lp, l2p, Uc, rp, False, bc, Xp, cc)
        else:
            R0 = _apply_const(bc, False, Xp)
            OUT = _solve_dud2_launch(lp, l2p, True, Uc, rp, False, R0)
            y = _apply_const(cc, False, OUT)
        ctx.have = (l is not None, r is not None)
        ctx.kept_tile = kept[2] if kept is not None else 0
        ctx.has_wadj = Wadj is not None
        ctx.save_for_backward(*([t for t in (lp, rp) if t is not None] + [l2p, Uc, OUT, Xp, bc, cc] +
                                (list(kept[:2]) if kept is not None else []) + ([Wadj] if Wadj is not None else [])))
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = list(ctx.saved_tensors)
        lp = saved.pop(0) if ctx.have[0] else None
        rp = saved.pop(0) if ctx.have[1] else None
        l2p, Uc, OUT, Xp, bc, cc, *kept = saved
        gyp = to_planar(gy.resolve_conj())
        Wadj = kept[-1] if ctx.has_wadj else None
        if Wadj is not None:   # A^-H c^H gy = w gy with w from the forward launch: formed inside the gradient kernel, no solve, no tensor
            gR = None
        elif kept:        # A^-H c^H gy from the forward solve's factors: a substitution, no second elimination
            gR = _solve_fdn_kept_adjoint_launch(kept[0], kept[1], ctx.kept_tile, cc, gyp, Uc.shape[0])
        elif ctx.in_solve:
            gR, _ = _solve_fdn_launch(lp, l2p, Uc, rp, True, cc, gyp, None)                           # A^-H c^H gy
        else:
            gR = _solve_dud2_launch(lp, l2p, False, Uc, rp, True, _apply_const(cc, True, gyp))
        need_b, need_c, need_l, need_U, need_r, need_X = (ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                                          ctx.needs_input_grad[2] and lp is not None, ctx.needs_input_grad[4],
                                                          ctx.needs_input_grad[5] and rp is not None, ctx.needs_input_grad[6])
        real = _rdtype(OUT)
        B, M, N, K, s_b, s_n, s_k = _bnk(OUT)
        dev = OUT.device
        L = _lib.lib()
        gl = _empty_rows((N,), M, OUT.dtype, dev) if need_l else None
        gr = _empty_rows((N,), M, OUT.dtype, dev) if need_r else None
        gR0 = _empty_planar(OUT.shape, OUT.dtype, dev) if need_X else None
        side = need_b or need_c
        real_gains = not bc.is_complex() and not cc.is_complex()
        part = gUS = side_real = None
        if need_U or side:
            cnt = N * N + (2 * N if side else 0)
            part = torch.empty((L.fl_solve_dud_grads_blocks(M, N), cnt), dtype=OUT.dtype, device=dev)
            gUS = torch.empty((cnt,), dtype=OUT.dtype, device=dev)
            if side and real_gains:
                side_real = torch.empty((2 * N,), dtype=real, device=dev)
        g_b = g_c = gU = gX = None
        if need_l or need_U or need_r or need_X or side:
            lptr, l_sn, l_sf = _diag_args(lp)
            l2ptr, l2_sn, l2_sf = _diag_args(l2p)
            rptr, r_sn, r_sf = _diag_args(rp)
            _, _, _, _, xs_b, _, _ = _bnk(Xp)
            _, _, _, _, gs_b, _, _ = _bnk(gyp)
            fn = L.fl_solve_dud2_grads_c64 if real == torch.float32 else L.fl_solve_dud2_grads_c128
            ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            with kernel_timer.span("solve_dud_grads"):
                if Wadj is not None:
                    _, _, _, _, _, w_sn, _ = _bnk(Wadj)
                    fnw = L.fl_solve_dud2_grads_w_c64 if real == torch.float32 else L.fl_solve_dud2_grads_w_c128
                    _lib.check(fnw(lptr, l_sn, l_sf, l2ptr, l2_sn, l2_sf, Uc.data_ptr(), rptr, r_sn, r_sf,
                                   Wadj.data_ptr(), w_sn, gyp.data_ptr(), gs_b, OUT.data_ptr(), s_b, s_n, s_k, B, M, N,
                                   ptr(gl), _pitch(M), ptr(gr), _pitch(M), ptr(part), ptr(gUS), ptr(gR0),
                                   Xp.data_ptr() if side else None, xs_b, gyp.data_ptr() if side else None, gs_b,
                                   ptr(side_real), _stream()), "solve_dud2_grads_w")
                else:
                    _lib.check(fn(lptr, l_sn, l_sf, l2ptr, l2_sn, l2_sf, Uc.data_ptr(), rptr, r_sn, r_sf, gR.data_ptr(), OUT.data_ptr(),
                                  s_b, s_n, s_k, B, M, N, K, ptr(gl), _pitch(M), ptr(gr), _pitch(M), ptr(part), ptr(gUS), ptr(gR0),
                                  Xp.data_ptr() if side else None, xs_b, gyp.data_ptr() if side else None, gs_b, ptr(side_real),
                                  _stream()), "solve_dud2_grads")
            if need_U:
                gU = gUS[:N * N].view(N, N)
            if side:
                tail = side_real if side_real is not None else gUS[N * N:]
                gb_full, gc_full = tail[:N].view(N, 1), tail[N:].view(1, N)
                if side_real is None:        # mixed real / complex gains: the real one takes the real part
                    gb_full = gb_full if bc.is_complex() else gb_full.real
                    gc_full = gc_full if cc.is_complex() else gc_full.real
                g_b = gb_full if need_b else None
                g_c = gc_full if need_c else None
            if need_X:
                gX = _apply_const(bc, True, gR0)
        return (g_b, g_c, (None if gl is None else gl.movedim(-1, 0)), None, gU, (None if gr is None else gr.movedim(-1, 0)), gX)


def fdn_core(b, c, l, l2, U, r, X):
    """One-channel spectrum X (B, M, 1) through  c (I - diag(l . l2) U diag(r))^-1 (l2 . (b X))  per bin -- see _FdnCore.
    b (N, 1), c (1, N): real (of X's precision) or complex constants; l, r: per-bin (M, N) or None; l2: per-bin (M, N) without
    gradient; U (N, N)."""
    if l2.requires_grad:
        raise ValueError("fdn_core: the feedforward diagonal must not require a gradient")
    cd = X.dtype
    rd = _rdtype(X)
    conv = lambda t: None if t is None else (t if t.dtype == cd else t.to(cd))  # noqa: E731
    gain = lambda t: t if (t.dtype == rd or t.dtype == cd) else t.to(cd if t.is_complex() else rd)  # noqa: E731
    return _FdnCore.apply(gain(b), gain(c), conv(l), conv(l2), conv(U), conv(r), X)


def _solve_dud2_launch(l, l2, rhs_l2, U, r, adjoint, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    fn = L.fl_solve_dud2_c64 if real == torch.float32 else L.fl_solve_dud2_c128
    lp, l_sn, l_sf = _diag_args(l)
    l2p, l2_sn, l2_sf = _diag_args(l2)
    rp, r_sn, r_sf = _diag_args(r)
    with kernel_timer.span("solve_dud_adj" if adjoint else "solve_dud"):
        _lib.check(fn(lp, l_sn, l_sf, l2p, l2_sn, l2_sf, int(rhs_l2), U.data_ptr(), rp, r_sn, r_sf, int(adjoint), R.data_ptr(),
                      rs_b, rs_n, rs_k, OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, _stream()), "solve_dud2")
    return OUT


def solve_dud2(l: Optional[torch.Tensor], l2: torch.Tensor, U: torch.Tensor, r: Optional[torch.Tensor], R0: torch.Tensor) -> torch.Tensor:
    """Per bin f: (I - diag(l[f] . l2[f]) U diag(r[f]))^-1 (l2[f] . R0[:, f]); l, r: per-bin (M, N) or None, l2: per-bin
    (M, N) without gradient (the diagonal of the feedforward path, which also scales the right-hand side).
    Replaces system.py:417-424 (Recursion.forward: A = I - fF(fB(I)), solve(A, fF(X))) for the FDN structure of
    reverb.py:117-199 / e8_fdn.py:60-100, where fF is the delay line alone."""
    if l2.requires_grad:
        raise ValueError("solve_dud2: the feedforward diagonal must not require a gradient (use solve_dud)")
    cd = R0.dtype
    conv = lambda t: None if t is None else (t if t.dtype == cd else t.to(cd))  # noqa: E731
    return _SolveDUD2.apply(conv(l), conv(l2), conv(U), conv(r), R0)


def _solve_scaled_launch(DU, g, adjoint, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    fn = L.fl_solve_scaled_c64 if real == torch.float32 else L.fl_solve_scaled_c128
    with kernel_timer.span("solve_adj" if adjoint else "solve"):
        _lib.check(fn(DU.data_ptr(), _lead_pitch(DU.movedim(0, -1)), g.data_ptr(), g.stride(0), int(adjoint), R.data_ptr(), rs_b,
                      rs_n, rs_k, OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, _stream()), "solve_scaled")
    return OUT


# The scaled loop's forward solve keeps its LU factors for the backward pass's adjoint system when the elimination dominates the
# solve (N > KEEP_LU_MIN_N) and the factors fit the budget: 8 N^2 bytes per bin in HBM (1.6 GB at the 32 x 32 chain of nfft =
# 384000) against a second (2/3) N^3 elimination per bin.  False / a smaller budget: the adjoint system is factored again.
KEEP_LU = True
KEEP_LU_MIN_N = 16
KEEP_LU_MAX_BYTES = 16 << 30


def _solve_scaled_keep_launch(DU, g, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    f64 = int(real == torch.float64)
    LU = torch.empty(L.fl_solve_kept_lu_elems(N, M, f64), dtype=R.dtype, device=R.device)
    piv = torch.empty(L.fl_solve_kept_piv_elems(N, M, f64), dtype=torch.int32, device=R.device)
    fn = L.fl_solve_scaled_keep_c64 if real == torch.float32 else L.fl_solve_scaled_keep_c128
    with kernel_timer.span("solve"):
        _lib.check(fn(DU.data_ptr(), _lead_pitch(DU.movedim(0, -1)), g.data_ptr(), g.stride(0), R.data_ptr(), rs_b, rs_n, rs_k,
                      OUT.data_ptr(), os_b, os_n, os_k, B, M, N, K, LU.data_ptr(), piv.data_ptr(), _stream()), "solve_scaled_keep")
    return OUT, LU, piv


def _solve_kept_adjoint_launch(LU, piv, R):
    real = _rdtype(R)
    B, M, N, K, rs_b, rs_n, rs_k = _bnk(R)
    OUT = _empty_planar(R.shape, R.dtype, R.device)
    _, _, _, _, os_b, os_n, os_k = _bnk(OUT)
    L = _lib.lib()
    fn = L.fl_solve_kept_adjoint_c64 if real == torch.float32 else L.fl_solve_kept_adjoint_c128
    with kernel_timer.span("solve_adj"):
        _lib.check(fn(LU.data_ptr(), piv.data_ptr(), R.data_ptr(), rs_b, rs_n, rs_k, OUT.data_ptr(), os_b,
                      os_n, os_k, B, M, N, K, _stream()), "solve_kept_adjoint")
    return OUT


class _SolveScaledLoop(torch.autograd.Function):
    """OUT = (I - diag(g) D[f] U)^-1 R per bin; D: per-bin (M, N, N) WITHOUT gradient, g: (N,), U: (N, N) constants.
    P' = D U is formed once; the gains scale its rows inside the solve.  Backward without the (M, N, N) gradient of the
    loop matrix: dP = gR (x) conj(out) is an outer product per bin, so
        g_g = sum_f gR . conj(P' out),      g_U = sum_f (D^H (conj(g) . gR)) (x) conj(out)
    are two per-bin matrix-vector passes and two small reductions."""

    @staticmethod
    def forward(ctx, g, D, U, R):
        _require_gpu(g, D, U, R)
        N = U.shape[0]
        if D.dim() != 3 or D.shape[1:] != (N, N) or U.shape != (N, N) or g.shape != (N,) or R.shape[2] != N:
            raise ValueError("solve_scaled_loop: expected g (N,), D (M, N, N), U (N, N), R (B, M, N, ...)")
        Dp = _h_planar(D.resolve_conj(), True)
        gc, Uc = g.resolve_conj().contiguous(), U.resolve_conj().contiguous()
        Rp = to_planar(R.resolve_conj())
        # P'[f] = D[f] U: rows of D as batch items of a signal, U^T as the constant matrix (planar memory of the result IS
        # the per-bin matrix (M, N, N))
        DU = _mimo_launch(Uc.transpose(-1, -2), False, False, False, to_planar(Dp.permute(1, 0, 2))).permute(1, 0, 2)
        M = Rp.shape[1]
        keep = (KEEP_LU and any(ctx.needs_input_grad) and KEEP_LU_MIN_N < N <= (64 if _rdtype(Rp) == torch.float32 else 32)
                and N * N * (M + 64) * Rp.element_size() <= KEEP_LU_MAX_BYTES)
        if keep:
            OUT, LU, piv = _solve_scaled_keep_launch(DU, gc, Rp)
            ctx.save_for_backward(gc, Dp, Uc, DU, OUT, LU, piv)
        else:
            OUT = _solve_scaled_launch(DU, gc, False, Rp)
            ctx.save_for_backward(gc, Dp, Uc, DU, OUT)
        return OUT

    @staticmethod
    def backward(ctx, gOUT):
        gc, Dp, Uc, DU, OUT, *kept = ctx.saved_tensors
        if kept:      # A^-H g from the forward solve's factors: a substitution, no second elimination
            gR = _solve_kept_adjoint_launch(kept[0], kept[1], to_planar(gOUT.resolve_conj()))
        else:
            gR = _solve_scaled_launch(DU, gc, True, to_planar(gOUT.resolve_conj()))    # A^-H g
        g_g = g_U = None
        if ctx.needs_input_grad[0]:
            v = _mimo_launch(DU, True, False, False, OUT)                             # P' out
            g_g = _gradh_launch(gR, v, True).sum(dim=-1)                              # (N,): sum_f,b gR conj(v)
        if ctx.needs_input_grad[2]:
            t = _mimo_launch(gc, False, True, True, gR)                               # conj(g) . gR
            w = _mimo_launch(Dp, True, False, True, t)                                # D^H (conj(g) . gR)
            g_U = _gradw_launch(w, OUT)                                               # sum_f,b w (x) conj(out)
        return g_g, None, g_U, (gR if ctx.needs_input_grad[3] else None)


def solve_scaled_loop(g: torch.Tensor, D: torch.Tensor, U: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    """Per bin f: (I - diag(g) D[f] U)^-1 R[:, f] -- a loop whose feedforward path is a per-bin matrix of delays followed
    by per-channel gains, around a constant mixing matrix (system.py:417-424 with fF = Series(Delay((N,N)), parallelGain(N)),
    fB = Matrix: the active-acoustics structure, e8_active_acoustics.py).  D must not require a gradient."""
    if D.requires_grad:
        raise ValueError("solve_scaled_loop: the per-bin factor must not require a gradient (use ops.solve)")
    cd = R.dtype
    conv = lambda t: t if t.dtype == cd else t.to(cd)  # noqa: E731
    return _SolveScaledLoop.apply(conv(g), conv(D), conv(U), R)


def solve_dud(l: Optional[torch.Tensor], U: torch.Tensor, r: Optional[torch.Tensor], R: torch.Tensor) -> torch.Tensor:
    """Per bin f: (I - diag(l[f]) U diag(r[f]))^-1 R[:, f] -- the closed loop of a feedback delay
    network, with the loop matrix kept in factored form (never materialised)."""
    cd = R.dtype
    conv = lambda t: None if t is None else (t if t.dtype == cd else t.to(cd))  # noqa: E731
    return _SolveDUD.apply(conv(l), conv(U), conv(r), R)


# ----------------------------------------------------------------------------- responses
def delay_response(m_int: torch.Tensor, amp: torch.Tensor, nfft: int) -> torch.Tensor:
    """Integer-delay response H[k, ...] = amp[...] * exp(-2 pi i ((k * m[...]) mod nfft) / nfft)
    for the local bin range.  m_int: integer tensor (any shape); amp: real, same shape."""
    dev = _require_gpu(m_int, amp)
    real = _rdtype(amp)
    bin0, m_local = _bin0_arg(nfft)
    shape = tuple(m_int.shape)
    C_ = max(_prod(shape), 1)
    m32 = m_int.to(torch.int32).contiguous()
    amp = amp.contiguous()
    H = _empty_rows(shape, m_local, _cdtype(real), dev)
    L = _lib.lib()
    fn = L.fl_delay_response_c64 if real == torch.float32 else L.fl_delay_response_c128
    _lib.check(fn(m32.data_ptr(), amp.data_ptr(), C_, twiddles(nfft, real, dev).data_ptr(), nfft, bin0, m_local,
                  H.data_ptr(), _pitch(m_local), _stream()), "delay_response")
    return H.movedim(-1, 0)


# float32 modules: evaluate the cascade forward in float (fl_sos_response_f32eval_c64: 3e-7 against 6e-8 of the double
# evaluation rounded once, about twice as fast) where that is safe for the gradients -- always for graphic-equaliser
# sections (benign parameter map: gradient within 2e-7 of the double evaluation's), for raw section coefficients only
# when no gradient is taken: the backward reuses the saved response and a parametric equaliser's map amplifies the extra
# error to 1e-5 .. 1e-4 (tools/dbg/archive/rc_fast_grad.py).  False = double everywhere.
FLOAT_CASCADE_EVAL = True

# float32 modules: mixed-precision backward of the cascade (see fl_sos_response_bwd_c64); False
# forces the all-double evaluation (tests compare the two)
SOS_BWD_MIXED = True


def _sos_forward_launch(bc, ac, gamma, nfft, real, float_eval=False):
    """bc, ac: contiguous float64 (3, S, chan...) on the GPU -> (H rows buffer (chan..., pitch)[..., :m_local], cfg)"""
    dev = bc.device
    S = bc.shape[1]
    chan = tuple(bc.shape[2:])
    C_ = max(_prod(chan), 1)
    bin0, m_local = _bin0_arg(nfft)
    H = _empty_rows(chan, m_local, _cdtype(real), dev)
    L = _lib.lib()
    if real == torch.float32:
        fn = L.fl_sos_response_f32eval_c64 if (float_eval and FLOAT_CASCADE_EVAL) else L.fl_sos_response_c64
    else:
        fn = L.fl_sos_response_c128
    Wd = twiddles(nfft, torch.float64, dev)
    with kernel_timer.span("sos_response"):
        _lib.check(fn(bc.data_ptr(), ac.data_ptr(), S, C_, float(gamma), Wd.data_ptr(), nfft, bin0, m_local,
                      H.data_ptr(), _pitch(m_local), _stream()), "sos_response")
    return H, (float(gamma), nfft, S, C_, bin0, m_local, real)


def _sos_backward_launch(gH, Hf, bc, ac, cfg):
    """-> part: float64 (nblk, 2, 3, S, C) per-bin-block partial sums of (dL/db, dL/da)"""
    gamma, nfft, S, C_, bin0, m_local, real = cfg
    g = _h_planar(gH.resolve_conj(), True)
    g_pitch = _lead_pitch(g.movedim(0, -1))
    L = _lib.lib()
    nblk = L.fl_sos_bwd_blocks(m_local, C_, S, int(real == torch.float32 and Hf is not None))
    part = torch.empty((nblk, 2, 3, S, C_), dtype=torch.float64, device=bc.device)   # every entry is written
    fn = L.fl_sos_response_bwd_c64 if real == torch.float32 else L.fl_sos_response_bwd_c128
    Wd = twiddles(nfft, torch.float64, bc.device)
    with kernel_timer.span("sos_response_bwd"):
        _lib.check(fn(g.data_ptr(), g_pitch, None if Hf is None else Hf.data_ptr(), _pitch(m_local), bc.data_ptr(),
                      ac.data_ptr(), S, C_, gamma, Wd.data_ptr(), nfft, bin0, m_local, part.data_ptr(), _stream()),
                   "sos_response_bwd")
    return part


class _Sos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, a, gamma, nfft, real):
        _require_gpu(b, a)
        if b.shape != a.shape or b.shape[0] != 3 or b.dim() < 2:
            raise ValueError("sos_response: b and a must both be (3, n_sections, ...)")
        if b.dtype != torch.float64 or a.dtype != torch.float64:
            raise TypeError("sos_response: coefficients are passed in float64")
        bc, ac = b.contiguous(), a.contiguous()
        # raw section coefficients: float evaluation only when no gradient will reuse the saved response (see FLOAT_CASCADE_EVAL)
        H, ctx.cfg = _sos_forward_launch(bc, ac, gamma, nfft, real, not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        # the backward reuses the forward output instead of re-evaluating the cascade
        keep = H if (real == torch.float64 or SOS_BWD_MIXED) else None
        ctx.save_for_backward(bc, ac, *([keep] if keep is not None else []))
        return H.movedim(-1, 0)

    @staticmethod
    def backward(ctx, gH):
        bc, ac, *kept = ctx.saved_tensors
        part = _sos_backward_launch(gH, kept[0] if kept else None, bc, ac, ctx.cfg)
        tot = part.sum(dim=0)
        return tot[0].view(bc.shape), tot[1].view(ac.shape), None, None, None


def _geq_in_kind(t: torch.Tensor, linear: bool, sig: bool = False) -> int:
    """fl_geq_sections' in_kind: 0 dB gains; 1 / 2 raw parameters under 20 log10|x| (f64 / f32); 3 / 4 raw parameters under
    20 log10(sigmoid(x))"""
    if not linear:
        return 0
    return (4 if sig else 2) if t.dtype == torch.float32 else (3 if sig else 1)


class _GeqSections(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gain_db, consts):
        dev = _require_gpu(gain_db, consts)
        gd = gain_db.to(torch.float64).contiguous()
        nb = gd.shape[0]
        chan = tuple(gd.shape[1:])
        C_ = max(_prod(chan), 1)
        b = torch.empty((3, nb, *chan), dtype=torch.float64, device=dev)
        a = torch.empty_like(b)
        _lib.check(_lib.lib().fl_geq_sections(gd.data_ptr(), 0, nb, C_, consts.data_ptr(), b.data_ptr(), a.data_ptr(),
                                              _stream()), "geq_sections")
        ctx.save_for_backward(gd, consts)
        ctx.in_dtype = gain_db.dtype
        return b, a

    @staticmethod
    def backward(ctx, gb, ga):
        gd, consts = ctx.saved_tensors
        nb = gd.shape[0]
        C_ = max(_prod(gd.shape[1:]), 1)
        out = torch.empty_like(gd)
        gb = torch.zeros_like(gd.new_empty((3, *gd.shape))) if gb is None else gb.to(torch.float64).contiguous()
        ga = torch.zeros_like(gb) if ga is None else ga.to(torch.float64).contiguous()
        _lib.check(_lib.lib().fl_geq_sections_bwd(gd.data_ptr(), 0, gb.data_ptr(), ga.data_ptr(), 0, 1, nb, C_,
                                                  consts.data_ptr(), out.data_ptr(), _stream()), "geq_sections_bwd")
        return out.to(ctx.in_dtype), None


class _GeqCascade(torch.autograd.Function):
    """GEQ / parallelGEQ under the default map: raw parameters -> response in two launches
    (design + cascade), gradient in two (cascade backward + design backward, which also sums the
    bin-block partials)."""

    @staticmethod
    def forward(ctx, x, consts, gamma, nfft, real, sig=False):
        dev = _require_gpu(x, consts)
        ctx.sig = bool(sig)
        if x.dtype not in (torch.float32, torch.float64):
            raise TypeError("geq_cascade expects float32 / float64 parameters")
        xc = x.contiguous()
        nb = xc.shape[0]
        chan = tuple(xc.shape[1:])
        C_ = max(_prod(chan), 1)
        kind = _geq_in_kind(xc, True, sig)
        b = torch.empty((3, nb, *chan), dtype=torch.float64, device=dev)
        a = torch.empty_like(b)
        if real == torch.float32:
            # design + cascade in one launch (the float kernel designs the sections in its prologue)
            bin0, m_local = _bin0_arg(nfft)
            H = _empty_rows(chan, m_local, torch.complex64, dev)
            with kernel_timer.span("sos_response"):
                _lib.check(_lib.lib().fl_geq_response_c64(xc.data_ptr(), kind, nb, consts.data_ptr(), b.data_ptr(), a.data_ptr(), C_, float(gamma),
                                                          twiddles(nfft, torch.float64, dev).data_ptr(), nfft, bin0, m_local, H.data_ptr(),
                                                          _pitch(m_local), int(bool(FLOAT_CASCADE_EVAL)), _stream()), "geq_response")
            ctx.cfg = (float(gamma), nfft, nb, C_, bin0, m_local, real)
        else:
            _lib.check(_lib.lib().fl_geq_sections(xc.data_ptr(), kind, nb, C_, consts.data_ptr(), b.data_ptr(), a.data_ptr(),
                                                  _stream()), "geq_sections")
            H, ctx.cfg = _sos_forward_launch(b, a, gamma, nfft, real, True)     # graphic-equaliser sections
        keep = H if (real == torch.float64 or SOS_BWD_MIXED) else None
        ctx.save_for_backward(xc, consts, b, a, *([keep] if keep is not None else []))
        return H.movedim(-1, 0)

    @staticmethod
    def backward(ctx, gH):
        xc, consts, b, a, *kept = ctx.saved_tensors
        nbx = _geq_lanes_blocks(ctx.cfg, 1, 0, 0) if kept else 0
        if nbx > 0:
            C_ = ctx.cfg[3]
            out, _ = _geq_backward_lanes(0, gH, kept[0], b, a, None, ctx.cfg, C_, 1, 0, nbx, xc, consts, ctx.sig)
            return out, None, None, None, None, None
        part = _sos_backward_launch(gH, kept[0] if kept else None, b, a, ctx.cfg)
        nblk = part.shape[0]
        nb = xc.shape[0]
        C_ = max(_prod(xc.shape[1:]), 1)
        st = nb * C_
        out = torch.empty_like(xc)
        esz = part.element_size()
        _lib.check(_lib.lib().fl_geq_sections_bwd(xc.data_ptr(), _geq_in_kind(xc, True, ctx.sig), part.data_ptr(),
                                                  part.data_ptr() + 3 * st * esz, 6 * st, nblk, nb, C_, consts.data_ptr(),
                                                  out.data_ptr(), _stream()), "geq_sections_bwd")
        return out, None, None, None, None, None


# ---- cascade response applied to a signal with few columns: Y = H X with dL/dH formed inside the cascade backward
def cascade_apply_supported(real: torch.dtype, X: torch.Tensor) -> bool:
    """float32 modules with the mixed-precision backward, vector signals (B, M, N)"""
    return real == torch.float32 and SOS_BWD_MIXED and X.dim() == 3 and X.dtype == torch.complex64 and X.is_cuda


def _sos_apply_forward(bc, ac, Xp, gamma, nfft, real, float_eval):
    """(H rows buffer (No, Ni, pitch)[..., :m_local], Y planar (B, M, No), cfg): cascade response and its product with the
    signal -- one launch when the float evaluation applies and the coefficient tables fit (fl_sos_response_apply_c64), else
    the response launch followed by the per-bin product"""
    S, No, Ni = bc.shape[1], bc.shape[2], bc.shape[3]
    B = Xp.shape[0]
    L = _lib.lib()
    if not (float_eval and FLOAT_CASCADE_EVAL and real == torch.float32 and B <= 2 and Ni <= L.fl_sos_response_apply_max_ni(S)):
        H, cfg = _sos_forward_launch(bc, ac, gamma, nfft, real, float_eval)
        return H, _mimo_launch(H.movedim(-1, 0), True, False, False, Xp), cfg
    dev = bc.device
    bin0, m_local = _bin0_arg(nfft)
    _, M, Nx, K, xs_b, xs_n, _ = _bnk(Xp)
    assert K == 1 and Nx == Ni and M == m_local
    H = _empty_rows((No, Ni), m_local, torch.complex64, dev)
    Y = _empty_planar((B, m_local, No), torch.complex64, dev)
    _, _, _, _, ys_b, ys_m, _ = _bnk(Y)
    with kernel_timer.span("sos_response"):
        _lib.check(L.fl_sos_response_apply_c64(bc.data_ptr(), ac.data_ptr(), S, No, Ni, Xp.data_ptr(), xs_b, xs_n, B, float(gamma),
                                               twiddles(nfft, torch.float64, dev).data_ptr(), nfft, bin0, m_local, H.data_ptr(),
                                               _pitch(m_local), Y.data_ptr(), ys_b, ys_m, _stream()), "sos_response_apply")
    return H, Y, (float(gamma), nfft, S, No * Ni, bin0, m_local, real)


def _sos_backward_outer_launch(gY, Xp, Hf, bc, ac, cfg, No, Ni):
    """part (nblk, 2, 3, S, C) with dL/dH[m][n] = sum_b gY[b][m] conj(X[b][n]) formed in the kernel"""
    gamma, nfft, S, C_, bin0, m_local, real = cfg
    B, M, _, K, xs_b, xs_n, _ = _bnk(Xp)
    _, _, _, _, gs_b, gs_n, _ = _bnk(gY)
    assert K == 1 and C_ == No * Ni and M == m_local
    L = _lib.lib()
    part = torch.empty((L.fl_sos_bwd_blocks(m_local, C_, S, 1), 2, 3, S, C_), dtype=torch.float64, device=bc.device)
    with kernel_timer.span("sos_response_bwd"):
        _lib.check(L.fl_sos_response_bwd_outer_c64(gY.data_ptr(), gs_b, gs_n, Xp.data_ptr(), xs_b, xs_n, B, No, Ni, Hf.data_ptr(),
                                                   _pitch(m_local), bc.data_ptr(), ac.data_ptr(), S, gamma,
                                                   twiddles(nfft, torch.float64, bc.device).data_ptr(), nfft, bin0, m_local,
                                                   part.data_ptr(), _stream()), "sos_response_bwd_outer")
    return part


class _SosApply(torch.autograd.Function):
    """Y[b,:,f] = sos_response(b, a)[f] X[b,:,f] for a full (N_out, N_in) cascade and a vector signal"""

    @staticmethod
    def forward(ctx, b, a, X, gamma, nfft, real):
        _require_gpu(b, a, X)
        if b.shape != a.shape or b.shape[0] != 3 or b.dim() != 4:
            raise ValueError("sos_response_apply: b and a must both be (3, n_sections, N_out, N_in)")
        bc, ac = b.contiguous(), a.contiguous()
        Xp = to_planar(X.resolve_conj())
        H, Y, ctx.cfg = _sos_apply_forward(bc, ac, Xp, gamma, nfft, real,
                                           not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        ctx.save_for_backward(bc, ac, H, Xp)
        return Y

    @staticmethod
    def backward(ctx, gY):
        bc, ac, H, Xp = ctx.saved_tensors
        gY = to_planar(gY.resolve_conj())
        gb = ga = gX = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            tot = _sos_backward_outer_launch(gY, Xp, H, bc, ac, ctx.cfg, bc.shape[2], bc.shape[3]).sum(dim=0)
            gb, ga = tot[0].view(bc.shape), tot[1].view(ac.shape)
        if ctx.needs_input_grad[2]:
            gX = _mimo_launch(H.movedim(-1, 0), True, False, True, gY)
        return gb, ga, gX, None, None, None


class _GeqCascadeApply(torch.autograd.Function):
    """Y = geq_cascade(x)[f] X[b,:,f]: design + cascade + product forward; cascade backward (outer product formed in the
    kernel) + design backward."""

    @staticmethod
    def forward(ctx, x, consts, X, gamma, nfft, real, sig=False):
        dev = _require_gpu(x, consts, X)
        ctx.sig = bool(sig)
        xc = x.contiguous()
        if xc.dim() != 3:
            raise ValueError("geq_cascade_apply expects full (n_bands, N_out, N_in) parameters")
        nb = xc.shape[0]
        chan = tuple(xc.shape[1:])
        C_ = _prod(chan)
        b = torch.empty((3, nb, *chan), dtype=torch.float64, device=dev)
        a = torch.empty_like(b)
        _lib.check(_lib.lib().fl_geq_sections(xc.data_ptr(), _geq_in_kind(xc, True, sig), nb, C_, consts.data_ptr(), b.data_ptr(),
                                              a.data_ptr(), _stream()), "geq_sections")
        Xp = to_planar(X.resolve_conj())
        H, Y, ctx.cfg = _sos_apply_forward(b, a, Xp, gamma, nfft, real, True)
        ctx.save_for_backward(xc, consts, b, a, H, Xp)
        return Y

    @staticmethod
    def backward(ctx, gY):
        xc, consts, b, a, H, Xp = ctx.saved_tensors
        gY = to_planar(gY.resolve_conj())
        out = gX = None
        if ctx.needs_input_grad[0]:
            part = _sos_backward_outer_launch(gY, Xp, H, b, a, ctx.cfg, xc.shape[1], xc.shape[2])
            nblk, nb = part.shape[0], xc.shape[0]
            C_ = _prod(xc.shape[1:])
            st = nb * C_
            out = torch.empty_like(xc)
            esz = part.element_size()
            _lib.check(_lib.lib().fl_geq_sections_bwd(xc.data_ptr(), _geq_in_kind(xc, True, ctx.sig), part.data_ptr(),
                                                      part.data_ptr() + 3 * st * esz, 6 * st, nblk, nb, C_, consts.data_ptr(),
                                                      out.data_ptr(), _stream()), "geq_sections_bwd")
        if ctx.needs_input_grad[2]:
            gX = _mimo_launch(H.movedim(-1, 0), True, False, True, gY)
        return out, None, gX, None, None, None, None


def sos_response_apply(b, a, X, gamma: float, nfft: int, dtype=torch.float32) -> torch.Tensor:
    """sos_response(b, a)[f] @ X[b, f] for a vector signal X (B, M, N_in) -- dsp.py:922-924 over the cascade tail
    dsp.py:1520-1526: the response's gradient never exists as a tensor (cascade_apply_supported)."""
    return _SosApply.apply(b.to(torch.float64), a.to(torch.float64), X, float(gamma), int(nfft), dtype)


def geq_cascade_apply(x, consts, X, gamma: float, nfft: int, dtype=torch.float32, gain_map: str = "abs") -> torch.Tensor:
    """geq_cascade(x, consts)[f] @ X[b, f] for a vector signal X (B, M, N_in) (dsp.py:922-924 over dsp.py:2573-2593)"""
    return _GeqCascadeApply.apply(x, consts, X, float(gamma), int(nfft), dtype, gain_map == "sigmoid")


def cascade_rc_supported(real: torch.dtype, n_in: int, n_mid: int = 1, n_sections: int = 1) -> bool:
    """cascade response (n_mid cascades of n_sections per output row) times a real constant matrix with n_in columns: the
    fused operator exists -- 2/4/8/16 columns, and fl_sos_response_rc_c64 / _c128's own limits (n_mid <= 32, <= 64 sections
    -- 12 in float64 --, coefficient tables of one output row within the default 64 KB of dynamic LDS: n_mid * 6 * sections
    doubles)."""
    if int(n_in) not in (2, 4, 8, 16):
        return False
    n_mid, n_sections = int(n_mid), int(n_sections)
    if n_mid < 1 or n_mid > 32 or n_sections < 1 or n_sections > 64:
        return False
    if real == torch.float64:      # the all-double kernels: the backward keeps one chunk of at most 12 sections in registers
        return n_sections <= 12 and n_mid * 6 * n_sections * 8 + n_mid * int(n_in) * 8 <= 64 * 1024
    if not (real == torch.float32 and SOS_BWD_MIXED):
        return False
    return n_mid * 6 * n_sections * 8 + n_mid * int(n_in) * 4 <= 64 * 1024


def _cascade_rc_forward(b, a, Wr, gamma, nfft, real, float_eval, geq=None):
    """G = cascade(b, a) (No, Nmid per bin), H = G @ Wr in one launch.  Returns (H view (M, No, Ni), G rows, cfg).
    geq = (raw gains, in_kind, band constants): b, a are OUTPUTS, designed by the same launch (fl_geq_response_rc_c64)."""
    if b.dim() != 4:
        raise ValueError("cascade_rc expects a full (N_out, N_mid) cascade")
    dev = b.device
    S, No, Nmid = b.shape[1], b.shape[2], b.shape[3]
    if Wr.dim() != 2 or Wr.shape[0] != Nmid:
        raise ValueError(f"cascade_rc: the constant factor must be ({Nmid}, N_in), got {tuple(Wr.shape)}")
    Ni = Wr.shape[1]
    bin0, m_local = _bin0_arg(nfft)
    f64 = real == torch.float64
    G = _empty_rows((No, Nmid), m_local, _cdtype(real), dev)
    H = _empty_rows((No, Ni), m_local, _cdtype(real), dev)
    Wc = Wr.contiguous()
    P = _pitch(m_local)
    with kernel_timer.span("sos_response_rc"):
        if f64 and geq is not None:
            xc, kind, consts = geq
            _lib.check(_lib.lib().fl_geq_response_rc_c128(xc.data_ptr(), kind, S, consts.data_ptr(), b.data_ptr(), a.data_ptr(), No, Nmid, Ni,
                                                          Wc.data_ptr(), float(gamma), twiddles(nfft, torch.float64, dev).data_ptr(), nfft,
                                                          bin0, m_local, G.data_ptr(), P, H.data_ptr(), P, _stream()), "geq_response_rc")
        elif f64:
            _lib.check(_lib.lib().fl_sos_response_rc_c128(b.data_ptr(), a.data_ptr(), S, No, Nmid, Ni, Wc.data_ptr(), float(gamma),
                                                          twiddles(nfft, torch.float64, dev).data_ptr(), nfft, bin0, m_local,
                                                          G.data_ptr(), P, H.data_ptr(), P, _stream()), "sos_response_rc")
        elif geq is not None:
            xc, kind, consts = geq
            _lib.check(_lib.lib().fl_geq_response_rc_c64(xc.data_ptr(), kind, S, consts.data_ptr(), b.data_ptr(), a.data_ptr(), No, Nmid, Ni,
                                                         Wc.data_ptr(), float(gamma), twiddles(nfft, torch.float64, dev).data_ptr(), nfft,
                                                         bin0, m_local, G.data_ptr(), P, H.data_ptr(), P,
                                                         int(bool(float_eval and FLOAT_CASCADE_EVAL)), _stream()), "geq_response_rc")
        else:
            _lib.check(_lib.lib().fl_sos_response_rc_c64(b.data_ptr(), a.data_ptr(), S, No, Nmid, Ni, Wc.data_ptr(), float(gamma),
                                                         twiddles(nfft, torch.float64, dev).data_ptr(), nfft, bin0, m_local,
                                                         G.data_ptr(), P, H.data_ptr(), P, int(bool(float_eval and FLOAT_CASCADE_EVAL)),
                                                         _stream()), "sos_response_rc")
    if kernel_timer.enabled and _lib.lib(pair_ok=True).fl_launch_pair_pending():
        kernel_timer.drop_last("sos_response_rc")       # recorded, not issued: it rides in the input's column pass
    return H.movedim(-1, 0), G, (float(gamma), nfft, S, No * Nmid, bin0, m_local, real)


def _cascade_rc_backward(gH, G, b, a, Wr, cfg):
    """-> (part: float64 (nblk, 2, 3, S, C), partW: (nblk, No, Nmid, Ni) in the response's real dtype)"""
    gamma, nfft, S, C_, bin0, m_local, real = cfg
    No, Nmid = G.shape[0], G.shape[1]
    Ni = Wr.shape[1]
    g = _h_planar(gH.resolve_conj(), True)
    L = _lib.lib()
    nblk = L.fl_sos_bwd_blocks(m_local, C_, S, 0 if real == torch.float64 else 1)
    part = torch.empty((nblk, 2, 3, S, C_), dtype=torch.float64, device=b.device)
    partW = torch.empty((nblk, No, Nmid, Ni), dtype=real, device=b.device)
    Wc = Wr.contiguous()
    fn = L.fl_sos_response_bwd_rc_c128 if real == torch.float64 else L.fl_sos_response_bwd_rc_c64
    with kernel_timer.span("sos_response_bwd_rc"):
        _lib.check(fn(g.data_ptr(), _lead_pitch(g.movedim(0, -1)), G.data_ptr(), _pitch(m_local),
                                                b.data_ptr(), a.data_ptr(), S, No, Nmid, Ni, Wc.data_ptr(), gamma,
                                                twiddles(nfft, torch.float64, b.device).data_ptr(), nfft, bin0, m_local,
                                                part.data_ptr(), partW.data_ptr(), _stream()), "sos_response_bwd_rc")
    return part, partW


def _geq_lanes_blocks(cfg, ppr: int, niw: int, mode: int) -> int:
    """bin blocks of the lanes-per-section backward (csrc/cascade2.hip) for this shape, 0 when it does not take it"""
    gamma, nfft, S, C_, bin0, m_local, real = cfg
    if not SOS_BWD_MIXED:
        return 0
    fn = _lib.lib().fl_geq_bwd_lanes_blocks if real == torch.float32 else _lib.lib().fl_geq_bwd_lanes_blocks_f64
    return int(fn(m_local, C_, S, nfft, bin0, int(ppr), int(niw), int(mode)))


def _geq_backward_lanes(mode, gH, G, b, a, Wr, cfg, No, Nmid, Ni, nbx, xc, consts, sig):
    """graphic equaliser, float32 / float64: cascade backward with one lane per (pair, section) + the launch that reduces its block
    partials and runs the design's backward.  -> (dL/dx in x's dtype, dL/dWr or None)"""
    gamma, nfft, S, C_, bin0, m_local, real = cfg
    dev = b.device
    g = _h_planar(gH.resolve_conj(), True)
    L = _lib.lib()
    f64 = real == torch.float64
    psum = torch.empty((S * C_, nbx, 4), dtype=real, device=dev)      # (band 0's entries stay unwritten and unread: closed form from pq)
    pq = torch.empty((C_, nbx), dtype=real, device=dev)
    # per workgroup (bin block x pair group) ONE (Nmid, Ni) matrix, summed over the group's output channels already
    wrows = (L.fl_geq_bwd_lanes_wrows_f64 if f64 else L.fl_geq_bwd_lanes_wrows)(m_local, C_, S, nfft, bin0, Nmid, Ni) if mode == 1 else 0
    partW = torch.empty((Nmid * Ni, wrows), dtype=real, device=dev) if mode == 1 else None
    Wc = Wr.to(real).contiguous() if mode == 1 else None
    with kernel_timer.span("sos_response_bwd_rc" if mode == 1 else "sos_response_bwd"):
        _lib.check((L.fl_geq_response_bwd_lanes_c128 if f64 else L.fl_geq_response_bwd_lanes_c64)(
            mode, g.data_ptr(), _lead_pitch(g.movedim(0, -1)), G.data_ptr(), _pitch(m_local), b.data_ptr(), a.data_ptr(), S, No, Nmid, Ni,
            None if Wc is None else Wc.data_ptr(), gamma, twiddles(nfft, torch.float64, dev).data_ptr(), nfft, bin0, m_local,
            psum.data_ptr(), pq.data_ptr(), None if partW is None else partW.data_ptr(), _stream()), "geq_response_bwd_lanes")
    out = torch.empty_like(xc)
    gW = torch.empty(Wr.shape, dtype=real, device=dev) if mode == 1 else None
    _lib.check((L.fl_geq_sections_bwd_lanes_f64 if f64 else L.fl_geq_sections_bwd_lanes)(
        xc.data_ptr(), _geq_in_kind(xc, True, sig), psum.data_ptr(), pq.data_ptr(), nbx, b.data_ptr(), a.data_ptr(), gamma, S, C_,
        consts.data_ptr(), out.data_ptr(), None if partW is None else partW.data_ptr(), wrows, Nmid * Ni if mode == 1 else 0,
        None if gW is None else gW.data_ptr(), _stream()), "geq_sections_bwd_lanes")
    if gW is not None and gW.dtype != Wr.dtype:
        gW = gW.to(Wr.dtype)
    return out, gW


class _SosRC(torch.autograd.Function):
    """sos_response(b, a) @ Wr with the composition's backward folded into the cascade's backward kernel"""

    @staticmethod
    def forward(ctx, b, a, Wr, gamma, nfft, real):
        _require_gpu(b, a, Wr)
        if b.shape != a.shape or b.shape[0] != 3 or b.dim() != 4:
            raise ValueError("sos_response_rc: b and a must both be (3, n_sections, N_out, N_mid)")
        bc, ac = b.contiguous(), a.contiguous()
        H, G, ctx.cfg = _cascade_rc_forward(bc, ac, Wr, gamma, nfft, real,
                                            not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        ctx.save_for_backward(bc, ac, G, Wr)
        return H

    @staticmethod
    def backward(ctx, gH):
        bc, ac, G, Wr = ctx.saved_tensors
        part, partW = _cascade_rc_backward(gH, G, bc, ac, Wr, ctx.cfg)
        tot = part.sum(dim=0)
        return tot[0].view(bc.shape), tot[1].view(ac.shape), partW.sum(dim=(0, 1)).to(Wr.dtype), None, None, None


class _GeqCascadeRC(torch.autograd.Function):
    """geq_cascade(x) @ Wr: design + cascade + product forward; cascade backward (with the composition folded in) +
    design backward."""

    @staticmethod
    def forward(ctx, x, consts, Wr, gamma, nfft, real, sig=False):
        dev = _require_gpu(x, consts, Wr)
        ctx.sig = bool(sig)
        xc = x.contiguous()
        nb = xc.shape[0]
        chan = tuple(xc.shape[1:])
        C_ = max(_prod(chan), 1)
        b = torch.empty((3, nb, *chan), dtype=torch.float64, device=dev)
        a = torch.empty_like(b)
        # the sections are designed in the response kernel's prologue (and written to b, a for the backward pass)
        H, G, ctx.cfg = _cascade_rc_forward(b, a, Wr, gamma, nfft, real, True, geq=(xc, _geq_in_kind(xc, True, sig), consts))
        ctx.save_for_backward(xc, consts, b, a, G, Wr)
        return H

    @staticmethod
    def backward(ctx, gH):
        xc, consts, b, a, G, Wr = ctx.saved_tensors
        nbx = _geq_lanes_blocks(ctx.cfg, G.shape[1], Wr.shape[1], 1)
        if nbx > 0:
            out, gW = _geq_backward_lanes(1, gH, G, b, a, Wr, ctx.cfg, G.shape[0], G.shape[1], Wr.shape[1], nbx, xc, consts, ctx.sig)
            return out, None, gW, None, None, None, None
        part, partW = _cascade_rc_backward(gH, G, b, a, Wr, ctx.cfg)
        nblk = part.shape[0]
        nb = xc.shape[0]
        C_ = max(_prod(xc.shape[1:]), 1)
        st = nb * C_
        out = torch.empty_like(xc)
        gW = torch.empty_like(Wr, memory_format=torch.contiguous_format)
        esz = part.element_size()
        # design backward (sums the bin-block partials) + the constant factor's partials, one launch
        fn = _lib.lib().fl_geq_sections_bwd_w64 if partW.dtype == torch.float64 else _lib.lib().fl_geq_sections_bwd_w
        _lib.check(fn(xc.data_ptr(), _geq_in_kind(xc, True, ctx.sig), part.data_ptr(),
                      part.data_ptr() + 3 * st * esz, 6 * st, nblk, nb, C_, consts.data_ptr(),
                      out.data_ptr(), partW.data_ptr(), partW.shape[0] * partW.shape[1],
                      partW.shape[2] * partW.shape[3], gW.data_ptr(), _stream()), "geq_sections_bwd_w")
        return out, None, gW, None, None, None, None


def sos_response_rc(b, a, Wr, gamma: float, nfft: int, dtype=torch.float32) -> torch.Tensor:
    """sos_response(b, a) (M, N_out, N_mid) times the real constant matrix Wr (N_mid, N_in) on the right, per bin."""
    return _SosRC.apply(b.to(torch.float64), a.to(torch.float64), Wr.to(dtype), float(gamma), int(nfft), dtype)


def geq_cascade_rc(x, consts, Wr, gamma: float, nfft: int, dtype=torch.float32, gain_map: str = "abs") -> torch.Tensor:
    """geq_cascade(x, ...) (M, N_out, N_mid) times the real constant matrix Wr (N_mid, N_in) on the right, per bin."""
    return _GeqCascadeRC.apply(x, consts, Wr.to(dtype), float(gamma), int(nfft), dtype, _is_sigmoid(gain_map))


def _is_sigmoid(gain_map: str) -> bool:
    if gain_map not in ("abs", "sigmoid"):
        raise ValueError("gain_map: 'abs' (20 log10|x|, the module default) or 'sigmoid' (20 log10(sigmoid(x)))")
    return gain_map == "sigmoid"


def geq_cascade(x: torch.Tensor, consts: torch.Tensor, gamma: float, nfft: int, dtype=torch.float32, gain_map: str = "abs") -> torch.Tensor:
    """Response (M, ...) of the graphic equaliser whose raw parameters x (n_bands, ...) go through
    the default map 20 log10|x| -- same result as sos_response(*geq_sections(20 log10|x|)), with
    the map and its backward folded into the design kernels.  gain_map="sigmoid": the map 20 log10(sigmoid(x)) of the
    FDN attenuation filters (e8_fdn.py:97) folded the same way."""
    return _GeqCascade.apply(x, consts, float(gamma), int(nfft), dtype, _is_sigmoid(gain_map))


def geq_sections(gain_db: torch.Tensor, consts: torch.Tensor):
    """Command gains in dB (n_bands, ...) -> (b, a), each (3, n_bands, ...) float64 tensors holding
    the float32-rounded sections of the graphic equaliser (one fused kernel each way)."""
    return _GeqSections.apply(gain_db, consts)


def sos_response(b: torch.Tensor, a: torch.Tensor, gamma: float, nfft: int, dtype=torch.float32) -> torch.Tensor:
    """H[k, ...] = prod_s B_s(k) / prod_s A_s(k) of a cascade of second-order sections with
    coefficients b, a: (3, n_sections, ...) real (anti-alias radius gamma applied to the taps).
    The cascade is evaluated in float64 (coefficients are promoted); ``dtype`` (float32 |
    float64) selects the precision H is stored in."""
    return _Sos.apply(b.to(torch.float64), a.to(torch.float64), float(gamma), int(nfft), dtype)


# ----------------------------------------------------------------------------- scalar objective
_ms_scratch = {}


def _ms_scratch_for(dev: torch.device) -> torch.Tensor:
    """Per-block partials, one buffer per (device, stream)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    s = _ms_scratch.get(key)
    if s is None:
        s = torch.empty(_lib.lib().fl_mean_square_scratch_bytes(), dtype=torch.uint8, device=dev)
        _ms_scratch[key] = s
    return s


def _rows_of(y: torch.Tensor):
    """(tensor with the same memory, rows, cols, pitch): y's memory as rows of `cols` contiguous
    elements `pitch` apart.  Works for contiguous tensors and for the signal-planar views the
    transforms return; anything else is made contiguous first."""
    if y.is_contiguous():
        return y, 1, y.numel(), y.numel()
    if y.dim() >= 2:
        mem = y.movedim(1, -1)
        P = _lead_pitch(mem)
        if P is not None:
            return y, _prod(mem.shape[:-1]), mem.shape[-1], P
    yc = y.contiguous()
    return yc, 1, yc.numel(), yc.numel()


class _MeanSquare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        dev = _require_gpu(y)
        if y.dtype not in (torch.float32, torch.float64):
            raise TypeError("mean_square expects a real float32/float64 tensor")
        if y.numel() == 0:
            raise ValueError("mean_square of an empty tensor")
        ym, rows, cols, pitch = _rows_of(y)
        loss = torch.empty((), dtype=y.dtype, device=dev)
        L = _lib.lib()
        fn = L.fl_mean_square_f32 if y.dtype == torch.float32 else L.fl_mean_square_f64
        with kernel_timer.span("mean_square"):
            _lib.check(fn(ym.data_ptr(), rows, cols, pitch, loss.data_ptr(), _ms_scratch_for(dev).data_ptr(), _stream()),
                       "mean_square")
        ctx.save_for_backward(ym)
        ctx.layout = (rows, cols, pitch)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (ym,) = ctx.saved_tensors
        rows, cols, pitch = ctx.layout
        gy = torch.empty_strided(ym.shape, ym.stride(), dtype=ym.dtype, device=ym.device)
        g = gloss.to(ym.dtype).contiguous()
        L = _lib.lib()
        fn = L.fl_mean_square_bwd_f32 if ym.dtype == torch.float32 else L.fl_mean_square_bwd_f64
        with kernel_timer.span("mean_square_bwd"):
            _lib.check(fn(ym.data_ptr(), g.data_ptr(), gy.data_ptr(), rows, cols, pitch, _stream()), "mean_square_bwd")
        return gy


def _is_pipeline_output(y: torch.Tensor) -> bool:
    """y still IS what _SpectralApply returned to autograd: its grad_fn is that node (a y produced under no_grad has none),
    no tensor hook, no retain_grad -- otherwise the fused objective would bypass something the caller can observe"""
    fn = y.grad_fn
    if fn is None or type(fn).__name__ != "_SpectralApplyBackward":
        return False
    if getattr(y, "retains_grad", False) or getattr(y, "_backward_hooks", None):
        return False
    return True


class _CAbs(torch.autograd.Function):
    """|z| of a complex tensor in one launch each way; the result has z's memory order (a bin-planar view stays one)"""

    @staticmethod
    def forward(ctx, z):
        dev = _require_gpu(z)
        z = z.resolve_conj()        # a lazily conjugated view: the kernels read the storage (the backward would return conj's gradient)
        zm, rows, cols, pitch = _rows_of(z)
        real = _rdtype(zm)
        if zm.is_contiguous():
            out = torch.empty(zm.shape, dtype=real, device=dev)
        else:
            mem = zm.movedim(1, -1)
            out = torch.empty_strided(mem.shape, mem.stride(), dtype=real, device=dev).movedim(-1, 1)
        L = _lib.lib()
        fn = L.fl_cabs_c64 if real == torch.float32 else L.fl_cabs_c128
        _lib.check(fn(zm.data_ptr(), out.data_ptr(), rows, cols, pitch, pitch, _stream()), "cabs")
        ctx.save_for_backward(zm)
        ctx.layout = (rows, cols, pitch)
        return out

    @staticmethod
    def backward(ctx, g):
        (zm,) = ctx.saved_tensors
        rows, cols, pitch = ctx.layout
        # the cotangent in the output's own memory order (same strides: rows `pitch` apart), else through a copy of that shape
        if zm.is_contiguous():
            gm = g.contiguous()
            gpitch = pitch
        else:
            mem = zm.movedim(1, -1)
            gv = g.movedim(1, -1)
            if tuple(gv.stride()) == tuple(mem.stride()):
                gm = g
            else:
                gm = torch.empty_strided(mem.shape, mem.stride(), dtype=g.dtype, device=g.device).movedim(-1, 1)
                gm.copy_(g)
            gpitch = pitch
        gz = torch.empty_strided(zm.shape, zm.stride(), dtype=zm.dtype, device=zm.device)
        L = _lib.lib()
        fn = L.fl_cabs_bwd_c64 if zm.dtype == torch.complex64 else L.fl_cabs_bwd_c128
        _lib.check(fn(zm.data_ptr(), gm.data_ptr(), gz.data_ptr(), rows, cols, pitch, gpitch, _stream()), "cabs_bwd")
        return gz


def cabs(z: torch.Tensor) -> torch.Tensor:
    """torch.abs of a complex device tensor (the magnitude output layer, dsp.py:27-66 wrapping `lambda x: torch.abs(x)`) in one
    launch each way."""
    if not z.is_complex() or z.dtype not in (torch.complex64, torch.complex128) or z.numel() == 0:
        raise ValueError("cabs: a non-empty complex64 / complex128 tensor")
    return _CAbs.apply(z)


class _Sparsity(torch.autograd.Function):
    """mean_c (sum |A_c| - N sqrt N) / (N (1 - sqrt N)) of (C, N, N) real matrices, one launch each way"""

    @staticmethod
    def forward(ctx, A):
        dev = _require_gpu(A)
        Ac = A.contiguous()
        C, N = (1 if Ac.dim() == 2 else Ac.shape[0]), Ac.shape[-1]
        loss = torch.empty((), dtype=A.dtype, device=dev)
        L = _lib.lib()
        fn = L.fl_sparsity_f32 if A.dtype == torch.float32 else L.fl_sparsity_f64
        _lib.check(fn(Ac.data_ptr(), C, N, loss.data_ptr(), _stream()), "sparsity")
        ctx.save_for_backward(Ac)
        ctx.cn = (C, N)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (Ac,) = ctx.saved_tensors
        C, N = ctx.cn
        gA = torch.empty_like(Ac)
        g = gloss.to(Ac.dtype).contiguous()
        L = _lib.lib()
        fn = L.fl_sparsity_bwd_f32 if Ac.dtype == torch.float32 else L.fl_sparsity_bwd_f64
        _lib.check(fn(Ac.data_ptr(), g.data_ptr(), C, N, gA.data_ptr(), _stream()), "sparsity_bwd")
        return gA


def sparsity(A: torch.Tensor) -> torch.Tensor:
    """The reference's sparsity criterion on a mixing matrix A (N, N) or a stack (C, N, N) -- flamo/optimize/loss.py:52-63:
    -(sum|A| - N sqrt N) / (N (sqrt N - 1)), the mean over C of the same expression for a stack."""
    if A.dim() not in (2, 3) or A.shape[-1] != A.shape[-2] or A.shape[-1] < 2 or A.dtype not in (torch.float32, torch.float64):
        raise ValueError("sparsity: a real (N, N) or (C, N, N) matrix with N >= 2")
    return _Sparsity.apply(A)


class _MSE(torch.autograd.Function):
    """mean((sum_c y[..., c] - t) ** 2) (ncols = 1: plain nn.MSELoss) in one streaming pass each way"""

    @staticmethod
    def forward(ctx, y, t, ncols):
        dev = _require_gpu(y, t)
        yc, tc = y.contiguous(), t.contiguous()
        rows = yc.numel() // ncols
        loss = torch.empty((), dtype=y.dtype, device=dev)
        L = _lib.lib()
        fn = L.fl_mse_f32 if y.dtype == torch.float32 else L.fl_mse_f64
        with kernel_timer.span("mse"):
            _lib.check(fn(yc.data_ptr(), tc.data_ptr(), rows, ncols, loss.data_ptr(), _ms_scratch_for(dev).data_ptr(), _stream()), "mse")
        ctx.save_for_backward(yc, tc)
        ctx.cfg = (rows, ncols, tuple(y.shape))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        yc, tc = ctx.saved_tensors
        rows, ncols, shape = ctx.cfg
        gy = torch.empty_like(yc)
        g = gloss.to(yc.dtype).contiguous()
        L = _lib.lib()
        fn = L.fl_mse_bwd_f32 if yc.dtype == torch.float32 else L.fl_mse_bwd_f64
        with kernel_timer.span("mse_bwd"):
            _lib.check(fn(yc.data_ptr(), tc.data_ptr(), g.data_ptr(), gy.data_ptr(), rows, ncols, _stream()), "mse_bwd")
        return gy.view(shape), None, None


MSE_MAX_COLS = 4096      # fl_mse_*: columns summed per row (reduce.hip)


def mse(y: torch.Tensor, target: torch.Tensor, sum_last: bool = False) -> torch.Tensor:
    """nn.MSELoss()(y, target) (examples/e7_biquad.py:82) -- or, with sum_last, the reference's criterion
    flamo/optimize/loss.py:101-102: nn.MSELoss()(y.sum(-1), target.squeeze(-1)) -- as one streaming pass each way.
    The target takes no gradient (a constant of the data set, trainer.py:172-175)."""
    if y.dtype not in (torch.float32, torch.float64) or not y.is_cuda:
        raise TypeError("mse expects a real float32 / float64 tensor on the GPU")
    if y.numel() == 0:
        raise ValueError("mse of an empty tensor")
    t = target.detach()
    ncols = 1
    if sum_last:
        ncols = y.shape[-1]
        if ncols > MSE_MAX_COLS:
            raise ValueError(f"mse: at most {MSE_MAX_COLS} summed columns, got {ncols}")
        if t.dim() == y.dim() and t.shape[-1] == 1:
            t = t.squeeze(-1)
        if tuple(t.shape) != tuple(y.shape[:-1]):
            raise ValueError(f"mse: the target must be {tuple(y.shape[:-1])} (or with a trailing 1), got {tuple(target.shape)}")
    elif tuple(t.shape) != tuple(y.shape):
        raise ValueError(f"mse: prediction {tuple(y.shape)} and target {tuple(target.shape)} differ in shape")
    return _MSE.apply(y, t.to(dtype=y.dtype, device=y.device), int(ncols))


def mean_square(y: torch.Tensor) -> torch.Tensor:
    """(y ** 2).mean() of a real tensor in one streaming pass each way (forward: one read of y;
    backward: one read + one write), in whatever layout y is stored."""
    tag = getattr(y, "_flamo_sa", None)
    if FUSE_OBJECTIVE and tag is not None and tag.parts is not None and y._version == tag.version:
        if torch.is_grad_enabled() and y.requires_grad and _is_pipeline_output(y):
            # y is the pipeline's own differentiable output, nobody has hung a hook on it or asked to keep its gradient:
            # the one-node form is indistinguishable from (y ** 2).mean() through y's node
            if (tag.x.requires_grad or tag.Hrm.requires_grad) and (tag.Xs is not None or not tag.Hrm.requires_grad):
                return _SpectralMeanSquare.apply(tag.x, tag.Hrm, y.detach(), tag)
        elif not (torch.is_grad_enabled() and y.requires_grad):
            # nothing to differentiate (evaluation under no_grad, validation steps): the value alone, from the partial sums
            loss = torch.empty((), dtype=y.dtype, device=y.device)
            fn = _lib.lib().fl_mean_square_final_f32 if y.dtype == torch.float32 else _lib.lib().fl_mean_square_final_f64
            with kernel_timer.span("mean_square_final"):
                _lib.check(fn(tag.parts.data_ptr(), tag.parts.numel(), 1.0 / y.numel(), loss.data_ptr(), _stream()), "mean_square_final")
            return loss
    return _MeanSquare.apply(y)


FUSE_OBJECTIVE = True      # mean_square(spectral_apply(...)) as one node (see _SpectralMeanSquare); False: always the two streaming passes


# ----------------------------------------------------------------------------- orthogonal parameter map
EXPM_MAX_N = 64


class _MatrixExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, skew, cplx=False):
        dev = _require_gpu(X)
        if X.dim() != 2 or X.shape[0] != X.shape[1]:
            raise ValueError("matrix_exp expects one square matrix")
        if X.dtype not in (torch.float32, torch.float64):
            raise TypeError("matrix_exp expects a float32 / float64 matrix")
        N = X.shape[0]
        if N > EXPM_MAX_N:
            raise ValueError(f"matrix_exp: N={N} exceeds the single-workgroup limit {EXPM_MAX_N}")
        Xc = X.contiguous()
        L = _lib.lib()
        E = torch.empty((N, N), dtype=_cdtype(X.dtype) if cplx else X.dtype, device=dev)
        stash = torch.empty(L.fl_matrix_exp_stash_elems(N), dtype=torch.float64, device=dev)
        if cplx:
            fn = L.fl_matrix_exp_cplx_f32 if X.dtype == torch.float32 else L.fl_matrix_exp_cplx_f64
        else:
            fn = L.fl_matrix_exp_f32 if X.dtype == torch.float32 else L.fl_matrix_exp_f64
        _lib.check(fn(Xc.data_ptr(), N, int(skew), E.data_ptr(), stash.data_ptr(), _stream()), "matrix_exp")
        ctx.save_for_backward(stash)
        ctx.cfg = (N, int(skew), X.dtype, bool(cplx))
        return E

    @staticmethod
    def backward(ctx, gE):
        (stash,) = ctx.saved_tensors
        N, skew, dt, cplx = ctx.cfg
        L = _lib.lib()
        if cplx:      # the kernel takes the real part of the complex gradient
            g = gE.resolve_conj().to(_cdtype(dt)).contiguous()
            fn = L.fl_matrix_exp_bwd_cplx_f32 if dt == torch.float32 else L.fl_matrix_exp_bwd_cplx_f64
        else:
            g = gE.to(dt).contiguous()
            fn = L.fl_matrix_exp_bwd_f32 if dt == torch.float32 else L.fl_matrix_exp_bwd_f64
        gX = torch.empty((N, N), dtype=dt, device=g.device)
        _lib.check(fn(g.data_ptr(), N, skew, stash.data_ptr(), gX.data_ptr(), _stream()), "matrix_exp_bwd")
        return gX, None, None


class _MatrixExpBoth(torch.autograd.Function):
    """(E real, E as the complex matrix (re, 0)) from one launch; one backward launch for both gradients"""

    @staticmethod
    def forward(ctx, X, skew):
        dev = _require_gpu(X)
        if X.dim() != 2 or X.shape[0] != X.shape[1] or X.dtype not in (torch.float32, torch.float64):
            raise ValueError("matrix_exp expects one square float32 / float64 matrix")
        N = X.shape[0]
        if N > EXPM_MAX_N:
            raise ValueError(f"matrix_exp: N={N} exceeds the single-workgroup limit {EXPM_MAX_N}")
        Xc = X.contiguous()
        L = _lib.lib()
        E = torch.empty((N, N), dtype=X.dtype, device=dev)
        Ec = torch.empty((N, N), dtype=_cdtype(X.dtype), device=dev)
        stash = torch.empty(L.fl_matrix_exp_stash_elems(N), dtype=torch.float64, device=dev)
        fn = L.fl_matrix_exp_both_f32 if X.dtype == torch.float32 else L.fl_matrix_exp_both_f64
        _lib.check(fn(Xc.data_ptr(), N, int(skew), E.data_ptr(), Ec.data_ptr(), stash.data_ptr(), _stream()), "matrix_exp")
        ctx.save_for_backward(stash)
        ctx.cfg = (N, int(skew), X.dtype)
        ctx.set_materialize_grads(False)        # an unused form's gradient arrives as None, not as a zero-filled tensor
        return E, Ec

    @staticmethod
    def backward(ctx, gE, gEc):
        (stash,) = ctx.saved_tensors
        N, skew, dt = ctx.cfg
        L = _lib.lib()
        if gE is None and gEc is None:
            return None, None
        g = None if gE is None else gE.to(dt).contiguous()
        gc = None if gEc is None else gEc.resolve_conj().to(_cdtype(dt)).contiguous()
        gX = torch.empty((N, N), dtype=dt, device=stash.device)
        fn = L.fl_matrix_exp_bwd_both_f32 if dt == torch.float32 else L.fl_matrix_exp_bwd_both_f64
        _lib.check(fn(None if g is None else g.data_ptr(), None if gc is None else gc.data_ptr(), N, skew, stash.data_ptr(),
                      gX.data_ptr(), _stream()), "matrix_exp_bwd")
        return gX, None


def matrix_exp_both(X: torch.Tensor, skew: bool = False):
    """(exp as a real matrix, the same as a complex matrix) from one launch each way -- for a step in which the model
    takes the complex form (dsp.py:649 then the cast of dsp.py:466-468) and a criterion the real one
    (optimize/loss.py:36-63)."""
    return _MatrixExpBoth.apply(X, bool(skew))


def matrix_exp(X: torch.Tensor, skew: bool = False, complex_out: bool = False) -> torch.Tensor:
    """exp(X), or exp(triu(X,1) - triu(X,1)^T) with skew=True, of one (N, N) parameter matrix
    (N <= 64) in one launch each way: float64 arithmetic, fixed scaling-and-squaring schedule, no
    host synchronisation (capturable).  ``complex_out``: the result as the complex matrix (re, 0) the per-bin kernels
    take, written by the same launch (and the real part of a complex gradient read by the backward launch)."""
    return _MatrixExp.apply(X, bool(skew), bool(complex_out))


# ----------------------------------------------------------------------------- per-bin eigenvalues
_eig_state = {"info": None}


def eigvals_info() -> Optional[torch.Tensor]:
    """int32 (bins,) convergence flags of the most recent ops.eigvals call (0 = converged; k > 0: the QR
    iteration gave up with k eigenvalues of that matrix unconverged).  A device tensor: reading it
    synchronises, which is why eigvals itself does not."""
    return _eig_state["info"]


class _Eigvals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A):
        dev = _require_gpu(A)
        if not A.is_complex():
            raise TypeError("eigvals expects a complex tensor")
        if A.dim() < 2 or A.shape[-1] != A.shape[-2]:
            raise ValueError("eigvals expects (..., N, N)")
        N = A.shape[-1]
        lead = tuple(A.shape[:-2])
        Mt = max(_prod(lead), 1) if lead else 1
        Ap = _h_planar(A.resolve_conj().reshape(Mt, N, N), True)          # (Mt, N, N) with the bin axis contiguous
        a_pitch = _lead_pitch(Ap.movedim(0, -1))
        need_v = ctx.needs_input_grad[0]
        lam = _empty_rows((N,), Mt, A.dtype, dev)
        Vv = _empty_rows((N, N), Mt, A.dtype, dev) if need_v else None
        info = torch.empty(Mt, dtype=torch.int32, device=dev)
        L = _lib.lib()
        fn = L.fl_eig_c64 if A.dtype == torch.complex64 else L.fl_eig_c128
        _lib.check(fn(Ap.data_ptr(), a_pitch, N, Mt, lam.data_ptr(), _pitch(Mt), None if Vv is None else Vv.data_ptr(),
                      _pitch(Mt), info.data_ptr(), _stream()), "eig")
        _eig_state["info"] = info
        if need_v:
            ctx.save_for_backward(Vv)
        ctx.meta = (lead, N, Mt)
        return lam.movedim(-1, 0).reshape(*lead, N)

    @staticmethod
    def backward(ctx, g):
        (Vv,) = ctx.saved_tensors                                          # planar (N, N, Mt): V[i, j, f]
        lead, N, Mt = ctx.meta
        V = Vv.movedim(-1, 0)                                              # (Mt, N, N), bin-planar view
        # g_A = V^-H diag(g) V^H  (eigenvector gradients are zero): solve V^H X = diag(g) V^H per bin
        R = g.resolve_conj().reshape(Mt, N, 1) * V.mH                      # (Mt, N, N)
        X = _solve_launch(V, False, True, to_planar(R.unsqueeze(0)))[0]      # (V^H)^-1 R, LU per bin
        return X.reshape(*lead, N, N)


def eigvals(A: torch.Tensor) -> torch.Tensor:
    """Eigenvalues of the (..., N, N) complex matrices, N <= 64, one wavefront per matrix (Hessenberg + shifted
    QR in LDS).  Order: position on the diagonal of the Schur form, not LAPACK's -- use order-independent
    functions of the result.  Differentiable (g_A = V^-H diag(g) V^H) for simple eigenvalues."""
    return _Eigvals.apply(A)
