"""Where the Matrix-then-cascade forward's time goes: graphic-equaliser entry (design in the prologue) against the raw-section
entry (taps read), and against a 2-section cascade (prologue + stores only)."""
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flamo_amd import _lib, ops
from flamo_amd.processor import dsp

N, nfft = 8, 96000
dev = torch.device("cuda:0")
torch.manual_seed(1)
L = _lib.lib()
geq = dsp.GEQ(size=(N, N), nfft=nfft, alias_decay_db=0.0, device=dev, dtype=torch.float32)
W = torch.randn(N, N, device=dev)
S = 12
spec = geq._cascade_spec(geq.param)
xc, consts = spec[1].contiguous(), spec[2]
b = torch.empty((3, S, N, N), dtype=torch.float64, device=dev)
a = torch.empty_like(b)
with ops.row_major_bins(nfft):
    bin0, m_local = ops._bin0_arg(nfft)
P = ops._pitch(m_local)
G = ops._empty_rows((N, N), m_local, torch.complex64, dev)
H = ops._empty_rows((N, N), m_local, torch.complex64, dev)
Wd = ops.twiddles(nfft, torch.float64, dev)
st = ops._stream()
gamma = float(geq._gamma_f)


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


kind = ops._geq_in_kind(xc, True, False)
for on in (3, 1):
    L.fl_debug_set_cascade_lanes(on, -1, -1)
    f_geq = lambda: _lib.check(L.fl_geq_response_rc_c64(xc.data_ptr(), kind, S, consts.data_ptr(), b.data_ptr(), a.data_ptr(), N, N, N, W.data_ptr(), gamma,
                                                        Wd.data_ptr(), nfft, bin0, m_local, G.data_ptr(), P, H.data_ptr(), P, 1, st), "f")
    f_geq()
    f_sos = lambda: _lib.check(L.fl_sos_response_rc_c64(b.data_ptr(), a.data_ptr(), S, N, N, N, W.data_ptr(), gamma, Wd.data_ptr(), nfft, bin0, m_local,
                                                        G.data_ptr(), P, H.data_ptr(), P, 1, st), "f")
    b2, a2 = b[:, :2].contiguous(), a[:, :2].contiguous()
    f_s2 = lambda: _lib.check(L.fl_sos_response_rc_c64(b2.data_ptr(), a2.data_ptr(), 2, N, N, N, W.data_ptr(), gamma, Wd.data_ptr(), nfft, bin0, m_local,
                                                       G.data_ptr(), P, H.data_ptr(), P, 1, st), "f")
    print(f"lanes={on}: graphic-equaliser entry {timed(f_geq):.1f} us, raw sections (12) {timed(f_sos):.1f} us, raw sections (2) {timed(f_s2):.1f} us")
z = torch.empty(49153024 // 4, dtype=torch.float32, device=dev)
print(f"49 MB fill: {timed(lambda: z.fill_(1.0)):.1f} us")
