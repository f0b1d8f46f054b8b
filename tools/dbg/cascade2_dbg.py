"""Compare the lanes backward's tap gradients with the first generation's and the all-double kernel's, per section / tap;
time the kernels through the C ABI on preallocated buffers."""
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
S, C = 12, N * N
M = nfft // 2 + 1
spec = geq._cascade_spec(geq.param)
xc, consts = spec[1].contiguous(), spec[2]
b = torch.empty((3, S, N, N), dtype=torch.float64, device=dev)
a = torch.empty_like(b)
with ops.row_major_bins(nfft):
    bin0, m_local = ops._bin0_arg(nfft)
    Hv, G, cfg = ops._cascade_rc_forward(b, a, W, geq._gamma_f, nfft, torch.float32, True, geq=(xc, ops._geq_in_kind(xc, True, False), consts))
P = ops._pitch(m_local)
gH = ops._empty_rows((N, N), m_local, torch.complex64, dev)
gH[..., :m_local] = torch.randn(N, N, m_local, device=dev) + 1j * torch.randn(N, N, m_local, device=dev)
Wd = ops.twiddles(nfft, torch.float64, dev)
st = ops._stream()
gamma = float(geq._gamma_f)


def timed(fn, reps=50):
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


# forward: second generation against the first
def fwd():
    with ops.row_major_bins(nfft):
        return ops._cascade_rc_forward(b, a, W, geq._gamma_f, nfft, torch.float32, True, geq=(xc, ops._geq_in_kind(xc, True, False), consts))
res = {}
for on in (3, 1):
    L.fl_debug_set_cascade_lanes(on, -1, -1)
    Hq, Gq, _ = fwd()
    res[on] = (Hq.clone(), Gq[..., :m_local].clone(), timed(lambda: fwd()))
L.fl_debug_set_rc_fast(0)
Hd, Gd, _ = fwd()
L.fl_debug_set_rc_fast(6)
rel = lambda x, y: ((x - y).norm() / y.norm()).item()
print(f"forward: first generation {res[3][2]:.1f} us, second {res[1][2]:.1f} us (eager call: host time included); second vs first: H {rel(res[1][0], res[3][0]):.2e} G {rel(res[1][1], res[3][1]):.2e};"
      f" vs the double evaluation: second H {rel(res[1][0], Hd):.2e} G {rel(res[1][1], Gd[..., :m_local]):.2e}, first H {rel(res[3][0], Hd):.2e} G {rel(res[3][1], Gd[..., :m_local]):.2e}")
Gp2 = ops._empty_rows((N, N), m_local, torch.complex64, dev)
Hp2 = ops._empty_rows((N, N), m_local, torch.complex64, dev)
def fwd_c(on):
    L.fl_debug_set_cascade_lanes(on, -1, -1)
    return lambda: _lib.check(L.fl_geq_response_rc_c64(xc.data_ptr(), ops._geq_in_kind(xc, True, False), S, consts.data_ptr(), b.data_ptr(), a.data_ptr(), N, N, N, W.data_ptr(), gamma,
                                                       Wd.data_ptr(), nfft, bin0, m_local, Gp2.data_ptr(), P, Hp2.data_ptr(), P, 1, st), "f")
Wd = ops.twiddles(nfft, torch.float64, dev)
P = ops._pitch(m_local)
st = ops._stream()
gamma = float(geq._gamma_f)
for on in (3, 1):
    f = fwd_c(on)
    print(f"forward through the C ABI, lanes={on}: {timed(f):.1f} us")
L.fl_debug_set_cascade_lanes(1, -1, -1)

# first generation
nblk = L.fl_sos_bwd_blocks(m_local, C, S, 1)
part = torch.empty((nblk, 2, 3, S, C), dtype=torch.float64, device=dev)
partW = torch.empty((nblk, N, N, N), dtype=torch.float32, device=dev)
f1 = lambda: _lib.check(L.fl_sos_response_bwd_rc_c64(gH.data_ptr(), P, G.data_ptr(), P, b.data_ptr(), a.data_ptr(), S, N, N, N, W.data_ptr(), gamma,
                                                      Wd.data_ptr(), nfft, bin0, m_local, part.data_ptr(), partW.data_ptr(), st), "g1")
f1()
tot1 = part.sum(0)
print(f"first generation cascade backward: {timed(f1):.1f} us")
out = torch.empty_like(xc)
gW = torch.empty_like(W)
esz = 8
f1d = lambda: _lib.check(L.fl_geq_sections_bwd_w(xc.data_ptr(), 2, part.data_ptr(), part.data_ptr() + 3 * S * C * esz, 6 * S * C, nblk, S, C, consts.data_ptr(),
                                                 out.data_ptr(), partW.data_ptr(), nblk * N, N * N, gW.data_ptr(), st), "d1")
f1d()
g1 = out.clone()
print(f"first generation design backward: {timed(f1d):.1f} us")
# all-double: complex128 copies
G128, gH128 = G.to(torch.complex128), gH.to(torch.complex128)
nblkd = L.fl_sos_bwd_blocks(m_local, C, S, 0)
partd = torch.empty((nblkd, 2, 3, S, C), dtype=torch.float64, device=dev)
gG128 = torch.einsum("mnf,jn->mjf", gH128[..., :m_local], W.double().to(torch.complex128)).contiguous()
gGp = torch.zeros((N, N, P), dtype=torch.complex128, device=dev)
gGp[..., :m_local] = gG128
G128p = torch.zeros((N, N, P), dtype=torch.complex128, device=dev)
G128p[..., :m_local] = G128[..., :m_local]
_lib.check(L.fl_sos_response_bwd_c128(gGp.data_ptr(), P, G128p.data_ptr(), P, b.data_ptr(), a.data_ptr(), S, C, gamma, Wd.data_ptr(), nfft, bin0, m_local,
                                      partd.data_ptr(), st), "dd")
totd = partd.sum(0)
print("first generation vs all-double, tap gradients per (poly, tap):", [[f"{((tot1[i, p] - totd[i, p]).norm() / totd[i, p].norm()).item():.1e}" for p in range(3)] for i in range(2)])

for bpc, tb in ((1, 0), (1, 20), (1, 12), (1, 10)):
    L.fl_debug_set_cascade_lanes(1, bpc, tb)
    nbx = L.fl_geq_bwd_lanes_blocks(m_local, C, S, nfft, bin0, N, N, 1)
    if nbx == 0:
        continue
    psum = torch.zeros((S * C, nbx, 4), dtype=torch.float32, device=dev)
    pq = torch.empty((C, nbx), dtype=torch.float32, device=dev)
    wrows = L.fl_geq_bwd_lanes_wrows(m_local, C, S, nfft, bin0, N, N)
    pW = torch.empty((N * N, wrows), dtype=torch.float32, device=dev)
    f2 = lambda: _lib.check(L.fl_geq_response_bwd_lanes_c64(1, gH.data_ptr(), P, G.data_ptr(), P, b.data_ptr(), a.data_ptr(), S, N, N, N, W.data_ptr(), gamma,
                                                            Wd.data_ptr(), nfft, bin0, m_local, psum.data_ptr(), pq.data_ptr(), pW.data_ptr(), st), "g2")
    out2 = torch.empty_like(xc)
    gW2 = torch.empty_like(W)
    f2d = lambda: _lib.check(L.fl_geq_sections_bwd_lanes(xc.data_ptr(), 2, psum.data_ptr(), pq.data_ptr(), nbx, b.data_ptr(), a.data_ptr(), gamma, S, C,
                                                         consts.data_ptr(), out2.data_ptr(), pW.data_ptr(), wrows, N * N, gW2.data_ptr(), st), "d2")
    f2(); f2d()
    t2, t2d = timed(f2), timed(f2d)
    # phase stamps (cycles per wavefront, averaged over the workgroups) and the phases alone
    nw = 11
    stamps = torch.zeros((nbx, nw, 6), dtype=torch.int64, device=dev)
    L.fl_debug_set_cascade_stamps(stamps.data_ptr(), 0)
    f2(); torch.cuda.synchronize()
    L.fl_debug_set_cascade_stamps(None, 0)
    sm = stamps.double().mean(0)
    print("   cycles per wavefront [bin work, barrier, mfma, section work, lifetime x10ns, kernel cycles]: waves 0/3/7/10:", [[int(v) for v in sm[w]] for w in (0, 3, 7, 10)],
          " max kernel", int(stamps[:, :, 5].max()), " core clock MHz", float((stamps[:, :, 5].double() / stamps[:, :, 4].double().clamp_min(1)).mean() * 100))
    L.fl_debug_set_cascade_stamps(None, 4); t_one_order = timed(f2)
    L.fl_debug_set_cascade_stamps(None, 0); t_alt = timed(f2)
    print(f"one order of the two phases in every wavefront {t_one_order:.1f} us, alternating by wavefront group {t_alt:.1f} us")
    L.fl_debug_set_cascade_stamps(None, 2); t_no2 = timed(f2)
    L.fl_debug_set_cascade_stamps(None, 3); t_no12 = timed(f2)
    L.fl_debug_set_cascade_stamps(None, 0)
    print(f"without the section phase {t_no2:.1f} us, without both {t_no12:.1f} us")
    print(f"lanes bpc {bpc} tile {tb:2d} blocks {nbx:4d}: cascade backward {t2:.1f} us, reduce + design {t2d:.1f} us;  gain grad vs gen1 {((out2 - g1).norm() / g1.norm()).item():.2e}  W {((gW2 - gW).norm() / gW.norm()).item():.2e}")
# tap gradients from the lanes sums (float64 on the host side)
ps = psum.double().sum(1).view(S, C, 4).permute(2, 0, 1)
Q = pq.double().sum(-1)
bb, aa = b.view(3, S, C), a.view(3, S, C)
tot2 = torch.zeros_like(tot1).view(2, 3, S, C)
for i, (co, sgn) in enumerate(((bb, 1.0), (aa, -1.0))):
    Sg, T, D = co[0] + gamma ** 2 * co[2], gamma * co[1], co[0] - gamma ** 2 * co[2]
    G0, G2 = sgn * ps[i], sgn * ps[2 + i]
    G1 = ((Sg + T) * G0 - D * G2 - sgn * Q) / Sg
    tot2[i, 0], tot2[i, 1], tot2[i, 2] = G0 - G1 - G2, gamma * G0, gamma ** 2 * (G0 - G1 + G2)
t1v, tdv = tot1.view(2, 3, S, C), totd.view(2, 3, S, C)
for s in range(1, S):
    print(f"section {s:2d}: lanes vs double " + " ".join(f"{((tot2[i, p, s] - tdv[i, p, s]).norm() / tdv[i, p, s].norm()).item():.1e}" for i in range(2) for p in range(3))
          + "   gen1 vs double " + " ".join(f"{((t1v[i, p, s] - tdv[i, p, s]).norm() / tdv[i, p, s].norm()).item():.1e}" for i in range(2) for p in range(3)))
# the combination the design takes: second difference b0 - 2 c b1 + b2 ~ (1 - c) G0 - G1 and first difference b0 - b2
for s in (2, 3, 6, 10):
    for i in range(2):
        d2 = lambda t: t[i, 0, s] - 2 * t[i, 1, s] + t[i, 2, s]
        d1 = lambda t: t[i, 0, s] - t[i, 2, s]
        print(f"section {s} poly {i}: second difference lanes {((d2(tot2) - d2(tdv)).norm() / d2(tdv).norm()).item():.1e} gen1 {((d2(t1v) - d2(tdv)).norm() / d2(tdv).norm()).item():.1e};"
              f" first difference lanes {((d1(tot2) - d1(tdv)).norm() / d1(tdv).norm()).item():.1e} gen1 {((d1(t1v) - d1(tdv)).norm() / d1(tdv).norm()).item():.1e}")
print("Q vs double:", ((Q - (gG128.conj() * G128[..., :m_local]).real.sum(-1).view(-1)).norm() / Q.norm()).item())
