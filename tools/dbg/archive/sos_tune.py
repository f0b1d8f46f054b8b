import sys, torch, time
sys.path.insert(0, '.')
from flamo_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
nfft, C, S = 96000, 64, 12
torch.manual_seed(0)
b = torch.randn(3, S, 8, 8, dtype=torch.float64, device=dev, requires_grad=True)
a = (torch.randn(3, S, 8, 8, dtype=torch.float64, device=dev) + torch.tensor([3., 0, 0], dtype=torch.float64, device=dev).view(3,1,1,1)).requires_grad_(True)
M = nfft // 2 + 1
gH = ops._empty_rows((8, 8), M, torch.complex64, dev); gH.copy_(torch.randn(8, 8, M, dtype=torch.complex64, device=dev))
H = ops.sos_response(b, a, 0.9999, nfft).detach()
Hp = ops._h_planar(H, True)
Wd = ops.twiddles(nfft, torch.float64, dev)
P = ops._pitch(M)
st = torch.cuda.current_stream().cuda_stream
def run(cfg, mixed):
    L.fl_debug_set_sos_chunk(cfg)
    nblk = L.fl_sos_bwd_blocks(M, C, S, int(mixed))
    part = torch.empty((nblk, 2, 3, S, C), dtype=torch.float64, device=dev)
    def go():
        _lib.check(L.fl_sos_response_bwd_c64(gH.data_ptr(), P, Hp.data_ptr() if mixed else None, P, b.data_ptr(), a.data_ptr(), S, C, 0.9999,
                                   Wd.data_ptr(), nfft, 0, M, part.data_ptr(), st))
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3, part.sum(0)
ref = None
for mixed in (False, True):
    for blocks in (16, 32, 47):
        for ch in (4, 6, 8, 12):
            if not mixed and ch == 8: continue
            us, tot = run(blocks * 100 + ch, mixed)
            if ref is None: ref = tot
            print("mixed", mixed, "blocks", blocks, "chunk", ch, "us %.1f" % us, "diff %.2e" % ((tot - ref).norm() / ref.norm()).item())
L.fl_debug_set_sos_chunk(0)
