"""Per-workgroup cycles of the forward walking kernel against (units, row-pair entries, self-mirrored units): least squares."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
nfft, N, B = 96000, 8, 32
M = nfft // 2 + 1
L = _lib.lib()
torch.manual_seed(0)
x = torch.randn(B, nfft, N, device=dev)
H = ops.permute_bins(torch.randn(M, N, N, device=dev, dtype=torch.complex64) / N ** 0.5, nfft)
Hp = ops._h_planar(H, True)
hp = ops._lead_pitch(Hp.movedim(0, -1))
W = ops.twiddles(nfft, torch.float32, dev)
S = ops._spec_cols_fwd(x, nfft, 0.0)
S2 = torch.empty_like(S)
Xp = torch.zeros(L.fl_spec_walk_spectrum_elems(nfft, B, N), dtype=torch.complex64, device=dev)
G = L.fl_spec_walk_workgroups(nfft, B)
P, U = 101, 101 * B
for mode in ("equal", "partition"):
    bounds = [U * w // G for w in range(G + 1)] if mode == "equal" else ops._walk_partition(nfft, B, dev).cpu().tolist()
    bt = torch.tensor(bounds, dtype=torch.int32, device=dev)
    rows = []
    for rep in range(3):
        buf = torch.zeros(G * 8, dtype=torch.int64, device=dev)
        L.fl_debug_set_walk(1, 0, 0, buf.data_ptr())
        L.fl_spec_mid_walk_f32(S.data_ptr(), S2.data_ptr(), Xp.data_ptr(), Hp.data_ptr(), N * hp, hp, 0, W.data_ptr(), nfft, B, N, N, 1.0, 0, 0,
                               bt.data_ptr(), ops._stream())
        torch.cuda.synchronize()
        L.fl_debug_set_walk(1, 0, 0, None)
        tt = buf.view(-1, 8).cpu().double()
        for blk in range(G):
            w = (blk & 7) * (G >> 3) + (blk >> 3)
            lo, hi = bounds[w], bounds[w + 1]
            if hi <= lo:
                continue
            rs = [u // B for u in range(lo, hi)]
            nself = sum(1 for r in rs if r == 0 or r == 100)
            rows.append([hi - lo - nself, len(set(rs)), nself, tt[blk, 1] - tt[blk, 0]])
    A = torch.tensor([r[:3] for r in rows], dtype=torch.float64)
    y = torch.tensor([r[3] for r in rows], dtype=torch.float64)
    sol = torch.linalg.lstsq(A, y.unsqueeze(1)).solution.squeeze()
    res = (A @ sol - y)
    print(f"{mode}: cycles ~ {sol[0]:.0f} * units + {sol[1]:.0f} * row-pair entries + {sol[2]:.0f} * self-mirrored units; rms residual {res.pow(2).mean().sqrt():.0f}; "
          f"body max {y.max():.0f} mean {y.mean():.0f}")
