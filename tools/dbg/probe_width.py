"""fl_hbm_probe with 16 and with 8 bytes per lane and access: read / write / copy, 98 MB and 1 GiB, k = 1, 4, 8 workgroups per CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flamo_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
GiB = 1 << 30
src = torch.empty(GiB // 4, device=dev).normal_()
dst = torch.empty(GiB // 4, device=dev)
part = torch.empty(65536, device=dev)
st = torch.cuda.current_stream().cuda_stream
for kind, name, mv in ((0, "read", 1), (1, "write", 1), (2, "copy", 2)):
    for label, nbytes in (("98MB", 98304000 // 32768 * 32768), ("1GiB", GiB)):
        row = []
        for flags in (0, 8):
            for k in (1, 4, 8):
                for _ in range(3):
                    L.fl_hbm_probe(kind, src.data_ptr(), dst.data_ptr(), nbytes, 256 * k, flags, part.data_ptr(), st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    L.fl_hbm_probe(kind, src.data_ptr(), dst.data_ptr(), nbytes, 256 * k, flags, part.data_ptr(), st)
                e1.record()
                torch.cuda.synchronize()
                row.append(20 * mv * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        print(f"{name:6s} {label:5s}  16 B: " + " ".join(f"{g:6.0f}" for g in row[:3]) + "   8 B: " + " ".join(f"{g:6.0f}" for g in row[3:]) + "  GB/s (k = 1, 4, 8)")
