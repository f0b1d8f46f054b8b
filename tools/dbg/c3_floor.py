#!/usr/bin/env python
"""Where config 3's float32 error comes from: the 16-channel FDN without attenuation at nfft = 192000, 30 dB --
the error of the core's SPECTRUM and of the time-domain output (behind the gamma^-t envelope) against the float64 oracle, for
the float32 and the float64 kernels, beside the floor that float32 STORAGE of the operands alone sets (oracle in float64 on
operands rounded to float32).    python tools/dbg/c3_floor.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import relerr  # noqa: E402
from oracle import hotpath as O  # noqa: E402
from test_hip_parity import _fdn_model  # noqa: E402
from flamo_amd.processor import dsp, system  # noqa: E402


def main():
    gpu = torch.device("cuda:0")
    torch.manual_seed(130709)
    N, nfft, db = 16, 192000, 30.0
    delays = [503, 593, 701, 811, 919, 1031, 1151, 1259, 1381, 1493, 1613, 1741, 1873, 2003, 2381, 2713]
    meta = dict(N=N, nfft=nfft, alias_decay_db=db, delays=delays, attn=False)
    a = {k: v.double() for k, v in dict(in_gain=torch.randn(N, 1), out_gain=torch.randn(1, N), U_param=torch.randn(N, N)).items()}
    a["delays_s"] = torch.tensor(delays, dtype=torch.float64) / 48000 * 100
    x = torch.zeros(1, nfft, 1, dtype=torch.float64)
    x[:, 0] = 1
    Yref = O.fdn_forward(x, a["in_gain"], a["out_gain"], a["U_param"], a["delays_s"], nfft, db, output="spectrum")
    yref = O.irfft(Yref, nfft, alias_decay_db=db)
    M = nfft // 2 + 1
    for dt in (torch.float32, torch.float64):
        model, _ = _fdn_model(dsp, system, meta, a, gpu, dt)
        with torch.no_grad():
            y = model(x.to(gpu, dt)).cpu()
            Y = model.get_core()(torch.ones(1, M, 1, device=gpu, dtype=torch.complex64 if dt == torch.float32 else torch.complex128)).cpu()
        half = nfft // 2
        print(f"{str(dt)[6:]}: spectrum {relerr(Y, Yref):.2e}  output {relerr(y, yref):.2e}  first half of the output {relerr(y[:, :half], yref[:, :half]):.2e}"
              f"  second half {relerr(y[:, half:], yref[:, half:]):.2e}")


if __name__ == "__main__":
    main()
