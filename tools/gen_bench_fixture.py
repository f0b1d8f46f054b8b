#!/usr/bin/env python
"""Parameters of the benchmark's configurations, drawn ONCE by the reference's own module constructors (flamo's init_param:
dsp.py:292 normal, dsp.py:2555-2556 GEQ uniform, dsp.py:3328 Delay randint) under torch.manual_seed(130709), so
that bench.py, the tools and the CPU baseline all run the same numbers.  Run in the build container (it imports
/root/reference); writes tests/golden/bench_params.npz (float32 values, ~60 KB).

    python tools/gen_bench_fixture.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_params.npz")
PRIMES16 = [503, 593, 701, 811, 919, 1031, 1151, 1259, 1381, 1493, 1613, 1741, 1873, 2003, 2381, 2713]


def main():
    dsp, system = refimport.load()
    torch.manual_seed(130709)
    f32 = torch.float32
    out = {}
    # configs[1]: Series(Matrix(8,8,"random"), GEQ((8,8)))
    out["c2_W"] = dsp.Matrix(size=(8, 8), nfft=960, matrix_type="random", requires_grad=True, dtype=f32).param
    out["c2_geq"] = dsp.GEQ(size=(8, 8), nfft=960, requires_grad=True, dtype=f32).param
    # configs[2] / configs[3]: 16-channel FDN (gains, orthogonal-matrix parameter, attenuation equaliser under 20 log10(sigmoid))
    out["c3_in_gain"] = dsp.Gain(size=(16, 1), nfft=960, requires_grad=True, dtype=f32).param
    out["c3_out_gain"] = dsp.Gain(size=(1, 16), nfft=960, requires_grad=True, dtype=f32).param
    out["c3_U"] = dsp.Matrix(size=(16, 16), nfft=960, matrix_type="orthogonal", requires_grad=True, dtype=f32).param
    out["c3_attn"] = torch.randn(12, 16) * 0.3 + 2.0          # e8_fdn.py draws the attenuation through its own helper; kept as before
    out["c3_delays"] = torch.tensor(PRIMES16, dtype=f32)
    # configs[4] structure: GEQ((32,32)), Delay((32,32), isint, max_len 2000) as the reference draws it, gains, mixing parameter
    out["c5_geq"] = dsp.GEQ(size=(32, 32), nfft=960, requires_grad=True, dtype=f32).param
    dly = dsp.Delay(size=(32, 32), max_len=2000, isint=True, nfft=960, dtype=f32)
    out["c5_delay_s"] = dly.param
    out["c5_gain"] = torch.rand(32) * 0.5 / 32 ** 0.5 + 0.01
    out["c5_U"] = dsp.Matrix(size=(32, 32), nfft=960, matrix_type="orthogonal", requires_grad=True, dtype=f32).param
    arrays = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()}
    meta = dict(seed=130709, source="gdalsanto/flamo v0.2.13 module constructors (tools/gen_bench_fixture.py)",
                shapes={k: list(v.shape) for k, v in arrays.items()})
    np.savez_compressed(OUT, meta=json.dumps(meta), **arrays)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
