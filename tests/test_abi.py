"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every
symbol include/flamo_hip.h declares; host logic of the operator API (no GPU compute)."""
import ctypes
import os
import re
import warnings
from collections import OrderedDict

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "flamo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from flamo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 29
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/flamo_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == syms          # the ctypes table binds exactly the header
    L = _lib.lib()
    assert L.fl_version() == 1


def test_plan_queries_and_errors():
    from flamo_amd import _lib
    L = _lib.lib()
    l1, l2 = ctypes.c_int(), ctypes.c_int()
    for nfft, f64 in ((96000, 0), (192000, 0), (384000, 0), (96000, 1), (2048, 0), (1500, 1)):
        assert L.fl_fft_plan(nfft, f64, ctypes.byref(l1), ctypes.byref(l2)) == 0
        assert l1.value * l2.value == nfft // 2
    assert L.fl_fft_plan(95, 0, None, None) == -2 and b"even" in L.fl_last_error()
    assert L.fl_fft_plan(2 * 17, 0, None, None) == -2 and b"prime factor" in L.fl_last_error()
    assert L.fl_fft_scratch_elems(96000, 0, 256) == 48000 * 256
    assert L.fl_fft_scratch_elems(2048, 0, 256) == 0
    assert L.fl_rfft_f32(None, 0, 0, None, 48032, None, None, 1, 96000, 1.0, 0.0, 0, None) == -1   # null pointers rejected


def test_ops_fail_loudly_without_gpu_tensors():
    from flamo_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rfft(torch.zeros(1, 8, 1), 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.mimo(torch.zeros(2, 2, dtype=torch.complex64), torch.zeros(1, 5, 2, dtype=torch.complex64))


def test_operator_api_host_logic():
    from flamo_amd.processor import dsp, system
    warnings.simplefilter("ignore")
    nfft = 512
    kw = dict(nfft=nfft, alias_decay_db=30.0)
    g1, g2 = dsp.Gain(size=(4, 1), **kw), dsp.Gain(size=(1, 4), **kw)
    dl = dsp.parallelDelay(size=(4,), max_len=100, isint=True, **kw)
    mx = dsp.Matrix(size=(4, 4), matrix_type="orthogonal", **kw)
    rec = system.Recursion(fF=dl, fB=mx)
    core = system.Series(OrderedDict({"input_gain": g1, "feedback_loop": rec, "output_gain": g2}))
    model = system.Shell(core, dsp.FFT(nfft), dsp.iFFTAntiAlias(nfft, alias_decay_db=30.0))
    assert list(model.state_dict()) == ["_Shell__core.input_gain.param", "_Shell__core.feedback_loop.feedforward.param",
                                        "_Shell__core.feedback_loop.feedback.param", "_Shell__core.output_gain.param"]
    assert (model.input_channels, model.output_channels, model.nfft) == (1, 1, nfft)
    assert rec.I.shape == (nfft // 2 + 1, 4, 4)
    # key rules of Series (system.py:127-209)
    s = system.Series(dsp.Gain(size=(2, 2)), OrderedDict({"a": dsp.Gain(size=(2, 2)), "7": dsp.Gain(size=(3, 2))}))
    assert list(s._modules) == ["0", "a", "2"] and (s.input_channels, s.output_channels) == (2, 3)
    s.prepend(dsp.Gain(size=(2, 5)))
    assert s.input_channels == 5 and len(s) == 4
    with pytest.raises(ValueError):
        system.Series(OrderedDict({"a": dsp.Gain()}), OrderedDict({"a": dsp.Gain()}))
    with pytest.raises(AssertionError):
        system.Series(dsp.Gain(size=(3, 2)), dsp.Gain(size=(2, 2)))          # channel mismatch
    with pytest.raises(ValueError):
        system.Series(dsp.Gain(size=(2, 2), nfft=64), dsp.Gain(size=(2, 2), nfft=128))
    with pytest.raises(AssertionError):
        dsp.Gain(size=(3,))
    with pytest.raises(AssertionError):
        dsp.DSP(size=[1, 2])
    with pytest.raises(AssertionError):
        system.Recursion(fF=dsp.Gain(size=(2, 3)), fB=dsp.Gain(size=(2, 2)))
    d = dsp.Delay(size=(2, 2), max_len=50, isint=True, nfft=64)
    d.assign_value(d.sample2s(torch.tensor([[1.0, 2.0], [3.0, 4.0]])))
    assert torch.allclose(d.s2sample(d.param), torch.tensor([[1.0, 2.0], [3.0, 4.0]])) and d.new_value == 1
    # parameter maps (host side) agree with the oracle's restatement
    from oracle import hotpath as O
    p = torch.randn(5, 5, dtype=torch.float64)
    m = dsp.Matrix(size=(5, 5), matrix_type="orthogonal", dtype=torch.float64)
    assert torch.allclose(m.map(p), O.orthogonal(p))
    bq = dsp.Biquad(size=(2, 2), n_sections=2, filter_type="bandpass", dtype=torch.float64)
    assert torch.allclose(bq.map(bq.param), O.biquad_map(bq.param.detach(), "bandpass"))
    geq = dsp.GEQ(size=(2, 2), dtype=torch.float64)
    b, a = geq._sos_coeffs(geq.map(geq.param))
    bo, ao = O.geq_sos(geq.map(geq.param), geq.center_freq, geq.shelving_crossover)
    assert b.dtype == torch.float32 and torch.equal(b, bo) and torch.equal(a, ao)


def test_walk_partition_covers_every_unit_once():
    """fl_spec_walk_partition (host side, no GPU): contiguous, monotone unit ranges that cover [0, row pairs * batch) exactly,
    with a smaller maximum cost than equal unit counts under the kernel's fitted cost model."""
    import ctypes
    from flamo_amd import _lib
    L = _lib.lib()
    for B, n_wg in ((32, 256), (4, 256), (7, 104), (33, 256), (1, 96)):
        host = (ctypes.c_int * (n_wg + 1))()
        assert L.fl_spec_walk_partition(96000, B, n_wg, host) == 0
        b = list(host)
        U = 101 * B
        assert b[0] == 0 and b[-1] == U and all(x <= y for x, y in zip(b, b[1:]))

        def cost(lo, hi):
            c, last = 0, None
            for u in range(lo, hi):
                r = u // B
                if r != last:
                    c, last = c + 31, r
                c += 17 if (r == 0 or 2 * r == 200) else 20
            return c
        got = max(cost(x, y) for x, y in zip(b, b[1:]))
        eq = max(cost(U * i // n_wg, U * (i + 1) // n_wg) for i in range(n_wg))
        assert got <= eq, (B, got, eq)
    assert L.fl_spec_walk_partition(12345, 4, 8, (ctypes.c_int * 9)()) != 0          # no plan for this length


def test_graphed_step_refuses_without_the_graph_packet_switch(monkeypatch):
    """GraphedStep raises (it used to warn) when DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 cannot be relied on: replayed reductions have
    returned wrong values with ROCm's pre-built graph packets.  The check comes before anything touches the GPU."""
    import flamo_amd
    from flamo_amd.graph import GraphedStep
    monkeypatch.setattr(flamo_amd, "_graph_packets_off", lambda: False)
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        GraphedStep(lambda x: x.sum(), (torch.zeros(2),), ())


def test_anonymous_gain_maps_are_recognised_by_probe():
    """The reference's examples hand the equalisers their parameter map as a lambda (examples/e8_fdn.py:97); the design
    kernels fold exactly two maps.  The contract (flamo_amd/processor/dsp.py "recognised callables"): a
    STATELESS plain function -- no closure, no defaults, only the names torch / abs / log10 / sigmoid and the constant 20 in
    its bytecode -- that reproduces one of the two bit for bit, values and gradient, on the probe vector is that map;
    anything else (a clamp, an offset, a captured scale, a callable object, a straight-through trick) is not."""
    import warnings
    from flamo_amd.processor import dsp
    kind = dsp._gain_map_kind
    assert kind(lambda x: 20 * torch.log10(torch.sigmoid(x))) == "sigmoid"
    assert kind(lambda x: 20 * torch.log10(torch.abs(x))) == "abs"
    assert kind(lambda v: 20.0 * v.sigmoid().log10()) == "sigmoid"
    assert kind(dsp.db_of_sigmoid) == "sigmoid" and kind(dsp._db_of_magnitude) == "abs"
    # the round-5 review's counter-examples: each would have passed a value probe of moderate arguments
    assert kind(lambda x: 20 * torch.log10(torch.sigmoid(x)).clamp(min=-200)) is None            # another name in the bytecode
    assert kind(lambda x: 20 * torch.log10(torch.abs(x)).clamp(max=100)) is None
    assert kind(lambda x: 20 * torch.log10(torch.abs(x)) if x.dtype == torch.float64 else x) is None   # dtype-dependent
    scale = [1.0]
    assert kind(lambda x: 20 * torch.log10(torch.abs(x)) * scale[0]) is None                    # closure over mutable state
    assert kind(lambda x, k=20: k * torch.log10(torch.abs(x))) is None                           # default argument

    class Stateful(torch.nn.Module):
        def forward(self, x):
            return 20 * torch.log10(torch.abs(x))
    assert kind(Stateful()) is None and kind(Stateful().forward) is None                         # callable objects, bound methods
    import functools
    assert kind(functools.partial(dsp._db_of_magnitude)) is None
    assert kind(lambda x: 20 * torch.log10(torch.sigmoid(x)) + 0.5) is None
    assert kind(lambda x: x + (20 * torch.log10(torch.sigmoid(x)) - x).detach()) is None       # same values, other gradient
    assert kind(lambda x: 10 * torch.log10(torch.sigmoid(x) ** 2)) is None                      # same function, other rounding
    assert kind(lambda x: 20 * torch.log10(torch.sigmoid(torch.abs(x)))) is None                # allowed names, another function
    assert kind(torch.abs) is None and kind(lambda x: x.reshape(2, -1)) is None and kind(lambda x: 1 / 0) is None
    geq = dsp.parallelGEQ(size=(4,), nfft=64)
    geq.map = lambda x: 20 * torch.log10(torch.sigmoid(x))                                       # set after construction, as e8_fdn.py does
    assert dsp._gain_map_kind(geq.map) == "sigmoid"
    # the substitution is announced once per function, and a module can opt out
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert geq._folded_map() == "sigmoid" and geq._folded_map() == "sigmoid"
    assert len([w for w in rec if "runs inside the HIP kernels" in str(w.message)]) == 1
    keep = dsp.parallelGEQ(size=(4,), nfft=64, map=lambda x: 20 * torch.log10(torch.sigmoid(x)), fold_map=False)
    assert keep._folded_map() is None
    geq.fold_map = False
    assert geq._folded_map() is None
    with warnings.catch_warnings(record=True) as rec:                                            # the named maps: no warning
        warnings.simplefilter("always")
        assert dsp.GEQ(size=(2, 2), nfft=64)._folded_map() == "abs"
    assert not rec


def test_criteria_and_magnitude_layer_host_logic():
    """Host-side logic of the round-5 drop-ins, no GPU: the magnitude-layer probe takes exactly the callables that are torch.abs
    (values and gradient bit for bit) and nothing else; flamo_amd.optimize.mse_loss / sparsity_loss on host tensors run the
    reference's own lines (flamo/optimize/loss.py:12-103) -- constructors, attributes and values."""
    import math
    from flamo_amd.optimize import mse_loss, sparsity_loss
    from flamo_amd.processor import dsp
    assert dsp._is_magnitude_map(torch.abs)
    assert dsp._is_magnitude_map(lambda x: torch.abs(x))
    assert dsp._is_magnitude_map(lambda z: z.abs())
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x) + 0.0 * x.real)      # another name (`real`) in the bytecode: refused on sight
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x).clamp(max=1e9))      # the review's counter-example (probe maximum was 2.5e7)
    gain = [1.0]
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x) * gain[0])           # closure over mutable state

    class Mag(torch.nn.Module):
        def forward(self, x):
            return torch.abs(x)
    assert not dsp._is_magnitude_map(Mag()) and not dsp._is_magnitude_map(Mag().forward)
    assert dsp.Transform(lambda x: torch.abs(x)).recognise and not dsp.Transform(lambda x: torch.abs(x), recognise=False).recognise
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x) + 1e-30)
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x) ** 2)
    assert not dsp._is_magnitude_map(lambda x: x)
    assert not dsp._is_magnitude_map(lambda x: torch.abs(x).float())
    assert not dsp._is_magnitude_map(lambda x: 1 / 0)
    layer = dsp.Transform(lambda x: torch.abs(x))
    z = torch.randn(2, 5, 3, dtype=torch.complex128)
    assert torch.equal(layer(z), torch.abs(z))          # host tensors: the callable itself
    crit = mse_loss(nfft=1024, device="cpu")
    assert crit.name == "MSE" and crit.nfft == 1024 and isinstance(crit.mse_loss, torch.nn.MSELoss)
    yp, yt = torch.randn(2, 7, 3, dtype=torch.float64, requires_grad=True), torch.randn(2, 7, 1, dtype=torch.float64)
    want = torch.nn.MSELoss()(yp.sum(-1), yt.squeeze(-1))
    assert torch.equal(crit(yp, yt), want)

    class _Mix:
        def __init__(self, A):
            self.param = A
            self.map = lambda x: x

    class _Model:
        def __init__(self, A):
            core = type("C", (), {})()
            core.feedback_loop = type("L", (), {})()
            core.feedback_loop.feedback = _Mix(A)
            self._core = core

        def get_core(self):
            return self._core

    for A in (torch.randn(6, 6, dtype=torch.float64), torch.randn(3, 4, 4, dtype=torch.float64)):
        N = A.shape[-1]
        got = sparsity_loss()(None, None, _Model(A))
        if A.dim() == 3:
            ref = torch.mean((torch.sum(torch.abs(A), dim=(-2, -1)) - N * math.sqrt(N)) / (N * (1 - math.sqrt(N))))
        else:
            ref = -(torch.sum(torch.abs(A)) - N * math.sqrt(N)) / (N * (math.sqrt(N) - 1))
        assert abs(float(got) - float(ref)) < 1e-14


def test_sparsity_loss_of_a_householder_feedback_against_the_reference():
    """flamo/optimize/loss.py:51-53: with a HouseholderMatrix as the mixing matrix the criterion is the sparsity of I - 2 u u^T,
    not of the stored unit vector (the drop-in of round 5 returned -inf here: N = 1).  Value and gradient recorded from the
    reference itself (tools/gen_golden.py::gen_householder_sparsity), reproduced by the drop-in criterion on a drop-in FDN core
    built on the host (no kernel runs: construction and the criterion's host lines only)."""
    import os
    from collections import OrderedDict
    import numpy as np
    from flamo_amd.optimize import sparsity_loss
    from flamo_amd.processor import dsp, system
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "householder_sparsity.npz"))
    N, nfft = 6, 256
    kw = dict(nfft=nfft, dtype=torch.float64)
    mix = dsp.HouseholderMatrix(size=(N, N), requires_grad=True, **kw)
    with torch.no_grad():
        mix.param.copy_(torch.from_numpy(gold["u"]))
    core = system.Series(OrderedDict(input_gain=dsp.Gain(size=(N, 1), **kw),
                                     feedback_loop=system.Recursion(fF=dsp.parallelDelay(size=(N,), max_len=40, isint=True, **kw), fB=mix),
                                     output_gain=dsp.Gain(size=(1, N), **kw)))
    model = system.Shell(core, dsp.FFT(nfft, dtype=torch.float64), dsp.iFFT(nfft, dtype=torch.float64))
    loss = sparsity_loss()(None, None, model)
    (g,) = torch.autograd.grad(loss, [mix.param])
    assert abs(loss.item() - float(gold["loss"])) < 1e-14, (loss.item(), float(gold["loss"]))
    assert np.allclose(g.numpy(), gold["g_u"], rtol=1e-12, atol=1e-15)
    # the nested form the reference also looks in: feedback = Series(mixing_matrix=Householder, ...)
    core2 = system.Series(OrderedDict(feedback_loop=system.Recursion(
        fF=dsp.parallelDelay(size=(N,), max_len=40, isint=True, **kw),
        fB=system.Series(OrderedDict(mixing_matrix=mix, attenuation=dsp.parallelGain(size=(N,), **kw))))))
    model2 = system.Shell(core2, dsp.FFT(nfft, dtype=torch.float64), dsp.iFFT(nfft, dtype=torch.float64))
    assert abs(sparsity_loss()(None, None, model2).item() - float(gold["loss"])) < 1e-14
