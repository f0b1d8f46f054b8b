"""Drop-in counterparts of ``flamo.processor.dsp`` for the frequency-sampling hot path, executed
by hand-written HIP kernels on MI355X (``flamo_amd.ops``).

Same class names, constructor signatures, attributes (``param``, ``map``, ``gamma``, ``nfft``,
``alias_decay_db``, ``freq_response``, ``freq_convolve``, ``input_channels``/``output_channels``),
tensor conventions (time ``(B, T, N, ...)``, frequency ``(B, M, N, ...)`` with ``M = nfft//2+1``)
and error behaviour as the reference (gdalsanto/flamo v0.2.13; citations are to its
``flamo/processor/dsp.py``).  What differs is where the arithmetic runs:

* ``FFT``/``iFFT``/``*AntiAlias``     -> ``ops.rfft`` / ``ops.irfft``   (torch.fft.rfft/irfft, dsp.py:88,114,161,204)
* every ``freq_convolve``            -> ``ops.mimo``                   (the four einsum patterns, dsp.py:466,552,922,1021)
* SOS-type responses (Biquad, GEQ)   -> ``ops.sos_response``           (rfft(3 taps)+prod/prod, dsp.py:1520-1526)
* integer delays                     -> ``ops.delay_response``         (exp(-j w m), dsp.py:3356-3374)

Parameter maps (matrix_exp, log10, softplus, RBJ / shelving formulas) stay in PyTorch: they act
on a few hundred scalars and autograd differentiates them for free.
"""
from __future__ import annotations

import warnings
import weakref
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..functional import (GEQDesign, accurate_geq, HadamardMatrix, RotationMatrix, bandpass_filter, eq_freqs, highpass_filter,
                          lowpass_filter, matrix_exp_capturable, rad2hertz, skew_matrix)
from ..utils import to_complex

_identity = lambda x: x  # noqa: E731


def _gamma(alias_decay_db, nfft, device=None, dtype=torch.float32) -> torch.Tensor:
    db = torch.as_tensor(alias_decay_db, device=device, dtype=dtype)
    return 10 ** (-torch.abs(db) / nfft / 20)


# ============================================================================ recognised callables
# The reference's examples hand over two small functions as ANONYMOUS callables: the magnitude output layer
# `lambda x: torch.abs(x)` (e7_biquad.py:76, e8_colorless_fdn.py:102) and the equalisers' gain map
# `lambda x: 20*torch.log10(torch.sigmoid(x))` (e8_fdn.py:97).  The drop-in runs those inside its kernels.  The contract
# ("Recognised callables"): a callable is replaced by a kernel only when
#   (1) it IS the named function (torch.abs, dsp._db_of_magnitude, dsp.db_of_sigmoid), or
#   (2) it is a STATELESS plain Python function -- a `def` / `lambda` with no closure cells, no default arguments, not a bound
#       method, not an nn.Module or other callable object -- whose bytecode refers to nothing but the names `torch`, `abs`,
#       `log10`, `sigmoid` and the constants 20 / 20.0 (so `.clamp(...)`, a dtype test, a captured or global scale are refused
#       on sight: they would need another name), AND it then reproduces the named function bit for bit, values and gradient,
#       on a probe vector that spans 1e-30 ... 1e30;
#   the module has not opted out (`Transform(..., recognise=False)`, `GEQ(..., fold_map=False)`, or the attributes of the same
#   names set later), and the first substitution of an anonymous callable says so once through `warnings.warn`.
# Everything else runs as the callable it is, as torch ops in front of / behind the kernels.
import types as _types

_PURE_NAMES = frozenset({"torch", "abs", "log10", "sigmoid"})
_PURE_CONSTS = (20, 20.0)


def _is_stateless_function_of(fn, names=_PURE_NAMES) -> bool:
    """(2) above, the structural half: nothing the function computes can depend on anything but its argument."""
    if not isinstance(fn, _types.FunctionType):
        return False            # builtins, bound methods, functools.partial, nn.Module instances, callable objects
    if fn.__closure__ or fn.__defaults__ or fn.__kwdefaults__:
        return False
    code = fn.__code__
    if code.co_argcount != 1 or code.co_kwonlyargcount or code.co_flags & 0x0C:      # exactly one positional argument, no *args / **kwargs
        return False
    if code.co_freevars or code.co_cellvars or not set(code.co_names) <= names:
        return False
    for c in code.co_consts:
        if c is None or isinstance(c, str):      # a docstring is harmless
            continue
        if isinstance(c, bool) or not isinstance(c, (int, float)) or c not in _PURE_CONSTS:
            return False
    return True


_SUBSTITUTION_WARNED = set()


def _warn_substitution(fn, what: str) -> None:
    key = (getattr(fn, "__code__", None) or id(fn), what)
    if key in _SUBSTITUTION_WARNED:
        return
    _SUBSTITUTION_WARNED.add(key)
    code = getattr(fn, "__code__", None)
    where = f"{code.co_filename}:{code.co_firstlineno}" if code is not None else repr(fn)
    warnings.warn(f"flamo_amd: the callable defined at {where} equals {what} bit for bit and runs inside the HIP kernels from now "
                  f"on (pass recognise=False / fold_map=False to the module to keep it as torch ops)", stacklevel=3)


# ============================================================================ transforms
MAGNITUDE_LAYER = True      # False: a recognised `torch.abs` output layer still runs as the callable (tests compare the two)
_MAGNITUDE_CACHE = weakref.WeakKeyDictionary()


def _is_magnitude_map(fn) -> bool:
    """True when `fn` IS torch.abs, or is a stateless plain function over {torch, abs} (`_is_stateless_function_of`) that
    reproduces it bit for bit -- values AND gradient -- on a complex128 probe vector (zeros, both signs, 1e-300 ... 1e30):
    the examples' output layer `lambda x: torch.abs(x)` (e7_biquad.py:76, e8_colorless_fdn.py:102).  Probed once per function
    object (a stateless function cannot change its answer); anything else runs as it is."""
    if fn is torch.abs:
        return True
    if not _is_stateless_function_of(fn):
        return False
    try:
        return _MAGNITUDE_CACHE[fn]
    except KeyError:
        pass
    ok = False
    try:
        re = torch.tensor([0.0, 1.0, -1.0, 0.0, 3.0, -0.5, 1e-12, 2.5e7, -7.25, 0.0, 1e30, -1e-30, 6e9, -3e15], dtype=torch.float64)
        im = torch.tensor([0.0, 0.0, 0.0, -2.0, 4.0, 0.125, -3e-13, 1.5e6, -7.25, 1e-300, -1e30, 1e-30, 8e9, 0.0], dtype=torch.float64)
        probe = torch.complex(re, im)
        w = torch.linspace(0.5, 1.5, probe.numel(), dtype=torch.float64)
        with torch.enable_grad():
            x = probe.clone().requires_grad_(True)
            y = fn(x)
            if torch.is_tensor(y) and y.shape == x.shape and y.dtype == torch.float64 and y.requires_grad:
                (g,) = torch.autograd.grad((y * w).sum(), [x])
                xr = probe.clone().requires_grad_(True)
                yr = torch.abs(xr)
                (gr,) = torch.autograd.grad((yr * w).sum(), [xr])
                ok = bool(torch.equal(y.detach(), yr.detach()) and torch.equal(torch.view_as_real(g), torch.view_as_real(gr)))
    except Exception:       # a function that does not take a complex128 host vector is simply not the magnitude
        ok = False
    _MAGNITUDE_CACHE[fn] = ok
    return ok


class Transform(nn.Module):
    """Wraps a callable as a layer (dsp.py:27-66)."""

    def __init__(self, transform: callable = _identity, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32, *, recognise: bool = True):
        super().__init__()
        self.transform = transform
        self.device = device
        self.dtype = dtype
        self.recognise = recognise      # False: the callable always runs as it is (see "recognised callables" above)

    def forward(self, x: torch.Tensor):
        if (torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.complex64, torch.complex128) and x.numel() > 0
                and MAGNITUDE_LAYER and self.recognise and _is_magnitude_map(self.transform)):
            if self.transform is not torch.abs:
                _warn_substitution(self.transform, "torch.abs")
            return ops.cabs(x)
        return self.transform(x)

    def probe(self, z: torch.Tensor):
        return None


class FFT(Transform):
    """Real FFT along dim 1, ``(B, T, N, ...) -> (B, nfft//2+1, N, ...)`` (dsp.py:69-93)."""

    def __init__(self, nfft: int = 2 ** 11, norm: str = "backward", dtype: torch.dtype = torch.float32):
        self.nfft, self.norm = nfft, norm
        super().__init__(transform=lambda x: ops.rfft(x, self.nfft, self.norm), dtype=dtype)
        self._own_transform = self.transform


class iFFT(Transform):
    """Inverse real FFT along dim 1 (dsp.py:96-119)."""

    def __init__(self, nfft: int = 2 ** 11, norm: str = "backward", dtype: torch.dtype = torch.float32):
        self.nfft, self.norm = nfft, norm
        super().__init__(transform=lambda x: ops.irfft(x, self.nfft, self.norm), dtype=dtype)
        self._own_transform = self.transform


class _AntiAliasMixin:
    def _setup(self, nfft, norm, alias_decay_db, device, dtype):
        self.nfft, self.norm = nfft, norm
        self._alias_db = float(alias_decay_db)
        gamma = _gamma(alias_decay_db, nfft, device, dtype)
        # kept for API parity; the kernels regenerate gamma^-t in registers from log2(gamma)
        self.alias_envelope = gamma ** torch.arange(0, -nfft, -1, device=device, dtype=dtype)

    def _check(self, x):
        # the reference applies the envelope with einsum("btm,t->btm"): 3-D input, T == nfft
        if x.dim() != 3 or x.shape[1] != self.nfft:
            raise RuntimeError(
                f"anti-alias transform expects a (B, {self.nfft}, N) tensor, got {tuple(x.shape)}")


class FFTAntiAlias(Transform, _AntiAliasMixin):
    """rfft(x * gamma^-t) -- the envelope is the *rising* one, as coded in dsp.py:153-162."""

    def __init__(self, nfft: int = 2 ** 11, norm: str = "backward", alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self._setup(nfft, norm, alias_decay_db, device, dtype)

        def transform(x):
            self._check(x)
            return ops.rfft(x, self.nfft, self.norm, self._alias_db)

        super().__init__(transform=transform, device=device, dtype=dtype)
        self._own_transform = self.transform


class iFFTAntiAlias(Transform, _AntiAliasMixin):
    """irfft(X) * gamma^-t (dsp.py:166-206)."""

    def __init__(self, nfft: int = 2 ** 11, norm: str = "backward", alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self._setup(nfft, norm, alias_decay_db, device, dtype)

        def transform(x):
            if x.dim() != 3:
                raise RuntimeError(f"anti-alias transform expects a (B, M, N) tensor, got {tuple(x.shape)}")
            return ops.irfft(x, self.nfft, self.norm, self._alias_db)

        super().__init__(transform=transform, device=device, dtype=dtype)
        self._own_transform = self.transform


def _cached_integer_delay(module, param, samples_fn):
    """Integer-delay response H[k] = gamma^m W_n^((k m) mod n), m = round(samples) (dsp.py:3356-3365).  `round` has a zero
    gradient, so the response is a constant of the parameter VALUES: it is evaluated once per (parameter tensor, version
    counter, bin range / order) and kept -- a feedback delay network's delays never change during training, and the
    eight parameter-sized launches in front of the kernel (cast, round, pow, casts) were a tenth of a batch-1 FDN step.
    (In-place edits through `param.data` do not move the version counter; use assign_value / copy_ under no_grad.)"""
    import weakref
    # While a HIP graph is being captured the cache is neither read nor written: a hit would record NO kernel, so the graph
    # would read an eager-pool tensor that the next eager call with another key (bin order, shard, bumped version) frees --
    # and its replays would ignore a later assign_value.  Inside a capture the kernel is launched, recorded, and its
    # output lives in the graph's own pool.
    # A flamo_amd.graph.GraphedStep capture (ops.capture_scope) is the exception: it holds the tensor for the graph's lifetime and
    # checks the parameter's version counter before every replay, so a hit left by its eager warm-up runs is taken -- the
    # response's launch and the eight small ones in front of it are 26 us of a 0.31 ms colorless-FDN training step.
    capturing = param.is_cuda and torch.cuda.is_current_stream_capturing()
    key = (param._version, ops.bin_shard(module.nfft), ops.bin_order(module.nfft), param.device, param.dtype)
    hit = module.__dict__.get("_int_delay_cache")
    if hit is not None and hit[0]() is param and hit[1] == key:
        if not capturing:
            return hit[2]
        sink = ops.capture_constants()
        if sink is not None and not param.requires_grad:
            sink.append((weakref.ref(param), param._version, hit[2]))
            return hit[2]
    with torch.no_grad():
        mi = samples_fn().round()             # half-to-even, as torch.round in the reference
        amp = (module._gamma_f ** mi).to(module.dtype)
        H = ops.delay_response(mi.to(torch.int64), amp, module.nfft)
    if not capturing:
        module.__dict__["_int_delay_cache"] = (weakref.ref(param), key, H)
    return H


# ============================================================================ core base class
class DSP(nn.Module):
    """Learnable LTI block: raw ``param`` -> ``map`` -> frequency response -> product with the
    input spectrum (dsp.py:212-352)."""

    def __init__(self, size: tuple, nfft: int = 2 ** 11, map: callable = _identity, requires_grad: bool = False,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        super().__init__()
        assert isinstance(size, tuple), "Size must be a tuple."
        self.size = size
        self.nfft = nfft
        self.map = map
        self.new_value = 0
        self.requires_grad = requires_grad
        self.device = device
        self.dtype = dtype
        self.param = nn.Parameter(torch.empty(self.size, device=device, dtype=dtype), requires_grad=requires_grad)
        # transforms along dim 0 (taps -> response), on the same HIP FFT as the signal path
        self.fft = lambda x: ops.rfft(x.unsqueeze(0), self.nfft).squeeze(0)
        self.ifft = lambda x: ops.irfft(x.unsqueeze(0), self.nfft).squeeze(0)
        self.alias_decay_db = torch.tensor(alias_decay_db, device=device, dtype=dtype)
        self.init_param()
        self.get_gamma()

    def forward(self, x, **kwargs):
        warnings.warn("Forward method not implemented. Input is returned.", UserWarning)
        return x

    def init_param(self):
        torch.nn.init.normal_(self.param)

    def get_gamma(self):
        self.gamma = 10 ** (-torch.abs(self.alias_decay_db) / self.nfft / 20)
        # float64 host copy used by the kernels: gamma is raised to powers up to nfft, so the
        # float32-rounded tensor above (kept for API parity) would cost 1e-4 of accuracy
        self._gamma_f = 10.0 ** (-abs(float(self.alias_decay_db)) / self.nfft / 20.0)

    def assign_value(self, new_value: torch.Tensor, indx: tuple = tuple([slice(None)])):
        assert (
            self.param[indx].shape == new_value.shape
        ), f"New values shape {new_value.shape} is not compatible with the parameter shape {self.param[indx].shape}."
        with torch.no_grad():
            self.param[indx].copy_(new_value)
            self.new_value = 1

    def probe(self, z: torch.Tensor):
        raise NotImplementedError(f"probe() not implemented for {self.__class__.__name__}")

    def probe_w(self, w: torch.Tensor):
        return self.probe(1 / w)

    # ---- shared by all subclasses
    def _run(self, x, ext_param):
        self.check_input_shape(x)
        if ext_param is None:
            return self.freq_convolve(x, self.param)
        with torch.no_grad():
            self.assign_value(ext_param)
        return self.freq_convolve(x, ext_param)

    def _gamma_on(self, t: torch.Tensor) -> torch.Tensor:
        return self.gamma.to(device=t.device)

    def _response_once(self, param):
        """freq_response(param), evaluated once per forward pass of the enclosing Shell: a Recursion applies
        its feedforward path twice (to the identity and to the signal) and the reference regenerates the
        (M, N_out, N_in) response each time.  Keyed by module, parameter tensor and its version counter; the
        memo lives in the Shell.forward scope (ops.fork_point), so nothing survives a parameter update.
        Reusing one response tensor at several places of the autograd graph is what autograd is for."""
        memo = ops.forward_memo()
        if memo is None or not torch.is_tensor(param):
            return self._response_in_order(param)
        key = (id(self), id(param), param._version, ops.bin_shard(self.nfft), ops.bin_order(self.nfft), torch.is_grad_enabled())
        hit = memo.get(key)
        if hit is None:
            H = self._response_in_order(param)
            ev = None
            if param.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(param.device))
            # the entry keeps `param` alive: the key holds its id(), which must not be recycled while the memo lives
            memo[key] = (H, ev, torch.cuda.current_stream(param.device).cuda_stream if param.is_cuda else 0, param)
            return H
        H, ev, produced_on, _ = hit
        if ev is not None:
            cur = torch.cuda.current_stream(param.device)
            if cur.cuda_stream != produced_on:
                cur.wait_event(ev)      # reuse on another stream (side-stream response build vs main): order it
        return H

    # Classes whose freq_response hands back the output of the cascade / integer-delay kernels untouched set this: those
    # kernels write the row-major bin order of the fused Shell pipeline themselves (ops.row_major_bins); every other
    # per-bin response is generated in natural order and reordered by one gather.
    _native_bin_order = False

    def _response_in_order(self, param):
        H = self.freq_response(param)
        if ops.bin_order(self.nfft) is None:
            return H
        per_bin = torch.is_tensor(H) and H.dim() >= 1 and H.shape[0] == self.nfft // 2 + 1 and H.dim() == (2 if getattr(self, "_diag", False) else 3)
        if not per_bin or (self._native_bin_order and self._native_now(param)):
            return H
        return ops.permute_bins(H, self.nfft)

    def _native_now(self, param) -> bool:
        """True when this call of freq_response went through the kernels that honour ops.row_major_bins"""
        return True

    # ---- protocol used by system.Series to fold adjacent per-bin modules into one pass
    def _bin_response(self, param):
        """(H, diag) such that forward(x) == ops.mimo(H, x, diag), or None if this module is not a
        plain per-bin product (or its freq_convolve was replaced by the user)."""
        return None

    def _fusable(self) -> bool:
        """True when applying this module is exactly ops.mimo(response, x): its freq_convolve is the library's own, its
        forward() is the library's own (a user subclass that overrides forward is called as the reference would call
        it) and no forward hooks are registered (the fused paths bypass Module.__call__)."""
        if getattr(self, "_own_convolve", None) is None or self.freq_convolve is not self._own_convolve:
            return False
        if type(self).forward not in _LIBRARY_FORWARDS:
            return False
        return not (self._forward_hooks or self._forward_pre_hooks)

    def _param_for_fusion(self, shape, ext_param):
        """Same checks and side effects as forward() (shape check, ext_param logging); returns the parameter
        tensor the response is to be built from."""
        from types import SimpleNamespace
        self.check_input_shape(SimpleNamespace(shape=tuple(shape)))
        if ext_param is None:
            return self.param
        with torch.no_grad():
            self.assign_value(ext_param)
        return ext_param

    def _response_for_fusion(self, shape, ext_param):
        """forward()'s checks, but returns the response instead of applying it."""
        return self._bin_response(self._param_for_fusion(shape, ext_param))


# ============================================================================ gains / matrices
class Gain(DSP):
    """Frequency-independent gain matrix, param (N_out, N_in) (dsp.py:357-496)."""

    _diag = False

    def __init__(self, size: tuple = (1, 1), nfft: int = 2 ** 11, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, nfft=nfft, map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)
        self.initialize_class()

    def forward(self, x, ext_param=None):
        return self._run(x, ext_param)

    def check_input_shape(self, x):
        if self.input_channels != x.shape[2]:
            raise ValueError(f"parameter shape = {self.size} not compatible with input signal of shape = ({x.shape}).")

    def check_param_shape(self):
        assert len(self.size) == 2, "gains must be 2D. For 1D (parallel) gains use parallelGain module."

    def get_freq_convolve(self):
        # (a real matrix goes to ops.mimo as it is: small ones are applied without forming their complex cast)
        self.freq_convolve = lambda x, param: ops.mimo(self._mapped(param), x, diag=self._diag)
        self._own_convolve = self.freq_convolve

    def _mapped(self, param):
        W = self.map(param)
        return W if (not self._diag and torch.is_tensor(W) and W.dim() == 2 and W.is_cuda) else to_complex(W)

    def _complex_mapped(self, param):
        """to_complex(map(param)) (dsp.py:466-468); the orthogonal map writes the complex matrix itself"""
        if (self.map is _orthogonal_map and param.is_cuda and param.dim() == 2 and param.shape[0] <= ops.EXPM_MAX_N
                and param.dtype in (torch.float32, torch.float64)):
            pair = _orthogonal_pair(param)
            return pair[1] if pair is not None else ops.matrix_exp(param, skew=True, complex_out=True)
        return to_complex(self.map(param))

    def _bin_response(self, param):
        # once per Shell.forward: a Recursion asks for its modules' responses once per loop structure it tries, and the
        # orthogonal map is a launch of its own (the memo entry keeps `param` alive, see DSP._response_once)
        memo = ops.forward_memo()
        if memo is None or not torch.is_tensor(param):
            return self._complex_mapped(param), self._diag
        key = (id(self), id(param), param._version, "mapped", torch.is_grad_enabled())
        hit = memo.get(key)
        if hit is None:
            hit = (self._complex_mapped(param), param)
            memo[key] = hit
        return hit[0], self._diag

    def _real_matrix(self, param):
        """the mapped parameter as the real matrix it is (the response is its complex cast, dsp.py:466-468)"""
        W = self.map(param)
        return None if (W.is_complex() or W.dim() != 2) else W

    def initialize_class(self):
        self.check_param_shape()
        self.get_io()
        self.get_freq_convolve()

    def get_io(self):
        self.input_channels = self.size[-1]
        self.output_channels = self.size[-2]

    def probe(self, z: torch.Tensor):
        return to_complex(self.map(self.param))


class parallelGain(Gain):
    """Per-channel gains, param (N,) (dsp.py:499-573)."""

    _diag = True

    def __init__(self, size: tuple = (1,), nfft: int = 2 ** 11, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, nfft=nfft, map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 1, "gains must be 1D, for 2D gains use Gain module."

    def get_io(self):
        self.input_channels = self.output_channels = self.size[-1]

    def probe(self, z: torch.Tensor):
        return torch.diag(to_complex(self.map(self.param)))


def _orthogonal_pair(x):
    """(real, complex) exp(skew(x)) shared by everybody who asks inside one ops.step_scope() -- the model (complex form, for
    the per-bin kernels) and e.g. a sparsity criterion on the same mixing matrix (real form); None outside a scope."""
    memo = ops.step_memo()
    if memo is None:
        return None
    key = ("orthogonal_map", id(x), x._version, torch.is_grad_enabled())
    hit = memo.get(key)
    if hit is None:
        hit = (*ops.matrix_exp_both(x, skew=True), x)       # the entry keeps x alive: its id() is in the key
        memo[key] = hit
    return hit


def _orthogonal_map(x):
    """exp of the skew part (dsp.py:649).  On the GPU: one fused launch each way (ops.matrix_exp);
    otherwise the same fixed schedule written with torch ops (torch.matrix_exp synchronises with
    the host and cannot be captured in a HIP graph)."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] <= ops.EXPM_MAX_N and x.dtype in (torch.float32, torch.float64):
        pair = _orthogonal_pair(x)      # inside ops.step_scope(): once per parameter version, real and complex forms
        return pair[0] if pair is not None else ops.matrix_exp(x, skew=True)
    return matrix_exp_capturable(skew_matrix(x))


class Matrix(Gain):
    """Gain matrix with a structural map: random | orthogonal | hadamard | rotation (dsp.py:579-676)."""

    def __init__(self, size: tuple = (1, 1), nfft: int = 2 ** 11, map: callable = _identity,
                 matrix_type: str = "random", iter: int = 1, requires_grad: bool = False,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self.matrix_type = matrix_type
        self.iter = iter
        super().__init__(size=size, nfft=nfft, map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)

    def matrix_gallery(self):
        N = self.size[0]
        kind = self.matrix_type
        if kind == "random":
            self.map = _identity
        elif kind == "orthogonal":
            assert N == self.size[1], "Matrix must be square to be orthogonal"
            # exp of the skew part (dsp.py:649); fixed-schedule evaluation so that a step can be
            # captured in a HIP graph (torch.matrix_exp synchronises with the host)
            self.map = _orthogonal_map
        elif kind == "hadamard":
            assert N == self.size[1], "Matrix must be square to be Hadamard"
            assert N % 2 == 0, "Matrix must have even dimensions to be Hadamard"
            self.map = lambda x: HadamardMatrix(self.size[0], device=x.device, dtype=self.dtype)(x)
        elif kind == "rotation":
            assert N == self.size[1], "Matrix must be square to be a rotation matrix"
            assert N % 2 == 0, "Matrix must have even dimensions to be a rotation matrix"
            self.map = lambda x: RotationMatrix(self.size[0], self.iter, device=x.device, dtype=self.dtype)([x[0][0]])

    def initialize_class(self):
        self.check_param_shape()
        self.get_io()
        self.matrix_gallery()
        self.get_freq_convolve()


class HouseholderMatrix(Gain):
    """U = I - 2 u u^T applied as two rank-1 products (dsp.py:679-782)."""

    def __init__(self, size: tuple = (1, 1), nfft: int = 2 ** 11, requires_grad: bool = False,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        assert size[0] == size[1], "Matrix must be square"
        unit = lambda x: to_complex(x) / torch.norm(x, dim=0, keepdim=True)  # noqa: E731
        super().__init__(size=(size[0], 1), nfft=nfft, map=unit, requires_grad=requires_grad,
                         alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def _bin_response(self, param):
        return None

    def _fusable(self) -> bool:
        return False

    def forward(self, x, ext_param=None):
        self.check_input_shape(x)
        if ext_param is None:
            u = self.map(self.param)
        else:
            with torch.no_grad():
                self.assign_value(ext_param)
            u = self.map(ext_param)
        uTx = ops.mimo(u.transpose(1, 0), x)       # (B, M, 1, ...)
        return x - 2 * ops.mimo(u, uTx)

    def check_input_shape(self, x):
        if self.size[0] != x.shape[2]:
            raise ValueError(f"parameter shape = {self.size} not compatible with input signal of shape = ({x.shape}).")

    def get_io(self):
        self.input_channels = self.output_channels = self.size[0]


# ============================================================================ filters
class Filter(DSP):
    """FIR filter matrix, param (taps, N_out, N_in) (dsp.py:788-962)."""

    _diag = False

    def __init__(self, size: tuple = (1, 1, 1), nfft: int = 2 ** 11, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, nfft=nfft, map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)
        self.initialize_class()

    def forward(self, x, ext_param=None):
        return self._run(x, ext_param)

    def check_input_shape(self, x):
        if (ops.bin_shard(self.nfft)[1], self.input_channels) != (x.shape[1], x.shape[2]):
            raise ValueError(f"parameter shape not compatible with input signal of shape = ({x.shape}).")

    def check_param_shape(self):
        assert len(self.size) == 3, "Filter must be 3D, for 2D (parallel) filters use ParallelFilter module."

    def get_freq_response(self):
        """taps * gamma^n  ->  rfft(nfft) along the tap axis (dsp.py:893-908)."""
        self.ir = lambda x: self.map(x)

        def response(param):
            h = self.ir(param)
            n = torch.arange(0, h.shape[0], device=h.device, dtype=torch.float64)
            env = (self._gamma_f ** n).to(h.dtype).view(-1, *([1] * (h.dim() - 1)))
            H = self.fft(h * env)
            bin0, m_local = ops.bin_shard(self.nfft)      # bin-sharded execution keeps the local range
            return H if m_local == H.shape[0] else H[bin0:bin0 + m_local]

        self.freq_response = response

    def get_freq_convolve(self):
        self.freq_convolve = lambda x, param: ops.mimo(self._response_once(param), x, diag=self._diag)
        self._own_convolve = self.freq_convolve

    def _bin_response(self, param):
        return self._response_once(param), self._diag

    def initialize_class(self):
        self.check_param_shape()
        self.get_io()
        self.get_freq_response()
        self.get_freq_convolve()

    def get_io(self):
        if self._diag:
            self.input_channels = self.output_channels = self.size[-1]
        else:
            self.input_channels = self.size[-1]
            self.output_channels = self.size[-2]

    def _probe_fir(self, z):
        coeff = self.map(self.param)
        k = torch.arange(coeff.shape[0], device=coeff.device, dtype=coeff.dtype)
        w = (self.gamma ** k) * z ** (-k)
        return (to_complex(coeff) * w.view(-1, *([1] * (coeff.dim() - 1)))).sum(dim=0)

    def probe(self, z: torch.Tensor):
        return self._probe_fir(z)


class parallelFilter(Filter):
    """Per-channel FIR filters, param (taps, N) (dsp.py:965-1049)."""

    _diag = True

    def __init__(self, size: tuple = (1, 1), nfft: int = 2 ** 11, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, nfft=nfft, map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 2, "Filter must be 1D, for 2D filters use Filter module."

    def probe(self, z: torch.Tensor):
        return torch.diag(self._probe_fir(z))


# cascade-type filters applied to few columns: see _SOSMixin._apply_narrow
NARROW_APPLY = True


class _SOSMixin:
    """Second-order-section cascades share one tail: weight the 3 taps by gamma^[0,1,2] and
    evaluate prod B / prod A per bin (dsp.py:1520-1526) -- here directly in ``ops.sos_response``,
    without building the (M, sections, ...) tensors."""

    _native_bin_order = True     # every response of these classes comes straight out of the cascade kernels

    def _sos_to_response(self, b, a):
        return ops.sos_response(b, a, self._gamma_f, self.nfft, dtype=self.dtype)

    def _response_times_matrix(self, param, Wr):
        """freq_response(param) @ Wr (a real constant matrix on the right) as one operator whose backward folds the
        composition into the cascade kernel; None when this module / dtype has no such path."""
        if self._diag or not param.is_cuda:
            return None
        if getattr(self, "_own_response", None) is not self.freq_response:
            return None
        spec = self._cascade_spec(param)
        # sections, cascades per output row, columns of the constant factor: the fused operator's kernel limits (LDS tables)
        n_sections = spec[1].shape[0] if spec[0] == "geq" else spec[1].shape[1]
        if not ops.cascade_rc_supported(self.dtype, Wr.shape[1], Wr.shape[0], n_sections):
            return None
        if spec[0] == "geq":
            return ops.geq_cascade_rc(spec[1], spec[2], Wr, self._gamma_f, self.nfft, dtype=self.dtype, gain_map=spec[3])
        return ops.sos_response_rc(spec[1], spec[2], Wr, self._gamma_f, self.nfft, dtype=self.dtype)

    def _cascade_spec(self, param):
        return ("sos", *self._sos_coeffs(self.map(param.double())))

    # A full cascade applied to a signal with this many columns or fewer, outside a loop: the product and the response's
    # gradient go through ops.*_apply (dL/dH = gY (x) conj(X) formed inside the cascade backward; an (M, N_out, N_in)
    # gradient tensor is 1.6 GB for a 32 x 32 equaliser at nfft = 384000)
    NARROW_APPLY_MAX_COLUMNS = 2
    NARROW_APPLY_MIN_PAIRS = 64

    def get_freq_convolve(self):
        def convolve(x, param):
            Y = self._apply_narrow(x, param)
            return Y if Y is not None else ops.mimo(self._response_once(param), x, diag=self._diag)
        self.freq_convolve = convolve
        self._own_convolve = self.freq_convolve

    def _apply_narrow(self, x, param):
        if (self._diag or not NARROW_APPLY or not torch.is_tensor(param) or not param.is_cuda or x.dim() != 3
                or x.shape[0] > self.NARROW_APPLY_MAX_COLUMNS or ops.in_loop() or ops.bin_order(self.nfft) is not None
                or self.input_channels * self.output_channels < self.NARROW_APPLY_MIN_PAIRS
                or not ops.cascade_apply_supported(self.dtype, x)
                or getattr(self, "_own_response", None) is not self.freq_response):
            return None
        spec = self._cascade_spec(param)
        if spec[0] == "geq":
            return ops.geq_cascade_apply(spec[1], spec[2], x, self._gamma_f, self.nfft, dtype=self.dtype, gain_map=spec[3])
        if spec[1].dim() != 4:
            return None
        return ops.sos_response_apply(spec[1], spec[2], x, self._gamma_f, self.nfft, dtype=self.dtype)

    def _sections_spectra(self, b, a):
        """B, A as the reference returns them from get_poly_coeff: rfft of the weighted taps."""
        env = self.alias_envelope_dcy.to(b.device).view(3, *([1] * (b.dim() - 1)))
        return self.fft(b.to(self.dtype) * env), self.fft(a.to(self.dtype) * env)

    def get_freq_response(self):
        # the few hundred parameter scalars are mapped to coefficients in float64 whatever the
        # module dtype: the cascade is ill-conditioned in its coefficients at low frequency and
        # the float64 reference is the parity target
        self.freq_response = lambda param: self._sos_to_response(*self._sos_coeffs(self.map(param.double())))
        self._own_response = self.freq_response

    def get_poly_coeff(self, param):
        """(H, B, A) for *mapped* parameters, as in the reference.  H comes from the fused kernel;
        B and A (per-section spectra) are only materialised by this inspection method."""
        b, a = self._sos_coeffs(param)
        B, A = self._sections_spectra(b, a)
        return self._sos_to_response(b, a), B, A


class Biquad(_SOSMixin, Filter):
    """Cascaded RBJ biquads, param (n_sections, 2|3, N_out, N_in): (fc/nyquist[, fc2], gain)
    (dsp.py:1353-1603)."""

    def __init__(self, size: tuple = (1, 1), n_sections: int = 1, filter_type: str = "lowpass",
                 nfft: int = 2 ** 11, fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        assert filter_type in ["lowpass", "highpass", "bandpass"], "Invalid filter type"
        self.n_sections = n_sections
        self.filter_type = filter_type
        self.fs = fs
        self.device = device
        self.dtype = dtype
        self.get_map()
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device,
                                                                                             dtype=dtype)
        super().__init__(size=(n_sections, *self.get_size(), *size), nfft=nfft, map=self.map,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def get_size(self):
        return (3,) if self.filter_type == "bandpass" else (2,)

    def get_map(self):
        """(fc, gain) -> clamp(stack(fc, 20 log10|gain|), [0,-60], [1,60]) (dsp.py:1528-1563)."""
        bp = self.filter_type == "bandpass"

        def mapping(x):
            dt, dev = x.dtype, x.device
            gain_db = 20 * torch.log10(torch.abs(x[:, -1]))
            if bp:
                e = torch.finfo(dt).eps
                y = torch.stack((x[:, 0], x[:, 1], gain_db), dim=1)
                lo = torch.tensor([0 + e, 0 + e, -60], device=dev, dtype=dt)
                hi = torch.tensor([1 - e, 1 - e, 60], device=dev, dtype=dt)
            else:
                y = torch.stack((x[:, 0], gain_db), dim=1)
                lo = torch.tensor([0, -60], device=dev, dtype=dt)
                hi = torch.tensor([1, 60], device=dev, dtype=dt)
            shp = [1, -1] + [1] * (x.dim() - 2)
            return torch.clamp(y, min=lo.view(shp).expand_as(y), max=hi.view(shp).expand_as(y))

        self.map = mapping

    def _sos_coeffs(self, p):
        """mapped params -> RBJ (b, a), each (3, n_sections, ...) (dsp.py:1494-1519)."""
        hz = lambda r: rad2hertz(r * torch.pi, fs=self.fs)  # noqa: E731
        kw = dict(fs=self.fs, device=p.device, dtype=p.dtype)
        if self.filter_type == "lowpass":
            return lowpass_filter(fc=hz(p[:, 0]), gain=p[:, 1], **kw)
        if self.filter_type == "highpass":
            return highpass_filter(fc=hz(p[:, 0]), gain=p[:, 1], **kw)
        return bandpass_filter(fc1=hz(p[:, 0]), fc2=hz(p[:, 1]), gain=p[:, 2], **kw)

    def init_param(self):
        torch.nn.init.uniform_(self.param[:, 0], a=0, b=0.5)
        if self.filter_type == "bandpass":
            torch.nn.init.uniform_(self.param[:, 1], a=self.param[:, 0].max().item(), b=1)
        torch.nn.init.uniform_(self.param[:, -1], a=-1, b=1)

    def check_param_shape(self):
        assert len(self.size) == 4, "Parameter size must be 4D, for 3D (parallel) biquads use parallelBiquad module."


class parallelBiquad(Biquad):
    """Per-channel biquad cascades, param (n_sections, 2|3, N) (dsp.py:1607-1764)."""

    _diag = True

    def __init__(self, size: tuple = (1,), n_sections: int = 1, filter_type: str = "lowpass", nfft: int = 2 ** 11,
                 fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        super().__init__(size=size, n_sections=n_sections, filter_type=filter_type, nfft=nfft, fs=fs,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 3, "Parameter size must be 3D, for 3D sapce use Biquad module."


def _db_of_magnitude(x):
    """GEQ's default parameter map (dsp.py:2526); a named function so that the module can tell the
    default from a user map and fold it into the design kernel."""
    return 20 * torch.log10(torch.abs(x))


def db_of_sigmoid(x):
    """20 log10(sigmoid(x)): the map the reference's FDN examples give their attenuation filters as a lambda
    (e8_fdn.py:97).  Passed as THIS function, the graphic equalisers fold it into the design kernel with its backward
    (fl_geq_sections in_kind 3 / 4) as they do their default map; any other callable runs as torch ops."""
    return 20 * torch.log10(torch.sigmoid(x))


_FOLDED_GAIN_MAPS = {_db_of_magnitude: "abs", db_of_sigmoid: "sigmoid"}
_MAP_KIND_CACHE = weakref.WeakKeyDictionary()


def _gain_map_kind(fn):
    """"abs" / "sigmoid" when `fn` IS one of the two maps the design kernels fold (fl_geq_sections in_kind 1..4), or -- an
    anonymous function, as the reference's examples pass them: `map=lambda x: 20 * torch.log10(torch.sigmoid(x))`,
    e8_fdn.py:97 -- when it is a stateless plain function over {torch, abs, log10, sigmoid, 20} (`_is_stateless_function_of`:
    no closure, no defaults, no other name in its bytecode) AND reproduces one of them bit for bit, values and gradient, on a
    float64 probe vector (both signs, 1e-30 ... 1e30).  None otherwise: the callable then runs as torch ops in front of the
    design kernel, whatever it does."""
    kind = _FOLDED_GAIN_MAPS.get(fn)
    if kind is not None or fn is None:
        return kind
    if not _is_stateless_function_of(fn):
        return None
    try:
        return _MAP_KIND_CACHE[fn]
    except KeyError:
        pass
    kind = None
    try:
        probe = torch.tensor([-200.0, -31.5, -7.25, -2.0, -0.75, -1e-3, -1e-30, 1e-30, 1e-6, 0.0625, 0.5, 0.9990234375, 1.0, 1.4142,
                              3.0, 12.5, 40.0, 200.0, 1e9, 1e30, -1e30], dtype=torch.float64)
        with torch.enable_grad():
            x = probe.clone().requires_grad_(True)
            y = fn(x)
            ok = torch.is_tensor(y) and y.shape == x.shape and y.dtype == x.dtype and y.requires_grad
            if ok:
                (g,) = torch.autograd.grad(y.sum(), [x])
                for named, k in _FOLDED_GAIN_MAPS.items():
                    xr = probe.clone().requires_grad_(True)
                    yr = named(xr)
                    (gr,) = torch.autograd.grad(yr.sum(), [xr])
                    # equal_nan: sigmoid saturates at the extremes of the probe (log10(0) = -inf, gradient nan) in both alike
                    if torch.equal(torch.nan_to_num(y.detach(), nan=7.0), torch.nan_to_num(yr.detach(), nan=7.0)) and \
                            torch.equal(torch.nan_to_num(g, nan=7.0), torch.nan_to_num(gr, nan=7.0)):
                        kind = k
                        break
    except Exception:       # a function that does not take a float64 host vector is simply not one of the two
        kind = None
    _MAP_KIND_CACHE[fn] = kind
    return kind


class GEQ(_SOSMixin, Filter):
    """Graphic equaliser: param (n_bands+3 command gains, N_out, N_in), default map 20 log10|x|
    (dsp.py:2467-2611).  Sections: flat gain, low shelf, octave peaks (R = 2.7), high shelf."""

    def __init__(self, size: tuple = (1, 1), octave_interval: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 map: callable = _db_of_magnitude, requires_grad: bool = False,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32,
                 *, fold_map: bool = True):
        self.fold_map = fold_map        # False: the map always runs as torch ops in front of the design kernel
        self.octave_interval = octave_interval
        self.fs = fs
        self.center_freq, self.shelving_crossover = eq_freqs(interval=octave_interval)
        self.n_gains = len(self.center_freq) + 3
        self._design = GEQDesign(self.center_freq, self.shelving_crossover, fs=fs, R=2.7)
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device,
                                                                                             dtype=dtype)
        super().__init__(size=(self.n_gains, *size), nfft=nfft, map=map, requires_grad=requires_grad,
                         alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def init_param(self):
        torch.nn.init.uniform_(self.param, a=10 ** (-6 / 20), b=10 ** (6 / 20))

    def check_param_shape(self):
        assert len(self.size) == 3, "Filter must be 3D, for 2D (parallel) filters use ParallelGEQ module."

    def _folded_map(self):
        """the design kernel's name for self.map ("abs" / "sigmoid"), or None: run the callable (see "recognised callables")"""
        if not getattr(self, "fold_map", True):
            return None
        gm = _gain_map_kind(self.map)
        if gm is not None and self.map not in _FOLDED_GAIN_MAPS:
            _warn_substitution(self.map, "20*log10(|x|)" if gm == "abs" else "20*log10(sigmoid(x))")
        return gm

    def get_freq_response(self):
        def response(param):
            gm = self._folded_map()
            if gm is not None and param.is_cuda and param.dtype in (torch.float32, torch.float64):
                # default map: 10^(map(x)/20) = |x| (or sigmoid(x)), folded into the design kernel with its backward
                return ops.geq_cascade(param, self._design.device_consts(param.device), self._gamma_f, self.nfft,
                                       dtype=self.dtype, gain_map=gm)
            return self._sos_to_response(*self._sos_coeffs(self.map(param.double())))
        self.freq_response = response
        self._own_response = response

    def _cascade_spec(self, param):
        gm = self._folded_map()
        if gm is not None and param.is_cuda and param.dtype in (torch.float32, torch.float64):
            return ("geq", param, self._design.device_consts(param.device), gm)
        return ("sos", *self._sos_coeffs(self.map(param.double())))

    def _sos_coeffs(self, gain_db):
        """command gains in dB -> float32 SOS (b, a) for every channel pair at once; the
        reference loops over pairs in Python calling eq.geq (dsp.py:2573-2585)."""
        if gain_db.is_cuda:
            return ops.geq_sections(gain_db, self._design.device_consts(gain_db.device))
        return self._design.sections(gain_db)      # host-side evaluation (inspection on CPU tensors)


class parallelGEQ(GEQ):
    """Per-channel graphic equaliser, param (n_bands+3, N) (dsp.py:2614-2692)."""

    _diag = True

    def __init__(self, size: tuple = (1,), octave_interval: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 map: callable = _db_of_magnitude, requires_grad: bool = False,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32,
                 *, fold_map: bool = True):
        super().__init__(size=size, octave_interval=octave_interval, nfft=nfft, fs=fs, map=map,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype,
                         fold_map=fold_map)

    def check_param_shape(self):
        assert len(self.size) == 2, "Filter must be 2D, for 3D filters use GEQ module."


class AccurateGEQ(_SOSMixin, Filter):
    """Graphic equaliser whose command gains are fitted so that the cascade interpolates the TARGET gains:
    param (n_bands + 2 target gains, N_out, N_in), map 20 log10(x); not learnable (dsp.py:3002-3123).
    The fit (functional.accurate_geq: bounded least squares by L-BFGS on a dozen numbers, float32 as in
    the reference) runs on the host once per parameter value -- the reference repeats it for every
    channel pair on every forward call; the cascade itself is evaluated by ops.sos_response."""

    def __init__(self, size: tuple = (1, 1), octave_interval: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 map: callable = lambda x: 20 * torch.log10(x), alias_decay_db: float = 0.0,
                 start_freq: float = 31.25, end_freq: float = 16000.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        self.octave_interval = octave_interval
        self.fs = fs
        self.center_freq, self.shelving_crossover = eq_freqs(interval=octave_interval, start_freq=start_freq,
                                                             end_freq=end_freq)
        self.n_gains = len(self.center_freq) + 2
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device,
                                                                                             dtype=dtype)
        self._design_cache = (None, None)
        super().__init__(size=(self.n_gains, *size), nfft=nfft, map=map, requires_grad=False,
                         alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def init_param(self):
        torch.nn.init.uniform_(self.param, a=10 ** (-6 / 20), b=10 ** (6 / 20))

    def check_param_shape(self):
        assert len(self.size) == 3, "Filter must be 3D, for 2D (parallel) filters use ParallelGEQ module."

    def _sos_coeffs(self, target_db):
        """target gains in dB (n_gains, chan...) -> float32-valued SOS (b, a), each (3, n_gains + 1, chan...)"""
        host = target_db.detach().to("cpu")
        key = (tuple(host.shape), str(host.dtype), host.contiguous().numpy().tobytes())
        if self._design_cache[0] != key:
            chan = tuple(host.shape[1:])
            flat = host.reshape(host.shape[0], -1)
            bs, as_ = [], []
            for c in range(flat.shape[1]):
                b, a = accurate_geq(flat[:, c], self.center_freq, self.shelving_crossover, fs=self.fs)
                bs.append(b)
                as_.append(a)
            b = torch.stack(bs, dim=-1).reshape(3, -1, *chan)
            a = torch.stack(as_, dim=-1).reshape(3, -1, *chan)
            self._design_cache = (key, (b, a))
        b, a = self._design_cache[1]
        return b.to(target_db.device), a.to(target_db.device)


class parallelAccurateGEQ(AccurateGEQ):
    """Per-channel AccurateGEQ, param (n_bands + 2, N) (dsp.py:3126-3220)."""

    _diag = True

    def __init__(self, size: tuple = (1,), octave_interval: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 map: callable = lambda x: 20 * torch.log10(x), alias_decay_db: float = 0.0,
                 start_freq: float = 31.25, end_freq: float = 16000.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, octave_interval=octave_interval, nfft=nfft, fs=fs, map=map,
                         alias_decay_db=alias_decay_db, start_freq=start_freq, end_freq=end_freq, device=device,
                         dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 2, "Filter must be 2D, for 3D filters use GEQ module."


class SOSFilter(_SOSMixin, Filter):
    """Cascade of second-order sections given directly by their coefficients, param
    (n_sections, 6, N_out, N_in) = [b0, b1, b2, a0, a1, a2]; optional a0 normalisation
    (dsp.py:1767-1964).  Not learnable in the reference (requires_grad=False)."""

    def __init__(self, size: tuple = (1, 1), n_sections: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32,
                 normalize_a0: bool = True):
        self.n_sections = n_sections
        self.fs = fs
        self.device = device
        self.dtype = dtype
        self.normalize_a0 = normalize_a0
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device)
        self.get_map()
        super().__init__(size=(n_sections, *self.get_size(), *size), nfft=nfft, map=self.map, requires_grad=False,
                         alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def get_size(self):
        return (6,)

    def get_map(self):
        def mapping(x: torch.Tensor) -> torch.Tensor:
            if not self.normalize_a0:
                return x
            a0 = x[:, 3:4]
            eps = torch.finfo(x.dtype).eps
            a0_safe = torch.where(torch.abs(a0) > eps, a0, eps * torch.ones_like(a0))
            y = x / a0_safe                       # every coefficient divided by a0 ...
            return torch.cat((y[:, :3], torch.ones_like(a0), y[:, 4:]), dim=1)   # ... and a0 itself set to 1

        self.map = mapping

    def init_param(self):
        with torch.no_grad():
            self.param.zero_()
            self.param[:, 0] = 1.0
            self.param[:, 3] = 1.0

    def check_param_shape(self):
        assert len(self.size) == 4, "Parameter size must be 4D, expected (K, 6, N_out, N_in)."
        assert self.size[1] == 6, "Second dimension must be 6: [b0,b1,b2,a0,a1,a2]."

    def _sos_coeffs(self, p):
        return p[:, 0:3].transpose(0, 1), p[:, 3:6].transpose(0, 1)     # (3, K, ...)

    def _probe_sos(self, z):
        p = self.map(self.param)
        g = self.alias_envelope_dcy.to(p.device)
        zi = z ** (-1)
        H = None
        for k in range(p.shape[0]):
            Bk = to_complex(p[k, 0]) * g[0] + to_complex(p[k, 1]) * g[1] * zi + to_complex(p[k, 2]) * g[2] * zi ** 2
            Ak = to_complex(p[k, 3]) * g[0] + to_complex(p[k, 4]) * g[1] * zi + to_complex(p[k, 5]) * g[2] * zi ** 2
            H = Bk / Ak if H is None else H * Bk / Ak
        return H

    def probe(self, z: torch.Tensor):
        return self._probe_sos(z)


class parallelSOSFilter(SOSFilter):
    """Per-channel SOS cascades, param (n_sections, 6, N) (dsp.py:1967-2073)."""

    _diag = True

    def __init__(self, size: tuple = (1,), n_sections: int = 1, nfft: int = 2 ** 11, fs: int = 48000,
                 alias_decay_db: float = 0.0, device: Optional[str] = None, dtype: torch.dtype = torch.float32,
                 normalize_a0: bool = True):
        super().__init__(size=size, n_sections=n_sections, nfft=nfft, fs=fs, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype, normalize_a0=normalize_a0)

    def check_param_shape(self):
        assert len(self.size) == 3, "Parameter size must be 3D, expected (K, 6, N)."
        assert self.size[1] == 6, "Second dimension must be 6: [b0,b1,b2,a0,a1,a2]."

    def probe(self, z: torch.Tensor):
        return torch.diag(self._probe_sos(z))


class SVF(_SOSMixin, Filter):
    """Cascaded state-variable filters, param (5, n_sections, N_out, N_in) = raw (f, R, mLP, mBP, mHP)
    (dsp.py:2076-2366).  The map turns the raw values into (tan(pi*sigmoid/2), softplus/ln2, mix)."""

    _TYPES = ["lowpass", "highpass", "bandpass", "lowshelf", "highshelf", "peaking", "notch", None]

    def __init__(self, size: tuple = (1, 1), n_sections: int = 1, filter_type: str = None, nfft: int = 2 ** 11,
                 fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self.fs = fs
        self.n_sections = n_sections
        assert filter_type in self._TYPES, "Invalid filter type"
        self.filter_type = filter_type
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device,
                                                                                             dtype=dtype)
        super().__init__(size=(5, self.n_sections, *size), nfft=nfft, map=self.map_param2svf,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 4, "Filter parameter space must be 4D, for 3D (parallel) filters use parallelSVF module."

    def param2freq(self, param):
        return torch.tan(torch.pi * torch.div(1, 1 + torch.exp(-param)) * 0.5)

    def param2R(self, param):
        one = torch.ones(1, device=param.device, dtype=param.dtype)
        return torch.div(torch.log(one + torch.exp(param)), torch.log(torch.tensor(2, device=param.device, dtype=param.dtype)))

    def param2mix(self, param, R=None):
        G = 10 ** (-nn.functional.softplus(param[0]))
        one, zero = torch.ones_like(G), torch.zeros_like(G)
        ft = self.filter_type
        if ft == "lowpass":
            return torch.stack((one, zero, zero))
        if ft == "highpass":
            return torch.stack((zero, zero, one))
        if ft == "bandpass":
            return torch.stack((zero, one, zero))
        if ft == "lowshelf":
            return torch.stack((one, 2 * R * torch.sqrt(G), G * one))
        if ft == "highshelf":
            return torch.stack((G * one, 2 * R * torch.sqrt(G), one))
        if ft in ("peaking", "notch"):
            return torch.stack((one, 2 * R * torch.sqrt(G), one))
        bias = torch.ones(param.shape, device=param.device, dtype=param.dtype)   # general SVF: raw mix + (1, 2, 1)
        bias[1] = 2 * torch.ones(param.shape[1:], device=param.device, dtype=param.dtype)
        return param + bias

    def map_param2svf(self, param):
        f = self.param2freq(param[0])
        r = self.param2R(param[1])
        if self.filter_type == "peaking":
            R = 1 / r
            m = self.param2mix(param[2:], r)
        else:
            R = r
            m = self.param2mix(param[2:], R)
        return f, R, m[0], m[1], m[2]

    def _sos_coeffs(self, mapped):
        f, R, mLP, mBP, mHP = mapped
        f2 = f ** 2
        b = torch.stack((f2 * mLP + f * mBP + mHP, 2 * f2 * mLP - 2 * mHP, f2 * mLP - f * mBP + mHP))
        a = torch.stack((f2 + 2 * R * f + 1, 2 * f2 - 2, f2 - 2 * R * f + 1))
        return b.float(), a.float()     # the reference stores the sections in float32 buffers (dsp.py:2217-2218)


class parallelSVF(SVF):
    """Per-channel SVF cascades, param (5, n_sections, N) (dsp.py:2369-2464)."""

    _diag = True

    def __init__(self, size: tuple = (1,), n_sections: int = 1, filter_type: str = None, nfft: int = 2 ** 11,
                 fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        super().__init__(size=size, n_sections=n_sections, filter_type=filter_type, nfft=nfft, fs=fs,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 3, "Filter parameter space must be 3D, for 4D filters use SVF module."


class PEQ(_SOSMixin, Filter):
    """Parametric equaliser: low shelf, n_bands-2 peaking filters, high shelf; param
    (n_bands, 3, N_out, N_in) = raw (frequency, resonance, gain dB); design "biquad" | "svf"
    (dsp.py:2695-2872)."""

    def __init__(self, size: tuple = (1, 1), n_bands: int = 10, f_min: float = 20, f_max: float = 20000,
                 design: str = "biquad", fs: int = 48000, nfft: int = 2 ** 11, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        self.n_bands = n_bands
        self.design = design
        self.fs = fs
        self.f_min = f_min
        self.f_max = f_max
        k = torch.arange(1, n_bands + 1, dtype=dtype)
        self.center_freq_bias = f_min * (f_max / f_min) ** ((k - 1) / (n_bands - 1))
        self.alias_envelope_dcy = _gamma(alias_decay_db, nfft, device, dtype) ** torch.arange(0, 3, 1, device=device,
                                                                                             dtype=dtype)
        super().__init__(size=(n_bands, 3, *size), nfft=nfft, map=map, requires_grad=requires_grad,
                         alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def init_param(self):
        torch.nn.init.uniform_(self.param)

    def check_param_shape(self):
        assert len(self.size) == 4, "Filter must be 3D, for 2D (parallel) filters use ParallelPEQ module."

    def map_eq(self, param):
        """raw (n_bands, 3, ...) -> (f, R, G), each (n_bands, ...) (dsp.py:2836-2855)."""
        R, G = param[:, 1], param[:, 2]
        cfb = self.center_freq_bias.to(device=param.device, dtype=param.dtype).view(-1, *([1] * (param.dim() - 2)))
        if self.design == "biquad":
            bias = cfb / self.fs * 2 * torch.pi
            f = torch.clamp(torch.sigmoid(param[:, 0]) + bias, min=2 * torch.pi * self.f_min / self.fs,
                            max=2 * torch.pi * self.f_max / self.fs)
        else:
            bias = torch.log(2 * cfb / self.fs / (1 - 2 * cfb / self.fs))
            f = torch.tan(torch.pi * torch.sigmoid(param[:, 0] + bias) * 0.5)
        return f, R, G

    def _band_coeffs(self, f, R, G, kind):
        """(a, b) taps of one band type for all channels at once (dsp.py:2775-2832)."""
        if self.design == "svf":
            G = 10 ** (G / 20)
            one = torch.ones_like(G)
            mBP = 2 * R * torch.sqrt(G)
            mLP, mHP = (one, one) if kind == "peaking" else ((G, one) if kind == "lowshelf" else (one, G))
            f2 = f ** 2
            b = (f2 * mLP + f * mBP + mHP, 2 * f2 * mLP - 2 * mHP, f2 * mLP - f * mBP + mHP)
            a = (f2 + 2 * R * f + 1, 2 * f2 - 2, f2 - 2 * R * f + 1)
        else:
            G = 10 ** (G / 40)
            c = torch.cos(f)
            if kind == "peaking":
                alpha = torch.sin(f) / (2 * R)
                b = (1 + alpha * G, -2 * c, 1 - alpha * G)
                a = (1 + alpha / G, -2 * c, 1 - alpha / G)
            else:
                alpha = torch.sin(f) * torch.sqrt((G ** 2 + 1) * (1 / R - 1) + 2 * G)
                sg = 1.0 if kind == "lowshelf" else -1.0
                b = (G * ((G + 1) - sg * (G - 1) * c + alpha), sg * 2 * G * ((G - 1) - sg * (G + 1) * c),
                     G * ((G + 1) - sg * (G - 1) * c - alpha))
                a = ((G + 1) + sg * (G - 1) * c + alpha, -sg * 2 * ((G - 1) + sg * (G + 1) * c),
                     (G + 1) + sg * (G - 1) * c - alpha)
        return torch.stack(a), torch.stack(b)

    def compute_biquad_coeff(self, f, R, G, type="peaking"):
        """(a, b), each (*f.shape, 3): the taps of one band type, as the reference's method of the same name returns
        them (dsp.py:2790-2842: float32 buffers, tap index last)."""
        a, b = self._band_coeffs(f, R, G, type)
        return a.movedim(0, -1).float(), b.movedim(0, -1).float()

    def _sos_coeffs(self, mapped):
        f, R, G = self.map_eq(mapped)
        a_lo, b_lo = self._band_coeffs(f[0], R[0], G[0], "lowshelf")
        a_hi, b_hi = self._band_coeffs(f[-1], R[-1], G[-1], "highshelf")
        a_pk, b_pk = self._band_coeffs(f[1:-1], R[1:-1], G[1:-1], "peaking")          # (3, n_bands-2, ...)
        b = torch.cat((b_lo.unsqueeze(1), b_pk, b_hi.unsqueeze(1)), dim=1)
        a = torch.cat((a_lo.unsqueeze(1), a_pk, a_hi.unsqueeze(1)), dim=1)
        return b.float(), a.float()     # float32 section buffers in the reference (dsp.py:2750-2751)


class parallelPEQ(PEQ):
    """Per-channel parametric equaliser, param (n_bands, 3, N) (dsp.py:2875-3000).  NB: the reference's
    parallel `map_eq` broadcasts its frequency bias as (n_bands, 1, 1) against (n_bands, N) parameters and
    cannot run; this class applies the bias per band, as the non-parallel class does."""

    _diag = True

    def __init__(self, size: tuple = (1,), n_bands: int = 10, f_min: float = 20, f_max: float = 20000,
                 design: str = "biquad", nfft: int = 2 ** 11, fs: int = 48000, map: callable = _identity,
                 requires_grad: bool = False, alias_decay_db: float = 0.0, device: Optional[str] = None,
                 dtype: torch.dtype = torch.float32):
        super().__init__(size=size, n_bands=n_bands, f_min=f_min, f_max=f_max, design=design, fs=fs, nfft=nfft,
                         map=map, requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device,
                         dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 3, "Filter must be 2D in the parallel configuration, for 3D filters use PEQ module."


# ============================================================================ delays
class Delay(DSP):
    """Delay matrix, param (N_out, N_in) in units of ``unit/fs`` seconds (dsp.py:3226-3450).

    ``isint=True``: the delay in samples is rounded (half to even) and the response
    gamma^m exp(-j 2 pi k m / nfft) is generated with the phase index (k*m) mod nfft reduced in
    integer arithmetic -- exact, where the reference evaluates exp(-j*omega*m) in floating point."""

    _diag = False

    def __init__(self, size: tuple = (1, 1), max_len: int = 2000, isint: bool = False, unit: int = 100,
                 nfft: int = 2 ** 11, fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self.fs = fs
        self.max_len = max_len
        self.unit = unit
        self.isint = isint
        super().__init__(size=size, nfft=nfft, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)
        self.initialize_class()

    def forward(self, x, ext_param=None):
        return self._run(x, ext_param)

    def init_param(self):
        if self.isint:
            delay_len = torch.randint(1, self.max_len, self.size, device=self.device)
        else:
            delay_len = torch.rand(self.size, device=self.device) * self.max_len
        self.assign_value(self.sample2s(delay_len))
        self.order = delay_len.max() + 1

    def s2sample(self, delay):
        return delay * self.fs / self.unit

    def sample2s(self, delay: torch.Tensor):
        return delay / self.fs * self.unit

    def get_delays(self):
        return lambda param: self.s2sample(self.map(param))

    _native_bin_order = True

    def _native_now(self, param) -> bool:
        return bool(self.isint)      # integer delays: the exact-phase kernel; fractional ones are torch expressions

    def get_freq_response(self):
        m = self.get_delays()

        def response(param):
            if self.isint:
                return _cached_integer_delay(self, param, lambda: m(param.double()))
            md = m(param.double())          # seconds -> samples in float64 whatever the module dtype
            # fractional (learnable) delays: phase = frac(k m / nfft) and gamma^m in float64
            bin0, m_local = ops.bin_shard(self.nfft)
            k = torch.arange(bin0, bin0 + m_local, device=md.device, dtype=torch.float64)
            k = k.view(-1, *([1] * md.dim()))
            ang = -2 * torch.pi * torch.remainder(k * md.unsqueeze(0) / self.nfft, 1.0)
            H = torch.polar((self._gamma_f ** md).unsqueeze(0).expand_as(ang).contiguous(), ang)
            return H.to(torch.complex64 if self.dtype == torch.float32 else torch.complex128)

        self.freq_response = response

    def check_input_shape(self, x):
        if (ops.bin_shard(self.nfft)[1], self.input_channels) != (x.shape[1], x.shape[2]):
            raise ValueError(
                f"parameter shape = {self.param.shape} not compatible with input signal of shape = ({x.shape}).")

    def check_param_shape(self):
        assert len(self.size) == 2, "delay must be 2D, for 1D (parallel) delay use parallelDelay module."

    def get_freq_convolve(self):
        self.freq_convolve = lambda x, param: ops.mimo(self._response_once(param), x, diag=self._diag)
        self._own_convolve = self.freq_convolve

    def _bin_response(self, param):
        return self._response_once(param), self._diag

    def initialize_class(self):
        self.check_param_shape()
        self.get_io()
        if self.requires_grad:
            self.map = lambda x: nn.functional.softplus(x)
        self.omega = (2 * torch.pi * torch.arange(0, self.nfft // 2 + 1, device=self.device, dtype=self.dtype)
                      / self.nfft).unsqueeze(1)
        self.get_freq_response()
        self.get_freq_convolve()

    def get_io(self):
        if self._diag:
            self.input_channels = self.output_channels = self.size[-1]
        else:
            self.input_channels = self.size[-1]
            self.output_channels = self.size[-2]

    def _probe_delay(self, z):
        m = self.s2sample(self.map(self.param))
        if self.isint:
            m = m.round()
        return (self.gamma ** m) * (1.0 / z) ** m

    def probe(self, z: torch.Tensor):
        return self._probe_delay(z)


class parallelDelay(Delay):
    """Per-channel delays, param (N,) (dsp.py:3453-3551)."""

    _diag = True

    def __init__(self, size: tuple = (1,), max_len: int = 2000, unit: int = 100, isint: bool = False,
                 nfft=2 ** 11, fs: int = 48000, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        super().__init__(size=size, max_len=max_len, isint=isint, unit=unit, nfft=nfft, fs=fs,
                         requires_grad=requires_grad, alias_decay_db=alias_decay_db, device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 1, "delays must be 1D, for 2D delays use Delay module."

    def probe(self, z: torch.Tensor):
        return torch.diag_embed(self._probe_delay(z))


class GainDelay(DSP):
    """Gain matrix and delay matrix in one module, param (2, N_out, N_in) = (gains, delays in unit/fs
    seconds): H[k] = g * gamma^m * exp(-j w_k m) (dsp.py:3554-3702)."""

    _diag = False

    def __init__(self, size: tuple = (1, 1), max_len: int = 2000, isint: bool = False, unit: int = 100,
                 nfft: int = 2 ** 11, fs: int = 48000, map_gain: Optional[callable] = None,
                 map_delay: Optional[callable] = None, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        self.fs = fs
        self.max_len = max_len
        self.unit = unit
        self.isint = isint
        self._custom_gain_map = map_gain is not None
        self._custom_delay_map = map_delay is not None
        self.map_gain = map_gain if map_gain is not None else _identity
        self.map_delay = map_delay if map_delay is not None else _identity
        super().__init__(size=(2, *size), nfft=nfft, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)
        self.initialize_class()

    def forward(self, x, ext_param=None):
        return self._run(x, ext_param)

    def init_param(self):
        shape = self.size[1:]
        with torch.no_grad():
            nn.init.ones_(self.param[0])
            if self.isint:
                d = torch.randint(1, self.max_len, shape, device=self.device, dtype=torch.int64).to(self.param.dtype)
            else:
                d = torch.rand(shape, device=self.device, dtype=self.dtype) * self.max_len
            self.param[1].copy_(self.sample2s(d))
        self.order = int(torch.ceil(d).max().item()) + 1

    def s2sample(self, delay: torch.Tensor):
        return delay * self.fs / self.unit

    def sample2s(self, delay: torch.Tensor):
        return delay / self.fs * self.unit

    def check_input_shape(self, x):
        if (ops.bin_shard(self.nfft)[1], self.input_channels) != (x.shape[1], x.shape[2]):
            raise ValueError(
                f"parameter shape = {self.param.shape} not compatible with input signal of shape = ({x.shape}).")

    def check_param_shape(self):
        assert len(self.size) == 3 and self.size[0] == 2, "GainDelay parameters must have shape (2, N_out, N_in)."

    def get_gains(self):
        return lambda param: to_complex(self.map_gain(param[0]))

    def get_delays(self):
        return lambda param: self.s2sample(self.map_delay(param[1]))

    _native_bin_order = True

    def _native_now(self, param) -> bool:
        return bool(self.isint)      # gain x integer-delay response (an elementwise product keeps the bin order)

    def get_freq_response(self):
        gains, delays = self.get_gains(), self.get_delays()

        def response(param):
            g = gains(param)
            samples = lambda: self.s2sample(self.map_delay(param[1].double()))      # noqa: E731  (samples, float64)
            cd = torch.complex64 if self.dtype == torch.float32 else torch.complex128
            if self.isint:
                D = _cached_integer_delay(self, param, samples)
            else:
                md = samples()
                bin0, m_local = ops.bin_shard(self.nfft)
                k = torch.arange(bin0, bin0 + m_local, device=md.device, dtype=torch.float64)
                k = k.view(-1, *([1] * md.dim()))
                ang = -2 * torch.pi * torch.remainder(k * md.unsqueeze(0) / self.nfft, 1.0)
                D = torch.polar((self._gamma_f ** md).unsqueeze(0).expand_as(ang).contiguous(), ang).to(cd)
            return g.to(cd).unsqueeze(0) * D

        self.freq_response = response

    def get_freq_convolve(self):
        self.freq_convolve = lambda x, param: ops.mimo(self._response_once(param), x, diag=self._diag)
        self._own_convolve = self.freq_convolve

    def _bin_response(self, param):
        return self._response_once(param), self._diag

    def initialize_class(self):
        self.check_param_shape()
        self.get_io()
        if self.requires_grad and not self._custom_delay_map:
            self.map_delay = lambda x: nn.functional.softplus(x)
        self.omega = (2 * torch.pi * torch.arange(0, self.nfft // 2 + 1, device=self.device, dtype=self.dtype)
                      / self.nfft).unsqueeze(1)
        self.get_freq_response()
        self.get_freq_convolve()

    def get_io(self):
        if self._diag:
            self.input_channels = self.output_channels = self.size[-1]
        else:
            self.input_channels = self.size[-1]
            self.output_channels = self.size[-2]


class parallelGainDelay(GainDelay):
    """Per-channel gain+delay, param (2, N) (dsp.py:3705-3779)."""

    _diag = True

    def __init__(self, size: tuple = (1,), max_len: int = 2000, isint: bool = False, unit: int = 100,
                 nfft: int = 2 ** 11, fs: int = 48000, map_gain: Optional[callable] = None,
                 map_delay: Optional[callable] = None, requires_grad: bool = False, alias_decay_db: float = 0.0,
                 device: Optional[str] = None, dtype: torch.dtype = torch.float32):
        super().__init__(size=size, max_len=max_len, isint=isint, unit=unit, nfft=nfft, fs=fs, map_gain=map_gain,
                         map_delay=map_delay, requires_grad=requires_grad, alias_decay_db=alias_decay_db,
                         device=device, dtype=dtype)

    def check_param_shape(self):
        assert len(self.size) == 2 and self.size[0] == 2, \
            "parallelGainDelay parameters must have shape (2, N), for MIMO use GainDelay module."


# forward() implementations of this library: a module is folded into fused paths only while its class still uses one
# of them (see DSP._fusable)
_LIBRARY_FORWARDS = {cls.__dict__["forward"] for cls in list(globals().values())
                     if isinstance(cls, type) and issubclass(cls, DSP) and "forward" in cls.__dict__}
