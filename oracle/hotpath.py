"""CPU oracle for the flamo frequency-sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``flamo_amd`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

This is a plain-torch (CPU, any real dtype, float64 by default) restatement of what the
reference computes on the path rFFT -> per-bin complex MIMO product -> Recursion solve ->
irFFT.  Each function cites the reference file:line it follows (paths relative to the
reference checkout, gdalsanto/flamo v0.2.13).  All arithmetic in the reference lives in
``torch`` primitives (``torch.fft.rfft/irfft``, ``torch.einsum``, ``torch.linalg.solve``,
``torch.matrix_exp``; pyproject.toml lists torch unpinned, this container has 2.10.0), so the
restatement uses the same primitives and autograd supplies the backward oracle.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the reference itself (CPU, float64)
and stores its inputs/outputs/gradients under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.
"""
from __future__ import annotations

import math

import torch

# --------------------------------------------------------------------------- helpers


def cdtype(real_dtype: torch.dtype) -> torch.dtype:
    return torch.complex128 if real_dtype == torch.float64 else torch.complex64


def to_complex(x: torch.Tensor) -> torch.Tensor:
    """flamo/utils.py:12-22 -- real tensor -> complex tensor with zero imaginary part."""
    return torch.complex(x, torch.zeros_like(x))


def gamma_of(alias_decay_db, nfft: int, dtype=torch.float64) -> torch.Tensor:
    """flamo/processor/dsp.py:307 -- gamma = 10 ** (-|dB| / nfft / 20), in the module dtype."""
    db = torch.as_tensor(alias_decay_db, dtype=dtype)
    return 10 ** (-torch.abs(db) / nfft / 20)


def alias_envelope(alias_decay_db, nfft: int, dtype=torch.float64) -> torch.Tensor:
    """flamo/processor/dsp.py:153-160 and 196-203 -- gamma ** arange(0, -nfft, -1).

    Both FFTAntiAlias and iFFTAntiAlias build this same *rising* envelope gamma^(-n)."""
    g = gamma_of(alias_decay_db, nfft, dtype)
    return g ** torch.arange(0, -nfft, -1, dtype=dtype)


# --------------------------------------------------------------------------- transforms (K1-K3)


def rfft(x: torch.Tensor, nfft: int, norm: str = "backward", alias_decay_db=None) -> torch.Tensor:
    """dsp.FFT (dsp.py:84-89) and dsp.FFTAntiAlias (dsp.py:141-163).

    x: real (B, T, N, ...) -> complex (B, nfft//2+1, N, ...).  T != nfft is zero-padded or
    truncated by torch.  With ``alias_decay_db`` the time signal is first multiplied by the
    envelope (einsum "btm,t->btm", dsp.py:162)."""
    if alias_decay_db is not None:
        env = alias_envelope(alias_decay_db, nfft, x.dtype)
        shape = [1, -1] + [1] * (x.dim() - 2)
        x = x * env.view(shape)
    return torch.fft.rfft(x, n=nfft, dim=1, norm=norm)


def irfft(X: torch.Tensor, nfft: int, norm: str = "backward", alias_decay_db=None) -> torch.Tensor:
    """dsp.iFFT (dsp.py:110-115) and dsp.iFFTAntiAlias (dsp.py:184-206)."""
    y = torch.fft.irfft(X, n=nfft, dim=1, norm=norm)
    if alias_decay_db is not None:
        env = alias_envelope(alias_decay_db, nfft, y.dtype)
        shape = [1, -1] + [1] * (y.dim() - 2)
        y = y * env.view(shape)
    return y


# --------------------------------------------------------------------------- per-bin products (K4-K7)


def mimo_const(W: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """Gain/Matrix: einsum("mn,bfn...->bfm...") (dsp.py:466-468)."""
    return torch.einsum("mn,bfn...->bfm...", W, X)


def mimo_const_diag(w: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """parallelGain: einsum("n,bfn...->bfn...") (dsp.py:552-554)."""
    return torch.einsum("n,bfn...->bfn...", w, X)


def mimo_full(H: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """Filter-type modules: einsum("fmn,bfn...->bfm...") (dsp.py:922-924) -- the metric's product."""
    return torch.einsum("fmn,bfn...->bfm...", H, X)


def mimo_diag(H: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """parallel* modules: einsum("fn,bfn...->bfn...") (dsp.py:1021-1023)."""
    return torch.einsum("fn,bfn...->bfn...", H, X)


# --------------------------------------------------------------------------- responses (K10, K11)


def fir_response(h: torch.Tensor, nfft: int, gamma: torch.Tensor) -> torch.Tensor:
    """Filter.get_freq_response (dsp.py:893-908): rfft(h * gamma^arange(taps), nfft, dim=0)."""
    taps = h.shape[0]
    env = (gamma ** torch.arange(0, taps)).view(-1, *([1] * (h.dim() - 1)))
    return torch.fft.rfft(h * env, n=nfft, dim=0)


def sos_response(b: torch.Tensor, a: torch.Tensor, nfft: int, gamma: torch.Tensor) -> torch.Tensor:
    """Tail shared by Biquad/SVF/GEQ/PEQ.get_poly_coeff (dsp.py:1520-1526, 2587-2593).

    b, a: (3, n_sections, ...) real.  The 3 taps are weighted by gamma^[0,1,2], transformed
    with rfft(nfft, dim=0), multiplied over the section axis and divided; bins where the
    denominator product is exactly zero get eps(dtype)."""
    env = gamma ** torch.arange(0, 3, 1, dtype=gamma.dtype)
    shape = [3] + [1] * (b.dim() - 1)
    # einsum("p,pomn->pomn") promotes float32 coefficients to the envelope (module) dtype
    B = torch.fft.rfft(b * env.view(shape), nfft, dim=0)
    A = torch.fft.rfft(a * env.view(shape), nfft, dim=0)
    Bp, Ap = torch.prod(B, dim=1), torch.prod(A, dim=1)
    H = Bp / Ap
    return torch.where(torch.abs(Ap) != 0, H, torch.finfo(H.dtype).eps * torch.ones_like(H))


def delay_samples(param_s: torch.Tensor, fs: int, unit: int, isint: bool) -> torch.Tensor:
    """Delay.s2sample (dsp.py:3334-3341) and the .round() of the isint branch (dsp.py:3358)."""
    m = param_s * fs / unit
    return m.round() if isint else m


def delay_response(m: torch.Tensor, nfft: int, gamma: torch.Tensor) -> torch.Tensor:
    """Delay/parallelDelay.get_freq_response (dsp.py:3352-3374, 3508-3530) as coded:
    gamma^m * exp(-1j * omega_k * m), omega_k = 2 pi k / nfft (dsp.py:3420-3425).
    m: (N_out, N_in) or (N,); result (M, ...) with M = nfft//2+1."""
    omega = 2 * torch.pi * torch.arange(0, nfft // 2 + 1, dtype=m.dtype) / nfft
    phase = omega.view(-1, *([1] * m.dim())) * m.unsqueeze(0)
    return (gamma ** m) * torch.exp(-1j * phase)


def delay_response_exact(m_int: torch.Tensor, nfft: int, gamma: torch.Tensor) -> torch.Tensor:
    """Integer-delay response with the phase reduced exactly, (k*m) mod nfft, before the
    sin/cos.  Mathematically identical to :func:`delay_response` for integer m; it is what
    the HIP path computes (twiddle table index) and reproduces the float64 reference to
    ~1e-12 where the reference's own float32 run is 1e-3 off."""
    k = torch.arange(0, nfft // 2 + 1, dtype=torch.int64).view(-1, *([1] * m_int.dim()))
    idx = (k * m_int.to(torch.int64).unsqueeze(0)) % nfft
    ang = -2 * math.pi * idx.to(torch.float64) / nfft
    g = gamma.to(torch.float64) ** m_int.to(torch.float64)
    return torch.polar(g.expand_as(ang).contiguous(), ang)


# --------------------------------------------------------------------------- parameter maps


def skew_matrix(X: torch.Tensor) -> torch.Tensor:
    """flamo/functional.py:42-56."""
    A = X.triu(1)
    return A - A.transpose(-1, -2)


def orthogonal(param: torch.Tensor) -> torch.Tensor:
    """Matrix(matrix_type="orthogonal") map: matrix_exp(skew_matrix(x)) (dsp.py:649)."""
    return torch.matrix_exp(skew_matrix(param))


def rbj_biquad(kind: str, fc_hz: torch.Tensor, gain_db: torch.Tensor, fs: int, fc2_hz=None):
    """RBJ low/high/band-pass coefficients (functional.py:376-552).  Returns (b, a), each
    (3, *fc.shape); the linear gain 10^(dB/20) multiplies b."""
    dt = fc_hz.dtype
    two = torch.tensor(2.0, dtype=dt)
    if kind == "bandpass":
        w1 = fc_hz / fs * 2 * torch.pi
        w2 = fc2_hz / fs * 2 * torch.pi
        wc = (w1 + w2) / 2
        bw = torch.log2(fc2_hz / fc_hz)
        alpha = torch.sin(wc) * torch.sinh(torch.log(two) / two * bw * (wc / torch.sin(wc)))
        c = torch.cos(wc)
        b = torch.stack([alpha, torch.zeros_like(alpha), -alpha])
    else:
        wc = fc_hz / fs * 2 * torch.pi
        alpha = torch.sin(wc) / 2 * torch.sqrt(two)
        c = torch.cos(wc)
        if kind == "lowpass":
            b = torch.stack([(1 - c) / 2, 1 - c, (1 - c) / 2])
        elif kind == "highpass":
            b = torch.stack([(1 + c) / 2, -(1 + c), (1 + c) / 2])
        else:
            raise ValueError(kind)
    a = torch.stack([1 + alpha, -2 * c, 1 - alpha])
    return 10 ** (gain_db / 20) * b, a


def biquad_map(x: torch.Tensor, kind: str) -> torch.Tensor:
    """Biquad.get_map (dsp.py:1528-1563): (fc_norm, gain) -> clamp(stack(fc, 20log10|gain|))."""
    dt = x.dtype
    if kind in ("lowpass", "highpass"):
        y = torch.stack((x[:, 0], 20 * torch.log10(torch.abs(x[:, 1]))), dim=1)
        lo = torch.tensor([0, -60], dtype=dt)
        hi = torch.tensor([1, 60], dtype=dt)
    else:
        e = torch.finfo(dt).eps
        y = torch.stack((x[:, 0], x[:, 1], 20 * torch.log10(torch.abs(x[:, -1]))), dim=1)
        lo = torch.tensor([0 + e, 0 + e, -60], dtype=dt)
        hi = torch.tensor([1 - e, 1 - e, 60], dtype=dt)
    shp = [1, -1] + [1] * (x.dim() - 2)
    return torch.clamp(y, min=lo.view(shp).expand_as(y), max=hi.view(shp).expand_as(y))


def biquad_response(param: torch.Tensor, kind: str, nfft: int, fs: int, gamma: torch.Tensor):
    """Biquad.get_poly_coeff (dsp.py:1464-1526) on the *mapped* parameters
    (n_sections, 2|3, ...): fc is given in units of pi rad (rad2hertz(param*pi), dsp.py:1497)."""
    p = biquad_map(param, kind)
    hz = lambda r: r * torch.pi * fs / (2 * torch.pi)  # rad2hertz, functional.py:322-335
    if kind == "bandpass":
        b, a = rbj_biquad(kind, hz(p[:, 0]), p[:, 2], fs, fc2_hz=hz(p[:, 1]))
    else:
        b, a = rbj_biquad(kind, hz(p[:, 0]), p[:, 1], fs)
    return sos_response(b, a, nfft, gamma)


# -- graphic equaliser (auxiliary/eq.py) -------------------------------------------------


def eq_freqs(interval: int = 1, start_freq: float = 31.25, end_freq: float = 16000.0):
    """auxiliary/eq.py:8-55 -- octave-band centres and the two shelving crossovers (float32)."""
    import numpy as np

    centres, c = [], start_freq
    while c < end_freq:
        c = c * np.power(2, 1 / interval)
        centres.append(c)
    cf = torch.tensor(centres, dtype=torch.float32)
    sc = torch.tensor([cf[0] / np.power(2, 1 / interval / 2), cf[-1] * np.power(2, 1 / interval / 2)],
                      dtype=torch.float32)
    return cf, sc


def geq_sos(gain_db: torch.Tensor, center_freq: torch.Tensor, shelving_freq: torch.Tensor,
            fs: int = 48000, R: float = 2.7, exact: bool = False):
    """auxiliary/eq.py:57-111 (geq) vectorised over trailing dims of ``gain_db``.

    gain_db: (n_bands, ...) with n_bands = len(center_freq)+3.  Returns (b, a), each
    (3, n_bands, ...) **float32** -- the reference allocates the SOS buffers without a dtype and
    calls geq() without one (dsp.py:2573-2585), so GEQ coefficients are float32 even in
    float64 mode.  Band 0: pure gain; band 1: low shelf; last: high shelf;
    others: peaking with Q = sqrt(R)/(R-1) (functional.py:555-675).

    ``exact=True`` is NOT the reference's arithmetic in the backward direction: the forward VALUES are the
    reference's float32 coefficients bit for bit, but the graph behind them is the same formulas kept in
    the dtype of ``gain_db`` (straight-through rounding: value = exact + (rounded - exact).detach()).  The
    reference's own gain gradient passes through the float32 section graph, where the cancelling tap
    contributions lose ~1e-4; this mode is the yardstick tests/ use to show which of (reference, HIP path)
    is closer to the gradient of the function the reference evaluates."""
    if exact:
        with torch.no_grad():
            b_ref, a_ref = geq_sos(gain_db.detach(), center_freq, shelving_freq, fs, R, exact=False)
        b_ex, a_ex = _geq_sos_impl(gain_db, center_freq, shelving_freq, fs, R, True)
        return b_ex + (b_ref.to(b_ex.dtype) - b_ex).detach(), a_ex + (a_ref.to(a_ex.dtype) - a_ex).detach()
    return _geq_sos_impl(gain_db, center_freq, shelving_freq, fs, R, False)


def _geq_sos_impl(gain_db, center_freq, shelving_freq, fs, R, exact):
    f32 = gain_db.dtype if exact else torch.float32
    nb = gain_db.shape[0]
    assert nb == len(center_freq) + len(shelving_freq) + 1
    # dtype choreography of the reference: the scalar formulas run in the dtype of the mapped
    # gains (0-dim float64 x 0-dim float32 -> float64) with tan()/cos() of the float32 band
    # frequencies evaluated in float32; every result is rounded to float32 when stored into the
    # float32 coefficient buffers; the later whole-vector scalings (g2*b, a*gain) are float32.
    gd = gain_db.dtype
    g = 10 ** (gain_db / 20)
    c32 = torch.float32                                   # the band constants: float32 in both modes
    Rt = torch.tensor(R, dtype=c32)
    Q = torch.sqrt(Rt) / (Rt - 1)
    bs, as_ = [], []
    for band in range(nb):
        gb = g[band]
        one, zero = torch.ones_like(gb, dtype=f32), torch.zeros_like(gb, dtype=f32)
        if band == 0:
            b = torch.stack([gb.to(f32), zero, zero])
            a = torch.stack([one, zero, zero])
        elif band in (1, nb - 1):
            fc = shelving_freq[0] if band == 1 else shelving_freq[1]
            t32 = torch.tan((fc.to(c32) / fs * 2 * torch.pi) / 2)
            t, t2 = t32.to(gd), (t32 ** 2).to(gd)            # t**2 is rounded in float32 first
            st = (torch.sqrt(torch.tensor(2.0, dtype=c32)) * t32).to(gd)  # sqrt(2)*t: float32 product
            g2, g4 = gb ** 0.5, gb ** 0.25
            b = torch.stack([g2 * t2 + st * g4 + 1, 2 * g2 * t2 - 2, g2 * t2 - st * g4 + 1]).to(f32)
            a = torch.stack([g2 + st * g4 + t2, 2 * t2 - 2 * g2, g2 - st * g4 + t2]).to(f32)
            b = g2.to(f32) * b
            if band == nb - 1:
                b, a = a * gb.to(f32), b
        else:
            wc = (center_freq[band - 2].to(c32) / fs * 2 * torch.pi)
            t = torch.tan(wc / Q / 2).to(gd)
            c = torch.cos(wc).to(gd)
            sg = torch.sqrt(gb)
            b = torch.stack([sg + gb * t, -2 * sg * c, sg - gb * t]).to(f32)
            a = torch.stack([sg + t, -2 * sg * c, sg - t]).to(f32)
        bs.append(b)
        as_.append(a)
    return torch.stack(bs, dim=1), torch.stack(as_, dim=1)


def geq_response(param: torch.Tensor, nfft: int, gamma: torch.Tensor, fs: int = 48000,
                 octave_interval: int = 1, map_fn=None, exact: bool = False) -> torch.Tensor:
    """GEQ / parallelGEQ.get_poly_coeff (dsp.py:2563-2593, 2657-2680): default map
    20*log10|x| (dsp.py:2529), float32 SOS, then the shared SOS tail with the module's gamma.
    NB the reference swaps the local names a/b but the result is numerator/denominator-correct."""
    cf, sc = eq_freqs(octave_interval)
    gain_db = (20 * torch.log10(torch.abs(param))) if map_fn is None else map_fn(param)
    b, a = geq_sos(gain_db, cf, sc, fs, exact=exact)
    return sos_response(b, a, nfft, gamma)


# --------------------------------------------------------------------------- closed loop (K9)


def recursion(F: torch.Tensor, Bk: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """system.Recursion.forward (system.py:397-425) for per-bin responses F (feedforward,
    (M,N,N)) and Bk (feedback, (M,N,N)):  A = I - F @ Bk;  out = solve(A, F @ X).
    X: (B, M, N) vector RHS or (B, M, N, K) matrix RHS."""
    N = F.shape[-1]
    A = torch.eye(N, dtype=F.dtype) - F @ Bk
    R = torch.einsum("fmn,bfn...->bfm...", F, X)
    A = A.unsqueeze(0).expand(X.shape[0], *A.shape)
    return torch.linalg.solve(A, R)


# --------------------------------------------------------------------------- composed configs


def config2_forward(x, W, geq_param, nfft, alias_decay_db=0.0, fs=48000):
    """BASELINE config 2: Shell(FFT -> Series(Matrix(N,N,"random"), GEQ((N,N))) -> iFFT)."""
    dt = x.dtype
    gamma = gamma_of(alias_decay_db, nfft, dt)
    X = rfft(x, nfft)
    X = mimo_const(to_complex(W), X)
    H = geq_response(geq_param, nfft, gamma, fs)
    X = mimo_full(H.to(cdtype(dt)), X)
    return irfft(X, nfft)


def fdn_forward(x, in_gain, out_gain, U_param, delays_s, nfft, alias_decay_db, fs=48000, unit=100,
                attn_param=None, attn_map=None, output="time", geq_exact=False):
    """e8_fdn-type FDN (reverb.py:117-199, examples/e8_fdn.py:106-123):
    FFT -> Gain(N,1) -> Recursion(fF=parallelDelay(isint), fB=Matrix(orthogonal)[ -> parallelGEQ])
    -> Gain(1,N) -> iFFTAntiAlias | abs."""
    dt = x.dtype
    gamma = gamma_of(alias_decay_db, nfft, dt)
    X = rfft(x, nfft)
    X = mimo_const(to_complex(in_gain), X)
    m = delay_samples(delays_s, fs, unit, True)
    D = delay_response(m, nfft, gamma)  # (M, N)
    U = to_complex(orthogonal(U_param))
    N = U.shape[0]
    M = nfft // 2 + 1
    Bk = U.unsqueeze(0).expand(M, N, N)
    if attn_param is not None:
        G = geq_response(attn_param, nfft, gamma, fs, map_fn=attn_map, exact=geq_exact).to(cdtype(dt))  # (M, N)
        Bk = G.unsqueeze(-1) * Bk
    F = torch.diag_embed(D)
    Y = recursion(F, Bk, X)
    Y = mimo_const(to_complex(out_gain), Y)
    if output == "time":
        return irfft(Y, nfft, alias_decay_db=alias_decay_db)
    if output == "abs":
        return torch.abs(Y)
    return Y


# --------------------------------------------------------------------------- the same responses on a subset of the bins
# (full-size checks: at nfft = 384000 a 32 x 32 chain has 192001 loop matrices -- minutes and tens of GB on the CPU -- while
# its bins are independent, so the oracle evaluates a sample of them.  The formulas are the ones above with the rfft of the
# three taps / the phase ramp written out for the chosen bins k: sum_p c_p gamma^p exp(-j 2 pi k p / nfft).)


def sos_response_at(b: torch.Tensor, a: torch.Tensor, nfft: int, gamma: torch.Tensor, bins: torch.Tensor) -> torch.Tensor:
    """sos_response (dsp.py:1520-1526, 2587-2593) at the bins `bins` (int64, (nb,)): (nb, ...)."""
    w = torch.exp(-2j * torch.pi * bins.to(torch.float64) / nfft).to(cdtype(gamma.dtype))       # (nb,)
    zp = torch.stack([torch.ones_like(w), gamma * w, (gamma * w) ** 2], 0)                       # (3, nb): (gamma w)^p
    shape = [3, -1] + [1] * (b.dim() - 1)
    B = (b.to(gamma.dtype).unsqueeze(1) * zp.view(shape)).sum(0)                                 # (nb, n_sections, ...)
    A = (a.to(gamma.dtype).unsqueeze(1) * zp.view(shape)).sum(0)
    Bp, Ap = torch.prod(B, dim=1), torch.prod(A, dim=1)
    H = Bp / Ap
    return torch.where(torch.abs(Ap) != 0, H, torch.finfo(H.dtype).eps * torch.ones_like(H))


def geq_response_at(param: torch.Tensor, nfft: int, gamma: torch.Tensor, bins: torch.Tensor, fs: int = 48000,
                    octave_interval: int = 1, map_fn=None, exact: bool = False) -> torch.Tensor:
    """geq_response (dsp.py:2563-2593) at the bins `bins`."""
    cf, sc = eq_freqs(octave_interval)
    gain_db = (20 * torch.log10(torch.abs(param))) if map_fn is None else map_fn(param)
    b, a = geq_sos(gain_db, cf, sc, fs, exact=exact)
    return sos_response_at(b, a, nfft, gamma, bins)


def delay_response_at(m: torch.Tensor, nfft: int, gamma: torch.Tensor, bins: torch.Tensor) -> torch.Tensor:
    """delay_response (dsp.py:3352-3374) at the bins `bins`, phase reduced exactly for integer m (delay_response_exact)."""
    k = bins.to(torch.int64).view(-1, *([1] * m.dim()))
    idx = (k * m.to(torch.int64).unsqueeze(0)) % nfft
    ang = -2 * math.pi * idx.to(torch.float64) / nfft
    g = gamma.to(torch.float64) ** m.to(torch.float64)
    return torch.polar(g.expand_as(ang).contiguous(), ang)


def recursion_at(F: torch.Tensor, Bk: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """recursion (system.py:397-425) for responses already restricted to a subset of bins: F, Bk (nb, N, N), X (B, nb, N[, K])."""
    return recursion(F, Bk, X)
