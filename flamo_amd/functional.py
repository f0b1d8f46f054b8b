"""Parameter-to-coefficient formulas used by the processor modules (the subset of
flamo/functional.py and flamo/auxiliary/eq.py that sits on the hot path's input side).

These run on a few hundred scalars per step and stay in PyTorch (autograd for free); the
per-bin work they feed is done by the HIP kernels.  Formulas follow the cited reference
lines; the code is written for batched tensors instead of per-channel Python loops.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

# ----------------------------------------------------------------------------- small maps


def get_magnitude(x: torch.Tensor) -> torch.Tensor:
    """|x| (flamo/functional.py:8-21)."""
    return torch.abs(x)


def get_eigenvalues(x: torch.Tensor) -> torch.Tensor:
    """Eigenvalues over the last two (equal) dimensions (flamo/functional.py:24-39).  On the GPU: one
    wavefront per matrix (ops.eigvals, N <= 64); the order of the eigenvalues of a matrix is the order
    on the diagonal of its Schur form, which is not torch.linalg.eigvals' order."""
    assert x.shape[-1] == x.shape[-2]
    if x.shape[-1] == 1:
        return x
    if x.is_cuda:
        from . import ops
        return ops.eigvals(x if x.is_complex() else x.to(torch.complex64 if x.dtype == torch.float32 else torch.complex128))
    return torch.linalg.eigvals(x)          # host tensors (inspection): the reference's own call


def skew_matrix(X: torch.Tensor) -> torch.Tensor:
    """Skew-symmetric matrix from the strictly upper triangle of X (flamo/functional.py:42-56)."""
    up = torch.triu(X, diagonal=1)
    return up - up.mT


def matrix_exp_capturable(X: torch.Tensor, squarings: int = 10, order: int = 10) -> torch.Tensor:
    """exp(X) by scaling-and-squaring with a FIXED schedule (2^-squarings scaling, Taylor series of
    the given order, then repeated squaring), evaluated in float64.  torch.matrix_exp picks its
    Pade degree from the matrix norm on the host, which forces a device synchronisation and cannot
    be captured in a HIP graph; for the small parameter matrices of the orthogonal map
    (|X| up to ~100) the fixed schedule agrees with it to ~1e-13 and is sync-free."""
    dt = X.dtype
    A = X.to(torch.float64) / float(2 ** squarings)
    n = A.shape[-1]
    E = torch.eye(n, dtype=torch.float64, device=X.device).expand_as(A).clone()
    term = E
    for k in range(1, order + 1):
        term = term @ A / k
        E = E + term
    for _ in range(squarings):
        E = E @ E
    return E.to(dt)


def db2mag(dB):
    return 10 ** (dB / 20)


def mag2db(mag):
    return 20 * torch.log10(torch.abs(mag))


def hertz2rad(hertz, fs):
    return torch.divide(hertz, fs) * 2 * torch.pi


def rad2hertz(rad, fs):
    return torch.divide(rad * fs, 2 * torch.pi)


class HadamardMatrix(nn.Module):
    """Normalised Sylvester-Hadamard matrix of order N, ignoring its input
    (flamo/functional.py:76-93)."""

    def __init__(self, N: int, device=None, dtype=torch.float32):
        super().__init__()
        self.N, self.device, self.dtype = N, device, dtype

    def forward(self, x):
        H = torch.ones(1, 1, device=self.device, dtype=self.dtype)
        base = torch.tensor([[1, 1], [1, -1]], device=self.device, dtype=self.dtype)
        while H.shape[0] < self.N:
            H = torch.kron(H, base) / (2.0 ** 0.5)
        return H


class RotationMatrix(nn.Module):
    """Kronecker powers of a 2x2 rotation (flamo/functional.py:96-138)."""

    def __init__(self, N: int, min_angle: float = 0, max_angle: float = torch.pi / 4, iter: Optional[int] = None,
                 device=None, dtype=torch.float32):
        super().__init__()
        self.N, self.min_angle, self.max_angle, self.iter = N, min_angle, max_angle, iter
        self.device, self.dtype = device, dtype

    def create_submatrix(self, angles, iters: int = 1):
        th = torch.clamp(angles[0], self.min_angle, self.max_angle)
        c, s = torch.cos(th), torch.sin(th)
        X = torch.stack([torch.stack([c, s]), torch.stack([-s, c])]).to(device=self.device, dtype=self.dtype)
        if iters is None:
            iters = int(np.log2(self.N)) - 1
        for i in range(iters):
            X = torch.kron(X, self.create_submatrix([angles[i]])) if len(angles) > 1 else torch.kron(X, X)
        return X

    def forward(self, theta):
        return self.create_submatrix(theta, self.iter)


# ----------------------------------------------------------------------------- test signals


def signal_gallery(batch_size: int, n_samples: int, n: int, signal_type: str = "impulse", fs: int = 48000,
                   rate: float = 1.0, reference=None, device=None, dtype=torch.float32) -> torch.Tensor:
    """(batch, n_samples, n) test signals (flamo/functional.py:164-270); the types the hot path's
    callers use: impulse, wgn/noise, sine, exp, reference."""
    if signal_type == "impulse":
        x = torch.zeros(batch_size, n_samples, n, dtype=dtype, device=device)
        x[:, 0, :] = 1
        return x
    if signal_type in ("wgn", "noise"):
        return torch.randn((batch_size, n_samples, n), device=device, dtype=dtype)
    if signal_type == "sine":
        t = torch.linspace(0, n_samples / fs, n_samples, dtype=dtype)
        return torch.sin(2 * np.pi * rate / fs * t).unsqueeze(-1).expand(batch_size, n_samples, n).to(device)
    if signal_type == "exp":
        t = torch.arange(n_samples, dtype=dtype)
        return torch.exp(-rate * t / fs).unsqueeze(-1).expand(batch_size, n_samples, n).to(device)
    if signal_type == "reference":
        ref = reference if isinstance(reference, torch.Tensor) else torch.tensor(reference, dtype=dtype)
        return ref.expand(batch_size, n_samples, n).to(device)
    raise ValueError(f"Signal type {signal_type} not recognized.")


# ----------------------------------------------------------------------------- RBJ biquads


def _rbj_den(alpha, cos_w):
    return torch.stack([1 + alpha, -2 * cos_w, 1 - alpha])


def lowpass_filter(fc, gain=0.0, fs: int = 48000, device=None, dtype=torch.float32):
    """RBJ low-pass (flamo/functional.py:376-428); returns (b, a) each (3, *fc.shape)."""
    w = hertz2rad(fc, fs)
    alpha = torch.sin(w) / 2 * torch.sqrt(torch.tensor(2, device=device, dtype=dtype))
    c = torch.cos(w)
    b = torch.stack([(1 - c) / 2, 1 - c, (1 - c) / 2])
    return 10 ** (gain / 20) * b, _rbj_den(alpha, c)


def highpass_filter(fc, gain=0.0, fs: int = 48000, device=None, dtype=torch.float32):
    """RBJ high-pass (flamo/functional.py:431-482)."""
    w = hertz2rad(fc, fs)
    alpha = torch.sin(w) / 2 * torch.sqrt(torch.tensor(2, device=device, dtype=dtype))
    c = torch.cos(w)
    b = torch.stack([(1 + c) / 2, -(1 + c), (1 + c) / 2])
    return 10 ** (gain / 20) * b, _rbj_den(alpha, c)


def bandpass_filter(fc1, fc2, gain=0.0, fs: int = 48000, device=None, dtype=torch.float32):
    """RBJ constant-skirt band-pass between fc1 and fc2 (flamo/functional.py:485-552)."""
    w = (hertz2rad(fc1, fs) + hertz2rad(fc2, fs)) / 2
    two = torch.tensor(2, device=device, dtype=dtype)
    bw = torch.log2(fc2 / fc1)
    alpha = torch.sin(w) * torch.sinh(torch.log(two) / two * bw * (w / torch.sin(w)))
    b = torch.stack([alpha, torch.zeros_like(alpha), -alpha])
    return 10 ** (gain / 20) * b, _rbj_den(alpha, torch.cos(w))


# ----------------------------------------------------------------------------- graphic equaliser


def octave_bands(interval: int = 1, start_freq: float = 31.25, end_freq: float = 16000.0):
    out, f = [], start_freq
    while f < end_freq:
        f = f * np.power(2, 1 / interval)
        out.append(f)
    return out


def eq_freqs(interval: int = 1, start_freq: float = 31.25, end_freq: float = 16000.0, device="cpu",
             dtype=torch.float32):
    """Band centres and the two shelving crossovers (flamo/auxiliary/eq.py:8-31)."""
    cf = torch.tensor(octave_bands(interval, start_freq, end_freq), device=device, dtype=dtype)
    half = np.power(2, 1 / interval / 2)
    sc = torch.tensor([cf[0] / half, cf[-1] * half], device=device, dtype=dtype)
    return cf, sc


class GEQDesign:
    """Vectorised restatement of auxiliary/eq.py:57-111 (`geq`) + functional.py:555-675
    (`shelving_filter`, `peak_filter`) for all channel pairs at once.

    The band constants (tan/cos of the float32 band frequencies, Q from R = 2.7) depend on
    nothing learnable and are evaluated ONCE on the host in float32, exactly as the reference
    evaluates them per call.  The gain-dependent arithmetic runs in float64 and every
    coefficient is then rounded to float32, because the reference stores the sections in
    float32 buffers even in float64 mode (dsp.py:2573-2585) and its float64 run is
    the parity target.  The whole-vector scalings that the reference performs on the already
    rounded float32 vectors (g^(1/2) * b, a * g) are float32 products here too."""

    def __init__(self, center_freq: torch.Tensor, shelving_freq: torch.Tensor, fs: int = 48000, R: float = 2.7):
        f32 = torch.float32
        cf, sf = center_freq.detach().cpu().to(f32), shelving_freq.detach().cpu().to(f32)
        self.n_bands = len(cf) + len(sf) + 1
        Rt = torch.tensor(R, dtype=f32)
        Q = torch.sqrt(Rt) / (Rt - 1)
        t_sh = torch.tan(hertz2rad(sf, fs) / 2)                       # (2,) float32
        self.sh_t = t_sh.double()
        self.sh_t2 = (t_sh ** 2).double()
        self.sh_st = (torch.sqrt(torch.tensor(2.0, dtype=f32)) * t_sh).double()
        wc = hertz2rad(cf, fs)
        self.pk_t = torch.tan(wc / Q / 2).double()                    # (n_peaks,)
        self.pk_c = torch.cos(wc).double()
        self._dev = {}

    def device_consts(self, device) -> torch.Tensor:
        """[t_lo, t_hi, t2_lo, t2_hi, st_lo, st_hi, pk_t..., pk_c...] as one float64 device tensor
        (argument of the fused HIP kernel ``ops.geq_sections``)."""
        key = ("flat", str(device))
        if key not in self._dev:
            self._dev[key] = torch.cat([self.sh_t, self.sh_t2, self.sh_st, self.pk_t, self.pk_c]).to(device)
        return self._dev[key]

    def _consts(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(v.to(device) for v in (self.sh_t, self.sh_t2, self.sh_st, self.pk_t, self.pk_c))
        return self._dev[key]

    def sections(self, gain_db: torch.Tensor):
        """gain_db: (n_bands, ...) -> (b, a) float32, each (3, n_bands, ...)."""
        assert gain_db.shape[0] == self.n_bands, "The number of gains must be equal to the number of frequencies."
        f32 = torch.float32
        sh_t, sh_t2, sh_st, pk_t, pk_c = self._consts(gain_db.device)
        g = 10 ** (gain_db.double() / 20)
        tail = (1,) * (g.dim() - 1)
        nb = self.n_bands
        # band 0: flat gain
        g0 = g[0].to(f32)
        b0 = torch.stack([g0, torch.zeros_like(g0), torch.zeros_like(g0)])
        a0 = torch.stack([torch.ones_like(g0), torch.zeros_like(g0), torch.zeros_like(g0)])

        def shelf(gs, i):
            t, t2, st = sh_t[i], sh_t2[i], sh_st[i]
            g2, g4 = gs ** 0.5, gs ** 0.25
            b = torch.stack([g2 * t2 + st * g4 + 1, 2 * g2 * t2 - 2, g2 * t2 - st * g4 + 1]).to(f32)
            a = torch.stack([g2 + st * g4 + t2, 2 * t2 - 2 * g2, g2 - st * g4 + t2]).to(f32)
            return g2.to(f32) * b, a

        b_lo, a_lo = shelf(g[1], 0)
        bh, ah = shelf(g[nb - 1], 1)
        b_hi, a_hi = ah * g[nb - 1].to(f32), bh
        # peaking bands 2 .. nb-2, all at once along dim 0
        gp = g[2:nb - 1]
        t = pk_t.view(-1, *tail)
        c = pk_c.view(-1, *tail)
        sg = torch.sqrt(gp)
        bp = torch.stack([sg + gp * t, -2 * sg * c, sg - gp * t]).to(f32)       # (3, n_peaks, ...)
        ap = torch.stack([sg + t, -2 * sg * c, sg - t]).to(f32)
        b = torch.cat([b0.unsqueeze(1), b_lo.unsqueeze(1), bp, b_hi.unsqueeze(1)], dim=1)
        a = torch.cat([a0.unsqueeze(1), a_lo.unsqueeze(1), ap, a_hi.unsqueeze(1)], dim=1)
        return b, a


def geq(center_freq, shelving_freq, R, gain_db, fs: int = 48000, device="cpu", dtype=torch.float32):
    """Second-order sections of the graphic equaliser for ONE channel: (b, a), each (3, n_bands)
    (flamo/auxiliary/eq.py:57-111)."""
    return GEQDesign(center_freq, shelving_freq, fs, float(R)).sections(gain_db)


# ----------------------------------------------------------------------------- accurate GEQ design
def _interp_clamped(xp: torch.Tensor, fp: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Piecewise-linear interpolation of (xp, fp) at x, held constant outside [xp[0], xp[-1]] -- what
    flamo.utils.RegularGridInterpolator does in one dimension (flamo/utils.py:51-140), weights written
    the same way: (f_left * d_right + f_right * d_left) / (d_left + d_right)."""
    n = xp.shape[0]
    right = torch.bucketize(x, xp).clamp(max=n - 1)
    left = (right - 1).clamp(0, n - 1)
    dl = (x - xp[left]).clamp(min=0)
    dr = (xp[right] - x).clamp(min=0)
    both = (dl == 0) & (dr == 0)
    dl = torch.where(both, torch.ones_like(dl), dl)
    dr = torch.where(both, torch.ones_like(dr), dr)
    return (fp[left] * dr + fp[right] * dl) / (dl + dr)


def _sos_magnitude_db(b: torch.Tensor, a: torch.Tensor, freqs: torch.Tensor, nfft: int, fs: int) -> torch.Tensor:
    """dB magnitude of every section (columns of b, a: (3, n)) at the control frequencies: the section is
    normalised by its a0, sampled on an nfft-point grid and interpolated linearly in dB
    (flamo/functional.py:933-979, probe_sos)."""
    f = torch.fft.rfftfreq(nfft, 1 / fs)
    out = torch.zeros((freqs.shape[0], b.shape[1]), dtype=b.dtype)
    for k in range(b.shape[1]):
        a0 = a[0, k].clone()
        B = torch.fft.rfft(b[:, k] / a0, nfft, dim=0)
        A = torch.fft.rfft(a[:, k] / a0, nfft, dim=0)
        h = B / (A + torch.tensor(1e-10))
        out[:, k] = _interp_clamped(f, 20 * torch.log10(torch.abs(h)), freqs)
    return out


def _bounded_least_squares_lbfgs(G: torch.Tensor, target: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor,
                                 steps: int = 100) -> torch.Tensor:
    """argmin mean((G x - target)^2) from x = 1 with torch's L-BFGS (default settings), the iterate
    clamped to [lower, upper] inside the closure: 100 optimiser steps (flamo/auxiliary/minimize.py:34-78)."""
    x = nn.Parameter(torch.ones(G.shape[1]))
    opt = torch.optim.LBFGS([x])

    def closure():
        opt.zero_grad()
        loss = torch.mean(torch.pow(torch.matmul(G, x) - target, 2))
        loss.backward()
        x.data.clamp_(lower, upper)
        return loss

    for _ in range(steps):
        opt.step(closure)
    return x


def _geq_sections_plain(center_freq, shelving_freq, R, gain_db, fs: int, dtype):
    """Graphic-equaliser sections evaluated band by band with tensor arithmetic in ``dtype``, the way
    flamo.auxiliary.eq.geq does when it is called directly (eq.py:57-111 with functional.py:555-675) --
    GEQDesign above restates the *module's* mixed-precision route instead.  (b, a), each (3, n_bands)."""
    nb = len(center_freq) + len(shelving_freq) + 1
    assert len(gain_db) == nb, "The number of gains must be equal to the number of frequencies."
    b = torch.zeros((3, nb), dtype=dtype)
    a = torch.zeros((3, nb), dtype=dtype)
    two = torch.tensor(2, dtype=dtype)
    for k in range(nb):
        g = db2mag(gain_db[k].reshape(-1)[0].to(dtype))
        if k == 0:
            bb = torch.stack([g, torch.zeros((), dtype=dtype), torch.zeros((), dtype=dtype)])
            aa = torch.tensor([1, 0, 0], dtype=dtype)
        elif k == 1 or k == nb - 1:
            t = torch.tan(hertz2rad(shelving_freq[0 if k == 1 else 1], fs) / 2)
            t2, g2, g4 = t ** 2, g ** 0.5, g ** 0.25
            num = torch.stack([g2 * t2 + torch.sqrt(two) * t * g4 + 1, 2 * g2 * t2 - 2, g2 * t2 - torch.sqrt(two) * t * g4 + 1])
            den = torch.stack([g2 + torch.sqrt(two) * t * g4 + t2, 2 * t2 - 2 * g2, g2 - torch.sqrt(two) * t * g4 + t2])
            num = g2 * num
            bb, aa = (num, den) if k == 1 else (den * g, num)
        else:
            w = hertz2rad(center_freq[k - 2], fs)
            t = torch.tan(w / (torch.sqrt(R) / (R - 1)) / 2)
            sg = torch.sqrt(g)
            bb = torch.stack([sg + g * t, -2 * sg * torch.cos(w), sg - g * t])
            aa = torch.stack([sg + t, -2 * sg * torch.cos(w), sg - t])
        b[:, k], a[:, k] = bb, aa
    return b, a


def accurate_geq(target_gain: torch.Tensor, center_freq: torch.Tensor, shelving_crossover: torch.Tensor, fs: int = 48000,
                 device="cpu", dtype=torch.float32):
    """Graphic-equaliser sections whose cascade interpolates the target gains (dB, one per band centre
    plus the two band edges): the command gains are fitted by bounded least squares on the dB interaction
    matrix of 10 dB prototype sections at 101 log-spaced control frequencies (Schlecht & Habets 2017;
    flamo/auxiliary/eq.py:114-182).  A host-side design on a dozen numbers: runs on the CPU in the
    reference's float32.  Returns (b, a), each (3, n_bands + 3)."""
    target_gain = target_gain.detach().to("cpu")
    center_freq = center_freq.detach().to("cpu", dtype)
    shelving_crossover = shelving_crossover.detach().to("cpu", dtype)
    assert len(target_gain) == len(center_freq) + 2, \
        "The number of target gains must be equal to the number of center frequencies + 2."
    nfft = 2 ** 16
    n_sec = len(center_freq) + len(shelving_crossover)
    R = torch.tensor(2.7, dtype=dtype)
    ctrl = torch.round(torch.logspace(np.log10(1), np.log10(fs / 2.1), 101, dtype=dtype))
    knots = torch.cat((torch.tensor([1], dtype=dtype), center_freq, torch.tensor([fs / 2.1], dtype=dtype)))
    wanted = _interp_clamped(knots, target_gain, ctrl)
    proto = 10.0                                                   # dB
    pb, pa = _geq_sections_plain(center_freq, shelving_crossover, R, torch.full((n_sec + 1, 1), proto, dtype=dtype), fs, dtype)
    G = _sos_magnitude_db(pb, pa, ctrl, nfft, fs) / proto
    upper = torch.tensor([torch.inf] + [2 * proto] * n_sec, dtype=dtype)
    gains = _bounded_least_squares_lbfgs(G, wanted, -upper, upper)
    b, a = _geq_sections_plain(center_freq, shelving_crossover, R, gains.detach(), fs, dtype)
    return b.to(device), a.to(device)
