// Device helpers shared by the response generators (response.hip) and the lanes-per-section cascade kernels (cascade2.hip):
// bin order, section polynomials, the graphic-equaliser design (flamo/auxiliary/eq.py:57-111).
#pragma once
#include "common.h"

namespace fl {

typedef float f2 __attribute__((ext_vector_type(2)));

// Bin number of element f of a response row.  bin0 >= 0: the contiguous range bin0, bin0+1, ... (bin-sharded execution).
// bin0 < 0: the whole spectrum in the ROW-MAJOR bin order of the fused Shell pipeline (spectral.hip) with row length
// L2 = -bin0: element f = k1*L2 + k2 holds bin k1 + L1*k2 (L1 = nfft/2/L2), element nfft/2 the Nyquist bin.
__device__ __forceinline__ int bin_of(int f, int bin0, int nfft) {
    if (bin0 >= 0) return bin0 + f;
    const int L2 = -bin0, L = nfft >> 1;
    if (f >= L) return L;
    const int k1 = f / L2;
    return k1 + (L / L2) * (f - k1 * L2);
}
static inline bool bin_range_ok(int bin0, int m_local, int nfft) {
    if (bin0 >= 0) return true;
    return nfft % 2 == 0 && (nfft / 2) % (-bin0) == 0 && m_local == nfft / 2 + 1;
}


// The section polynomials are evaluated in DOUBLE precision whatever the storage type T: at low
// frequencies b0 + b1 w + b2 w^2 cancels to ~1e-5 of its terms (shelving sections at 44 Hz), so
// float32 evaluation -- what the reference's float32 mode does -- loses 3 digits there.  The
// point w = exp(-2 pi i k / n) comes from the float64 master twiddle table.
// stage the (3, S) taps of channel c of b and a into LDS: lb = [3][S], la = [3][S]
__device__ inline void stage_taps(const double* __restrict__ b, const double* __restrict__ a, int S, int C, int c,
                                  double* lb, double* la) {
    for (int i = threadIdx.x; i < 3 * S; i += blockDim.x) {
        lb[i] = b[(size_t)i * C + c];
        la[i] = a[(size_t)i * C + c];
    }
    __syncthreads();
}

struct SosEval {
    cx<double> z1, z2;  // g*w, g^2*w^2
    // co: this channel's taps staged in LDS as [3][S] (broadcast reads, no scalar-load latency)
    __device__ inline cx<double> poly(const double* co, int S, int s) const {
        const double c0 = co[s];
        const double c1 = co[S + s];
        const double c2 = co[2 * S + s];
        return cx<double>(c0 + c1 * z1.x + c2 * z2.x, c1 * z1.y + c2 * z2.y);
    }
};

__device__ inline SosEval sos_point(const cx<double>* __restrict__ Wd, int nfft, int k, double g) {
    SosEval e;
    // 0 <= k <= nfft/2 on this path, so the two indices reduce with a compare instead of a modulo
    const int k2 = 2 * k;
    const cx<double> w1 = Wd[k < nfft ? k : k - nfft];
    const cx<double> w2 = Wd[k2 < nfft ? k2 : k2 - nfft];
    e.z1 = cx<double>(g * w1.x, g * w1.y);
    e.z2 = cx<double>(g * g * w2.x, g * g * w2.y);
    return e;
}

template <typename T> __device__ inline T eps_of();
template <> __device__ inline float eps_of<float>() { return 1.1920928955078125e-07f; }
template <> __device__ inline double eps_of<double>() { return 2.220446049250313e-16; }


// ---------------------------------------------------------------- graphic-equaliser design
// Command gains (dB) -> second-order sections of the GEQ, all channel pairs at once, and the
// backward of that map.  Restates flamo/auxiliary/eq.py:57-111 (geq) with
// flamo/functional.py:555-675 (shelving_filter, peak_filter): band 0 flat gain, band 1 low
// shelf, bands 2..nb-2 peaking (R = 2.7), band nb-1 high shelf.  The reference stores the
// sections in float32 buffers even in float64 mode (dsp.py:2573-2585), so every coefficient is
// rounded to float32 exactly where the reference rounds it; the band constants (tan/cos of the
// float32 band frequencies) are supplied by the host.  consts layout (double):
//   [t_lo, t_hi, t2_lo, t2_hi, st_lo, st_hi, pk_t[nb-3], pk_c[nb-3]]
// One thread per (band, channel): ~1 KB of work replaces ~150 tiny elementwise launches per step.
__device__ inline double f32r(double x) { return (double)(float)x; }

// in_kind: 0 = command gains in dB (double); 1 / 2 (and 3 / 4, see below) = LINEAR command gains x (double / float),
// i.e. the module's raw parameters under its default map 20 log10|x| -- then g = 10^(map/20) = |x|
// and the map, its backward and the dtype casts (ten tiny launches per step) fold into these two.
__device__ inline double geq_linear_gain(const void* gain, int in_kind, int idx, double* raw) {
    if (in_kind == 0) {
        const double v = reinterpret_cast<const double*>(gain)[idx];
        *raw = v;
        return pow(10.0, v / 20.0);
    }
    const double v = (in_kind == 2 || in_kind == 4) ? (double)reinterpret_cast<const float*>(gain)[idx]
                                                    : reinterpret_cast<const double*>(gain)[idx];
    *raw = v;
    // 3 / 4: raw parameters under the map 20 log10(sigmoid(x)) (the attenuation filters of e8_fdn.py:97): g = sigmoid(x)
    if (in_kind >= 3) return 1.0 / (1.0 + exp(-v));
    return fabs(v);
}

// raw command value of (band, channel pair) idx, as stored (in_kind as above) -- the load alone, so that a kernel can request it
// together with its other operands
__device__ inline double geq_raw_gain(const void* gain, int in_kind, int idx) {
    return (in_kind == 2 || in_kind == 4) ? (double)reinterpret_cast<const float*>(gain)[idx]
                                          : reinterpret_cast<const double*>(gain)[idx];
}
// ... and the linear gain from it (geq_linear_gain's arithmetic)
__device__ inline double geq_gain_of_raw(double v, int in_kind) {
    if (in_kind == 0) return pow(10.0, v / 20.0);
    if (in_kind >= 3) return 1.0 / (1.0 + exp(-v));
    return fabs(v);
}
// the two band constants a section's design reads: shelves (t2, st q), peaking bands (t, c); band 0: none (a valid element)
__device__ inline void geq_band_const_idx(int band, int nb, int* ia, int* ib) {
    if (band == 1 || band == nb - 1) {
        const int i = (band == 1) ? 0 : 1;
        *ia = 2 + i; *ib = 4 + i;
    } else if (band == 0) {
        *ia = 0; *ib = 1;
    } else {
        *ia = 6 + band - 2; *ib = 6 + (nb - 3) + band - 2;
    }
}
// one section of the equaliser from its linear gain g and its two band constants (ka, kb), taps into bb[3], aa[3]
__device__ inline void geq_section_vals(double g, int band, int nb, double ka, double kb, double* bb, double* aa) {
    double b0, b1, b2, a0, a1, a2;
    if (band == 0) {
        b0 = f32r(g); b1 = 0; b2 = 0; a0 = 1; a1 = 0; a2 = 0;
    } else if (band == 1 || band == nb - 1) {
        const double t2 = ka, stq = kb;
        // (g^(1/4) as the root of the root: two correctly rounded square roots, 0.75 ulp -- ocml's pow is ~500 float64
        // instructions, which inside the launch pair were a third of a response workgroup's life)
        const double u = sqrt(g), q = sqrt(u);
        const double p0 = f32r(u * t2 + stq * q + 1), p1 = f32r(2 * u * t2 - 2), p2 = f32r(u * t2 - stq * q + 1);
        const double d0 = f32r(u + stq * q + t2), d1 = f32r(2 * t2 - 2 * u), d2 = f32r(u - stq * q + t2);
        const float uf = (float)u, gf = (float)g;
        const float s0 = uf * (float)p0, s1 = uf * (float)p1, s2 = uf * (float)p2;   // float32 products
        if (band == 1) {
            b0 = s0; b1 = s1; b2 = s2; a0 = d0; a1 = d1; a2 = d2;
        } else {
            b0 = (float)d0 * gf; b1 = (float)d1 * gf; b2 = (float)d2 * gf; a0 = s0; a1 = s1; a2 = s2;
        }
    } else {
        const double t = ka, c = kb;
        const double sg = sqrt(g);
        b0 = f32r(sg + g * t); b1 = f32r(-2 * sg * c); b2 = f32r(sg - g * t);
        a0 = f32r(sg + t); a1 = b1; a2 = f32r(sg - t);
    }
    bb[0] = b0; bb[1] = b1; bb[2] = b2;
    aa[0] = a0; aa[1] = a1; aa[2] = a2;
}
// one section of the equaliser: band `band` of channel pair idx - band * C (idx = band * C + c), taps into bb[3], aa[3]
__device__ inline void geq_section_of(const void* __restrict__ gain, int in_kind, int idx, int band, int nb,
                                      const double* __restrict__ k, double* bb, double* aa) {
    double raw;
    const double g = geq_linear_gain(gain, in_kind, idx, &raw);
    int ia, ib;
    geq_band_const_idx(band, nb, &ia, &ib);
    geq_section_vals(g, band, nb, k[ia], k[ib], bb, aa);
}


// dL/dg (linear gain) of one band from the gradients of its six taps: the backward of geq_section_of (double; the float32
// roundings of the design are identities for the gradient, as in the reference's autograd graph)
__device__ inline double geq_design_bwd(int band, int nb, double g, const double* __restrict__ k, double B0, double B1,
                                        double B2, double A0, double A1, double A2) {
    if (band == 0) return B0;
    if (band == 1 || band == nb - 1) {
        const int i = (band == 1) ? 0 : 1;
        const double t2 = k[2 + i], stq = k[4 + i];
        const double u = sqrt(g), q = sqrt(u);
        const double du = 0.5 / u, dq = 0.25 * q / g;
        const double p0 = u * t2 + stq * q + 1, p1 = 2 * u * t2 - 2, p2 = u * t2 - stq * q + 1;
        const double d0 = u + stq * q + t2, d1 = 2 * t2 - 2 * u, d2 = u - stq * q + t2;
        const double dp0 = t2 * du + stq * dq, dp1 = 2 * t2 * du, dp2 = t2 * du - stq * dq;
        const double dd0 = du + stq * dq, dd1 = -2 * du, dd2 = du - stq * dq;
        // s = u * p (scaled numerator-form), d = denominator-form
        const double ds0 = du * p0 + u * dp0, ds1 = du * p1 + u * dp1, ds2 = du * p2 + u * dp2;
        if (band == 1) return B0 * ds0 + B1 * ds1 + B2 * ds2 + A0 * dd0 + A1 * dd1 + A2 * dd2;
        // b = d * g, a = s
        return B0 * (d0 + g * dd0) + B1 * (d1 + g * dd1) + B2 * (d2 + g * dd2) + A0 * ds0 + A1 * ds1 + A2 * ds2;
    }
    const int np = nb - 3;
    const double t = k[6 + band - 2], c = k[6 + np + band - 2];
    const double dsg = 0.5 / sqrt(g);
    return B0 * (dsg + t) + B1 * (-2 * c * dsg) + B2 * (dsg - t) + A0 * dsg + A1 * (-2 * c * dsg) + A2 * dsg;
}
// ... and through the parameter map (in_kind as geq_linear_gain), stored in the parameter's dtype
__device__ inline void geq_store_gain_grad(void* __restrict__ ggain, int in_kind, int idx, double dg, double g, double raw) {
    if (in_kind == 0) {
        reinterpret_cast<double*>(ggain)[idx] = dg * g * (2.302585092994045684 / 20.0);   // dg/dgain_db = g ln(10) / 20
    } else {
        const double v = in_kind >= 3 ? dg * g * (1.0 - g)                                 // g = sigmoid(x)
                                      : dg * (raw > 0 ? 1.0 : (raw < 0 ? -1.0 : 0.0));     // g = |x|
        if (in_kind == 2 || in_kind == 4) reinterpret_cast<float*>(ggain)[idx] = (float)v;
        else reinterpret_cast<double*>(ggain)[idx] = v;
    }
}

// The section polynomial turned by half a sample: with w = exp(-i omega), z = g w,
//     B(z) conj(w) = b0 conj(w) + g b1 + g^2 b2 w = [S cos(omega) + T] + i [D sin(omega)],   S = b0 + g^2 b2, T = g b1, D = b0 - g^2 b2
// -- real and imaginary part are ONE multiply-add and ONE multiply (the monomial and the 1 -+ w forms take four), every
// section of B and of A carries the same unit factor conj(w), so prod B / prod A, |B|^2 and the quotients the backward
// pass forms are unchanged.  Float-safe as the 1 -+ w form is: about the nearer of omega = 0 / pi,
//     Re = (S + T) - S x,  x = 1 - cos(omega)   (bins below nfft/4)     |     Re = (T - S) + S x,  x = 1 + cos(omega)
// with the sums formed in double and x formed in double from the float64 twiddle before rounding.  Table entry of a
// section: (c0, c1, c2) with Re = c0 + c1 x, Im = c2 sin(omega), `pitch` floats apart.
__device__ inline void half_turn_tables(double t0, double t1, double t2, double g, float* lo, float* hi, int pitch) {
    const double S = t0 + g * g * t2, T = g * t1, D = t0 - g * g * t2;
    lo[0] = (float)(S + T); lo[pitch] = (float)(-S); lo[2 * pitch] = (float)D;
    hi[0] = (float)(T - S); hi[pitch] = (float)S;    hi[2 * pitch] = (float)D;
}

struct GeqDesign {
    const void* gain;      // (nb, C) command gains / raw parameters, or null: sections are read from b, a
    int in_kind;
    const double* k;       // band constants
    double* b_out;         // (3, nb, C) each
    double* a_out;
};


// csrc/cascade2.hip: the Matrix-then-cascade forward with (numerator, denominator) in the packed halves
int rc_ba_launch(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd,
                 int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch, void* stream, GeqDesign gd);

}  // namespace fl
