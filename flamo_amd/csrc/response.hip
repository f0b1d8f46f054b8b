// Frequency-response generators evaluated directly per bin (gfx950 / MI355X).
//
//  * integer delay lines: H[c,k] = amp[c] * W_n^((k*m_c) mod n)  -- exact integer phase index
//    (Delay/parallelDelay with isint=True, flamo/processor/dsp.py:3356-3365, 3512-3521);
//  * second-order-section cascades: H[c,k] = prod_s B_s(k) / prod_s A_s(k) with
//    B_s(k) = b0 + b1 g w + b2 g^2 w^2, w = W_n^k -- what the reference obtains from
//    rfft(3 taps, nfft) per section followed by prod/prod (dsp.py:1520-1526, 2587-2593),
//    without ever materialising the (M, sections, N_out, N_in) tensors; plus its backward.
#include "common.h"
#include "response_common.h"

namespace fl {

template <typename T>
__global__ void __launch_bounds__(256) delay_response_kernel(const int32_t* __restrict__ m, const T* __restrict__ amp,
                                                            const cx<T>* __restrict__ W, int nfft, double inv_nfft,
                                                            int bin0, int m_local, cx<T>* __restrict__ H, long h_pitch) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const int c = blockIdx.y;
    // (k m) mod nfft exactly, without the 64-bit integer division (~100 instructions, most of this kernel's time):
    // |k m| < 2^53 is exact in double, the quotient estimate is off by at most one
    const long long prod = (long long)bin_of(f, bin0, nfft) * (long long)m[c];
    long long idx = prod - (long long)((double)prod * inv_nfft) * nfft;
    idx += (idx < 0) ? nfft : 0;
    idx -= (idx >= nfft) ? nfft : 0;
    const cx<T> w = W[idx];
    const T a = amp[c];
    H[(size_t)c * h_pitch + f] = cx<T>(a * w.x, a * w.y);
}

template <typename T>
__global__ void __launch_bounds__(256) sos_response_kernel(const double* __restrict__ b, const double* __restrict__ a, int S, int C,
                                                          double g, const cx<double>* __restrict__ Wd, int nfft,
                                                          int bin0, int m_local, cx<T>* __restrict__ H, long h_pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lb = reinterpret_cast<double*>(smem);
    double* la = lb + 3 * S;
    const int c = blockIdx.y;
    stage_taps(b, a, S, C, c, lb, la);
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const SosEval e = sos_point(Wd, nfft, bin_of(f, bin0, nfft), g);
    cx<double> Bp(1, 0), Ap(1, 0);
    for (int s = 0; s < S; ++s) {
        Bp = Bp * e.poly(lb, S, s);
        Ap = Ap * e.poly(la, S, s);
    }
    cx<double> h = (Ap.x != 0 || Ap.y != 0) ? cdiv(Bp, Ap) : cx<double>((double)eps_of<T>(), 0);
    H[(size_t)c * h_pitch + f] = cx<T>((T)h.x, (T)h.y);
}

// Cascade response times a real constant matrix on the right, H[m][n] = sum_j G[m][j] W[j][n] (Series of Matrix then a
// cascade-type filter): one thread per (output channel m, bin) walks the Nmid cascades of its row, stores G (the
// backward pass needs it) and the product -- the composition pass over the response-sized tensors and the real -> complex
// conversion of W (three tiny launches) never run.  G is rounded to float before the product, as the separate passes do.
// (T = double: the same operator in complex128 with a float64 constant factor, what the reference's float64 examples run)
template <int NIW, typename T = float>
__global__ void __launch_bounds__(256) sos_response_rc_kernel(const double* __restrict__ b, const double* __restrict__ a, int S,
                                                             int C, int Nmid, const T* __restrict__ Wr, double g,
                                                             const cx<double>* __restrict__ Wd, int nfft, int bin0, int m_local,
                                                             cx<T>* __restrict__ G, long g_pitch, cx<T>* __restrict__ H,
                                                             long h_pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lb = reinterpret_cast<double*>(smem);            // [Nmid][3][S]
    double* la = lb + Nmid * 3 * S;
    T* lw = reinterpret_cast<T*>(la + Nmid * 3 * S);   // [Nmid][NIW]
    const int m = blockIdx.y;
    for (int i = threadIdx.x; i < Nmid * 3 * S; i += 256) {
        const int j = i / (3 * S), e = i - j * 3 * S;
        lb[i] = b[(size_t)e * C + m * Nmid + j];
        la[i] = a[(size_t)e * C + m * Nmid + j];
    }
    for (int i = threadIdx.x; i < Nmid * NIW; i += 256) lw[i] = Wr[i];
    __syncthreads();
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const SosEval e = sos_point(Wd, nfft, bin_of(f, bin0, nfft), g);
    cx<T> acc[NIW];
#pragma unroll
    for (int n = 0; n < NIW; ++n) acc[n] = cx<T>((T)0, (T)0);
#pragma unroll 2
    for (int j = 0; j < Nmid; ++j) {
        const double* tb = lb + j * 3 * S;
        const double* ta = la + j * 3 * S;
        cx<double> Bp(1, 0), Ap(1, 0);
        // (unrolled by 4: the LDS reads of four sections' taps are issued together -- one latency per four sections instead of
        // one per section; two cascades interleaved by the outer unroll: the products are dependent chains)
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            Bp = Bp * e.poly(tb, S, s);
            Ap = Ap * e.poly(ta, S, s);
        }
        const cx<double> h = (Ap.x != 0 || Ap.y != 0) ? cdiv(Bp, Ap) : cx<double>((double)eps_of<T>(), 0);
        const cx<T> hf((T)h.x, (T)h.y);
        G[(size_t)(m * Nmid + j) * g_pitch + f] = hf;
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const T w = lw[j * NIW + n];
            acc[n].x += w * hf.x;
            acc[n].y += w * hf.y;
        }
    }
#pragma unroll
    for (int n = 0; n < NIW; ++n) H[(size_t)(m * NIW + n) * h_pitch + f] = acc[n];
}

// The same operator with the cascade evaluated in FLOAT, two sections per packed instruction.  What makes float safe:
// every section polynomial is evaluated about the nearer of w = +1 / w = -1,  B = c0 + c1 x + c2 x^2  with  x = 1 - w
// (bins below nfft/4) or x = 1 + w, the coefficient sums c0 = b0 +- b1 + b2, ... formed in double and x itself formed in
// double from the float64 twiddle before rounding -- the cancellation that costs plain float evaluation 3 digits at low
// frequency (shelving sections at 44 Hz) happens in exact arithmetic, what is left is a well-conditioned Horner step.
// (The backward kernel's float stage uses the same basis.)  The 2 x S/2 running products are two independent chains per
// packed register.  Measured against the double kernel: response 3e-7 relative, 49 -> ~30 us at config 2.
// Graphic-equaliser mode of the kernel below (gain != null): the sections are DESIGNED in the prologue from the command
// gains (eq.py:57-111, the arithmetic of geq_sections_kernel) instead of being read -- every workgroup needs the 12 sections
// of its N_mid cascades, one per thread -- and the first bin block of each output channel writes them out for the backward
// pass: the design launch in front of this one (5 us and a dispatch gap in the config-2 step) is gone.
template <int NIW, int UNR = 1>
__global__ void __launch_bounds__(256) sos_response_rc_fast_kernel(const double* __restrict__ b, const double* __restrict__ a, int S,
                                                                  int C, int Nmid, const float* __restrict__ Wr, double g,
                                                                  const cx<double>* __restrict__ Wd, int nfft, int bin0,
                                                                  int m_local, cx<float>* __restrict__ G, long g_pitch,
                                                                  cx<float>* __restrict__ H, long h_pitch, GeqDesign gd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SP = (S + 1) & ~1;                               // even table pitch: a section pair is one 8-byte read
    float* cf = reinterpret_cast<float*>(smem);                // [Nmid][basis 2][poly 2][3][SP]
    float* lw = cf + (size_t)Nmid * 12 * SP;                   // [Nmid][NIW]
    const int m = blockIdx.y;
    for (int i = threadIdx.x; i < Nmid * SP; i += 256) {
        const int j = i / SP, sidx = i - j * SP;
        const int c = m * Nmid + j;
        const bool real = sidx < S;                            // padding section: b = a = (1, 0, 0)
        double tb[3] = {1.0, 0.0, 0.0}, ta[3] = {1.0, 0.0, 0.0};
        if (real) {
            if (gd.gain) {
                geq_section_of(gd.gain, gd.in_kind, sidx * C + c, sidx, S, gd.k, tb, ta);
                if (blockIdx.x == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        gd.b_out[(size_t)(q * S + sidx) * C + c] = tb[q];
                        gd.a_out[(size_t)(q * S + sidx) * C + c] = ta[q];
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    tb[q] = b[(size_t)(q * S + sidx) * C + c];
                    ta[q] = a[(size_t)(q * S + sidx) * C + c];
                }
            }
        }
#pragma unroll
        for (int poly = 0; poly < 2; ++poly) {
            const double t0 = poly ? ta[0] : tb[0], t1 = poly ? ta[1] : tb[1], t2 = poly ? ta[2] : tb[2];
            float* lo = cf + ((size_t)j * 4 + 0 * 2 + poly) * 3 * SP;
            float* hi = cf + ((size_t)j * 4 + 1 * 2 + poly) * 3 * SP;
            half_turn_tables(t0, t1, t2, g, lo + sidx, hi + sidx, SP);
        }
    }
    for (int i = threadIdx.x; i < Nmid * NIW; i += 256) lw[i] = Wr[i];
    __syncthreads();
    // A thread takes TWO ADJACENT BINS (k, k + 1) through the cascades: the twelve table reads of a section pair serve both
    // (one bin per thread had the LDS return path as busy as the vector ALU: 48 bytes per lane and section pair against 16
    // packed instructions).  Adjacent bins share the expansion point except for the one pair that straddles nfft/4, which
    // walks the tables twice.  Natural order: elements 2p, 2p + 1; row-major order: (row 2r, column c) and the element one
    // row below it (bin + 1); the Nyquist element is a pair of its own.
    const int p = blockIdx.x * 256 + threadIdx.x;
    int e[2];
    bool two;
    if (bin0 >= 0) {
        e[0] = 2 * p;
        if (e[0] >= m_local) return;
        two = e[0] + 1 < m_local;
        e[1] = two ? e[0] + 1 : e[0];
    } else {
        const int L2 = -bin0, L = nfft >> 1, L1 = L / L2, main = ((L1 + 1) >> 1) * L2;
        if (p > main) return;
        if (p == main) {
            e[0] = e[1] = L;
            two = false;
        } else {
            const int r = p / L2, c2 = p - r * L2;
            e[0] = 2 * r * L2 + c2;
            two = 2 * r + 1 < L1;
            e[1] = two ? e[0] + L2 : e[0];
        }
    }
    constexpr int UNR_REST = UNR > 1 ? UNR - 1 : 1;      // pairs per trip behind the first one (which starts the products)
    bool low[2];
    float xr[2], xi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = bin_of(e[q], bin0, nfft);
        const cx<double> w1 = Wd[k < nfft ? k : k - nfft];
        low[q] = 4 * (long)k < nfft;
        xr[q] = (float)(low[q] ? 1.0 - w1.x : 1.0 + w1.x);      // 1 -+ cos(omega), formed in double
        xi[q] = (float)(-w1.y);                                  // sin(omega)
    }
    const bool same = low[0] == low[1];
    cx<float> acc[2][NIW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < NIW; ++n) acc[q][n] = cx<float>(0.f, 0.f);
    for (int j = 0; j < Nmid; ++j) {
        f2 pbr[2], pbi[2], par[2], pai[2];
        // one section pair of bin q: (c0 + c1 x) + i c2 sin(omega) (half_turn_tables) times the running products
        auto step = [&](int q, f2 b0, f2 b1, f2 b2, f2 a0, f2 a1, f2 a2) {
            const f2 Br = b0 + b1 * xr[q], Bi = b2 * xi[q];
            const f2 Ar = a0 + a1 * xr[q], Ai = a2 * xi[q];
            const f2 nbr = pbr[q] * Br - pbi[q] * Bi, nbi = pbr[q] * Bi + pbi[q] * Br;
            const f2 nar = par[q] * Ar - pai[q] * Ai, nai = par[q] * Ai + pai[q] * Ar;
            pbr[q] = nbr; pbi[q] = nbi; par[q] = nar; pai[q] = nai;
        };
        if (same) {
            const float* cb = cf + ((size_t)j * 4 + (low[0] ? 0 : 2)) * 3 * SP;
            const float* ca = cb + 3 * SP;
            {   // the first pair STARTS the running products (what a multiplication of (1, 0) by it gives, bit for bit): sixteen
                // moves and sixteen packed operations per cascade less
                const f2 b0 = *reinterpret_cast<const f2*>(cb), b1 = *reinterpret_cast<const f2*>(cb + SP), b2 = *reinterpret_cast<const f2*>(cb + 2 * SP);
                const f2 a0 = *reinterpret_cast<const f2*>(ca), a1 = *reinterpret_cast<const f2*>(ca + SP), a2 = *reinterpret_cast<const f2*>(ca + 2 * SP);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    pbr[q] = b0 + b1 * xr[q]; pbi[q] = b2 * xi[q];
                    par[q] = a0 + a1 * xr[q]; pai[q] = a2 * xi[q];
                }
            }
            // (UNR section pairs per trip with the first: their table reads are issued together -- one LDS latency per trip, not per pair)
#pragma unroll UNR_REST
            for (int s = 2; s < SP; s += 2) {
                const f2 b0 = *reinterpret_cast<const f2*>(cb + s), b1 = *reinterpret_cast<const f2*>(cb + SP + s),
                         b2 = *reinterpret_cast<const f2*>(cb + 2 * SP + s);
                const f2 a0 = *reinterpret_cast<const f2*>(ca + s), a1 = *reinterpret_cast<const f2*>(ca + SP + s),
                         a2 = *reinterpret_cast<const f2*>(ca + 2 * SP + s);
                step(0, b0, b1, b2, a0, a1, a2);
                step(1, b0, b1, b2, a0, a1, a2);
            }
        } else {
            for (int q = 0; q < 2; ++q) {
                pbr[q] = (f2)(1.f); pbi[q] = (f2)(0.f); par[q] = (f2)(1.f); pai[q] = (f2)(0.f);
                const float* cb = cf + ((size_t)j * 4 + (low[q] ? 0 : 2)) * 3 * SP;
                const float* ca = cb + 3 * SP;
                for (int s = 0; s < SP; s += 2) {
                    const f2 b0 = *reinterpret_cast<const f2*>(cb + s), b1 = *reinterpret_cast<const f2*>(cb + SP + s),
                             b2 = *reinterpret_cast<const f2*>(cb + 2 * SP + s);
                    const f2 a0 = *reinterpret_cast<const f2*>(ca + s), a1 = *reinterpret_cast<const f2*>(ca + SP + s),
                             a2 = *reinterpret_cast<const f2*>(ca + 2 * SP + s);
                    if (q == 0) step(0, b0, b1, b2, a0, a1, a2);
                    else step(1, b0, b1, b2, a0, a1, a2);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // the two chains (even / odd sections) of each product
            const float Bx = pbr[q].x * pbr[q].y - pbi[q].x * pbi[q].y, By = pbr[q].x * pbi[q].y + pbi[q].x * pbr[q].y;
            const float Ax = par[q].x * par[q].y - pai[q].x * pai[q].y, Ay = par[q].x * pai[q].y + pai[q].x * par[q].y;
            cx<float> hf;
            if (Ax != 0.f || Ay != 0.f) {
                const float inv = __builtin_amdgcn_rcpf(Ax * Ax + Ay * Ay);      // (1 ulp; an IEEE division is ten instructions per bin and cascade)
                hf = cx<float>((Bx * Ax + By * Ay) * inv, (By * Ax - Bx * Ay) * inv);
            } else {
                hf = cx<float>(eps_of<float>(), 0.f);
            }
            if (q == 0 || two) G[(size_t)(m * Nmid + j) * g_pitch + e[q]] = hf;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const float w = lw[j * NIW + n];
                acc[q][n].x += w * hf.x;
                acc[q][n].y += w * hf.y;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (q == 0 || two) {
#pragma unroll
            for (int n = 0; n < NIW; ++n) H[(size_t)(m * NIW + n) * h_pitch + e[q]] = acc[q][n];
        }
}

// The cascade applied to a signal with BX <= 2 columns in the same launch: thread (output channel m, bin) walks the Ni
// cascades of its row as above, stores G[m][j] (the backward pass reads it) and accumulates Y[b][m] += G[m][j] X[b][j] --
// the product's own pass over the (M, No, Ni) response (0.3 ms at 32 x 32, nfft = 384000) never runs.
template <int BX>
__global__ void __launch_bounds__(256) sos_response_apply_fast_kernel(const double* __restrict__ b, const double* __restrict__ a, int S,
                                                                     int C, int Nmid, const cx<float>* __restrict__ X, long xs_b,
                                                                     long xs_n, double g, const cx<double>* __restrict__ Wd, int nfft,
                                                                     int bin0, int m_local, cx<float>* __restrict__ G, long g_pitch,
                                                                     cx<float>* __restrict__ Y, long ys_b, long ys_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SP = (S + 1) & ~1;
    float* cf = reinterpret_cast<float*>(smem);                // [Nmid][basis 2][poly 2][3][SP]
    const int m = blockIdx.y;
    for (int i = threadIdx.x; i < Nmid * 2 * SP; i += 256) {
        const int j = i / (2 * SP), rem = i - j * 2 * SP;
        const int poly = rem / SP, sidx = rem - poly * SP;
        const double* t = poly ? a : b;
        const int c = m * Nmid + j;
        const bool real = sidx < S;
        const double t0 = real ? t[(size_t)sidx * C + c] : 1.0, t1 = real ? t[(size_t)(S + sidx) * C + c] : 0.0,
                     t2 = real ? t[(size_t)(2 * S + sidx) * C + c] : 0.0;
        half_turn_tables(t0, t1, t2, g, cf + ((size_t)j * 4 + 0 * 2 + poly) * 3 * SP + sidx, cf + ((size_t)j * 4 + 1 * 2 + poly) * 3 * SP + sidx, SP);
    }
    __syncthreads();
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const int k = bin_of(f, bin0, nfft);
    const cx<double> w1 = Wd[k < nfft ? k : k - nfft];
    const bool low = 4 * (long)k < nfft;
    const float xr = (float)(low ? 1.0 - w1.x : 1.0 + w1.x), xi = (float)(-w1.y);      // 1 -+ cos(omega) (formed in double), sin(omega)
    cx<float> acc[BX];
#pragma unroll
    for (int n = 0; n < BX; ++n) acc[n] = cx<float>(0.f, 0.f);
    for (int j = 0; j < Nmid; ++j) {
        cx<float> xv[BX];
#pragma unroll
        for (int n = 0; n < BX; ++n) xv[n] = X[(size_t)n * xs_b + (size_t)j * xs_n + f];      // requested ahead of the cascade
        const float* cb = cf + ((size_t)j * 4 + (low ? 0 : 2)) * 3 * SP;
        const float* ca = cb + 3 * SP;
        f2 pbr = (f2)(1.f), pbi = (f2)(0.f), par = (f2)(1.f), pai = (f2)(0.f);
        for (int s = 0; s < SP; s += 2) {
            const f2 b0 = *reinterpret_cast<const f2*>(cb + s), b1 = *reinterpret_cast<const f2*>(cb + SP + s),
                     b2 = *reinterpret_cast<const f2*>(cb + 2 * SP + s);
            const f2 a0 = *reinterpret_cast<const f2*>(ca + s), a1 = *reinterpret_cast<const f2*>(ca + SP + s),
                     a2 = *reinterpret_cast<const f2*>(ca + 2 * SP + s);
            const f2 Br = b0 + b1 * xr, Bi = b2 * xi;      // (c0 + c1 x) + i c2 sin(omega): half_turn_tables
            const f2 Ar = a0 + a1 * xr, Ai = a2 * xi;
            const f2 nbr = pbr * Br - pbi * Bi, nbi = pbr * Bi + pbi * Br;
            const f2 nar = par * Ar - pai * Ai, nai = par * Ai + pai * Ar;
            pbr = nbr; pbi = nbi; par = nar; pai = nai;
        }
        const float Bx = pbr.x * pbr.y - pbi.x * pbi.y, By = pbr.x * pbi.y + pbi.x * pbr.y;
        const float Ax = par.x * par.y - pai.x * pai.y, Ay = par.x * pai.y + pai.x * par.y;
        cx<float> hf;
        if (Ax != 0.f || Ay != 0.f) {
            const float inv = 1.0f / (Ax * Ax + Ay * Ay);
            hf = cx<float>((Bx * Ax + By * Ay) * inv, (By * Ax - Bx * Ay) * inv);
        } else {
            hf = cx<float>(eps_of<float>(), 0.f);
        }
        G[(size_t)(m * Nmid + j) * g_pitch + f] = hf;
#pragma unroll
        for (int n = 0; n < BX; ++n) fma_cx(acc[n], hf, xv[n]);
    }
#pragma unroll
    for (int n = 0; n < BX; ++n) Y[(size_t)n * ys_b + (size_t)m * ys_m + f] = acc[n];
}

// The plain cascade response (one channel per blockIdx.y) with the same float evaluation: what the float32 modules
// outside the Matrix-then-cascade operator run (the FDN attenuation filters, a full GEQ matrix in front of a loop).
// A block stages its channel's two coefficient tables once and walks BINS_PER_BLOCK bins.
constexpr int kSosFastBins = 1024;
__global__ void __launch_bounds__(256) sos_response_fast_kernel(const double* __restrict__ b, const double* __restrict__ a, int S, int C,
                                                               double g, const cx<double>* __restrict__ Wd, int nfft, int bin0,
                                                               int m_local, cx<float>* __restrict__ H, long h_pitch, GeqDesign gd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SP = (S + 1) & ~1;
    float* cf = reinterpret_cast<float*>(smem);                // [basis 2][poly 2][3][SP]
    const int c = blockIdx.y;
    for (int sidx = threadIdx.x; sidx < SP; sidx += 256) {
        const bool real = sidx < S;                            // padding section: b = a = (1, 0, 0)
        double tb[3] = {1.0, 0.0, 0.0}, ta[3] = {1.0, 0.0, 0.0};
        if (real) {
            if (gd.gain) {                                     // graphic equaliser: designed here (see GeqDesign)
                geq_section_of(gd.gain, gd.in_kind, sidx * C + c, sidx, S, gd.k, tb, ta);
                if (blockIdx.x == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        gd.b_out[(size_t)(q * S + sidx) * C + c] = tb[q];
                        gd.a_out[(size_t)(q * S + sidx) * C + c] = ta[q];
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    tb[q] = b[(size_t)(q * S + sidx) * C + c];
                    ta[q] = a[(size_t)(q * S + sidx) * C + c];
                }
            }
        }
#pragma unroll
        for (int poly = 0; poly < 2; ++poly) {
            const double t0 = poly ? ta[0] : tb[0], t1 = poly ? ta[1] : tb[1], t2 = poly ? ta[2] : tb[2];
            half_turn_tables(t0, t1, t2, g, cf + (0 * 2 + poly) * 3 * SP + sidx, cf + (1 * 2 + poly) * 3 * SP + sidx, SP);
        }
    }
    __syncthreads();
    const int f_end = min(m_local, (int)(blockIdx.x + 1) * kSosFastBins);
    for (int f = blockIdx.x * kSosFastBins + threadIdx.x; f < f_end; f += 256) {
        const int k = bin_of(f, bin0, nfft);
        const cx<double> w1 = Wd[k < nfft ? k : k - nfft];
        const bool low = 4 * (long)k < nfft;
        const float xr = (float)(low ? 1.0 - w1.x : 1.0 + w1.x), xi = (float)(-w1.y);      // 1 -+ cos(omega) (formed in double), sin(omega)
        const float* cb = cf + (low ? 0 : 6 * SP);
        const float* ca = cb + 3 * SP;
        f2 pbr = (f2)(1.f), pbi = (f2)(0.f), par = (f2)(1.f), pai = (f2)(0.f);
        for (int s = 0; s < SP; s += 2) {
            const f2 b0 = *reinterpret_cast<const f2*>(cb + s), b1 = *reinterpret_cast<const f2*>(cb + SP + s),
                     b2 = *reinterpret_cast<const f2*>(cb + 2 * SP + s);
            const f2 a0 = *reinterpret_cast<const f2*>(ca + s), a1 = *reinterpret_cast<const f2*>(ca + SP + s),
                     a2 = *reinterpret_cast<const f2*>(ca + 2 * SP + s);
            const f2 Br = b0 + b1 * xr, Bi = b2 * xi;      // (c0 + c1 x) + i c2 sin(omega): half_turn_tables
            const f2 Ar = a0 + a1 * xr, Ai = a2 * xi;
            const f2 nbr = pbr * Br - pbi * Bi, nbi = pbr * Bi + pbi * Br;
            const f2 nar = par * Ar - pai * Ai, nai = par * Ai + pai * Ar;
            pbr = nbr; pbi = nbi; par = nar; pai = nai;
        }
        const float Bx = pbr.x * pbr.y - pbi.x * pbi.y, By = pbr.x * pbi.y + pbi.x * pbr.y;
        const float Ax = par.x * par.y - pai.x * pai.y, Ay = par.x * pai.y + pai.x * par.y;
        cx<float> hf;
        if (Ax != 0.f || Ay != 0.f) {
            const float inv = 1.0f / (Ax * Ax + Ay * Ay);
            hf = cx<float>((Bx * Ax + By * Ay) * inv, (By * Ax - Bx * Ay) * inv);
        } else {
            hf = cx<float>(eps_of<float>(), 0.f);
        }
        H[(size_t)c * h_pitch + f] = hf;
    }
}

// 1/x in double from a float32 hardware reciprocal refined by two Newton steps (|x| within float
// range, which |B_s|^2 of a filter section always is): ~8 instructions instead of a full divide.
__device__ inline double fast_rcp(double x) {
    double r = (double)__frcp_rn((float)x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
// a / b = a conj(b) / |b|^2
__device__ inline cx<double> cdiv_fast(cx<double> a, cx<double> b) {
    const double inv = fast_rcp(b.x * b.x + b.y * b.y);
    return cx<double>((a.x * b.x + a.y * b.y) * inv, (a.y * b.x - a.x * b.y) * inv);
}

// Backward: dL/db[p,s,c] = sum_k Re(conj(gH) * H/B_s * z_p),  dL/da[p,s,c] = -sum_k Re(conj(gH) * H/A_s * z_p)
// One thread walks bins of one channel and keeps the 6*SCH running sums of a chunk of SCH sections
// in registers (GEQ: 12 sections = one chunk, nothing is recomputed).  Everything stays in double:
// the three tap sums of a section are nearly collinear at low frequency and the parameter maps
// combine them with cancellation, so single-precision sums cost 3 digits of the final gradient.
// Hs (or null): the forward output.  With it the cascade product is not re-evaluated per bin (24 polynomial values and 24
// complex products for a graphic equaliser, per section chunk); the section loop shares q_p = conj(gH) H z_p between the
// sections and takes ONE reciprocal per section for both quotients, 1/|B|^2 = |A|^2 / (|B|^2 |A|^2), as the mixed kernel does.
// NIW > 0: "right constant factor" mode of the all-double kernel (see SosRC below: Hs = G is the cascade's own response, gH
// holds dL/dH of H = G W, planes m * NIW + n) -- the float64 form of the fused Matrix-then-cascade operator.
struct SosRCd {
    int Nmid, nbx;       // nbx: bin blocks (the grid is 1-D in this mode)
    const double* Wr;    // (Nmid, NIW) row-major
    double* partW;       // (nbx, C, NIW)
};
template <typename T, int SCH, int NIW = 0>
__global__ void __launch_bounds__(256) sos_response_bwd_kernel(const cx<T>* __restrict__ gH, long g_pitch, const cx<T>* __restrict__ Hs,
                                                              long h_pitch, const double* __restrict__ b,
                                                              const double* __restrict__ a, int S, int C, double g,
                                                              const cx<double>* __restrict__ Wd, int nfft, int bin0,
                                                              int m_local, double* __restrict__ part, SosRCd rc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lb = reinterpret_cast<double*>(smem);
    double* la = lb + 3 * S;
    // block -> (bin block bx, channel pair c); constant-factor mode: the Nmid pairs of one output channel on one XCD, as in
    // the mixed kernel below (they read the same NIW gradient planes)
    int bx, c, nbx;
    if (NIW > 0) {
        nbx = rc.nbx;
        const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
        const int j = t % rc.Nmid, pr = (t / rc.Nmid) * 8 + xcd;        // pr = m * nbx + bx
        if (pr >= (C / rc.Nmid) * nbx) return;
        bx = pr % nbx;
        c = (pr / nbx) * rc.Nmid + j;
    } else {
        bx = blockIdx.x;
        nbx = gridDim.x;
        c = blockIdx.y;
    }
    stage_taps(b, a, S, C, c, lb, la);
    const int s0 = blockIdx.z * SCH;
    double acc[6 * SCH];   // [(i*3 + p)*SCH + q]: i = b|a, p = tap, q = section of the chunk
#pragma unroll
    for (int v = 0; v < 6 * SCH; ++v) acc[v] = 0.0;
    const T eps = eps_of<T>();
    constexpr int NW = NIW > 0 ? NIW : 1;
    double wrow[NW], accw[NW];
    const cx<T>* gbase = gH + (size_t)c * g_pitch;
    if (NIW > 0) {
        const int mrow = c / rc.Nmid, j = c - mrow * rc.Nmid;
        gbase = gH + (size_t)mrow * NIW * g_pitch;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            wrow[n] = rc.Wr[j * NIW + n];
            accw[n] = 0.0;
        }
    }

    for (int f = bx * 256 + threadIdx.x; f < m_local; f += nbx * 256) {
        const SosEval e = sos_point(Wd, nfft, bin_of(f, bin0, nfft), g);
        cx<T> gin;
        if (NIW > 0) {
            // dL/dG = sum_n W[j][n] dL/dH[m][n];  dL/dW[j][n] += Re(conj(G) dL/dH[m][n])
            const cx<T> hv = Hs[(size_t)c * h_pitch + f];
            gin = cx<T>((T)0, (T)0);
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const cx<T> t = gbase[(size_t)n * g_pitch + f];
                gin.x += (T)wrow[n] * t.x;
                gin.y += (T)wrow[n] * t.y;
                if (blockIdx.z == 0) accw[n] += (double)hv.x * (double)t.x + (double)hv.y * (double)t.y;
            }
        } else {
            gin = gbase[f];
        }
        cx<double> h, Ap(1, 0);
        if (Hs) {
            const cx<T> hv = Hs[(size_t)c * h_pitch + f];
            if (hv.x == eps && hv.y == (T)0) continue;   // guarded bin (prod A == 0): the constant eps, zero gradient
            h = cx<double>((double)hv.x, (double)hv.y);
        } else {
            cx<double> Bp(1, 0);
            for (int s = 0; s < S; ++s) {
                Bp = Bp * e.poly(lb, S, s);
                Ap = Ap * e.poly(la, S, s);
            }
            if (Ap.x == 0 && Ap.y == 0) continue;  // guarded bins are the constant eps: zero gradient
            h = cdiv_fast(Bp, Ap);
        }
        const cx<double> gc((double)gin.x, -(double)gin.y);
        const cx<double> q0 = gc * h, q1 = q0 * e.z1, q2 = q0 * e.z2;      // conj(gH) H z_p
#pragma unroll
        for (int q = 0; q < SCH; ++q) {
            const int s = s0 + q;
            if (s < S) {
                const cx<double> Bs = e.poly(lb, S, s), As = e.poly(la, S, s);
                const double nb = Bs.x * Bs.x + Bs.y * Bs.y, na = As.x * As.x + As.y * As.y;
                const double nn = nb * na;
                if (nn > 1e-30 && nn < 1e30) {      // (the reciprocal's seed is a float: the product has to sit in its range)
                    const double inv = fast_rcp(nn);
                    const double ib = inv * na, ia = -(inv * nb);
                    const double Brs = Bs.x * ib, Bis = Bs.y * ib, Ars = As.x * ia, Ais = As.y * ia;
                    // Re(q_p / B_s) = (q_p.x Br + q_p.y Bi) / |B_s|^2
                    acc[0 * SCH + q] = fma(Bis, q0.y, fma(Brs, q0.x, acc[0 * SCH + q]));
                    acc[1 * SCH + q] = fma(Bis, q1.y, fma(Brs, q1.x, acc[1 * SCH + q]));
                    acc[2 * SCH + q] = fma(Bis, q2.y, fma(Brs, q2.x, acc[2 * SCH + q]));
                    acc[3 * SCH + q] = fma(Ais, q0.y, fma(Ars, q0.x, acc[3 * SCH + q]));
                    acc[4 * SCH + q] = fma(Ais, q1.y, fma(Ars, q1.x, acc[4 * SCH + q]));
                    acc[5 * SCH + q] = fma(Ais, q2.y, fma(Ars, q2.x, acc[5 * SCH + q]));
                } else {      // a section value vanishes (or leaves the range) at this bin
                    cx<double> tb;
                    if (nb > 1e-290) {
                        tb = cdiv(q0, Bs);
                    } else {  // numerator section vanishes at this bin: product of the others
                        cx<double> o(1, 0), Aq(1, 0);
                        for (int t = 0; t < S; ++t) {
                            if (t != s) o = o * e.poly(lb, S, t);
                            Aq = Aq * e.poly(la, S, t);
                        }
                        tb = gc * cdiv(o, Aq);
                    }
                    const cx<double> ta = cdiv(q0, As);
                    acc[0 * SCH + q] += tb.x;
                    acc[1 * SCH + q] += tb.x * e.z1.x - tb.y * e.z1.y;     // Re(tb * z_p)
                    acc[2 * SCH + q] += tb.x * e.z2.x - tb.y * e.z2.y;
                    acc[3 * SCH + q] -= ta.x;
                    acc[4 * SCH + q] -= ta.x * e.z1.x - ta.y * e.z1.y;
                    acc[5 * SCH + q] -= ta.x * e.z2.x - ta.y * e.z2.y;
                }
            }
        }
    }
    // block reduction: wavefront reduce-scatter, then the 4 wave partials through LDS
    __shared__ double red[4][6 * SCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = 0, cnt = 0, dup = 0;
    wave_reduce_scatter<double, 6 * SCH, 32>(acc, lane, off, cnt, dup);
    if ((lane & dup) == 0) {
#pragma unroll
        for (int v = 0; v < 6 * SCH; ++v)
            if (v < cnt) red[wave][off + v] = acc[v];
    }
    __syncthreads();
    if (threadIdx.x < 6 * SCH) {
        const int i = threadIdx.x / (3 * SCH), p = (threadIdx.x / SCH) % 3, q = threadIdx.x % SCH;
        const int s = s0 + q;
        if (s < S) {
            const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            part[((((size_t)bx * 2 + i) * 3 + p) * S + s) * C + c] = v;
        }
    }
    if (NIW > 0 && blockIdx.z == 0) {
        __shared__ double redw[4][NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            double v = accw[n];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) redw[wave][n] = v;
        }
        __syncthreads();
        if (threadIdx.x < NW)
            rc.partW[((size_t)bx * C + c) * NW + threadIdx.x] =
                redw[0][threadIdx.x] + redw[1][threadIdx.x] + redw[2][threadIdx.x] + redw[3][threadIdx.x];
    }
}

// Mixed-precision backward for float32 storage (H, gH are c64): float arithmetic arranged so that
// nothing is lost where the monomial form b0 + b1 w + b2 w^2 cancels, and written on section PAIRS so
// that it compiles to packed-float instructions.
//  * Section values B_s(k), A_s(k): float, each polynomial turned by half a sample (half_turn_tables: B conj(w) =
//    [S cos + T] + i D sin, a multiply-add and a multiply) and its real part expanded (in double) about omega = 0 for
//    the lower half of the bins and about pi for the upper half.  Near DC the monomial terms cancel to ~theta^2 of their
//    size (1e-5 for a 31 Hz band: float would keep 2 digits); the expanded form's terms are each of the size of the
//    result times the section's Q.  A value that vanishes or leaves the float range is flagged and redone in double after
//    the loop.  (Until round 4 the polynomials were expanded in x = 1 -+ w without the turn: c0 + c1 x + c2 x^2, four
//    packed operations per section pair and polynomial instead of two.)
//  * Quotients conj(gH) H / (B_s conj(w)): float, relative error ~1e-7 per bin.  H is the saved forward output
//    (evaluated in double, rounded once), so the cascade product is not re-evaluated.
//  * Running sums: float, in the basis {Re t, (1 - cos) Re t, sin Im t} instead of the three monomial sums
//    {Re(t conj(w)), g Re t, g^2 Re(t w)}: at low frequency those are nearly equal and the parameter maps downstream
//    take first and second differences of them (factors 1/theta, 1/theta^2 ~ 1e4..1e5), which float sums would not
//    survive, while the basis sums ARE those differences (second difference = -2 G1, first = 2 G2).  They are
//    converted back to (b0, b1, b2) gradients in double:
//      d/db0 = G0 - G1 - G2,   d/db1 = g G0,   d/db2 = g^2 (G0 - G1 + G2).
// Agreement with the all-double kernel ~1e-6 (tests/test_hip_kernels.py).
// Rare route of the mixed kernel: a section value that vanishes or leaves the float range (e.g. a
// band-pass numerator at DC).  All double, product of the other sections as in the kernel above;
// (inlined: an out-of-line call costs the hot loop more in saved registers than the code size does).
__device__ inline void sos_bwd_slow_section(const SosEval& e, const double* lb, const double* la, int S,
                                                               int s, cx<float> gc, cx<double> Bs, cx<double> As,
                                                               cx<float>& tb, cx<float>& ta) {
    cx<double> o(1, 0), Ap(1, 0);
    for (int t = 0; t < S; ++t) {
        if (t != s) o = o * e.poly(lb, S, t);
        Ap = Ap * e.poly(la, S, t);
    }
    const cx<double> gcd((double)gc.x, (double)gc.y);
    const cx<double> tbd = gcd * cdiv(o, Ap);      // conj(gH) H / B_s
    const cx<double> tad = cdiv(tbd * Bs, As);     // conj(gH) H / A_s
    tb = cx<float>((float)tbd.x, (float)tbd.y);
    ta = cx<float>((float)tad.x, (float)tad.y);
}

// NIW > 0: "right constant factor" mode.  The cascade's response G (channel pair c = m*Nmid + j) was multiplied by a
// real constant matrix W (Nmid x NIW) on the right, H[m][n] = sum_j G[m][j] W[j][n] (a Series of Matrix then a
// cascade-type filter): gH then holds dL/dH (planes (m*NIW + n)), and the kernel forms
//     dL/dG[m][j] = sum_n dL/dH[m][n] W[j][n]                      on the fly (no (M, No, Nmid) gradient tensor), and
//     dL/dW[j][n] += Re(conj(G[m][j]) dL/dH[m][n])                 summed over its bins (per-block partials in partW)
// -- the two composition-backward passes over the response-sized tensors disappear into this ALU-bound kernel.
struct SosRC {
    int Nmid, nbx;       // nbx: bin blocks (the grid is 1-D in this mode)
    const float* Wr;     // (Nmid, NIW) row-major
    float* partW;        // (gridDim.x, C, NIW)
    // "outer" mode (NIW == 0, oG != null): the response was applied to a signal with few columns, Y[b] = H X[b], and its
    // gradient dL/dH[m][n][f] = sum_b gY[b][m][f] conj(X[b][n][f]) is formed here from the two signals instead of being
    // read from an (M, No, Ni) tensor nobody else needs (1.6 GB written and re-read for a 32 x 32 equaliser at
    // nfft = 384000).  Channel pair c = m * oNi + n.
    const cx<float>* oG = nullptr;   // gY planes: b * o_gb + m * o_gn + f
    const cx<float>* oX = nullptr;   // X planes:  b * o_xb + n * o_xn + f
    long o_gb = 0, o_gn = 0, o_xb = 0, o_xn = 0;
    int oB = 0, oNi = 1;
};

template <int SCH, int NIW>
__global__ void __launch_bounds__(256, 2) sos_response_bwd_mixed_kernel(
    const cx<float>* __restrict__ gH, long g_pitch, const cx<float>* __restrict__ H, long h_pitch,
    const double* __restrict__ b, const double* __restrict__ a, int S, int C, double g,
    const cx<double>* __restrict__ Wd, int nfft, int bin0, int m_local, double* __restrict__ part, SosRC rc) {
    static_assert(SCH % 2 == 0, "sections are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lb = reinterpret_cast<double*>(smem);
    double* la = lb + 3 * S;
    // float tables of THIS block's section chunk, [basis][b|a][3][SCH]: the pitch is a compile-time constant, so the
    // twelve reads of a section pair are one base register (basis chosen per bin) plus immediate offsets
    float* cf = reinterpret_cast<float*>(la + 3 * S + (S & 1));   // 8-byte aligned
    // block -> (bin block bx, channel pair c).  Constant-factor mode: the Nmid pairs (m, j) of one output channel m read
    // the same NIW gradient planes; their blocks get ids 8 apart (block q runs on XCD q % 8), i.e. consecutive slots
    // of ONE XCD, so those planes cross the fabric once and come from that L2 for the other Nmid - 1 blocks
    int bx, c, nbx;
    if (NIW > 0) {
        nbx = rc.nbx;
        const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
        const int j = t % rc.Nmid, pr = (t / rc.Nmid) * 8 + xcd;        // pr = m * nbx + bx
        if (pr >= (C / rc.Nmid) * nbx) return;
        bx = pr % nbx;
        c = (pr / nbx) * rc.Nmid + j;
    } else {
        bx = blockIdx.x;
        nbx = gridDim.x;
        c = blockIdx.y;
    }
    stage_taps(b, a, S, C, c, lb, la);
    // the section polynomials turned by half a sample, about omega = 0 for the lower half of the bins and about pi for the
    // upper half (half_turn_tables: real part one multiply-add, imaginary part one multiply; sums formed in double, stored
    // in float).  Sections past the cascade's end are padding, b = a = (1, 0, 0): their sums are computed and never written
    const int s0 = blockIdx.z * SCH;
    for (int i = threadIdx.x; i < 2 * SCH; i += blockDim.x) {
        const int poly = i / SCH, q = i - poly * SCH, sidx = s0 + q;
        const double* t = poly ? la : lb;
        const bool real = sidx < S;
        const double t0 = real ? t[sidx] : 1.0, t1 = real ? t[S + sidx] : 0.0, t2 = real ? t[2 * S + sidx] : 0.0;
        half_turn_tables(t0, t1, t2, g, cf + (0 * 2 + poly) * 3 * SCH + q, cf + (1 * 2 + poly) * 3 * SCH + q, SCH);
    }
    __syncthreads();
    // running sums of section PAIRS (x: section s0+2u, y: section s0+2u+1): every operation below is a
    // packed-float instruction (v_pk_*), the only kind that issues two lanes' worth per cycle slot --
    // unpacked FP32 and FP64 FMAs issue at the same rate, so "float instead of double" alone buys nothing
    f2 acc[6][SCH / 2];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int u = 0; u < SCH / 2; ++u) acc[p][u] = (f2)(0.f);
    const float eps = eps_of<float>();
    constexpr int NW = NIW > 0 ? NIW : 1;
    float wrow[NW];
    f2 accw[NW];
    const cx<float>* gbase = gH + (size_t)c * g_pitch;
    if (NIW > 0) {
        const int mrow = c / rc.Nmid, j = c - mrow * rc.Nmid;
        gbase = gH + (size_t)mrow * NIW * g_pitch;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            wrow[n] = rc.Wr[j * NIW + n];
            accw[n] = (f2)(0.f);
        }
    }

    const int fstride = nbx * 256;
    const bool outer = NIW == 0 && rc.oG != nullptr;
    // The operands of bin f + fstride are requested before bin f is worked on (two wavefronts per SIMD: nothing else would
    // cover the round trip -- the twiddle's address depends on the bin number, its load used to sit alone in front of the
    // arithmetic).  Row-major bin order: the (row, column) pair of the element advances incrementally instead of by a
    // division per bin.
    struct Operands {
        cx<float> h;
        f2 gv[NW];
        cx<double> w1;
        int k;
    };
    int fn = bx * 256 + threadIdx.x;                 // element the next fetch is for
    const int L2r = bin0 < 0 ? -bin0 : 1, Lh = nfft >> 1, L1r = Lh / L2r;
    int k1 = fn / L2r, k2 = fn - k1 * L2r;
    const int dk1 = fstride / L2r, dk2 = fstride - dk1 * L2r;
    auto fetch = [&](Operands& o) {
        const int f = fn < m_local ? fn : m_local - 1;        // (past the end: a valid address, the values are not used)
        o.k = bin0 >= 0 ? bin0 + f : (fn >= Lh ? Lh : k1 + L1r * k2);
        o.w1 = Wd[o.k < nfft ? o.k : o.k - nfft];
        o.h = H[(size_t)c * h_pitch + f];
        if (NIW > 0) {
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const cx<float> t = gbase[(size_t)n * g_pitch + f];
                o.gv[n] = f2{t.x, t.y};
            }
        } else if (!outer) {
            const cx<float> t = gbase[f];
            o.gv[0] = f2{t.x, t.y};
        }
        fn += fstride;
        k1 += dk1;
        k2 += dk2;
        if (k2 >= L2r) {
            k2 -= L2r;
            ++k1;
        }
    };
    Operands cur, nxt;
    fetch(cur);
    for (int f = bx * 256 + threadIdx.x; f < m_local; f += fstride, cur = nxt) {
        fetch(nxt);
        const cx<float> h = cur.h;
        cx<float> gin;
        if (NIW > 0) {
            // dL/dG = sum_n W[j][n] dL/dH[m][n] and the dL/dW sums, one packed instruction per plane each
            f2 gi = (f2)(0.f);
            const f2 hv = {h.x, h.y};
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                gi += cur.gv[n] * wrow[n];
                if (blockIdx.z == 0) accw[n] += hv * cur.gv[n];     // (Re, Im halves; added after the loop)
            }
            gin = cx<float>(gi.x, gi.y);
        } else if (outer) {
            const int mo = c / rc.oNi, no = c - mo * rc.oNi;
            gin = cx<float>(0.f, 0.f);
            for (int bb = 0; bb < rc.oB; ++bb)
                fma_cxc(gin, rc.oG[(size_t)bb * rc.o_gb + (size_t)mo * rc.o_gn + f], rc.oX[(size_t)bb * rc.o_xb + (size_t)no * rc.o_xn + f]);
        } else {
            gin = cx<float>(cur.gv[0].x, cur.gv[0].y);
        }
        if (h.x == eps && h.y == 0.f) continue;   // guarded bin (prod A == 0): constant, zero gradient
        const int k = cur.k;
        const cx<double> w1 = cur.w1;
        const bool low = 4 * (long)k < nfft;      // omega nearer to 0 than to pi
        const float xr = (float)(low ? 1.0 - w1.x : 1.0 + w1.x), xi = (float)(-w1.y);      // 1 -+ cos(omega) (formed in double), sin(omega)
        const float uu = (float)(1.0 - w1.x);
        const f2* tb = reinterpret_cast<const f2*>(cf + (low ? 0 : 6 * SCH));      // [b|a][3][SCH / 2] section pairs
        const cx<float> gc(gin.x, -gin.y);
        // With t = conj(gH) H / (B_s conj(w)) the three tap gradients are sums of Re(t conj(w)), g Re(t), g^2 Re(t w); the
        // running sums are kept in the basis {Re t, (1 - cos) Re t, sin Im t} -- well-scaled quantities, combined in double at
        // the end -- through three per-bin vectors shared by all sections:  sum_p += (q_p.x Br + q_p.y Bi) / |B_s|^2
        const cx<float> q0 = gc * h, q1(uu * q0.x, uu * q0.y), q2(xi * q0.y, -(xi * q0.x));
        unsigned slow = 0;
        // The range test of a section pair -- |B|^2 |A|^2 a normal number in both halves -- used to cost two v_cmp_class, two
        // selects of the reciprocal, the flag bits and their hazard slots per pair (a fifth of the loop beside its 33 packed
        // instructions).  Now the pairs are taken in groups of GP: the group's values are formed with the RAW reciprocals (a
        // lane with a bad section holds inf / NaN there, nothing is accumulated yet), one packed min and one `nn * 0` sum per
        // pair watch the range (min >= FLT_MIN: no zero / denormal; 0 * nn == 0: no inf / NaN), and ONE test per group and lane
        // decides between the plain accumulation and the careful per-section form below (the lanes that need it: rare).
        constexpr int GP = (SCH / 2) % 3 == 0 ? 3 : (SCH / 2);
#pragma unroll
        for (int g0 = 0; g0 < SCH / 2; g0 += GP) {
            f2 sBr[GP], sBi[GP], sAr[GP], sAi[GP];
            f2 mn = {3.0e38f, 3.0e38f}, zz = {0.f, 0.f};
#pragma unroll
            for (int v = 0; v < GP; ++v) {
                const int u = g0 + v;
                const f2 b0 = tb[u], b1 = tb[SCH / 2 + u], b2 = tb[SCH + u];
                const f2 a0 = tb[3 * SCH / 2 + u], a1 = tb[2 * SCH + u], a2 = tb[5 * SCH / 2 + u];
                const f2 Br = b0 + b1 * xr, Bi = b2 * xi;
                const f2 Ar = a0 + a1 * xr, Ai = a2 * xi;
                const f2 nb = Br * Br + Bi * Bi, na = Ar * Ar + Ai * Ai;
                const f2 nn = nb * na;
                mn = __builtin_elementwise_min(mn, nn);
                zz = __builtin_elementwise_fma(nn, (f2)(0.f), zz);
                f2 inv;
                inv.x = __builtin_amdgcn_rcpf(nn.x);
                inv.y = __builtin_amdgcn_rcpf(nn.y);
                const f2 ib = inv * na, ia = -(inv * nb);
                sBr[v] = Br * ib; sBi[v] = Bi * ib; sAr[v] = Ar * ia; sAi[v] = Ai * ia;
            }
            const bool okg = fminf(mn.x, mn.y) >= 1.17549435e-38f && (zz.x + zz.y) == 0.f;
            if (__builtin_expect(okg, 1)) {
#pragma unroll
                for (int v = 0; v < GP; ++v) {
                    const int u = g0 + v;
                    // (one fused multiply-add per statement: "acc += x + y" would be a multiply, an fma and an add)
                    acc[0][u] = sBr[v] * q0.x + acc[0][u]; acc[0][u] = sBi[v] * q0.y + acc[0][u];
                    acc[1][u] = sBr[v] * q1.x + acc[1][u]; acc[1][u] = sBi[v] * q1.y + acc[1][u];
                    acc[2][u] = sBr[v] * q2.x + acc[2][u]; acc[2][u] = sBi[v] * q2.y + acc[2][u];
                    acc[3][u] = sAr[v] * q0.x + acc[3][u]; acc[3][u] = sAi[v] * q0.y + acc[3][u];
                    acc[4][u] = sAr[v] * q1.x + acc[4][u]; acc[4][u] = sAi[v] * q1.y + acc[4][u];
                    acc[5][u] = sAr[v] * q2.x + acc[5][u]; acc[5][u] = sAi[v] * q2.y + acc[5][u];
                }
            } else {
#pragma unroll
                for (int v = 0; v < GP; ++v) {
                    const int u = g0 + v;
                    const f2 b0 = tb[u], b1 = tb[SCH / 2 + u], b2 = tb[SCH + u];
                    const f2 a0 = tb[3 * SCH / 2 + u], a1 = tb[2 * SCH + u], a2 = tb[5 * SCH / 2 + u];
                    const f2 Br = b0 + b1 * xr, Bi = b2 * xi;
                    const f2 Ar = a0 + a1 * xr, Ai = a2 * xi;
                    const f2 nb = Br * Br + Bi * Bi, na = Ar * Ar + Ai * Ai;
                    // one reciprocal per section for both quotients: 1/|B|^2 = |A|^2 / (|B|^2 |A|^2).  A product that is not a
                    // normal number (a norm vanished, or left the float range) is flagged for the double route
                    const f2 nn = nb * na;
                    const bool ok0 = __builtin_amdgcn_classf(nn.x, 0x100);             // +normal
                    const bool ok1 = __builtin_amdgcn_classf(nn.y, 0x100);
                    slow |= (ok0 ? 0u : (1u << (2 * u))) | (ok1 ? 0u : (2u << (2 * u)));
                    f2 inv;
                    inv.x = ok0 ? __builtin_amdgcn_rcpf(nn.x) : 0.f;
                    inv.y = ok1 ? __builtin_amdgcn_rcpf(nn.y) : 0.f;
                    const f2 ib = inv * na, ia = -(inv * nb);
                    const f2 Brs = Br * ib, Bis = Bi * ib, Ars = Ar * ia, Ais = Ai * ia;
                    acc[0][u] = Brs * q0.x + acc[0][u]; acc[0][u] = Bis * q0.y + acc[0][u];
                    acc[1][u] = Brs * q1.x + acc[1][u]; acc[1][u] = Bis * q1.y + acc[1][u];
                    acc[2][u] = Brs * q2.x + acc[2][u]; acc[2][u] = Bis * q2.y + acc[2][u];
                    acc[3][u] = Ars * q0.x + acc[3][u]; acc[3][u] = Ais * q0.y + acc[3][u];
                    acc[4][u] = Ars * q1.x + acc[4][u]; acc[4][u] = Ais * q1.y + acc[4][u];
                    acc[5][u] = Ars * q2.x + acc[5][u]; acc[5][u] = Ais * q2.y + acc[5][u];
                }
            }
        }
        if (slow) {   // rare: redo the flagged sections in double
            SosEval e;
            e.z1 = cx<double>(g * w1.x, g * w1.y);
            e.z2 = e.z1 * e.z1;
            const float cwf = (float)w1.x;
#pragma unroll
            for (int q = 0; q < SCH; ++q) {
                if (((slow >> q) & 1u) && s0 + q < S) {
                    const int s = s0 + q;
                    const cx<double> Bs = e.poly(lb, S, s), As = e.poly(la, S, s);
                    cx<float> tb2, ta2;
                    sos_bwd_slow_section(e, lb, la, S, s, gc, Bs, As, tb2, ta2);
                    // the careful route returns conj(gH) H / B_s: times w is the quotient by the turned polynomial
                    const cx<float> tt(tb2.x * cwf + tb2.y * xi, tb2.y * cwf - tb2.x * xi), ua(ta2.x * cwf + ta2.y * xi, ta2.y * cwf - ta2.x * xi);
                    const float v[6] = {tt.x, uu * tt.x, xi * tt.y, -ua.x, -(uu * ua.x), -(xi * ua.y)};
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        if (q & 1) acc[p][q / 2].y += v[p];
                        else acc[p][q / 2].x += v[p];
                    }
                }
            }
        }
    }
    // flatten to [(i*3 + p)*SCH + q] for the reduction
    float flat[6 * SCH];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int u = 0; u < SCH / 2; ++u) {
            flat[p * SCH + 2 * u] = acc[p][u].x;
            flat[p * SCH + 2 * u + 1] = acc[p][u].y;
        }
    __shared__ float red[4][6 * SCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = 0, cnt = 0, dup = 0;
    wave_reduce_scatter<float, 6 * SCH, 32>(flat, lane, off, cnt, dup);
    if ((lane & dup) == 0) {
#pragma unroll
        for (int v = 0; v < 6 * SCH; ++v)
            if (v < cnt) red[wave][off + v] = flat[v];
    }
    __syncthreads();
    if (threadIdx.x < 2 * SCH) {
        const int i = threadIdx.x / SCH, q = threadIdx.x % SCH;
        const int s = s0 + q;
        if (s < S) {
            double G[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int j = (i * 3 + p) * SCH + q;
                G[p] = (double)red[0][j] + (double)red[1][j] + (double)red[2][j] + (double)red[3][j];
            }
            // Re(t conj(w)) = (1 - u) Re t - sin Im t,   Re(t w) = (1 - u) Re t + sin Im t,   u = 1 - cos
            const double out[3] = {G[0] - G[1] - G[2], g * G[0], g * g * (G[0] - G[1] + G[2])};
#pragma unroll
            for (int p = 0; p < 3; ++p) part[((((size_t)bx * 2 + i) * 3 + p) * S + s) * C + c] = out[p];
        }
    }
    if (NIW > 0 && blockIdx.z == 0) {
        __shared__ float redw[4][NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            float v = accw[n].x + accw[n].y;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) redw[wave][n] = v;
        }
        __syncthreads();
        if (threadIdx.x < NW)
            rc.partW[((size_t)bx * C + c) * NW + threadIdx.x] =
                redw[0][threadIdx.x] + redw[1][threadIdx.x] + redw[2][threadIdx.x] + redw[3][threadIdx.x];
    }
}

__global__ void __launch_bounds__(256) geq_sections_kernel(const void* __restrict__ gain, int in_kind, int nb, int C,
                                                          const double* __restrict__ k, double* __restrict__ b,
                                                          double* __restrict__ a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nb * C) return;
    const int st = nb * C;
    double bb[3], aa[3];
    geq_section_of(gain, in_kind, idx, idx / C, nb, k, bb, aa);
    b[idx] = bb[0]; b[idx + st] = bb[1]; b[idx + 2 * st] = bb[2];
    a[idx] = aa[0]; a[idx + st] = aa[1]; a[idx + 2 * st] = aa[2];
}

// gb / ga: (nblk, 3, nb, C) partial sums blk_stride elements apart (nblk = 1: plain gradients);
// summed here in block order, so the bin-block partials of the cascade backward need no separate
// reduction launch.
// The launch can carry a second small reduction in its tail blocks (wrows > 0): gW[e] = sum over wrows rows of
// partW[row * wn + e] -- the constant factor's gradient partials of the fused Matrix-then-GEQ operator -- one wavefront
// per entry (lanes stride over the rows, fixed butterfly: deterministic); a launch of its own costs more than the sum.
__global__ void __launch_bounds__(256) geq_sections_bwd_kernel(const void* __restrict__ gain, int in_kind,
                                                              const double* __restrict__ gb,
                                                              const double* __restrict__ ga, long blk_stride, int nblk,
                                                              int nb, int C, const double* __restrict__ k,
                                                              void* __restrict__ ggain, int main_blocks,
                                                              const void* __restrict__ partW_, int wrows, int wn,
                                                              void* __restrict__ gW_, int w_f64) {
    if ((int)blockIdx.x >= main_blocks) {
        const int e = ((int)blockIdx.x - main_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (e >= wn) return;
        if (w_f64) {
            const double* partW = reinterpret_cast<const double*>(partW_);
            double v = 0.0;
            for (int r = lane; r < wrows; r += 64) v += partW[(size_t)r * wn + e];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) reinterpret_cast<double*>(gW_)[e] = v;
            return;
        }
        const float* partW = reinterpret_cast<const float*>(partW_);
        float v = 0.f;
        for (int r = lane; r < wrows; r += 64) v += partW[(size_t)r * wn + e];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) reinterpret_cast<float*>(gW_)[e] = v;
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nb * C) return;
    const int band = idx / C;
    const int st = nb * C;
    double raw;
    const double g = geq_linear_gain(gain, in_kind, idx, &raw);
    double B0 = 0, B1 = 0, B2 = 0, A0 = 0, A1 = 0, A2 = 0;
    int blk = 0;
    for (; blk + 4 <= nblk; blk += 4) {   // 24 independent loads in flight, summed in block order
        double v[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* pb = gb + (size_t)(blk + u) * blk_stride + idx;
            const double* pa = ga + (size_t)(blk + u) * blk_stride + idx;
            v[u][0] = pb[0]; v[u][1] = pb[st]; v[u][2] = pb[2 * st];
            v[u][3] = pa[0]; v[u][4] = pa[st]; v[u][5] = pa[2 * st];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            B0 += v[u][0]; B1 += v[u][1]; B2 += v[u][2];
            A0 += v[u][3]; A1 += v[u][4]; A2 += v[u][5];
        }
    }
    for (; blk < nblk; ++blk) {
        const double* pb = gb + (size_t)blk * blk_stride + idx;
        const double* pa = ga + (size_t)blk * blk_stride + idx;
        B0 += pb[0]; B1 += pb[st]; B2 += pb[2 * st];
        A0 += pa[0]; A1 += pa[st]; A2 += pa[2 * st];
    }
    const double dg = geq_design_bwd(band, nb, g, k, B0, B1, B2, A0, A1, A2);   // dL/dg
    geq_store_gain_grad(ggain, in_kind, idx, dg, g, raw);
}

static int g_sos_chunk = 0;
static int g_rc_fast = 6;   // cascade-times-matrix forward: float evaluation in the 1 -+ w basis, section pairs per loop trip (1 | 2 | 3 | 6; 0: the double kernel)
static int g_sos_blocks = 0;

// Bin blocks per channel pair of the backward kernels: the whole grid (blocks x C channel pairs x section chunks) is ONE
// round of resident workgroups when the channel count allows it (two workgroups of 256 threads per CU: the kernels run at
// two wavefronts per SIMD), so that every thread walks many bins and the prologue / reduction of a workgroup is paid once
// per CU slot -- at config 2 (64 pairs, 48001 bins) 8 blocks per pair (512 workgroups, 24 bins per thread) run in 57 us,
// the 32 blocks of before (2048 workgroups, 2.7 rounds) in 72.
static int sos_chunk_of(int S, bool mixed) {
    if (mixed) {
        const int want = g_sos_chunk > 0 ? g_sos_chunk : 12;
        return (S <= 4 || want <= 4) ? 4 : (S <= 6 || want <= 6) ? 6 : (S <= 8 || want <= 8) ? 8 : 12;
    }
    const int sch = (g_sos_chunk == 3 || g_sos_chunk == 4 || g_sos_chunk == 6 || g_sos_chunk == 12) ? g_sos_chunk : 6;
    return S > 4 ? sch : 4;
}

static int sos_blocks(int m_local, int C, int S, bool mixed) {
    int cus = 256;       // of the CURRENT device (asked per call: a few hundred nanoseconds beside a launch)
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int nz = cdiv_i(S, sos_chunk_of(S, mixed));
    int nb = g_sos_blocks > 0 ? g_sos_blocks : (2 * cus) / (C * nz > 0 ? C * nz : 1);
    const int cap = cdiv_i(m_local, 256);
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

template <typename T>
static int delay_impl(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local, void* H,
                      long h_pitch, void* stream) {
    FL_REQUIRE(m && amp && W && H, "delay_response: null pointer");
    FL_REQUIRE(h_pitch >= m_local, "delay_response: h_pitch must be >= m_local");
    FL_REQUIRE(C > 0 && C <= 65535 && nfft > 0 && bin_range_ok(bin0, m_local, nfft) && m_local >= 0, "delay_response: bad sizes");
    if (m_local == 0) return FL_OK;
    dim3 grid(cdiv_i(m_local, 256), C);
    hipLaunchKernelGGL((delay_response_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, m, (const T*)amp,
                       (const cx<T>*)W, nfft, 1.0 / (double)nfft, bin0, m_local, (cx<T>*)H, h_pitch);
    FL_CHECK_LAUNCH("delay_response");
    return FL_OK;
}

template <typename T>
static int sos_impl(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                    int m_local, void* H, long h_pitch, void* stream, bool float_eval = false,
                    GeqDesign gd = GeqDesign{nullptr, 0, nullptr, nullptr, nullptr}) {
    FL_REQUIRE(b && a && H && Wd, "sos_response: null pointer");
    FL_REQUIRE(h_pitch >= m_local, "sos_response: h_pitch must be >= m_local");
    FL_REQUIRE(S > 0 && S <= 1024 && C > 0 && C <= 65535 && nfft > 0 && bin_range_ok(bin0, m_local, nfft) && m_local >= 0, "sos_response: bad sizes");
    if (m_local == 0) return FL_OK;
    if constexpr (sizeof(T) == 4) {
        if (g_rc_fast && float_eval) {      // float evaluation in the 1 -+ w basis, section pairs packed
            const int SP = (S + 1) & ~1;
            hipLaunchKernelGGL(sos_response_fast_kernel, dim3(cdiv_i(m_local, kSosFastBins), C), dim3(256), (size_t)12 * SP * sizeof(float),
                               (hipStream_t)stream, (const double*)b, (const double*)a, S, C, gamma, (const cx<double>*)Wd, nfft, bin0,
                               m_local, (cx<float>*)H, h_pitch, gd);
            FL_CHECK_LAUNCH("sos_response_fast");
            return FL_OK;
        }
    }
    if (gd.gain) {      // the double kernel reads its sections: design them first
        hipLaunchKernelGGL(geq_sections_kernel, dim3(cdiv_i(S * C, 256)), dim3(256), 0, (hipStream_t)stream, gd.gain, gd.in_kind, S, C, gd.k,
                           gd.b_out, gd.a_out);
        FL_CHECK_LAUNCH("geq_sections");
    }
    dim3 grid(cdiv_i(m_local, 256), C);
    hipLaunchKernelGGL((sos_response_kernel<T>), grid, dim3(256), (size_t)6 * S * sizeof(double), (hipStream_t)stream, (const double*)b, (const double*)a, S, C,
                       gamma, (const cx<double>*)Wd, nfft, bin0, m_local, (cx<T>*)H, h_pitch);
    FL_CHECK_LAUNCH("sos_response");
    return FL_OK;
}

template <typename T>
static int sos_bwd_impl(const void* gH, long g_pitch, const void* H, long h_pitch, const void* b, const void* a, int S, int C,
                        double gamma, const void* Wd, int nfft, int bin0, int m_local, void* part, void* stream,
                        int rc_ni = 0, SosRC rc = SosRC{0, 0, nullptr, nullptr}, SosRCd rcd = SosRCd{0, 0, nullptr, nullptr}) {
    FL_REQUIRE((gH || rc.oG) && b && a && part && Wd, "sos_response_bwd: null pointer");
    FL_REQUIRE(g_pitch >= m_local && (!H || h_pitch >= m_local), "sos_response_bwd: g_pitch / h_pitch must be >= m_local");
    FL_REQUIRE(S > 0 && S <= 1024 && C > 0 && C <= 65535 && nfft > 0 && bin_range_ok(bin0, m_local, nfft) && m_local > 0, "sos_response_bwd: bad sizes");
    if constexpr (sizeof(T) == 4) {
        if (H) {
#define FL_SOS_MIX(SC)                                                                                              \
    {                                                                                                               \
        dim3 grid(sos_blocks(m_local, C, S, true), C, cdiv_i(S, SC));                                               \
        if (rc_ni > 0) {                                                                                           \
            rc.nbx = sos_blocks(m_local, C, S, true);                                                              \
            grid = dim3(cdiv_i((C / rc.Nmid) * rc.nbx, 8) * 8 * rc.Nmid, 1, cdiv_i(S, SC));                        \
        }                                                                                                          \
        const size_t lds = (size_t)(6 * S + 2) * sizeof(double) + (size_t)12 * SC * sizeof(float);    \
        FL_SOS_MIX_N(SC, 0) else FL_SOS_MIX_N(SC, 2) else FL_SOS_MIX_N(SC, 4) else FL_SOS_MIX_N(SC, 8)            \
        else FL_SOS_MIX_N(SC, 16) else {                                                                          \
            set_error("sos_response_bwd: no kernel for %d input channels of the constant factor", rc_ni);          \
            return FL_ERR_UNSUPPORTED;                                                                             \
        }                                                                                                          \
    }
#define FL_SOS_MIX_N(SC, NIW_)                                                                                      \
    if (rc_ni == NIW_)                                                                                              \
        hipLaunchKernelGGL((sos_response_bwd_mixed_kernel<SC, NIW_>), grid, dim3(256), lds, (hipStream_t)stream,   \
                           (const cx<float>*)gH, g_pitch, (const cx<float>*)H, h_pitch, (const double*)b,           \
                           (const double*)a, S, C, gamma, (const cx<double>*)Wd, nfft, bin0, m_local, (double*)part, rc);
            const int sc = sos_chunk_of(S, true);
            if (sc == 4) FL_SOS_MIX(4)
            else if (sc == 6) FL_SOS_MIX(6)
            else if (sc == 8) FL_SOS_MIX(8)
            else FL_SOS_MIX(12)
#undef FL_SOS_MIX
#undef FL_SOS_MIX_N
            FL_CHECK_LAUNCH("sos_response_bwd");
            return FL_OK;
        }
    }
    FL_REQUIRE(!rc.oG, "sos_response_bwd: the outer-product mode needs float32 and the saved forward response");
    if (rc_ni > 0) {      // constant-factor mode of the all-double kernel: one section chunk (the NIW gradient planes are read once)
        if constexpr (sizeof(T) == 8) {
            FL_REQUIRE(H && rcd.Wr && rcd.partW && rcd.Nmid > 0, "sos_response_bwd: constant-factor mode needs the saved response");
            rcd.nbx = sos_blocks(m_local, C, S, false);
            const dim3 grid(cdiv_i((C / rcd.Nmid) * rcd.nbx, 8) * 8 * rcd.Nmid, 1, cdiv_i(S, 6));
#define FL_SOS_BWD_RC(NIW_)                                                                                              \
    if (rc_ni == NIW_)                                                                                                   \
        hipLaunchKernelGGL((sos_response_bwd_kernel<double, 6, NIW_>), grid, dim3(256), (size_t)6 * S * sizeof(double), \
                           (hipStream_t)stream, (const cx<double>*)gH, g_pitch, (const cx<double>*)H, h_pitch,           \
                           (const double*)b, (const double*)a, S, C, gamma, (const cx<double>*)Wd, nfft, bin0, m_local,  \
                           (double*)part, rcd);
            FL_SOS_BWD_RC(2) else FL_SOS_BWD_RC(4) else FL_SOS_BWD_RC(8) else FL_SOS_BWD_RC(16) else {
                set_error("sos_response_bwd: no kernel for %d input channels of the constant factor", rc_ni);
                return FL_ERR_UNSUPPORTED;
            }
#undef FL_SOS_BWD_RC
            FL_CHECK_LAUNCH("sos_response_bwd_rc");
            return FL_OK;
        }
        set_error("sos_response_bwd: the float32 constant-factor mode needs the saved forward response");
        return FL_ERR_BAD_ARG;
    }
    const int sch = sos_chunk_of(S, false);
#define FL_SOS_BWD(SC)                                                                                              \
    {                                                                                                               \
        dim3 grid(sos_blocks(m_local, C, S, false), C, cdiv_i(S, SC));                                              \
        hipLaunchKernelGGL((sos_response_bwd_kernel<T, SC>), grid, dim3(256), (size_t)6 * S * sizeof(double),      \
                           (hipStream_t)stream, (const cx<T>*)gH, g_pitch, (const cx<T>*)H, h_pitch, (const double*)b, (const double*)a, S, C, gamma, \
                           (const cx<double>*)Wd, nfft, bin0, m_local, (double*)part, SosRCd{0, 0, nullptr, nullptr}); \
    }
    if (sch == 12) FL_SOS_BWD(12)
    else if (sch == 6) FL_SOS_BWD(6)
    else if (sch == 3) FL_SOS_BWD(3)
    else FL_SOS_BWD(4)
#undef FL_SOS_BWD
    FL_CHECK_LAUNCH("sos_response_bwd");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_delay_response_c64(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local,
                          void* H, long h_pitch, void* stream) {
    return delay_impl<float>(m, amp, C, W, nfft, bin0, m_local, H, h_pitch, stream);
}
int fl_delay_response_c128(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local,
                          void* H, long h_pitch, void* stream) {
    return delay_impl<double>(m, amp, C, W, nfft, bin0, m_local, H, h_pitch, stream);
}
int fl_sos_response_c64(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                        int m_local, void* H, long h_pitch, void* stream) {
    return sos_impl<float>(b, a, S, C, gamma, Wd, nfft, bin0, m_local, H, h_pitch, stream);
}
int fl_sos_response_f32eval_c64(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                                int m_local, void* H, long h_pitch, void* stream) {
    return sos_impl<float>(b, a, S, C, gamma, Wd, nfft, bin0, m_local, H, h_pitch, stream, true);
}
int fl_geq_response_c64(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int C, double gamma, const void* Wd,
                        int nfft, int bin0, int m_local, void* H, long h_pitch, int float_eval, void* stream) {
    FL_REQUIRE(gain && consts && b && a, "geq_response: null pointer");
    FL_REQUIRE(in_kind >= 0 && in_kind <= 4 && nb >= 4, "geq_response: in_kind in [0, 4], at least four bands");
    const GeqDesign gd{gain, in_kind, (const double*)consts, (double*)b, (double*)a};
    if (m_local == 0) {
        hipLaunchKernelGGL(geq_sections_kernel, dim3(cdiv_i(nb * C, 256)), dim3(256), 0, (hipStream_t)stream, gain, in_kind, nb, C,
                           (const double*)consts, (double*)b, (double*)a);
        FL_CHECK_LAUNCH("geq_sections");
        return FL_OK;
    }
    return sos_impl<float>(b, a, nb, C, gamma, Wd, nfft, bin0, m_local, H, h_pitch, stream, float_eval != 0, gd);
}
int fl_sos_response_c128(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                        int m_local, void* H, long h_pitch, void* stream) {
    return sos_impl<double>(b, a, S, C, gamma, Wd, nfft, bin0, m_local, H, h_pitch, stream);
}
int fl_sos_bwd_blocks(int m_local, int C, int S, int mixed) { return sos_blocks(m_local, C, S, mixed != 0); }
int fl_debug_set_rc_fast(int on) {
    g_rc_fast = on;
    return FL_OK;
}
int fl_debug_set_sos_chunk(int sections_per_thread) {
    g_sos_blocks = sections_per_thread / 100;      // hundreds digit(s): blocks per channel (0 = default)
    sections_per_thread %= 100;
    g_sos_chunk = sections_per_thread;
    return FL_OK;
}

int fl_geq_sections(const void* gain, int in_kind, int nb, int C, const void* consts, void* b, void* a, void* stream) {
    FL_REQUIRE(gain && consts && b && a, "geq_sections: null pointer");
    FL_REQUIRE(in_kind >= 0 && in_kind <= 4, "geq_sections: in_kind must be 0 (dB, f64), 1 / 2 (|x|, f64 / f32) or 3 / 4 (sigmoid(x), f64 / f32)");
    FL_REQUIRE(nb >= 4 && C > 0, "geq_sections: need >= 4 bands (gain, two shelves, one peak) and C > 0");
    hipLaunchKernelGGL(geq_sections_kernel, dim3(cdiv_i((long)nb * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gain, in_kind, nb, C, (const double*)consts, (double*)b, (double*)a);
    FL_CHECK_LAUNCH("geq_sections");
    return FL_OK;
}
int fl_geq_sections_bwd(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                        int C, const void* consts, void* ggain, void* stream) {
    FL_REQUIRE(gain && gb && ga && consts && ggain, "geq_sections_bwd: null pointer");
    FL_REQUIRE(in_kind >= 0 && in_kind <= 4, "geq_sections_bwd: bad in_kind");
    FL_REQUIRE(nb >= 4 && C > 0 && nblk >= 1 && blk_stride >= 0, "geq_sections_bwd: bad sizes");
    const int mb = cdiv_i((long)nb * C, 256);
    hipLaunchKernelGGL(geq_sections_bwd_kernel, dim3(mb), dim3(256), 0, (hipStream_t)stream,
                       gain, in_kind, (const double*)gb, (const double*)ga, blk_stride, nblk, nb, C,
                       (const double*)consts, ggain, mb, (const void*)nullptr, 0, 0, (void*)nullptr, 0);
    FL_CHECK_LAUNCH("geq_sections_bwd");
    return FL_OK;
}
static int geq_bwd_w_impl(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                          int C, const void* consts, void* ggain, const void* partW, int wrows, int wn, void* gW, int w_f64, void* stream) {
    FL_REQUIRE(gain && gb && ga && consts && ggain && partW && gW, "geq_sections_bwd_w: null pointer");
    FL_REQUIRE(nb >= 4 && C > 0 && nblk > 0 && in_kind >= 0 && in_kind <= 4 && wrows > 0 && wn > 0, "geq_sections_bwd_w: bad sizes");
    const int mb = cdiv_i((long)nb * C, 256);
    hipLaunchKernelGGL(geq_sections_bwd_kernel, dim3(mb + cdiv_i(wn, 4)), dim3(256), 0, (hipStream_t)stream,
                       gain, in_kind, (const double*)gb, (const double*)ga, blk_stride, nblk, nb, C,
                       (const double*)consts, ggain, mb, partW, wrows, wn, gW, w_f64);
    FL_CHECK_LAUNCH("geq_sections_bwd_w");
    return FL_OK;
}
int fl_geq_sections_bwd_w(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                          int C, const void* consts, void* ggain, const void* partW, int wrows, int wn, void* gW, void* stream) {
    return geq_bwd_w_impl(gain, in_kind, gb, ga, blk_stride, nblk, nb, C, consts, ggain, partW, wrows, wn, gW, 0, stream);
}
int fl_geq_sections_bwd_w64(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                            int C, const void* consts, void* ggain, const void* partW, int wrows, int wn, void* gW, void* stream) {
    return geq_bwd_w_impl(gain, in_kind, gb, ga, blk_stride, nblk, nb, C, consts, ggain, partW, wrows, wn, gW, 1, stream);
}
int fl_sos_response_bwd_c64(const void* gH, long g_pitch, const void* H, long h_pitch, const void* b, const void* a, int S,
                            int C, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* part, void* stream) {
    return sos_bwd_impl<float>(gH, g_pitch, H, h_pitch, b, a, S, C, gamma, Wd, nfft, bin0, m_local, part, stream);
}
}  // extern "C"
template <typename T>
static int rc_impl(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma,
                   const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch,
                   int float_eval, void* stream, GeqDesign gd) {
    FL_REQUIRE(b && a && Wr && Wd && G && H, "sos_response_rc: null pointer");
    FL_REQUIRE(g_pitch >= m_local && h_pitch >= m_local, "sos_response_rc: pitches must be >= m_local");
    FL_REQUIRE(S > 0 && S <= 64 && No > 0 && No <= 65535 && Nmid > 0 && Nmid <= 32 && nfft > 0 && bin_range_ok(bin0, m_local, nfft) &&
                   m_local >= 0, "sos_response_rc: bad sizes");
    if (m_local == 0) return FL_OK;
    dim3 grid(cdiv_i(m_local, 256), No);
    const size_t lds = (size_t)Nmid * 6 * S * sizeof(double) + (size_t)Nmid * Ni * sizeof(T);
    FL_REQUIRE(lds <= 64 * 1024, "sos_response_rc: the coefficient tables of one output row exceed 64 KB of LDS");
    const size_t lds_fast = ((size_t)Nmid * 12 * ((S + 1) & ~1) + (size_t)Nmid * Ni) * sizeof(float);
    // the float kernel's threads take bin PAIRS: half the elements (+ the Nyquist element as a pair of its own in row-major order)
    const int npairs = bin0 >= 0 ? cdiv_i(m_local, 2) : (((nfft / 2 / (-bin0)) + 1) / 2) * (-bin0) + 1;
    const dim3 grid_fast(cdiv_i(npairs, 256), No);
    if constexpr (sizeof(T) == 4) {
        if (g_rc_fast && float_eval) {      // second generation (cascade2.hip); shapes it does not take fall through
            const int rc2 = rc_ba_launch(b, a, S, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H, h_pitch, stream, gd);
            if (rc2 != FL_ERR_UNSUPPORTED) return rc2;
        }
    }
#define FL_RC_FWD(NIW_)                                                                                                      \
    if constexpr (sizeof(T) == 4) if (Ni == NIW_ && g_rc_fast && float_eval) {                                                                             \
        if (g_rc_fast == 2)                                                                                                  \
            hipLaunchKernelGGL((sos_response_rc_fast_kernel<NIW_, 2>), grid_fast, dim3(256), lds_fast, (hipStream_t)stream,       \
                               (const double*)b, (const double*)a, S, No * Nmid, Nmid, (const float*)Wr, gamma,              \
                               (const cx<double>*)Wd, nfft, bin0, m_local, (cx<float>*)G, g_pitch, (cx<float>*)H, h_pitch, gd);  \
        else if (g_rc_fast == 3)                                                                                             \
            hipLaunchKernelGGL((sos_response_rc_fast_kernel<NIW_, 3>), grid_fast, dim3(256), lds_fast, (hipStream_t)stream,       \
                               (const double*)b, (const double*)a, S, No * Nmid, Nmid, (const float*)Wr, gamma,              \
                               (const cx<double>*)Wd, nfft, bin0, m_local, (cx<float>*)G, g_pitch, (cx<float>*)H, h_pitch, gd);  \
        else if (g_rc_fast == 6)                                                                                             \
            hipLaunchKernelGGL((sos_response_rc_fast_kernel<NIW_, 6>), grid_fast, dim3(256), lds_fast, (hipStream_t)stream,       \
                               (const double*)b, (const double*)a, S, No * Nmid, Nmid, (const float*)Wr, gamma,              \
                               (const cx<double>*)Wd, nfft, bin0, m_local, (cx<float>*)G, g_pitch, (cx<float>*)H, h_pitch, gd);  \
        else                                                                                                                 \
            hipLaunchKernelGGL((sos_response_rc_fast_kernel<NIW_>), grid_fast, dim3(256), lds_fast, (hipStream_t)stream,          \
                               (const double*)b, (const double*)a, S, No * Nmid, Nmid, (const float*)Wr, gamma,              \
                               (const cx<double>*)Wd, nfft, bin0, m_local, (cx<float>*)G, g_pitch, (cx<float>*)H, h_pitch, gd);  \
        FL_CHECK_LAUNCH("sos_response_rc_fast");                                                                             \
        return FL_OK;                                                                                                        \
    }                                                                                                                        \
    if (Ni == NIW_) {                                                                                                        \
        if (gd.gain) {      /* the double kernel reads its sections: design them first */                                        \
            hipLaunchKernelGGL(geq_sections_kernel, dim3(cdiv_i(S * No * Nmid, 256)), dim3(256), 0, (hipStream_t)stream, gd.gain, \
                               gd.in_kind, S, No * Nmid, gd.k, gd.b_out, gd.a_out);                                              \
            FL_CHECK_LAUNCH("geq_sections");                                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL((sos_response_rc_kernel<NIW_, T>), grid, dim3(256), lds, (hipStream_t)stream, (const double*)b,   \
                           (const double*)a, S, No * Nmid, Nmid, (const T*)Wr, gamma, (const cx<double>*)Wd, nfft, bin0,     \
                           m_local, (cx<T>*)G, g_pitch, (cx<T>*)H, h_pitch);                                                 \
        FL_CHECK_LAUNCH("sos_response_rc");                                                                                  \
        return FL_OK;                                                                                                        \
    }
    FL_RC_FWD(2) FL_RC_FWD(4) FL_RC_FWD(8) FL_RC_FWD(16)
#undef FL_RC_FWD
    set_error("sos_response_rc: no kernel for %d input channels of the constant factor", Ni);
    return FL_ERR_UNSUPPORTED;
}
extern "C" {
int fl_sos_response_rc_c64(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma,
                           const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch,
                           int float_eval, void* stream) {
    return rc_impl<float>(b, a, S, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H, h_pitch, float_eval, stream,
                          GeqDesign{nullptr, 0, nullptr, nullptr, nullptr});
}
int fl_sos_response_rc_c128(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma,
                            const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch,
                            void* stream) {
    return rc_impl<double>(b, a, S, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H, h_pitch, 0, stream,
                           GeqDesign{nullptr, 0, nullptr, nullptr, nullptr});
}
}  // extern "C"
template <typename T>
static int geq_rc_impl(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int No, int Nmid, int Ni,
                       const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch,
                       void* H, long h_pitch, int float_eval, void* stream) {
    FL_REQUIRE(gain && consts, "geq_response_rc: null pointer");
    FL_REQUIRE(in_kind >= 0 && in_kind <= 4 && nb >= 4, "geq_response_rc: in_kind in [0, 4], at least four bands");
    if (m_local == 0) {      // nothing to evaluate: the sections are still an output
        hipLaunchKernelGGL(geq_sections_kernel, dim3(cdiv_i(nb * No * Nmid, 256)), dim3(256), 0, (hipStream_t)stream, gain, in_kind, nb,
                           No * Nmid, (const double*)consts, (double*)b, (double*)a);
        FL_CHECK_LAUNCH("geq_sections");
        return FL_OK;
    }
    return rc_impl<T>(b, a, nb, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H, h_pitch, float_eval, stream,
                      GeqDesign{gain, in_kind, (const double*)consts, (double*)b, (double*)a});
}
extern "C" {
int fl_geq_response_rc_c64(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int No, int Nmid, int Ni,
                           const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch,
                           void* H, long h_pitch, int float_eval, void* stream) {
    return geq_rc_impl<float>(gain, in_kind, nb, consts, b, a, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H,
                              h_pitch, float_eval, stream);
}
int fl_geq_response_rc_c128(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int No, int Nmid, int Ni,
                            const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch,
                            void* H, long h_pitch, void* stream) {
    return geq_rc_impl<double>(gain, in_kind, nb, consts, b, a, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, G, g_pitch, H,
                               h_pitch, 0, stream);
}
int fl_sos_response_apply_max_ni(int S) {      // cascades per row whose coefficient tables fit the default 64 KB of dynamic LDS
    const int SP = (S + 1) & ~1;
    return (int)(65536 / (12 * SP * sizeof(float)));
}
int fl_sos_response_apply_c64(const void* b, const void* a, int S, int No, int Ni, const void* X, long xs_b, long xs_n, int BX,
                              double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* Y, long ys_b,
                              long ys_m, void* stream) {
    FL_REQUIRE(b && a && X && Wd && G && Y, "sos_response_apply: null pointer");
    FL_REQUIRE(g_pitch >= m_local, "sos_response_apply: g_pitch must be >= m_local");
    FL_REQUIRE(S > 0 && No > 0 && No <= 65535 && Ni > 0 && Ni <= fl_sos_response_apply_max_ni(S) && (BX == 1 || BX == 2) && nfft > 0 &&
                   bin_range_ok(bin0, m_local, nfft) && m_local >= 0,
               "sos_response_apply: bad sizes (one or two columns; N_in up to fl_sos_response_apply_max_ni(S))");
    if (m_local == 0) return FL_OK;
    dim3 grid(cdiv_i(m_local, 256), No);
    const size_t lds = (size_t)Ni * 12 * ((S + 1) & ~1) * sizeof(float);
    if (BX == 1)
        hipLaunchKernelGGL((sos_response_apply_fast_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, (const double*)b,
                           (const double*)a, S, No * Ni, Ni, (const cx<float>*)X, xs_b, xs_n, gamma, (const cx<double>*)Wd, nfft, bin0,
                           m_local, (cx<float>*)G, g_pitch, (cx<float>*)Y, ys_b, ys_m);
    else
        hipLaunchKernelGGL((sos_response_apply_fast_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, (const double*)b,
                           (const double*)a, S, No * Ni, Ni, (const cx<float>*)X, xs_b, xs_n, gamma, (const cx<double>*)Wd, nfft, bin0,
                           m_local, (cx<float>*)G, g_pitch, (cx<float>*)Y, ys_b, ys_m);
    FL_CHECK_LAUNCH("sos_response_apply");
    return FL_OK;
}
int fl_sos_response_bwd_rc_c64(const void* gHfull, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                               int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                               int bin0, int m_local, void* part, void* partW, void* stream) {
    FL_REQUIRE(G && Wr && partW && No > 0 && Nmid > 0 && Ni > 0, "sos_response_bwd_rc: bad arguments");
    SosRC rc{Nmid, 0, (const float*)Wr, (float*)partW};
    return sos_bwd_impl<float>(gHfull, g_pitch, G, h_pitch, b, a, S, No * Nmid, gamma, Wd, nfft, bin0, m_local, part, stream,
                               Ni, rc);
}
int fl_sos_response_bwd_rc_c128(const void* gHfull, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                int bin0, int m_local, void* part, void* partW, void* stream) {
    FL_REQUIRE(G && Wr && partW && No > 0 && Nmid > 0 && Ni > 0, "sos_response_bwd_rc: bad arguments");
    SosRCd rcd{Nmid, 0, (const double*)Wr, (double*)partW};
    return sos_bwd_impl<double>(gHfull, g_pitch, G, h_pitch, b, a, S, No * Nmid, gamma, Wd, nfft, bin0, m_local, part, stream,
                                Ni, SosRC{0, 0, nullptr, nullptr}, rcd);
}
int fl_sos_response_bwd_outer_c64(const void* gY, long gy_sb, long gy_sn, const void* X, long x_sb, long x_sn, int B, int No, int Ni,
                                  const void* H, long h_pitch, const void* b, const void* a, int S, double gamma, const void* Wd,
                                  int nfft, int bin0, int m_local, void* part, void* stream) {
    FL_REQUIRE(gY && X && H && B > 0 && No > 0 && Ni > 0, "sos_response_bwd_outer: bad arguments");
    SosRC rc{0, 0, nullptr, nullptr};
    rc.oG = (const cx<float>*)gY; rc.oX = (const cx<float>*)X;
    rc.o_gb = gy_sb; rc.o_gn = gy_sn; rc.o_xb = x_sb; rc.o_xn = x_sn; rc.oB = B; rc.oNi = Ni;
    return sos_bwd_impl<float>(nullptr, m_local, H, h_pitch, b, a, S, No * Ni, gamma, Wd, nfft, bin0, m_local, part, stream, 0, rc);
}
int fl_sos_response_bwd_c128(const void* gH, long g_pitch, const void* H, long h_pitch, const void* b, const void* a, int S,
                             int C, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* part, void* stream) {
    return sos_bwd_impl<double>(gH, g_pitch, H, h_pitch, b, a, S, C, gamma, Wd, nfft, bin0, m_local, part, stream);
}
}
