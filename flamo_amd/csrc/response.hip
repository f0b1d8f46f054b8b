// Frequency-response generators evaluated directly per bin (gfx950 / MI355X).
//
//  * integer delay lines: H[c,k] = amp[c] * W_n^((k*m_c) mod n)  -- exact integer phase index
//    (Delay/parallelDelay with isint=True, flamo/processor/dsp.py:3356-3365, 3512-3521);
//  * second-order-section cascades: H[c,k] = prod_s B_s(k) / prod_s A_s(k) with
//    B_s(k) = b0 + b1 g w + b2 g^2 w^2, w = W_n^k -- what the reference obtains from
//    rfft(3 taps, nfft) per section followed by prod/prod (dsp.py:1520-1526, 2587-2593),
//    without ever materialising the (M, sections, N_out, N_in) tensors; plus its backward.
#include "common.h"

namespace fl {

template <typename T>
__global__ void __launch_bounds__(256) delay_response_kernel(const int32_t* __restrict__ m, const T* __restrict__ amp,
                                                            const cx<T>* __restrict__ W, int nfft, int bin0,
                                                            int m_local, cx<T>* __restrict__ H) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const int c = blockIdx.y;
    const long long k = bin0 + f;
    long long idx = (k * (long long)m[c]) % nfft;
    if (idx < 0) idx += nfft;
    const cx<T> w = W[idx];
    const T a = amp[c];
    H[(size_t)c * m_local + f] = cx<T>(a * w.x, a * w.y);
}

// The section polynomials are evaluated in DOUBLE precision whatever the storage type T: at low
// frequencies b0 + b1 w + b2 w^2 cancels to ~1e-5 of its terms (shelving sections at 44 Hz), so
// float32 evaluation -- what the reference's float32 mode does -- loses 3 digits there.  The
// point w = exp(-2 pi i k / n) comes from the float64 master twiddle table.
struct SosEval {
    cx<double> z1, z2;  // g*w, g^2*w^2
    __device__ inline cx<double> poly(const double* co, int S, int C, int s, int c) const {
        const double c0 = co[((size_t)0 * S + s) * C + c];
        const double c1 = co[((size_t)1 * S + s) * C + c];
        const double c2 = co[((size_t)2 * S + s) * C + c];
        return cx<double>(c0 + c1 * z1.x + c2 * z2.x, c1 * z1.y + c2 * z2.y);
    }
};

__device__ inline SosEval sos_point(const cx<double>* __restrict__ Wd, int nfft, int k, double g) {
    SosEval e;
    const cx<double> w1 = Wd[k % nfft];
    const cx<double> w2 = Wd[(2 * (long long)k) % nfft];
    e.z1 = cx<double>(g * w1.x, g * w1.y);
    e.z2 = cx<double>(g * g * w2.x, g * g * w2.y);
    return e;
}

template <typename T> __device__ inline T eps_of();
template <> __device__ inline float eps_of<float>() { return 1.1920928955078125e-07f; }
template <> __device__ inline double eps_of<double>() { return 2.220446049250313e-16; }

template <typename T>
__global__ void __launch_bounds__(256) sos_response_kernel(const double* __restrict__ b, const double* __restrict__ a, int S, int C,
                                                          double g, const cx<double>* __restrict__ Wd, int nfft,
                                                          int bin0, int m_local, cx<T>* __restrict__ H) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= m_local) return;
    const int c = blockIdx.y;
    const SosEval e = sos_point(Wd, nfft, bin0 + f, g);
    cx<double> Bp(1, 0), Ap(1, 0);
    for (int s = 0; s < S; ++s) {
        Bp = Bp * e.poly(b, S, C, s, c);
        Ap = Ap * e.poly(a, S, C, s, c);
    }
    cx<double> h = (Ap.x != 0 || Ap.y != 0) ? cdiv(Bp, Ap) : cx<double>((double)eps_of<T>(), 0);
    H[(size_t)c * m_local + f] = cx<T>((T)h.x, (T)h.y);
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
template <typename T, int SCH>
__global__ void __launch_bounds__(256) sos_response_bwd_kernel(const cx<T>* __restrict__ gH, const double* __restrict__ b,
                                                              const double* __restrict__ a, int S, int C, double g,
                                                              const cx<double>* __restrict__ Wd, int nfft, int bin0,
                                                              int m_local, double* __restrict__ part) {
    const int c = blockIdx.y;
    const int s0 = blockIdx.z * SCH;
    double acc[2][3][SCH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < SCH; ++q) acc[i][p][q] = 0.0;

    for (int f = blockIdx.x * 256 + threadIdx.x; f < m_local; f += gridDim.x * 256) {
        const SosEval e = sos_point(Wd, nfft, bin0 + f, g);
        cx<double> Bp(1, 0), Ap(1, 0);
        for (int s = 0; s < S; ++s) {
            Bp = Bp * e.poly(b, S, C, s, c);
            Ap = Ap * e.poly(a, S, C, s, c);
        }
        if (Ap.x == 0 && Ap.y == 0) continue;  // guarded bins are the constant eps: zero gradient
        const cx<double> h = cdiv_fast(Bp, Ap);
        const cx<T> gin = gH[(size_t)c * m_local + f];
        const cx<double> gc((double)gin.x, -(double)gin.y);
        const cx<double> gh = gc * h;            // conj(gH) * H
#pragma unroll
        for (int q = 0; q < SCH; ++q) {
            const int s = s0 + q;
            if (s < S) {
                const cx<double> Bs = e.poly(b, S, C, s, c), As = e.poly(a, S, C, s, c);
                cx<double> tb;
                if (Bs.x != 0 || Bs.y != 0) {
                    tb = cdiv_fast(gh, Bs);
                } else {  // numerator section vanishes at this bin: product of the others
                    cx<double> o(1, 0);
                    for (int t = 0; t < S; ++t)
                        if (t != s) o = o * e.poly(b, S, C, t, c);
                    tb = gc * cdiv_fast(o, Ap);
                }
                const cx<double> ta = cdiv_fast(gh, As);
                acc[0][0][q] += tb.x;
                acc[0][1][q] += tb.x * e.z1.x - tb.y * e.z1.y;     // Re(tb * z_p)
                acc[0][2][q] += tb.x * e.z2.x - tb.y * e.z2.y;
                acc[1][0][q] -= ta.x;
                acc[1][1][q] -= ta.x * e.z1.x - ta.y * e.z1.y;
                acc[1][2][q] -= ta.x * e.z2.x - ta.y * e.z2.y;
            }
        }
    }
    // block reduction: wavefront shuffles, then the 4 wave partials through LDS
    __shared__ double red[4][6 * SCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < SCH; ++q) {
                double v = acc[i][p][q];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0) red[wave][(i * 3 + p) * SCH + q] = v;
            }
    __syncthreads();
    if (threadIdx.x < 6 * SCH) {
        const int i = threadIdx.x / (3 * SCH), p = (threadIdx.x / SCH) % 3, q = threadIdx.x % SCH;
        const int s = s0 + q;
        if (s < S) {
            const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            part[((((size_t)blockIdx.x * 2 + i) * 3 + p) * S + s) * C + c] = v;
        }
    }
}

static int sos_blocks(int m_local) {
    int nb = cdiv_i(m_local, 256);
    if (nb > 64) nb = 64;
    if (nb < 1) nb = 1;
    return nb;
}

template <typename T>
static int delay_impl(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local, void* H,
                      void* stream) {
    FL_REQUIRE(m && amp && W && H, "delay_response: null pointer");
    FL_REQUIRE(C > 0 && C <= 65535 && nfft > 0 && bin0 >= 0 && m_local >= 0, "delay_response: bad sizes");
    if (m_local == 0) return FL_OK;
    dim3 grid(cdiv_i(m_local, 256), C);
    hipLaunchKernelGGL((delay_response_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, m, (const T*)amp,
                       (const cx<T>*)W, nfft, bin0, m_local, (cx<T>*)H);
    FL_CHECK_LAUNCH("delay_response");
    return FL_OK;
}

template <typename T>
static int sos_impl(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                    int m_local, void* H, void* stream) {
    FL_REQUIRE(b && a && H && Wd, "sos_response: null pointer");
    FL_REQUIRE(S > 0 && C > 0 && C <= 65535 && nfft > 0 && bin0 >= 0 && m_local >= 0, "sos_response: bad sizes");
    if (m_local == 0) return FL_OK;
    dim3 grid(cdiv_i(m_local, 256), C);
    hipLaunchKernelGGL((sos_response_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const double*)b, (const double*)a, S, C,
                       gamma, (const cx<double>*)Wd, nfft, bin0, m_local, (cx<T>*)H);
    FL_CHECK_LAUNCH("sos_response");
    return FL_OK;
}

template <typename T>
static int sos_bwd_impl(const void* gH, const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                        int nfft, int bin0, int m_local, void* part, void* stream) {
    FL_REQUIRE(gH && b && a && part && Wd, "sos_response_bwd: null pointer");
    FL_REQUIRE(S > 0 && C > 0 && C <= 65535 && nfft > 0 && bin0 >= 0 && m_local > 0, "sos_response_bwd: bad sizes");
    if (S > 4) {
        dim3 grid(sos_blocks(m_local), C, cdiv_i(S, 12));
        hipLaunchKernelGGL((sos_response_bwd_kernel<T, 12>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)gH,
                           (const double*)b, (const double*)a, S, C, gamma, (const cx<double>*)Wd, nfft, bin0, m_local,
                           (double*)part);
    } else {
        dim3 grid(sos_blocks(m_local), C, 1);
        hipLaunchKernelGGL((sos_response_bwd_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)gH,
                           (const double*)b, (const double*)a, S, C, gamma, (const cx<double>*)Wd, nfft, bin0, m_local,
                           (double*)part);
    }
    FL_CHECK_LAUNCH("sos_response_bwd");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_delay_response_c64(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local,
                          void* H, void* stream) {
    return delay_impl<float>(m, amp, C, W, nfft, bin0, m_local, H, stream);
}
int fl_delay_response_c128(const int32_t* m, const void* amp, int C, const void* W, int nfft, int bin0, int m_local,
                           void* H, void* stream) {
    return delay_impl<double>(m, amp, C, W, nfft, bin0, m_local, H, stream);
}
int fl_sos_response_c64(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                        int m_local, void* H, void* stream) {
    return sos_impl<float>(b, a, S, C, gamma, Wd, nfft, bin0, m_local, H, stream);
}
int fl_sos_response_c128(const void* b, const void* a, int S, int C, double gamma, const void* Wd, int nfft, int bin0,
                        int m_local, void* H, void* stream) {
    return sos_impl<double>(b, a, S, C, gamma, Wd, nfft, bin0, m_local, H, stream);
}
int fl_sos_bwd_blocks(int m_local) { return sos_blocks(m_local); }
int fl_sos_response_bwd_c64(const void* gH, const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                            int nfft, int bin0, int m_local, void* part, void* stream) {
    return sos_bwd_impl<float>(gH, b, a, S, C, gamma, Wd, nfft, bin0, m_local, part, stream);
}
int fl_sos_response_bwd_c128(const void* gH, const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                            int nfft, int bin0, int m_local, void* part, void* stream) {
    return sos_bwd_impl<double>(gH, b, a, S, C, gamma, Wd, nfft, bin0, m_local, part, stream);
}
}
