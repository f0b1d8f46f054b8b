// Per-frequency-bin complex MIMO products for gfx950 (MI355X).
//
//   Y[b,m,k,f] = sum_n H[f,m,n] X[b,n,k,f]            (flamo "fmn,bfn...->bfm...", dsp.py:922-924)
//
// All tensors are bin-planar (the bin axis f is contiguous), so a wavefront of 64 lanes owns
// 64 adjacent bins and every global access is a fully coalesced 512 B (c64) / 1 KiB (c128)
// segment.  The op is HBM-bound (N/2 flop per byte at N channels, far below the CDNA4 ridge),
// so the kernels are plain VALU FMA streams: no LDS, no MFMA -- the (No x Ni) contraction per
// bin is far too small for a 16x16/32x32 MFMA tile and reshaping bins into a GEMM would only
// add traffic.  Each lane keeps an (MT x BT) accumulator tile: MT output channels x BT
// batch/trailing columns, so one H element loaded is reused BT times and one X element MT
// times from registers.
#include "common.h"

namespace fl {

template <typename T, int MT, int BT, int NU>
__global__ void __launch_bounds__(256) mimo_full_kernel(
    const cx<T>* __restrict__ H, long hs_f, long hs_m, long hs_n, int conj_h,
    const cx<T>* __restrict__ X, long xs_b, long xs_n, long xs_k,
    cx<T>* __restrict__ Y, long ys_b, long ys_m, long ys_k,
    int B, int M, int No, int Ni, int K, int nct, int nmt) {
    // XCD-aware block order (block q runs on XCD q % 8, each XCD has its own L2): the blocks that
    // share one bin tile -- all batch-column tiles and channel tiles, which re-read the same H[f]
    // -- get consecutive slots on the SAME XCD, so H comes from HBM once and from that L2 after
    // (measured: 369 MB -> 232 MB of fabric traffic per launch at config 2, algorithmic 221 MB).
    const int inner = nct * nmt;
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int ft = (r / inner) * 8 + xcd, rem = r % inner;
    const int f = ft * 256 + threadIdx.x;
    if (f >= M) return;
    const int col0 = (rem % nct) * BT;
    const int m0 = (rem / nct) * MT;
    const int ncols = B * K;
    long xoff[BT], yoff[BT];
    bool cv[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        const int col = col0 + c;
        cv[c] = col < ncols;
        const int b = cv[c] ? col / K : 0, k = cv[c] ? col - b * K : 0;
        xoff[c] = (long)b * xs_b + (long)k * xs_k + f;
        yoff[c] = (long)b * ys_b + (long)k * ys_k + f;
    }
    cx<T> acc[BT][MT];
#pragma unroll
    for (int c = 0; c < BT; ++c)
#pragma unroll
        for (int mm = 0; mm < MT; ++mm) acc[c][mm] = cx<T>(0, 0);
    const cx<T>* Hf = H + (long)f * hs_f;
    // NU input channels per trip: (MT + BT) * NU loads are issued before their FMAs
    for (int n0 = 0; n0 < Ni; n0 += NU) {
        cx<T> h[NU][MT], x[NU][BT];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = n0 + u;
            const bool nv = (NU == 1) || n < Ni;
#pragma unroll
            for (int mm = 0; mm < MT; ++mm) {
                const int m = m0 + mm;
                h[u][mm] = (nv && m < No) ? Hf[(long)m * hs_m + (long)n * hs_n] : cx<T>(0, 0);
                if (conj_h) h[u][mm].y = -h[u][mm].y;
            }
#pragma unroll
            for (int c = 0; c < BT; ++c) x[u][c] = (nv && cv[c]) ? X[xoff[c] + (long)n * xs_n] : cx<T>(0, 0);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int c = 0; c < BT; ++c)
#pragma unroll
                for (int mm = 0; mm < MT; ++mm) fma_cx(acc[c][mm], h[u][mm], x[u][c]);
    }
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        if (!cv[c]) continue;
#pragma unroll
        for (int mm = 0; mm < MT; ++mm) {
            const int m = m0 + mm;
            if (m < No) Y[yoff[c] + (long)m * ys_m] = acc[c][mm];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) mimo_diag_kernel(
    const cx<T>* __restrict__ h, long hs_f, long hs_n, int conj_h,
    const cx<T>* __restrict__ X, long xs_b, long xs_n, long xs_k,
    cx<T>* __restrict__ Y, long ys_b, long ys_n, long ys_k, int B, int M, int N, int K) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= M) return;
    const int n = blockIdx.y;
    cx<T> hv = h[(long)f * hs_f + (long)n * hs_n];
    if (conj_h) hv.y = -hv.y;
    const int ncols = B * K;
    for (int col = blockIdx.z; col < ncols; col += gridDim.z) {
        const int b = col / K, k = col - b * K;
        Y[(long)b * ys_b + (long)n * ys_n + (long)k * ys_k + f] =
            hv * X[(long)b * xs_b + (long)n * xs_n + (long)k * xs_k + f];
    }
}

// dH[m,n,f] = scale * sum_{b,k} G[b,m,k,f] conj(X[b,n,k,f])
template <typename T, int MT, int NT>
__global__ void __launch_bounds__(256) mimo_gradh_kernel(
    const cx<T>* __restrict__ G, long gs_b, long gs_m, long gs_k,
    const cx<T>* __restrict__ X, long xs_b, long xs_n, long xs_k,
    cx<T>* __restrict__ dH, long dh_pitch, T scale, int B, int M, int No, int Ni, int K) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= M) return;
    const int m0 = blockIdx.y * MT, n0 = blockIdx.z * NT;
    cx<T> acc[MT][NT];
#pragma unroll
    for (int mm = 0; mm < MT; ++mm)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) acc[mm][nn] = cx<T>(0, 0);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) {
            const cx<T>* g = G + (long)b * gs_b + (long)k * gs_k + f;
            const cx<T>* x = X + (long)b * xs_b + (long)k * xs_k + f;
            cx<T> gv[MT], xv[NT];
#pragma unroll
            for (int mm = 0; mm < MT; ++mm) gv[mm] = (m0 + mm < No) ? g[(long)(m0 + mm) * gs_m] : cx<T>(0, 0);
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) xv[nn] = (n0 + nn < Ni) ? x[(long)(n0 + nn) * xs_n] : cx<T>(0, 0);
#pragma unroll
            for (int mm = 0; mm < MT; ++mm)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) fma_cxc(acc[mm][nn], gv[mm], xv[nn]);
        }
#pragma unroll
    for (int mm = 0; mm < MT; ++mm)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const int m = m0 + mm, n = n0 + nn;
            if (m < No && n < Ni) dH[((long)m * Ni + n) * dh_pitch + f] = cx<T>(scale * acc[mm][nn].x, scale * acc[mm][nn].y);
        }
}

template <typename T>
__global__ void __launch_bounds__(256) mimo_gradh_diag_kernel(
    const cx<T>* __restrict__ G, long gs_b, long gs_n, long gs_k,
    const cx<T>* __restrict__ X, long xs_b, long xs_n, long xs_k,
    cx<T>* __restrict__ dh, long dh_pitch, int B, int M, int N, int K) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= M) return;
    const int n = blockIdx.y;
    cx<T> acc(0, 0);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k)
            fma_cxc(acc, G[(long)b * gs_b + (long)n * gs_n + (long)k * gs_k + f],
                    X[(long)b * xs_b + (long)n * xs_n + (long)k * xs_k + f]);
    dh[(long)n * dh_pitch + f] = acc;
}

// ---------------------------------------------------------------- host dispatch
static int g_mimo_variant = 0;   // tuning hook: mt*100 + bt*10 + nu (0 = default choice)

template <typename T, int MT, int BT, int NU>
static void launch_full_one(dim3 grid, int nct, int nmt, hipStream_t st, const cx<T>* H, long hs_f, long hs_m, long hs_n, int conj_h,
                            const cx<T>* X, long xs_b, long xs_n, long xs_k, cx<T>* Y, long ys_b, long ys_m, long ys_k,
                            int B, int M, int No, int Ni, int K) {
    hipLaunchKernelGGL((mimo_full_kernel<T, MT, BT, NU>), grid, dim3(256), 0, st, H, hs_f, hs_m, hs_n, conj_h, X, xs_b, xs_n,
                       xs_k, Y, ys_b, ys_m, ys_k, B, M, No, Ni, K, nct, nmt);
}

template <typename T>
static int mimo_impl(const void* H, long hs_f, long hs_m, long hs_n, int conj_h, const void* X, long xs_b, long xs_n,
                     long xs_k, void* Y, long ys_b, long ys_m, long ys_k, int B, int M, int No, int Ni, int K,
                     void* stream) {
    FL_REQUIRE(H && X && Y, "mimo: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && No > 0 && Ni > 0 && K > 0, "mimo: bad sizes");
    if (B == 0 || M == 0) return FL_OK;
    const int ncols = B * K;
    int bt = ncols >= 4 ? 4 : (ncols >= 2 ? 2 : 1);
    int mt = No >= 8 ? 8 : (No >= 4 ? 4 : (No >= 2 ? 2 : 1));
    int nu = 1;
    if (g_mimo_variant > 0) {
        mt = g_mimo_variant / 100;
        bt = (g_mimo_variant / 10) % 10;
        nu = g_mimo_variant % 10;
    }
    const int nct = cdiv_i(ncols, bt), nmt = cdiv_i(No, mt);
    const size_t nblk = (size_t)cdiv_i(cdiv_i(M, 256), 8) * 8 * nct * nmt;
    FL_REQUIRE(nblk < (1ull << 31), "mimo: grid too large");
    dim3 grid((unsigned)nblk);
    hipStream_t st = (hipStream_t)stream;
    const cx<T>* h = (const cx<T>*)H;
    const cx<T>* x = (const cx<T>*)X;
    cx<T>* y = (cx<T>*)Y;
#define FL_MIMO_CASE(MT_, BT_, NU_)                                                                                   \
    if (mt == MT_ && bt == BT_ && nu == NU_) {                                                                        \
        launch_full_one<T, MT_, BT_, NU_>(grid, nct, nmt, st, h, hs_f, hs_m, hs_n, conj_h, x, xs_b, xs_n, xs_k, y, ys_b, \
                                          ys_m, ys_k, B, M, No, Ni, K);                                               \
        FL_CHECK_LAUNCH("mimo_full");                                                                                 \
        return FL_OK;                                                                                                 \
    }
    FL_MIMO_CASE(8, 4, 1) FL_MIMO_CASE(8, 2, 1) FL_MIMO_CASE(8, 1, 1)
    FL_MIMO_CASE(4, 4, 1) FL_MIMO_CASE(4, 2, 1) FL_MIMO_CASE(4, 1, 1)
    FL_MIMO_CASE(2, 4, 1) FL_MIMO_CASE(2, 2, 1) FL_MIMO_CASE(2, 1, 1)
    FL_MIMO_CASE(1, 4, 1) FL_MIMO_CASE(1, 2, 1) FL_MIMO_CASE(1, 1, 1)
    // tuning variants
    FL_MIMO_CASE(8, 4, 2) FL_MIMO_CASE(8, 2, 2) FL_MIMO_CASE(8, 2, 4) FL_MIMO_CASE(4, 4, 2) FL_MIMO_CASE(4, 4, 4)
    FL_MIMO_CASE(4, 8, 1) FL_MIMO_CASE(4, 8, 2) FL_MIMO_CASE(8, 8, 1) FL_MIMO_CASE(4, 2, 4) FL_MIMO_CASE(4, 2, 2)
#undef FL_MIMO_CASE
    set_error("mimo: no kernel variant mt=%d bt=%d nu=%d", mt, bt, nu);
    return FL_ERR_UNSUPPORTED;
}

template <typename T>
static int mimo_diag_impl(const void* h, long hs_f, long hs_n, int conj_h, const void* X, long xs_b, long xs_n, long xs_k,
                          void* Y, long ys_b, long ys_n, long ys_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(h && X && Y, "mimo_diag: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0 && N <= 65535, "mimo_diag: bad sizes");
    if (B == 0 || M == 0) return FL_OK;
    int gz = B * K;
    if (gz > 1024) gz = 1024;
    dim3 grid(cdiv_i(M, 256), N, gz);
    hipLaunchKernelGGL((mimo_diag_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)h, hs_f, hs_n, conj_h,
                       (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)Y, ys_b, ys_n, ys_k, B, M, N, K);
    FL_CHECK_LAUNCH("mimo_diag");
    return FL_OK;
}

template <typename T>
static int gradh_impl(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                      void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream) {
    FL_REQUIRE(G && X && dH, "mimo_gradh: null pointer");
    FL_REQUIRE(dh_pitch >= M, "mimo_gradh: dh_pitch must be >= M");
    FL_REQUIRE(B >= 0 && M >= 0 && No > 0 && Ni > 0 && K > 0, "mimo_gradh: bad sizes");
    if (M == 0) return FL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (No >= 4 && Ni >= 4) {
        dim3 grid(cdiv_i(M, 256), cdiv_i(No, 4), cdiv_i(Ni, 4));
        FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradh: too many channels");
        hipLaunchKernelGGL((mimo_gradh_kernel<T, 4, 4>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                           (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K);
    } else {
        dim3 grid(cdiv_i(M, 256), No, Ni);
        FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradh: too many channels");
        hipLaunchKernelGGL((mimo_gradh_kernel<T, 1, 1>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                           (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K);
    }
    FL_CHECK_LAUNCH("mimo_gradh");
    return FL_OK;
}

template <typename T>
static int gradh_diag_impl(const void* G, long gs_b, long gs_n, long gs_k, const void* X, long xs_b, long xs_n,
                           long xs_k, void* dh, long dh_pitch, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(G && X && dh, "mimo_gradh_diag: null pointer");
    FL_REQUIRE(dh_pitch >= M, "mimo_gradh_diag: dh_pitch must be >= M");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0 && N <= 65535, "mimo_gradh_diag: bad sizes");
    if (M == 0) return FL_OK;
    dim3 grid(cdiv_i(M, 256), N);
    hipLaunchKernelGGL((mimo_gradh_diag_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)G, gs_b, gs_n,
                       gs_k, (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dh, dh_pitch, B, M, N, K);
    FL_CHECK_LAUNCH("mimo_gradh_diag");
    return FL_OK;
}


// dW[m,n] = sum_{b,k,f} G[b,m,k,f] conj(X[b,n,k,f]) -- the gradient of a frequency-INDEPENDENT
// matrix (Gain/Matrix, the FDN mixing matrix) reduced over bins inside the kernel: each block
// walks bins with a grid stride, keeps a 4x4 (8x8) tile of sums per lane, reduces across the block and
// writes one partial tile; the host adds the <= 256 partials.  No (M, No, Ni) tensor is built.
template <typename T, int TM, int TN>
__global__ void __launch_bounds__(256) mimo_gradw_kernel(
    const cx<T>* __restrict__ G, long gs_b, long gs_m, long gs_k,
    const cx<T>* __restrict__ X, long xs_b, long xs_n, long xs_k,
    cx<T>* __restrict__ part, int B, int M, int No, int Ni, int K) {
    const int m0 = blockIdx.y * TM, n0 = blockIdx.z * TN;
    // accumulators as 2*TM*TN scalars: [2*(mm*TN + nn)] real, [+1] imaginary
    T acc[2 * TM * TN];
#pragma unroll
    for (int v = 0; v < 2 * TM * TN; ++v) acc[v] = (T)0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < M; f += gridDim.x * 256)
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) {
                const cx<T>* g = G + (long)b * gs_b + (long)k * gs_k + f;
                const cx<T>* x = X + (long)b * xs_b + (long)k * xs_k + f;
                cx<T> gv[TM], xv[TN];
#pragma unroll
                for (int mm = 0; mm < TM; ++mm) gv[mm] = (m0 + mm < No) ? g[(long)(m0 + mm) * gs_m] : cx<T>(0, 0);
#pragma unroll
                for (int nn = 0; nn < TN; ++nn) xv[nn] = (n0 + nn < Ni) ? x[(long)(n0 + nn) * xs_n] : cx<T>(0, 0);
#pragma unroll
                for (int mm = 0; mm < TM; ++mm)
#pragma unroll
                    for (int nn = 0; nn < TN; ++nn) {   // += g * conj(x)
                        acc[2 * (mm * TN + nn)] += gv[mm].x * xv[nn].x + gv[mm].y * xv[nn].y;
                        acc[2 * (mm * TN + nn) + 1] += gv[mm].y * xv[nn].x - gv[mm].x * xv[nn].y;
                    }
            }
    __shared__ T red[4][2 * TM * TN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = 0, cnt = 0, dup = 0;
    wave_reduce_scatter<T, 2 * TM * TN, 32>(acc, lane, off, cnt, dup);
    if ((lane & dup) == 0) {
#pragma unroll
        for (int v = 0; v < 2 * TM * TN; ++v)
            if (v < cnt) red[wave][off + v] = acc[v];
    }
    __syncthreads();
    if (threadIdx.x < TM * TN) {
        const int mm = threadIdx.x / TN, nn = threadIdx.x % TN;
        const int m = m0 + mm, n = n0 + nn;
        if (m < No && n < Ni) {
            T vr = 0, vi = 0;
            for (int w = 0; w < 4; ++w) {
                vr += red[w][threadIdx.x * 2];
                vi += red[w][threadIdx.x * 2 + 1];
            }
            part[((size_t)blockIdx.x * No + m) * Ni + n] = cx<T>(vr, vi);
        }
    }
}

// dW[m,n] = sum over the nblk partial tiles: one wavefront per entry, lanes stride over the blocks and
// combine with a fixed butterfly (deterministic; a serial loop over 256 partials costs 30 us of
// dependent L2 latency)
template <typename T>
__global__ void __launch_bounds__(256) mimo_gradw_final_kernel(const cx<T>* __restrict__ part, int nblk, int count,
                                                              cx<T>* __restrict__ dW) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= count) return;
    T vr = 0, vi = 0;
    for (int b = lane; b < nblk; b += 64) {
        const cx<T> v = part[(size_t)b * count + e];
        vr += v.x;
        vi += v.y;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        vr += __shfl_xor(vr, off, 64);
        vi += __shfl_xor(vi, off, 64);
    }
    if (lane == 0) dW[e] = cx<T>(vr, vi);
}

static int g_gradw_cap = 0;
static int gradw_blocks(int M) {
    int nb = cdiv_i(M, 256);
    const int cap = g_gradw_cap > 0 ? g_gradw_cap : 256;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

template <typename T>
static int gradw_impl(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                      void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream) {
    FL_REQUIRE(G && X && part && dW, "mimo_gradw: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && No > 0 && Ni > 0 && K > 0, "mimo_gradw: bad sizes");
    // 8x8 tiles where the matrix allows: every element of G and X is then read No/8 (Ni/8) times
    // instead of No/4 -- at N = 32 with matrix-valued signals that is 12 GB instead of 25 GB per launch
    const bool big = No >= 8 && Ni >= 8 && sizeof(T) == 4;
    const int tm = big ? 8 : 4;
    dim3 grid(gradw_blocks(M), cdiv_i(No, tm), cdiv_i(Ni, tm));
    FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradw: too many channels");
    if (big)
        hipLaunchKernelGGL((mimo_gradw_kernel<T, 8, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)G, gs_b, gs_m,
                           gs_k, (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)part, B, M, No, Ni, K);
    else
        hipLaunchKernelGGL((mimo_gradw_kernel<T, 4, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)G, gs_b, gs_m,
                           gs_k, (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)part, B, M, No, Ni, K);
    FL_CHECK_LAUNCH("mimo_gradw");
    hipLaunchKernelGGL((mimo_gradw_final_kernel<T>), dim3(cdiv_i((long)No * Ni, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const cx<T>*)part, (int)grid.x, No * Ni, (cx<T>*)dW);
    FL_CHECK_LAUNCH("mimo_gradw_final");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {

int fl_mimo_gradw_blocks(int M) { return gradw_blocks(M); }
int fl_debug_set_mimo_variant(int variant, int gradw_cap) {
    g_mimo_variant = variant;
    g_gradw_cap = gradw_cap;
    return FL_OK;
}
int fl_mimo_gradw_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                      void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream) {
    return gradw_impl<float>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, part, dW, B, M, No, Ni, K, stream);
}
int fl_mimo_gradw_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                       void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream) {
    return gradw_impl<double>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, part, dW, B, M, No, Ni, K, stream);
}

int fl_mimo_c64(const void* H, long hs_f, long hs_m, long hs_n, int conj_h, const void* X, long xs_b, long xs_n, long xs_k,
                void* Y, long ys_b, long ys_m, long ys_k, int B, int M, int No, int Ni, int K, void* stream) {
    return mimo_impl<float>(H, hs_f, hs_m, hs_n, conj_h, X, xs_b, xs_n, xs_k, Y, ys_b, ys_m, ys_k, B, M, No, Ni, K, stream);
}
int fl_mimo_c128(const void* H, long hs_f, long hs_m, long hs_n, int conj_h, const void* X, long xs_b, long xs_n, long xs_k,
                 void* Y, long ys_b, long ys_m, long ys_k, int B, int M, int No, int Ni, int K, void* stream) {
    return mimo_impl<double>(H, hs_f, hs_m, hs_n, conj_h, X, xs_b, xs_n, xs_k, Y, ys_b, ys_m, ys_k, B, M, No, Ni, K, stream);
}
int fl_mimo_diag_c64(const void* h, long hs_f, long hs_n, int conj_h, const void* X, long xs_b, long xs_n, long xs_k,
                     void* Y, long ys_b, long ys_n, long ys_k, int B, int M, int N, int K, void* stream) {
    return mimo_diag_impl<float>(h, hs_f, hs_n, conj_h, X, xs_b, xs_n, xs_k, Y, ys_b, ys_n, ys_k, B, M, N, K, stream);
}
int fl_mimo_diag_c128(const void* h, long hs_f, long hs_n, int conj_h, const void* X, long xs_b, long xs_n, long xs_k,
                      void* Y, long ys_b, long ys_n, long ys_k, int B, int M, int N, int K, void* stream) {
    return mimo_diag_impl<double>(h, hs_f, hs_n, conj_h, X, xs_b, xs_n, xs_k, Y, ys_b, ys_n, ys_k, B, M, N, K, stream);
}
int fl_mimo_gradh_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                      void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream) {
    return gradh_impl<float>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, dH, dh_pitch, scale, B, M, No, Ni, K, stream);
}
int fl_mimo_gradh_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                       void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream) {
    return gradh_impl<double>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, dH, dh_pitch, scale, B, M, No, Ni, K, stream);
}
int fl_mimo_gradh_diag_c64(const void* G, long gs_b, long gs_n, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                           void* dh, long dh_pitch, int B, int M, int N, int K, void* stream) {
    return gradh_diag_impl<float>(G, gs_b, gs_n, gs_k, X, xs_b, xs_n, xs_k, dh, dh_pitch, B, M, N, K, stream);
}
int fl_mimo_gradh_diag_c128(const void* G, long gs_b, long gs_n, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                            void* dh, long dh_pitch, int B, int M, int N, int K, void* stream) {
    return gradh_diag_impl<double>(G, gs_b, gs_n, gs_k, X, xs_b, xs_n, xs_k, dh, dh_pitch, B, M, N, K, stream);
}

}  // extern "C"
