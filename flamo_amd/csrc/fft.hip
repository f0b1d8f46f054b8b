// Batched real FFT / inverse real FFT for gfx950 (MI355X), mixed radix 2/3/4/5 (+7/11/13).
//
// A real transform of length n is computed as a complex FFT of the packed half-length
// sequence z[j] = x[2j] + i x[2j+1] (L = n/2) plus a split step.  The complex FFT is a
// "four-step" factorisation L = L1*L2 when L does not fit in LDS:
//
//   pass 1 (fft_cols): for a tile of CT adjacent columns c, Stockham FFT of length L1 over
//                      z[t1*L2 + c] in LDS, multiply by W_L^(c*k1), store A[k1*L2 + c].
//   pass 2 (fft_rows): for a tile of rows k1 (contiguous L2 elements), Stockham FFT of length
//                      L2 in LDS, then the epilogue writes natural order k = k1 + L1*k2.
//
// Forward (rfft): pass 2's epilogue is the real-FFT split step, which couples bins k and
// L-k; a workgroup therefore owns rows {r, L1-r} together and emits both bins.  Inverse
// (irfft): the Hermitian pre-step is fused into pass 1's loader (reads X[j] and X[L-j]) and
// the envelope/scale/real-pair store into pass 2's epilogue.  All global traffic is in
// >= 64-byte contiguous segments; one read + one write of the data per pass.
//
// The anti-alias envelope gamma^-t of FFTAntiAlias / iFFTAntiAlias (dsp.py:158-162,201-205)
// is applied in the loader / epilogue (no separate pass).
#define FL_PACKED_COMPLEX 1
#include "common.h"
#include "regfft.h"

namespace fl {

enum { LOAD_PACK = 0, LOAD_IRFFT_PRE = 1, LOAD_SCRATCH = 2, LOAD_PACK_FAST = 3 };
enum { EPI_RFFT_POST = 0, EPI_IRFFT_STORE = 1 };

struct Rad {
    int n;
    int r[24];
};

template <typename T>
struct FftArgs {
    const T* xr;        // LOAD_PACK source (real): signal-planar, or channel-innermost when ci_n > 0
    long xr_stride;
    int t_in;
    int ci_n;           // > 0: x is (B, ci_T, ci_n) contiguous, signal sig = b*ci_n + n, sample t at ((b*ci_T + t)*ci_n + n)
    int ci_T;
    int pack_fast;      // LOAD_PACK: t_in >= n, rows 2-element aligned: unguarded pair loads
    const cx<T>* Xc;    // LOAD_IRFFT_PRE source (half spectrum, L+1 bins per signal)
    cx<T>* Xout;        // EPI_RFFT_POST destination
    T* yr;              // EPI_IRFFT_STORE destination
    long yr_stride;
    int t_out;
    long xc_stride;     // elements between consecutive signals of Xc / Xout (>= L+1)
    cx<T>* scratch;
    const cx<T>* W;     // W_n^j, j in [0,n)
    int n, L, L1, L2, L1P, L2P, CT, RT, ntiles;
    int per;            // LDS slots per primary row in pass 2 (2 when mirror rows are held)
    int nsig;
    T scale;
    double env_log2;
    int interior;       // rfft: double interior bins; irfft: halve interior bins
    Rad rad1, rad2;
};


// One Stockham (decimation-in-frequency, autosort) stage of radix R on nseq sequences of
// length len stored at stride lenP in LDS:  y[q + s(Rp + k)] = w_p^k sum_j x[q + s(p + m j)] w_R^{jk}
template <typename T, int R, bool INV>
__device__ void stockham_stage(const cx<T>* __restrict__ a, cx<T>* __restrict__ b, const cx<T>* __restrict__ tw,
                               int nseq, int len, int lenP, int s) {
    const int nb = len / R;
    const int m = nb / s;
    for (int idx = threadIdx.x; idx < nseq * nb; idx += blockDim.x) {
        const int seq = idx / nb, w = idx - seq * nb;
        const int p = w / s, q = w - p * s;
        const cx<T>* x = a + seq * lenP;
        cx<T>* y = b + seq * lenP;
        cx<T> v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = x[q + s * (p + m * j)];
        Bfly<T, R, INV>::run(v, tw, nb);
        y[q + s * (R * p)] = v[0];
#pragma unroll
        for (int k = 1; k < R; ++k) {
            cx<T> t = tw[p * k * s];
            if (INV) t = conj(t);
            y[q + s * (R * p + k)] = v[k] * t;
        }
    }
}

// Full in-LDS FFT of nseq sequences; returns the buffer holding the result.
template <typename T, bool INV>
__device__ cx<T>* lds_fft(cx<T>* a, cx<T>* b, const cx<T>* tw, int nseq, int len, int lenP, const Rad& rad) {
    int s = 1;
    for (int i = 0; i < rad.n; ++i) {
        const int r = rad.r[i];
        switch (r) {
            case 2: stockham_stage<T, 2, INV>(a, b, tw, nseq, len, lenP, s); break;
            case 3: stockham_stage<T, 3, INV>(a, b, tw, nseq, len, lenP, s); break;
            case 4: stockham_stage<T, 4, INV>(a, b, tw, nseq, len, lenP, s); break;
            case 5: stockham_stage<T, 5, INV>(a, b, tw, nseq, len, lenP, s); break;
            case 7: stockham_stage<T, 7, INV>(a, b, tw, nseq, len, lenP, s); break;
            case 11: stockham_stage<T, 11, INV>(a, b, tw, nseq, len, lenP, s); break;
            default: stockham_stage<T, 13, INV>(a, b, tw, nseq, len, lenP, s); break;
        }
        __syncthreads();
        cx<T>* t = a;
        a = b;
        b = t;
        s *= r;
    }
    return a;
}

// ---------------------------------------------------------------- loaders
template <typename T>
__device__ inline T envelope(double env_log2, int t) {
    return (T)exp2((T)(env_log2 * (double)t));
}
template <>
__device__ inline float envelope<float>(double env_log2, int t) {
    return exp2f((float)(env_log2 * (double)t));
}

// packed half-length sequence of a real signal: z[j] = x[2j] e(2j) + i x[2j+1] e(2j+1)
template <typename T>
__device__ inline cx<T> load_pack(const FftArgs<T>& a, int sig, int j) {
    const int t = 2 * j;
    T re, im;
    if (a.ci_n > 0) {   // channel-innermost source: the layout conversion is fused into this load
        const int b = sig / a.ci_n, n = sig - b * a.ci_n;
        const T* x = a.xr + ((size_t)b * a.ci_T) * a.ci_n + n;
        re = (t < a.t_in) ? x[(size_t)t * a.ci_n] : (T)0;
        im = (t + 1 < a.t_in) ? x[(size_t)(t + 1) * a.ci_n] : (T)0;
    } else {
        const T* x = a.xr + (size_t)sig * a.xr_stride;
        re = (t < a.t_in) ? x[t] : (T)0;
        im = (t + 1 < a.t_in) ? x[t + 1] : (T)0;
    }
    if (a.env_log2 != 0.0) {
        re *= envelope<T>(a.env_log2, t);
        im *= envelope<T>(a.env_log2, t + 1);
    }
    return cx<T>(re, im);
}

// whole-length, aligned, signal-planar rows (the usual case, selected on the host): the sample pair is
// one unguarded 2-element load -- no per-element layout / length branches in the unrolled loaders
template <typename T>
__device__ inline cx<T> load_pack_fast(const FftArgs<T>& a, int sig, int j) {
    const int t = 2 * j;
    cx<T> v = *reinterpret_cast<const cx<T>*>(a.xr + (size_t)sig * a.xr_stride + t);
    if (a.env_log2 != 0.0) {
        v.x *= envelope<T>(a.env_log2, t);
        v.y *= envelope<T>(a.env_log2, t + 1);
    }
    return v;
}

// Hermitian pre-step of the inverse real FFT:
//   Zf[j] = (X[j] + conj X[L-j]) + i conj(W_n^j) (X[j] - conj X[L-j])
template <typename T>
__device__ inline cx<T> load_irfft_pre(const FftArgs<T>& a, int sig, int j) {
    const cx<T>* X = a.Xc + (size_t)sig * a.xc_stride;
    cx<T> xa = X[j], xb = X[a.L - j];
    if (j == 0) {  // DC and Nyquist: imaginary parts are ignored (C2R semantics)
        xa.y = 0;
        xb.y = 0;
    } else if (a.interior) {
        xa = (T)0.5 * xa;
        xb = (T)0.5 * xb;
    }
    xb = conj(xb);
    cx<T> sum = xa + xb, dif = xa - xb;
    cx<T> w = conj(a.W[j]);
    return sum + mul_i(w * dif);
}

template <typename T, int LOAD>
__device__ inline cx<T> load_any(const FftArgs<T>& a, int sig, int j) {
    if (LOAD == LOAD_PACK) return load_pack<T>(a, sig, j);
    if (LOAD == LOAD_PACK_FAST) return load_pack_fast<T>(a, sig, j);
    if (LOAD == LOAD_IRFFT_PRE) return load_irfft_pre<T>(a, sig, j);
    return a.scratch[(size_t)sig * a.L + j];
}

// Block -> (column tile, signal).  With a channel-innermost source the ci_n signals of one batch
// item read the SAME cache lines (each uses 1/ci_n of every line), so their blocks are given
// consecutive slots on the same XCD (block q runs on XCD q % 8): the first one pulls the lines
// from HBM into that XCD's L2, the others hit there.
template <typename T, int LOAD>
__device__ inline bool cols_block(const FftArgs<T>& a, int nsig, int& tile, int& sig) {
    if (a.ci_n > 0) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        const int n = r % a.ci_n, p = (r / a.ci_n) * 8 + xcd;   // p indexes (batch item, tile) pairs
        tile = p % a.ntiles;
        const int b = p / a.ntiles;
        sig = b * a.ci_n + n;
        return sig < nsig;
    }
    if (LOAD == LOAD_IRFFT_PRE) {
        // inverse transform: a tile also reads the mirror columns X[L-j] that another tile of the
        // same signal owns.  Give each XCD (block q runs on XCD q % 8) a CONTIGUOUS range of
        // (signal, tile) items so both reads meet in one L2 (fabric traffic 452 -> ~300 MB; the
        // forward pass, which has no such sharing, is faster with the plain round-robin order)
        const int per_xcd = (int)(gridDim.x >> 3);
        const int idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        tile = idx % a.ntiles;
        sig = idx / a.ntiles;
        return sig < nsig;
    }
    tile = blockIdx.x % a.ntiles;
    sig = blockIdx.x / a.ntiles;
    return sig < nsig;
}

// ---------------------------------------------------------------- pass 1: column FFTs
template <typename T, int LOAD, bool INV>
__global__ void __launch_bounds__(256) fft_cols(FftArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* buf0 = reinterpret_cast<cx<T>*>(smem);
    cx<T>* buf1 = buf0 + a.CT * a.L1P;
    cx<T>* tw = buf1 + a.CT * a.L1P;
    int tile, sig;
    if (!cols_block<T, LOAD>(a, a.nsig, tile, sig)) return;
    const int c0 = tile * a.CT;
    const int nc = min(a.CT, a.L2 - c0);
    const int twstep = a.n / a.L1;
    for (int j = threadIdx.x; j < a.L1; j += blockDim.x) tw[j] = a.W[j * twstep];
    for (int e = threadIdx.x; e < nc * a.L1; e += blockDim.x) {
        const int t1 = e / nc, c = e - t1 * nc;
        buf0[c * a.L1P + t1] = load_any<T, LOAD>(a, sig, t1 * a.L2 + c0 + c);
    }
    __syncthreads();
    cx<T>* res = lds_fft<T, INV>(buf0, buf1, tw, nc, a.L1, a.L1P, a.rad1);
    cx<T>* out = a.scratch + (size_t)sig * a.L;
    for (int e = threadIdx.x; e < nc * a.L1; e += blockDim.x) {
        const int k1 = e / nc, c = e - k1 * nc;
        cx<T> w = a.W[2 * (c0 + c) * k1];  // W_L^(c k1) = W_n^(2 c k1), index < n
        if (INV) w = conj(w);
        out[k1 * a.L2 + c0 + c] = res[c * a.L1P + k1] * w;
    }
}

// ---------------------------------------------------------------- pass 2: row FFTs + epilogue
template <typename T, int LOAD, int EPI, bool INV>
__global__ void __launch_bounds__(256) fft_rows(FftArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x % a.ntiles, sig = blockIdx.x / a.ntiles;
    const int P = (EPI == EPI_RFFT_POST) ? (a.L1 / 2 + 1) : a.L1;  // primary rows
    const int r0 = tile * a.RT;
    const int nr = min(a.RT, P - r0);
    const int nslots = a.per * nr;
    const int maxslots = a.per * a.RT;
    cx<T>* buf0 = reinterpret_cast<cx<T>*>(smem);
    cx<T>* buf1 = buf0 + maxslots * a.L2P;
    cx<T>* tw = buf1 + maxslots * a.L2P;
    const int twstep = a.n / a.L2;
    for (int j = threadIdx.x; j < a.L2; j += blockDim.x) tw[j] = a.W[j * twstep];
    for (int e = threadIdx.x; e < nslots * a.L2; e += blockDim.x) {
        const int sl = e / a.L2, t2 = e - sl * a.L2;
        int row;
        bool valid = true;
        if (a.per == 2) {
            const int r = r0 + (sl >> 1);
            if (sl & 1) {
                row = (a.L1 - r) % a.L1;
                valid = (row != r);
            } else {
                row = r;
            }
        } else {
            row = r0 + sl;
        }
        cx<T> v(0, 0);
        if (valid) v = load_any<T, LOAD>(a, sig, row * a.L2 + t2);
        buf0[sl * a.L2P + t2] = v;
    }
    __syncthreads();
    cx<T>* res = lds_fft<T, INV>(buf0, buf1, tw, nslots, a.L2, a.L2P, a.rad2);

    if (EPI == EPI_RFFT_POST) {
        // X[k] = 1/2 [ (Z[k] + conj Z[L-k]) - i W_n^k (Z[k] - conj Z[L-k]) ],  Z[L] := Z[0]
        cx<T>* X = a.Xout + (size_t)sig * a.xc_stride;
        const T hs = (T)0.5 * a.scale;
        const T wi = a.interior ? (T)2 : (T)1;
        for (int e = threadIdx.x; e < nr * a.L2; e += blockDim.x) {
            const int k2 = e / nr, i = e - k2 * nr;
            const int r = r0 + i;
            const int k = r + a.L1 * k2;
            const int km = (k == 0) ? 0 : a.L - k;
            const int rowm = km % a.L1, colm = km / a.L1;
            const int slm = (rowm == r) ? a.per * i : a.per * i + 1;
            const cx<T> zk = res[(a.per * i) * a.L2P + k2];
            const cx<T> zm = res[slm * a.L2P + colm];
            {
                const cx<T> p = zk + conj(zm), d = zk - conj(zm);
                const cx<T> o = p + mul_mi(a.W[k] * d);
                const T sc = (k == 0) ? hs : hs * wi;
                X[k] = cx<T>(sc * o.x, sc * o.y);
                if (k == 0) {  // Nyquist bin k = L: W_n^L = -1
                    const cx<T> o2 = p + mul_i(d);
                    X[a.L] = cx<T>(hs * o2.x, hs * o2.y);
                }
            }
            if (k != 0 && rowm != r) {  // partner bin L-k lives in a row only this workgroup holds
                const cx<T> p = zm + conj(zk), d = zm - conj(zk);
                const cx<T> o = p + mul_mi(a.W[km] * d);
                const T sc = hs * wi;
                X[km] = cx<T>(sc * o.x, sc * o.y);
            }
        }
    } else {
        // y[2j] = scale e(2j) Re z[j],  y[2j+1] = scale e(2j+1) Im z[j],  j = k1 + L1 k2
        T* y = a.yr + (size_t)sig * a.yr_stride;
        for (int e = threadIdx.x; e < nr * a.L2; e += blockDim.x) {
            const int k2 = e / nr, i = e - k2 * nr;
            const int j = (r0 + i) + a.L1 * k2;
            const cx<T> z = res[i * a.L2P + k2];
            const int t = 2 * j;
            T re = a.scale * z.x, im = a.scale * z.y;
            if (a.env_log2 != 0.0) {
                re *= envelope<T>(a.env_log2, t);
                im *= envelope<T>(a.env_log2, t + 1);
            }
            if (t < a.t_out) y[t] = re;
            if (t + 1 < a.t_out) y[t + 1] = im;
        }
    }
}


// ================================================================ fast path: two register stages
// For the production lengths (nfft/2 = L1*L2 with L1, L2 in {200, 240, 300, 320, 400, 480, ...})
// each sub-FFT of length len = A*B is done as TWO in-register stages with ONE LDS exchange:
//   stage 1: B work items per sequence, each an A-point FFT over the stride-B samples, loaded
//            straight from global memory, multiplied by W_len^(tb*ka), written to LDS;
//   stage 2: A work items per sequence, each a B-point FFT read from LDS; results leave for
//            global memory (pass 1) or go back to LDS in natural order for the epilogue (pass 2).
// The A- and B-point FFTs are fully unrolled mixed-radix networks on registers whose twiddles
// are compile-time constants.  Compared with the generic Stockham path (one LDS round trip per
// radix-2..5 stage) this does a quarter of the LDS traffic and a third of the barriers.


// Columns per workgroup in pass 1: CT x 8 B (c64) contiguous per row -- 128-byte segments for
// CT = 16, 256-byte for CT = 32.
//
// Inter-pass twiddle W_L^(col * k1), k1 = ka + A kb: factored as W_L^(col ka) * W_L^(col A kb).
// The first factor is one value per stage-2 thread, the second a (CT x B) table in LDS filled
// with CT*B gathers per workgroup -- instead of one gather from the master table per OUTPUT
// element (lanes of a wavefront hit 64 different cache lines each time: the twiddle gathers cost
// the L1 four times the cycles of the data itself).
template <typename T, int A, int B, int LOAD, bool INV, int CT>
__global__ void __launch_bounds__(256) fft_cols_fast(FftArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = LEN | 1;
    cx<T>* U = reinterpret_cast<cx<T>*>(smem);   // [CT][LENP]
    cx<T>* tw = U + CT * LENP;                    // W_LEN^m
    cx<T>* t2 = tw + LEN;                         // [CT][B]: W_L^(col * A * kb)
    cx<T>* wr = t2 + CT * B;                      // LOAD_IRFFT_PRE only: W_n^(row * L2), row < LEN
    cx<T>* wc = wr + LEN;                         //                      W_n^(c0 + c)
    int tile, sig;
    if (!cols_block<T, LOAD>(a, a.nsig, tile, sig)) return;
    const int c0 = tile * CT;
    const int nc = min(CT, a.L2 - c0);
    const int twstep = a.n / LEN;
    for (int j = threadIdx.x; j < LEN; j += 256) tw[j] = a.W[j * twstep];
    for (int j = threadIdx.x; j < CT * B; j += 256) {
        const int c = j / B, kb = j - c * B;
        t2[j] = (c < nc) ? a.W[2 * (c0 + c) * A * kb] : cx<T>(1, 0);
    }
    if constexpr (LOAD == LOAD_IRFFT_PRE) {
        for (int j = threadIdx.x; j < LEN; j += 256) wr[j] = a.W[j * a.L2];
        if (threadIdx.x < CT) wc[threadIdx.x] = (threadIdx.x < nc) ? a.W[c0 + threadIdx.x] : cx<T>(1, 0);
    }
    __syncthreads();
    // stage 1: A-point FFTs over t_a for every (t_b, column).  A thread owns up to NR items and
    // issues the loads of all of them before the first butterfly: the pass is bound by how many
    // bytes a CU keeps in flight (few workgroups fit beside the LDS tile), not by arithmetic.
    constexpr int NR = (B * CT + 255) / 256;
    {
        cx<T> v[NR][A];
        if constexpr (LOAD == LOAD_IRFFT_PRE) {
            // Hermitian pre-step fused into the load (see load_irfft_pre): both halves of every
            // pair are requested before anything is combined; W_n^j, j = row*L2 + col, comes
            // from the two small LDS tables instead of a third global load per element.
            cx<T> xb[NR][A];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int item = threadIdx.x + r * 256;
                const int tb = item / CT, c = item % CT;
                if (item < B * CT && c < nc) {
                    const cx<T>* X = a.Xc + (size_t)sig * a.xc_stride;
#pragma unroll
                    for (int ta = 0; ta < A; ++ta) {
                        const int j = (ta * B + tb) * a.L2 + c0 + c;
                        v[r][ta] = X[j];
                        xb[r][ta] = X[a.L - j];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int item = threadIdx.x + r * 256;
                const int tb = item / CT, c = item % CT;
                if (item < B * CT && c < nc) {
                    const cx<T> wcol = wc[c];
#pragma unroll
                    for (int ta = 0; ta < A; ++ta) {
                        const int row = ta * B + tb;
                        cx<T> xa = v[r][ta], xm = xb[r][ta];
                        if (row == 0 && c0 + c == 0) {  // DC and Nyquist: imaginary parts ignored (C2R)
                            xa.y = 0;
                            xm.y = 0;
                        } else if (a.interior) {
                            xa = (T)0.5 * xa;
                            xm = (T)0.5 * xm;
                        }
                        xm = conj(xm);
                        const cx<T> sum = xa + xm, dif = xa - xm;
                        const cx<T> w = conj(wr[row] * wcol);
                        v[r][ta] = sum + mul_i(w * dif);
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int item = threadIdx.x + r * 256;
                const int tb = item / CT, c = item % CT;
                if (item < B * CT && c < nc) {
#pragma unroll
                    for (int ta = 0; ta < A; ++ta) v[r][ta] = load_any<T, LOAD>(a, sig, (ta * B + tb) * a.L2 + c0 + c);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int item = threadIdx.x + r * 256;
            const int tb = item / CT, c = item % CT;
            if (item < B * CT && c < nc) {
                RegFFT<T, A, INV>::run(v[r]);
                cx<T>* u = U + c * LENP + tb;
                u[0] = v[r][0];
#pragma unroll
                for (int ka = 1; ka < A; ++ka) {
                    cx<T> t = tw[ka * tb];
                    if (INV) t = conj(t);
                    u[ka * B] = v[r][ka] * t;
                }
            }
        }
    }
    __syncthreads();
    // stage 2: B-point FFTs over t_b for every (k_a, column); inter-pass twiddle; store
    cx<T>* out = a.scratch + (size_t)sig * a.L;
    for (int item = threadIdx.x; item < A * CT; item += 256) {
        const int ka = item / CT, c = item % CT;
        if (c < nc) {
            cx<T> v[B];
            const cx<T>* u = U + c * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
            RegFFT<T, B, INV>::run(v);
            const cx<T> w1 = a.W[2 * (c0 + c) * ka];
            const cx<T>* w2 = t2 + c * B;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) {
                const int k1 = ka + A * kb;
                cx<T> w = w1 * w2[kb];
                if (INV) w = conj(w);
                out[(size_t)k1 * a.L2 + c0 + c] = v[kb] * w;
            }
        }
    }
}

// Inverse column pass with mirror columns paired in one workgroup.  The Hermitian pre-step needs X[j] and X[L-j]
// for every element: the kernel above fetches both for each of its 32 columns, so every element of X crosses the
// CU boundary twice (once as itself, once as the partner of an element of another tile) -- 196 MB of loads for 98 MB
// of data, and at ~10 B/cycle/CU that, not HBM, is what holds the pass at 63 us against the forward pass's 45.
// With j = row L2 + col the partner of (row, col) is (L1-1-row, L2-col): a whole other column, rows reversed.  Here a
// workgroup owns 16 columns c <= L2/2 AND their 16 partners L2-c; a thread that holds X[j] and X[L-j] produces both
//     Zf[j]   = s + t,     Zf[L-j] = conj(s - t),      s = xa + conj(xm),  t = i conj(W_n^j) (xa - conj(xm))
// (W_n^(L-j) = -conj(W_n^j)), and the eight rows ta B + tb it holds for column c are, reversed, the eight rows
// (A-1-ta) B + (B-1-tb) of item (B-1-tb) of column L2-c: a complete input of that column's first-stage FFT.
// Columns 0 and L2/2 are their own partners (within the column, other rows): they are processed as plain columns.
// CD = 4 (CT = 8): few signals -- see launch_fft: the tiles shrink until the launch has enough workgroups.
template <typename T, int A, int B, int CD = 16>
__global__ void __launch_bounds__(256) fft_cols_ipair(FftArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CT = 2 * CD, LEN = A * B, LENP = LEN | 1;
    cx<T>* U = reinterpret_cast<cx<T>*>(smem);   // [CT][LENP]: local columns 0..15 = c, 16..31 = partner of column lc-16
    cx<T>* tw = U + CT * LENP;                    // W_LEN^m
    cx<T>* t2 = tw + LEN;                         // [CT][B]: W_L^(col * A * kb)
    cx<T>* wr = t2 + CT * B;                      // W_n^(row * L2)
    cx<T>* wc = wr + LEN;                         // W_n^(d0 + c), c < CD
    int tile, sig;
    if (!cols_block<T, LOAD_IRFFT_PRE>(a, a.nsig, tile, sig)) return;
    const int ndirect = a.L2 / 2 + 1;             // columns 0 .. L2/2
    const int d0 = tile * CD;
    const int nd = min(CD, ndirect - d0);
    auto paired = [&](int c) { return c >= 1 && 2 * c != a.L2; };          // c = global direct column
    auto gcol = [&](int lc) { return lc < CD ? d0 + lc : a.L2 - (d0 + lc - CD); };
    auto live = [&](int lc) { return lc < CD ? lc < nd : (lc - CD < nd && paired(d0 + lc - CD)); };
    const int twstep = a.n / LEN;
    for (int j = threadIdx.x; j < LEN; j += 256) {
        tw[j] = a.W[j * twstep];
        wr[j] = a.W[j * a.L2];
    }
    for (int j = threadIdx.x; j < CT * B; j += 256) {
        const int lc = j / B, kb = j - lc * B;
        t2[j] = live(lc) ? a.W[2 * gcol(lc) * A * kb] : cx<T>(1, 0);
    }
    if (threadIdx.x < CD) wc[threadIdx.x] = (threadIdx.x < nd) ? a.W[d0 + threadIdx.x] : cx<T>(1, 0);
    __syncthreads();
    // stage 1: every load of the thread's items is requested before anything is combined
    constexpr int NR = (B * CD + 255) / 256;
    {
        cx<T> v[NR][A], xb[NR][A];
        const cx<T>* X = a.Xc + (size_t)sig * a.xc_stride;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int item = threadIdx.x + r * 256;
            const int tb = item / CD, c = item % CD;
            if (item < B * CD && c < nd) {
#pragma unroll
                for (int ta = 0; ta < A; ++ta) {
                    const int j = (ta * B + tb) * a.L2 + d0 + c;
                    v[r][ta] = X[j];
                    xb[r][ta] = X[a.L - j];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int item = threadIdx.x + r * 256;
            const int tb = item / CD, c = item % CD;
            if (item < B * CD && c < nd) {
                const cx<T> wcol = wc[c];
                cx<T> zd[A], zm[A];
#pragma unroll
                for (int ta = 0; ta < A; ++ta) {
                    const int row = ta * B + tb;
                    cx<T> xa = v[r][ta], xm = xb[r][ta];
                    if (row == 0 && d0 + c == 0) {  // DC and Nyquist: imaginary parts ignored (C2R)
                        xa.y = 0;
                        xm.y = 0;
                    } else if (a.interior) {
                        xa = (T)0.5 * xa;
                        xm = (T)0.5 * xm;
                    }
                    xm = conj(xm);
                    const cx<T> sum = xa + xm, dif = xa - xm;
                    const cx<T> t = mul_i(conj(wr[row] * wcol) * dif);
                    zd[ta] = sum + t;
                    zm[A - 1 - ta] = conj(sum - t);
                }
                RegFFT<T, A, true>::run(zd);
                cx<T>* u = U + c * LENP + tb;
                u[0] = zd[0];
#pragma unroll
                for (int ka = 1; ka < A; ++ka) u[ka * B] = zd[ka] * conj(tw[ka * tb]);
                if (paired(d0 + c)) {
                    const int tbm = B - 1 - tb;
                    RegFFT<T, A, true>::run(zm);
                    cx<T>* um = U + (c + CD) * LENP + tbm;
                    um[0] = zm[0];
#pragma unroll
                    for (int ka = 1; ka < A; ++ka) um[ka * B] = zm[ka] * conj(tw[ka * tbm]);
                }
            }
        }
    }
    __syncthreads();
    // stage 2: B-point FFTs over t_b for every (k_a, local column); inter-pass twiddle; store
    cx<T>* out = a.scratch + (size_t)sig * a.L;
    for (int item = threadIdx.x; item < A * CT; item += 256) {
        const int ka = item / CT, lc = item % CT;
        if (live(lc)) {
            cx<T> v[B];
            const cx<T>* u = U + lc * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
            RegFFT<T, B, true>::run(v);
            const int col = gcol(lc);
            const cx<T> w1 = a.W[2 * col * ka];
            const cx<T>* w2 = t2 + lc * B;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) {
                const int k1 = ka + A * kb;
                out[(size_t)k1 * a.L2 + col] = v[kb] * conj(w1 * w2[kb]);
            }
        }
    }
}

// pass 2 fast: rows of length LEN = A*B; nslots*A <= 256 so every thread owns at most one
// stage-2 item and the natural-order result can be written back into the same LDS buffer.
template <typename T, int A, int B, int LOAD, int EPI, bool INV>
__global__ void __launch_bounds__(512) fft_rows_fast(FftArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = LEN | 1;
    const int tile = blockIdx.x % a.ntiles, sig = blockIdx.x / a.ntiles;
    const int P = (EPI == EPI_RFFT_POST) ? (a.L1 / 2 + 1) : a.L1;
    const int r0 = tile * a.RT;
    const int nr = min(a.RT, P - r0);
    const int nslots = a.per * nr;
    cx<T>* U = reinterpret_cast<cx<T>*>(smem);    // [per*RT][LENP]
    cx<T>* tw = U + a.per * a.RT * LENP;
    const int twstep = a.n / LEN;
    for (int j = threadIdx.x; j < LEN; j += blockDim.x) tw[j] = a.W[j * twstep];
    __syncthreads();
    for (int item = threadIdx.x; item < nslots * B; item += blockDim.x) {
        const int sl = item / B, tb = item % B;
        int row;
        bool valid = true;
        if (a.per == 2) {
            const int r = r0 + (sl >> 1);
            if (sl & 1) {
                row = (a.L1 - r) % a.L1;
                valid = (row != r);
            } else {
                row = r;
            }
        } else {
            row = r0 + sl;
        }
        cx<T> v[A];
        if (valid) {   // one guarded region around all A loads (a per-element select guards each load separately)
#pragma unroll
            for (int ta = 0; ta < A; ++ta) v[ta] = load_any<T, LOAD>(a, sig, row * LEN + ta * B + tb);
        } else {
#pragma unroll
            for (int ta = 0; ta < A; ++ta) v[ta] = cx<T>(0, 0);
        }
        RegFFT<T, A, INV>::run(v);
        cx<T>* u = U + sl * LENP + tb;
        u[0] = v[0];
#pragma unroll
        for (int ka = 1; ka < A; ++ka) {
            cx<T> t = tw[ka * tb];
            if (INV) t = conj(t);
            u[ka * B] = v[ka] * t;
        }
    }
    __syncthreads();
    {
        const int item = threadIdx.x;
        const bool act = item < nslots * A;
        const int sl = item / A, ka = item % A;
        cx<T> v[B];
        if (act) {
            const cx<T>* u = U + sl * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
            RegFFT<T, B, INV>::run(v);
        }
        __syncthreads();
        if (act) {
            cx<T>* z = U + sl * LENP + ka;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) z[A * kb] = v[kb];
        }
        __syncthreads();
    }
    const cx<T>* res = U;
    if (EPI == EPI_RFFT_POST) {
        // split-step twiddle W_n^k, k = r + L1*k2, as W_n^r * W_n^(L1 k2) from two small LDS tables
        // (the stage-1 table is free by now) instead of one load from the master table per output --
        // as many bytes again as the outputs themselves; the mirror bin's twiddle is W_n^(L-k) = -conj(W_n^k)
        for (int j = threadIdx.x; j < LEN; j += blockDim.x) tw[j] = a.W[a.L1 * j];
        if (threadIdx.x < nr) tw[LEN + threadIdx.x] = a.W[r0 + threadIdx.x];
        __syncthreads();
        cx<T>* X = a.Xout + (size_t)sig * a.xc_stride;
        const T hs = (T)0.5 * a.scale;
        const T wi = a.interior ? (T)2 : (T)1;
        const bool full = (nr == 16);   // index arithmetic without integer divisions in the common case
        for (int e = threadIdx.x; e < nr * LEN; e += blockDim.x) {
            const int k2 = full ? (e >> 4) : e / nr, i = e - k2 * nr;
            const int r = r0 + i;
            const int k = r + a.L1 * k2;
            const int km = (k == 0) ? 0 : a.L - k;
            // km = L - k = rowm + L1*colm:  r > 0: (L1 - r, L2 - 1 - k2);  r == 0: (0, L2 - k2)
            const int rowm = (r == 0) ? 0 : a.L1 - r;
            const int colm = (r == 0) ? ((k2 == 0) ? 0 : LEN - k2) : LEN - 1 - k2;
            const int slm = (rowm == r) ? a.per * i : a.per * i + 1;
            const cx<T> zk = res[(a.per * i) * LENP + k2];
            const cx<T> zm = res[slm * LENP + colm];
            const cx<T> wk = tw[LEN + i] * tw[k2];
            {
                const cx<T> p = zk + conj(zm), d = zk - conj(zm);
                const cx<T> o = p + mul_mi(wk * d);
                const T sc = (k == 0) ? hs : hs * wi;
                X[k] = cx<T>(sc * o.x, sc * o.y);
                if (k == 0) {
                    const cx<T> o2 = p + mul_i(d);
                    X[a.L] = cx<T>(hs * o2.x, hs * o2.y);
                }
            }
            if (k != 0 && rowm != r) {
                const cx<T> p = zm + conj(zk), d = zm - conj(zk);
                const cx<T> wm(-wk.x, wk.y);          // W_n^(L-k) = -conj(W_n^k)
                const cx<T> o = p + mul_mi(wm * d);
                const T sc = hs * wi;
                X[km] = cx<T>(sc * o.x, sc * o.y);
            }
        }
    } else {
        T* y = a.yr + (size_t)sig * a.yr_stride;
        // the (re, im) pair of z[j] is the sample pair (2j, 2j+1): one 2-element store when the row is aligned
        const bool pair_ok = (reinterpret_cast<uintptr_t>(y) % (2 * sizeof(T))) == 0;
        const bool full = (nr == 16);
        for (int e = threadIdx.x; e < nr * LEN; e += blockDim.x) {
            const int k2 = full ? (e >> 4) : e / nr, i = e - k2 * nr;
            const int j = (r0 + i) + a.L1 * k2;
            const cx<T> z = res[i * LENP + k2];
            const int t = 2 * j;
            T re = a.scale * z.x, im = a.scale * z.y;
            if (a.env_log2 != 0.0) {
                re *= envelope<T>(a.env_log2, t);
                im *= envelope<T>(a.env_log2, t + 1);
            }
            if (pair_ok && t + 1 < a.t_out) {
                *reinterpret_cast<cx<T>*>(y + t) = cx<T>(re, im);
            } else {
                if (t < a.t_out) y[t] = re;
                if (t + 1 < a.t_out) y[t + 1] = im;
            }
        }
    }
}

// (A, B) split of a sub-FFT length handled by the fast path; 0 = not supported
struct FastSplit { int len, A, B; };
static const FastSplit kFastSplits[] = {{200, 8, 25}, {240, 16, 15}, {300, 12, 25}, {320, 16, 20},
                                        {400, 16, 25}, {480, 32, 15}};
static const FastSplit* fast_split(int len) {
    for (const FastSplit& f : kFastSplits)
        if (f.len == len) return &f;
    return nullptr;
}

template <typename T, int A, int B>
static void launch_cols_fast(bool inverse, const FftArgs<T>& a, unsigned nblk, size_t lds, hipStream_t st) {
    if (a.CT == 4) {        // few signals (launch_fft): four columns per workgroup
        if (inverse)
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_IRFFT_PRE, true, 4>), dim3(nblk), dim3(256), lds, st, a);
        else if (a.pack_fast)
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_PACK_FAST, false, 4>), dim3(nblk), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_PACK, false, 4>), dim3(nblk), dim3(256), lds, st, a);
    } else if (a.CT == 32) {
        if (inverse)
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_IRFFT_PRE, true, 32>), dim3(nblk), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_PACK, false, 32>), dim3(nblk), dim3(256), lds, st, a);
    } else {
        if (inverse)
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_IRFFT_PRE, true, 16>), dim3(nblk), dim3(256), lds, st, a);
        else if (a.pack_fast)
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_PACK_FAST, false, 16>), dim3(nblk), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL((fft_cols_fast<T, A, B, LOAD_PACK, false, 16>), dim3(nblk), dim3(256), lds, st, a);
    }
}
template <typename T, int A, int B>
static void launch_cols_ipair(const FftArgs<T>& a, unsigned nblk, size_t lds, hipStream_t st) {
    if (a.CT == 8) hipLaunchKernelGGL((fft_cols_ipair<T, A, B, 4>), dim3(nblk), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((fft_cols_ipair<T, A, B, 16>), dim3(nblk), dim3(256), lds, st, a);
}
template <typename T, int A, int B>
static void launch_rows_fast(bool inverse, const FftArgs<T>& a, unsigned nblk, size_t lds, int nthreads, hipStream_t st) {
    if (inverse)
        hipLaunchKernelGGL((fft_rows_fast<T, A, B, LOAD_SCRATCH, EPI_IRFFT_STORE, true>), dim3(nblk), dim3(nthreads), lds, st, a);
    else
        hipLaunchKernelGGL((fft_rows_fast<T, A, B, LOAD_SCRATCH, EPI_RFFT_POST, false>), dim3(nblk), dim3(nthreads), lds, st, a);
}

#define FL_FAST_DISPATCH(FN, len, ...)                                   \
    switch (len) {                                                       \
        case 200: FN<T, 8, 25>(__VA_ARGS__); break;                      \
        case 240: FN<T, 16, 15>(__VA_ARGS__); break;                     \
        case 300: FN<T, 12, 25>(__VA_ARGS__); break;                     \
        case 320: FN<T, 16, 20>(__VA_ARGS__); break;                     \
        case 400: FN<T, 16, 25>(__VA_ARGS__); break;                     \
        case 480: FN<T, 32, 15>(__VA_ARGS__); break;                     \
        default: break;                                                  \
    }

// ---------------------------------------------------------------- twiddles, transpose
template <typename T>
__global__ void twiddle_fill(cx<T>* W, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        double s, c;
        sincospi(2.0 * (double)j / (double)n, &s, &c);
        W[j] = cx<T>((T)c, (T)(-s));
    }
}

template <typename E>
__global__ void __launch_bounds__(256) transpose_kernel(const E* __restrict__ src, E* __restrict__ dst, int rows, int cols,
                                                        long pitch) {
    __shared__ E tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const size_t dbase = (size_t)blockIdx.z * cols * pitch;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = src[base + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) dst[dbase + (size_t)c * pitch + r] = tile[tx][i];
    }
}

// Tall-and-narrow case (rows >> cols, cols <= 64): the channel-innermost (T, N) -> planar (N, T)
// conversion.  A workgroup moves TR consecutive rows: one fully contiguous TR*cols read, cols
// contiguous TR-element writes.  TALL=false is the mirror image (planar -> channel-innermost).
template <typename E, bool TALL>
__global__ void __launch_bounds__(256) transpose_narrow_kernel(const E* __restrict__ src, E* __restrict__ dst,
                                                              int nlong, int nshort, int TR, long pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    E* tile = reinterpret_cast<E*>(smem);  // [TR][nshort + 1]
    const size_t base = (size_t)blockIdx.y * nlong * nshort;
    const int l0 = blockIdx.x * TR;
    const int nl = min(TR, nlong - l0);
    const int tpitch = nshort + 1;
    if (TALL) {
        // src[(l0 + i) * nshort + c] contiguous in (i, c); dst[c * nlong + l0 + i]
        const E* s = src + base + (size_t)l0 * nshort;
        for (int e = threadIdx.x; e < nl * nshort; e += 256) {
            const int i = e / nshort, c = e - i * nshort;
            tile[i * tpitch + c] = s[e];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nl * nshort; e += 256) {
            const int c = e / nl, i = e - c * nl;
            dst[(size_t)blockIdx.y * nshort * pitch + (size_t)c * pitch + l0 + i] = tile[i * tpitch + c];
        }
    } else {
        // src[c * nlong + l0 + i]; dst[(l0 + i) * nshort + c] contiguous in (i, c)
        for (int e = threadIdx.x; e < nl * nshort; e += 256) {
            const int c = e / nl, i = e - c * nl;
            tile[i * tpitch + c] = src[base + (size_t)c * nlong + l0 + i];
        }
        __syncthreads();
        E* d = dst + base + (size_t)l0 * nshort;
        for (int e = threadIdx.x; e < nl * nshort; e += 256) {
            const int i = e / nshort, c = e - i * nshort;
            d[e] = tile[i * tpitch + c];
        }
    }
}

// Vectorised tall-and-narrow transpose for 4-byte elements and NS = 4 / 8 / 16 channels (the (T, N) ->
// (N, T) conversion of every real input): 16-byte loads of the contiguous source, transposed into
// LDS, 16-byte stores of NS contiguous runs.  No integer divisions (NS is a power of two), eight
// 16-byte loads in flight per lane.  TR rows per workgroup, TR % 1024 == 0.
template <int NS>
__global__ void __launch_bounds__(256) transpose_tall4_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                             int nlong, int TR, long pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem);   // [NS][TR + 4]
    const int tp = TR + 4;
    const int l0 = blockIdx.x * TR;
    const int nl = min(TR, nlong - l0);                    // multiple of 4 (host checks nlong % 4 == 0)
    const uint4* s = reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.y * nlong + l0) * NS);
    const int nvec = nl * NS / 4;
    for (int v0 = 0; v0 < nvec; v0 += 256 * 8) {
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int v = v0 + threadIdx.x + 256 * u;
            if (v < nvec) q[u] = s[v];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int v = v0 + threadIdx.x + 256 * u;
            if (v < nvec) {
                const int e = 4 * v, i = e / NS, c = e % NS;
                tile[(c + 0) * tp + i] = q[u].x;
                tile[(c + 1) * tp + i] = q[u].y;
                tile[(c + 2) * tp + i] = q[u].z;
                tile[(c + 3) * tp + i] = q[u].w;
            }
        }
    }
    __syncthreads();
    const int rv = nl / 4;   // 16-byte vectors per output run
    for (int v = threadIdx.x; v < rv * NS; v += 256) {
        const int c = v / rv, iv = v - c * rv;
        const uint4 o = *reinterpret_cast<const uint4*>(tile + c * tp + 4 * iv);
        *reinterpret_cast<uint4*>(dst + ((size_t)blockIdx.y * NS + c) * pitch + l0 + 4 * iv) = o;
    }
}

// ---------------------------------------------------------------- host-side planning
static int g_max_single = 0;
static int g_fast_enabled = 1;
static int g_fast_ct = 0;   // 0 = default choice, 16 / 32 = forced (tuning hook)
static int g_fast_pair = 1; // inverse column pass with mirror columns paired (tuning hook: mode 2 switches it off)
static int g_fast_rt = 0;   // rows per workgroup in pass 2 (0 = default)

static bool factorize(int n, Rad& rad) {
    rad.n = 0;
    const int cand[] = {4, 2, 3, 5, 7, 11, 13};
    for (int r : cand)
        while (n % r == 0 && rad.n < 24) {
            rad.r[rad.n++] = r;
            n /= r;
        }
    return n == 1;
}

struct Plan {
    int n, L, L1, L2;
    Rad rad1, rad2;
};

static const int LDS_BUDGET = 64 * 1024;

static int max_single(bool f64) {
    if (g_max_single > 0) return g_max_single;
    return f64 ? 1024 : 2048;
}

static int make_plan(int nfft, bool f64, Plan& p) {
    if (nfft < 2 || (nfft & 1)) {
        set_error("nfft=%d: only even transform lengths are supported", nfft);
        return FL_ERR_UNSUPPORTED;
    }
    p.n = nfft;
    p.L = nfft / 2;
    const int ms = max_single(f64);
    const int maxrow = f64 ? 512 : 1024;
    if (p.L <= ms) {
        p.L1 = 1;
        p.L2 = p.L;
    } else {
        p.L1 = 0;
        for (int d = (int)floor(sqrt((double)p.L)); d >= 2; --d) {
            if (p.L % d == 0) {
                if (p.L / d <= maxrow) {
                    p.L1 = d;
                    p.L2 = p.L / d;
                } else if (p.L / d <= 2 * maxrow) {  // long columns, short rows
                    p.L1 = p.L / d;
                    p.L2 = d;
                }
                break;
            }
        }
        if (!p.L1) {
            set_error("nfft=%d: no two-pass factorisation with row length <= %d", nfft, maxrow);
            return FL_ERR_UNSUPPORTED;
        }
    }
    if (!factorize(p.L1, p.rad1) || !factorize(p.L2, p.rad2)) {
        set_error("nfft=%d: half-length %d has a prime factor > 13", nfft, p.L);
        return FL_ERR_UNSUPPORTED;
    }
    return FL_OK;
}

template <typename T>
static size_t cols_grid(const FftArgs<T>& a, int nsig) {
    if (a.ci_n > 0) {
        const size_t pairs = (size_t)a.ntiles * (nsig / a.ci_n);
        return (pairs + 7) / 8 * 8 * a.ci_n;
    }
    return ((size_t)a.ntiles * nsig + 7) / 8 * 8;
}

template <typename T>
static int launch_fft(bool inverse, FftArgs<T> a, const Plan& p, int nsig, hipStream_t st) {
    const int esz = (int)sizeof(cx<T>);
    const int budget = LDS_BUDGET / esz;  // complex elements of LDS
    a.n = p.n; a.L = p.L; a.L1 = p.L1; a.L2 = p.L2;
    a.L1P = p.L1 | 1; a.L2P = p.L2 | 1;
    a.rad1 = p.rad1; a.rad2 = p.rad2;
    a.nsig = nsig;
    if (nsig <= 0) return FL_OK;
    const bool use_fast = g_fast_enabled && p.L1 > 1 && fast_split(p.L1) && fast_split(p.L2);
    if (p.L1 > 1) {
        FL_REQUIRE(a.scratch != nullptr, "two-pass FFT (nfft=%d) needs a scratch buffer", p.n);
        if (use_fast) {
            const FastSplit* f1 = fast_split(p.L1);
            // 32-column tiles pay off for the inverse (its mirror reads straddle lines), not forward
            const bool pair = inverse && sizeof(T) == 4 && g_fast_pair && g_fast_ct == 0 && a.ci_n == 0;
            a.CT = (sizeof(T) == 4 && (g_fast_ct == 32 || (g_fast_ct == 0 && inverse))) ? 32 : 16;
            // Few signals (a batch-1 feedback delay network transforms ONE, config 3: six such launches per step): with
            // 16-column tiles the pass is 20 workgroups, each moving its 77 KB at the ~10 B/cycle a single CU sustains --
            // 3 us of an 11 us launch.  Four columns per workgroup (32-byte pieces: irrelevant at 1.5 MB) spread it over four
            // times as many CUs.  (round 5; the row pass below shrinks its tiles the same way)
            const bool few = g_fast_ct == 0 && a.ci_n == 0 && (long)cdiv_i(p.L2, 16) * nsig < 64;
            int cd = 16;
            if (few) {
                cd = 4;
                a.CT = pair ? 8 : 4;
            }
            a.ntiles = pair ? cdiv_i(p.L2 / 2 + 1, cd) : cdiv_i(p.L2, a.CT);
            const size_t lds = (size_t)(a.CT * a.L1P + p.L1 + a.CT * f1->B + (inverse ? p.L1 + a.CT : 0)) * esz;
            const size_t nblk = cols_grid(a, nsig);
            FL_REQUIRE(nblk < (1ull << 31), "grid too large");
            if (pair) {
                FL_FAST_DISPATCH(launch_cols_ipair, p.L1, a, (unsigned)nblk, lds, st)
            } else {
                FL_FAST_DISPATCH(launch_cols_fast, p.L1, inverse, a, (unsigned)nblk, lds, st)
            }
            FL_CHECK_LAUNCH("fft_cols_fast");
        } else {
            int ct = (budget - p.L1) / (2 * a.L1P);
            if (ct > 32) ct = 32;
            FL_REQUIRE(ct >= 1, "column pass does not fit in LDS (L1=%d)", p.L1);
            a.CT = ct;
            a.ntiles = cdiv_i(p.L2, ct);
            const size_t lds = (size_t)(2 * ct * a.L1P + p.L1) * esz;
            const size_t nblk = cols_grid(a, nsig);
            FL_REQUIRE(nblk < (1ull << 31), "grid too large");
            if (inverse)
                hipLaunchKernelGGL((fft_cols<T, LOAD_IRFFT_PRE, true>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else
                hipLaunchKernelGGL((fft_cols<T, LOAD_PACK, false>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            FL_CHECK_LAUNCH("fft_cols");
        }
    }
    if (use_fast) {
        const FastSplit* fs = fast_split(p.L2);
        const int per = inverse ? 1 : 2;
        a.per = per;
        // rows per workgroup: 16 (128-byte output segments in c64) if LDS allows, every stage-2
        // item (slot, k_a) on its own thread of a 256- or 512-thread workgroup
        const int P = inverse ? p.L1 : (p.L1 / 2 + 1);
        int rt = 512 / (per * fs->A);
        const int lds_rows = (LDS_BUDGET / esz - p.L2 - 16) / (per * a.L2P);
        if (rt > lds_rows) rt = lds_rows;
        if (rt > 16) rt = 16;
        if (g_fast_rt > 0 && g_fast_rt < rt) rt = g_fast_rt;
        if (g_fast_rt == 0) {      // few signals: tiles of at least two rows, as many workgroups as the part has CUs (0.346 -> 0.334 ms per FDN step)
            int want = (int)(((long)P * nsig + 255) / 256);
            if (want < 2) want = 2;
            if (want < rt) rt = want;
        }
        if (rt > P) rt = P;
        FL_REQUIRE(rt >= 1, "row pass does not fit (L2=%d)", p.L2);
        const int nthreads = (per * rt * fs->A > 256) ? 512 : 256;
        a.RT = rt;
        a.ntiles = cdiv_i(P, rt);
        const size_t lds = (size_t)(per * rt * a.L2P + p.L2 + rt) * esz;   // + rt: the W_n^r table of the split step
        const size_t nblk = (size_t)a.ntiles * nsig;
        FL_REQUIRE(nblk < (1ull << 31), "grid too large");
        FL_FAST_DISPATCH(launch_rows_fast, p.L2, inverse, a, (unsigned)nblk, lds, nthreads, st)
        FL_CHECK_LAUNCH("fft_rows_fast");
    } else {
        const int per = (inverse || p.L1 == 1) ? 1 : 2;  // slots per primary row
        a.per = per;
        int rt = (budget - p.L2) / (2 * per * a.L2P);
        if (rt > 32) rt = 32;
        FL_REQUIRE(rt >= 1, "row pass does not fit in LDS (L2=%d)", p.L2);
        const int P = inverse ? p.L1 : (p.L1 / 2 + 1);
        if (rt > P) rt = P;
        a.RT = rt;
        a.ntiles = cdiv_i(P, rt);
        const size_t lds = (size_t)(2 * per * rt * a.L2P + p.L2) * esz;
        const size_t nblk = (size_t)a.ntiles * nsig;
        FL_REQUIRE(nblk < (1ull << 31), "grid too large");
        if (!inverse) {
            if (p.L1 > 1)
                hipLaunchKernelGGL((fft_rows<T, LOAD_SCRATCH, EPI_RFFT_POST, false>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else
                hipLaunchKernelGGL((fft_rows<T, LOAD_PACK, EPI_RFFT_POST, false>), dim3((unsigned)nblk), dim3(256), lds, st, a);
        } else {
            if (p.L1 > 1)
                hipLaunchKernelGGL((fft_rows<T, LOAD_SCRATCH, EPI_IRFFT_STORE, true>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else
                hipLaunchKernelGGL((fft_rows<T, LOAD_IRFFT_PRE, EPI_IRFFT_STORE, true>), dim3((unsigned)nblk), dim3(256), lds, st, a);
        }
        FL_CHECK_LAUNCH("fft_rows");
    }
    return FL_OK;
}

template <typename T>
static int rfft_impl(const void* x, long x_sig_stride, int ci_n, int t_in, void* X, long X_sig_stride, void* scratch,
                     const void* W, int nsig, int nfft, double scale, double env_log2, int interior_x2, void* stream) {
    Plan p;
    int rc = make_plan(nfft, sizeof(T) == 8, p);
    if (rc) return rc;
    if (nsig == 0) return FL_OK;
    FL_REQUIRE(x && X && W, "rfft: null pointer");
    FL_REQUIRE(t_in >= 0 && nsig >= 0, "rfft: bad sizes");
    FftArgs<T> a = {};
    a.xr = (const T*)x;
    a.xr_stride = x_sig_stride;
    a.t_in = t_in < nfft ? t_in : nfft;
    if (ci_n > 0) {
        FL_REQUIRE(nsig % ci_n == 0, "rfft: nsig must be a multiple of the channel count for a channel-innermost source");
        a.ci_n = ci_n;
        a.ci_T = t_in;
    }
    a.pack_fast = ci_n <= 0 && t_in >= nfft && x_sig_stride % 2 == 0 && reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0;
    a.Xout = (cx<T>*)X;
    FL_REQUIRE(X_sig_stride >= nfft / 2 + 1, "rfft: X_sig_stride must be >= nfft/2+1");
    a.xc_stride = X_sig_stride;
    a.scratch = (cx<T>*)scratch;
    a.W = (const cx<T>*)W;
    a.scale = (T)scale;
    a.env_log2 = env_log2;
    a.interior = interior_x2;
    return launch_fft<T>(false, a, p, nsig, (hipStream_t)stream);
}

template <typename T>
static int irfft_impl(const void* X, long X_sig_stride, void* y, long y_sig_stride, int t_out, void* scratch, const void* W,
                      int nsig, int nfft, double scale, double env_log2, int interior_half, void* stream) {
    Plan p;
    int rc = make_plan(nfft, sizeof(T) == 8, p);
    if (rc) return rc;
    if (nsig == 0) return FL_OK;
    FL_REQUIRE(X && y && W, "irfft: null pointer");
    FL_REQUIRE(t_out >= 0 && t_out <= nfft && nsig >= 0, "irfft: t_out must be in [0, nfft]");
    FftArgs<T> a = {};
    a.Xc = (const cx<T>*)X;
    FL_REQUIRE(X_sig_stride >= nfft / 2 + 1, "irfft: X_sig_stride must be >= nfft/2+1");
    a.xc_stride = X_sig_stride;
    a.yr = (T*)y;
    a.yr_stride = y_sig_stride;
    a.t_out = t_out;
    a.scratch = (cx<T>*)scratch;
    a.W = (const cx<T>*)W;
    a.scale = (T)scale;
    a.env_log2 = env_log2;
    a.interior = interior_half;
    return launch_fft<T>(true, a, p, nsig, (hipStream_t)stream);
}

template <typename T>
static int twiddle_impl(void* W, int nfft, void* stream) {
    FL_REQUIRE(W && nfft > 0, "twiddle: bad arguments");
    hipLaunchKernelGGL((twiddle_fill<T>), dim3(cdiv_i(nfft, 256)), dim3(256), 0, (hipStream_t)stream, (cx<T>*)W, nfft);
    FL_CHECK_LAUNCH("twiddle_fill");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {

int fl_twiddle_fill_f32(void* W, int nfft, void* stream) { return twiddle_impl<float>(W, nfft, stream); }
int fl_twiddle_fill_f64(void* W, int nfft, void* stream) { return twiddle_impl<double>(W, nfft, stream); }

int fl_fft_plan(int nfft, int is_f64, int* L1, int* L2) {
    Plan p;
    int rc = make_plan(nfft, is_f64 != 0, p);
    if (rc) return rc;
    if (L1) *L1 = p.L1;
    if (L2) *L2 = p.L2;
    return FL_OK;
}

size_t fl_fft_scratch_elems(int nfft, int is_f64, int nsig) {
    Plan p;
    if (make_plan(nfft, is_f64 != 0, p) != FL_OK) return 0;
    if (p.L1 == 1 || nsig <= 0) return 0;
    return (size_t)p.L * (size_t)nsig;
}

int fl_debug_set_fft_max_single(int max_half_len) {
    g_max_single = max_half_len > 0 ? max_half_len : 0;
    return FL_OK;
}

int fl_debug_set_fft_fast(int enabled) {
    g_fast_enabled = enabled != 0;
    g_fast_rt = enabled / 1000;
    enabled %= 1000;
    g_fast_enabled = enabled != 0;
    g_fast_ct = (enabled == 32) ? 32 : (enabled == 16 ? 16 : 0);   // tuning: force 16- / 32-column tiles in pass 1
    g_fast_pair = enabled != 2;                                     // 2: inverse column pass without mirror pairing
    return FL_OK;
}

int fl_rfft_f32(const void* x, long xs, int t_in, void* X, long Xs, void* scratch, const void* W, int nsig, int nfft,
                double scale, double env_log2, int interior_x2, void* stream) {
    return rfft_impl<float>(x, xs, 0, t_in, X, Xs, scratch, W, nsig, nfft, scale, env_log2, interior_x2, stream);
}
int fl_rfft_ci_f32(const void* x, int n_chan, int t_in, void* X, long Xs, void* scratch, const void* W, int nsig, int nfft,
                   double scale, double env_log2, int interior_x2, void* stream) {
    FL_REQUIRE(n_chan > 0, "rfft_ci: n_chan must be positive");
    return rfft_impl<float>(x, 0, n_chan, t_in, X, Xs, scratch, W, nsig, nfft, scale, env_log2, interior_x2, stream);
}
int fl_rfft_f64(const void* x, long xs, int t_in, void* X, long Xs, void* scratch, const void* W, int nsig, int nfft,
                double scale, double env_log2, int interior_x2, void* stream) {
    return rfft_impl<double>(x, xs, 0, t_in, X, Xs, scratch, W, nsig, nfft, scale, env_log2, interior_x2, stream);
}
int fl_rfft_ci_f64(const void* x, int n_chan, int t_in, void* X, long Xs, void* scratch, const void* W, int nsig, int nfft,
                   double scale, double env_log2, int interior_x2, void* stream) {
    FL_REQUIRE(n_chan > 0, "rfft_ci: n_chan must be positive");
    return rfft_impl<double>(x, 0, n_chan, t_in, X, Xs, scratch, W, nsig, nfft, scale, env_log2, interior_x2, stream);
}
int fl_irfft_f32(const void* X, long Xs, void* y, long ys, int t_out, void* scratch, const void* W, int nsig, int nfft,
                 double scale, double env_log2, int interior_half, void* stream) {
    return irfft_impl<float>(X, Xs, y, ys, t_out, scratch, W, nsig, nfft, scale, env_log2, interior_half, stream);
}
int fl_irfft_f64(const void* X, long Xs, void* y, long ys, int t_out, void* scratch, const void* W, int nsig, int nfft,
                 double scale, double env_log2, int interior_half, void* stream) {
    return irfft_impl<double>(X, Xs, y, ys, t_out, scratch, W, nsig, nfft, scale, env_log2, interior_half, stream);
}

int fl_transpose(const void* src, void* dst, int nbatch, int rows, int cols, long dst_pitch, int elem_bytes, void* stream) {
    FL_REQUIRE(src && dst, "transpose: null pointer");
    FL_REQUIRE(nbatch >= 0 && rows >= 0 && cols >= 0 && dst_pitch >= rows, "transpose: bad sizes (dst_pitch >= rows)");
    if (nbatch == 0 || rows == 0 || cols == 0) return FL_OK;
    FL_REQUIRE(nbatch <= 65535, "transpose: batch too large");
    hipStream_t st = (hipStream_t)stream;
    const bool tall = cols <= 64 && rows >= 4 * cols, wide = rows <= 64 && cols >= 4 * rows && dst_pitch == rows;
    if (tall && elem_bytes == 4 && (cols == 4 || cols == 8 || cols == 16) && rows % 4 == 0 && dst_pitch % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
        const int tr = (cols == 4) ? 2048 : (cols == 8 ? 1024 : 512);   // 32 KB of LDS per workgroup
        dim3 g2(cdiv_i(rows, tr), nbatch);
        const size_t lds = (size_t)cols * (tr + 4) * 4;
        if (cols == 4) hipLaunchKernelGGL((transpose_tall4_kernel<4>), g2, dim3(256), lds, st, (const uint32_t*)src, (uint32_t*)dst, rows, tr, dst_pitch);
        else if (cols == 8) hipLaunchKernelGGL((transpose_tall4_kernel<8>), g2, dim3(256), lds, st, (const uint32_t*)src, (uint32_t*)dst, rows, tr, dst_pitch);
        else hipLaunchKernelGGL((transpose_tall4_kernel<16>), g2, dim3(256), lds, st, (const uint32_t*)src, (uint32_t*)dst, rows, tr, dst_pitch);
        FL_CHECK_LAUNCH("transpose_tall4");
        return FL_OK;
    }
    if (tall || wide) {
        const int nshort = tall ? cols : rows, nlong = tall ? rows : cols;
        int TR = (32 * 1024) / ((nshort + 1) * elem_bytes);   // ~32 KB of LDS
        if (TR > 2048) TR = 2048;
        TR = (TR / 64) * 64;
        if (TR < 64) TR = 64;
        dim3 g2(cdiv_i(nlong, TR), nbatch);
        const size_t lds = (size_t)TR * (nshort + 1) * elem_bytes;
#define FL_TN(E) \
        if (tall) hipLaunchKernelGGL((transpose_narrow_kernel<E, true>), g2, dim3(256), lds, st, (const E*)src, (E*)dst, nlong, nshort, TR, dst_pitch); \
        else hipLaunchKernelGGL((transpose_narrow_kernel<E, false>), g2, dim3(256), lds, st, (const E*)src, (E*)dst, nlong, nshort, TR, dst_pitch)
        switch (elem_bytes) {
            case 4: FL_TN(uint32_t); break;
            case 8: FL_TN(uint64_t); break;
            case 16: FL_TN(uint4); break;
            default: set_error("transpose: elem_bytes must be 4, 8 or 16"); return FL_ERR_BAD_ARG;
        }
#undef FL_TN
        FL_CHECK_LAUNCH("transpose_narrow");
        return FL_OK;
    }
    FL_REQUIRE(cdiv_i(rows, 32) <= 65535, "transpose: too many rows");
    dim3 grid(cdiv_i(cols, 32), cdiv_i(rows, 32), nbatch);
    switch (elem_bytes) {
        case 4: hipLaunchKernelGGL((transpose_kernel<uint32_t>), grid, dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, rows, cols, dst_pitch); break;
        case 8: hipLaunchKernelGGL((transpose_kernel<uint64_t>), grid, dim3(256), 0, st, (const uint64_t*)src, (uint64_t*)dst, rows, cols, dst_pitch); break;
        case 16: hipLaunchKernelGGL((transpose_kernel<uint4>), grid, dim3(256), 0, st, (const uint4*)src, (uint4*)dst, rows, cols, dst_pitch); break;
        default: set_error("transpose: elem_bytes must be 4, 8 or 16"); return FL_ERR_BAD_ARG;
    }
    FL_CHECK_LAUNCH("transpose");
    return FL_OK;
}

}  // extern "C"
