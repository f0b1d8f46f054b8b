// exp of a small parameter matrix in one launch, and its backward in one launch (gfx950 / MI355X).
//
//   E = exp(A),  A = X  or  A = triu(X,1) - triu(X,1)^T
//
// the "orthogonal" parameter map of dsp.Matrix (flamo/processor/dsp.py:649:
// torch.matrix_exp(skew_matrix(x))) -- the mixing matrix of every feedback delay network.  On the
// device torch.matrix_exp picks its Pade degree from the norm on the HOST (a synchronisation, not
// capturable in a HIP graph); a sync-free fixed schedule written with torch ops is ~45 tiny launches
// forward and ~100 backward, which at batch 1 is a third of a 16-channel FDN training step.  Here:
// one workgroup, matrices in LDS, float64 arithmetic whatever the parameter type,
//
//   forward   s = smallest count with |A|_1 / 2^s <= 1/4 (found by the kernel itself: no host round trip), A_s = A / 2^s;
//             Horner  P_k = I + A_s P_{k+1} / k  (k = ORDER..1, P_{ORDER+1} = I);  E_0 = P_1;  E_{i+1} = E_i^2  (s times)
//   backward  the same schedule reversed, from the stashed s, P_k and E_i:
//             G <- G E_i^T + E_i^T G;   dA_s += G P_{k+1}^T / k,  G <- A_s^T G / k
//
// The order-10 series on |A_s| <= 1/4 truncates at 6e-15; typical mixing-matrix parameters (|A|_1 ~ 4-16) take
// 4-6 squarings, |A|_1 up to 2.6e5 is covered (s <= EXPM_SQ).
#include "common.h"

namespace fl {

constexpr int EXPM_ORDER = 10;
constexpr int EXPM_SQ = 20;                            // most squarings the stash has room for
constexpr int EXPM_SLOTS = EXPM_ORDER + EXPM_SQ + 1;   // stash: A_s, P_2..P_{ORDER+1}, E_0..E_{s-1}; then the count s

// All LDS matrices have rows of NP = N | 1 doubles (odd pitch: transposed reads hit distinct banks).
// C = alpha * op(A) op(B) [+ C] [+ I];  every thread of the workgroup takes outputs idx, idx+nthreads, ...
// (pa: row pitch of A -- NP for an LDS matrix, N for a dense one read straight from global memory)
// NT > 0: the size as a compile-time constant (4, 8, 16, 32 -- the mixing matrices in use): the dot products unroll and
// the index arithmetic folds; with a runtime N each of the ~15-30 products of a launch spent most of its ~1.4 us in
// integer divisions and address multiplies of a 16-term loop.
template <int NT = 0>
__device__ inline void mm_small(const double* __restrict__ A, bool tA, const double* __restrict__ B, bool tB,
                                double* __restrict__ C, double alpha, int Nrt, bool add_identity, bool accumulate,
                                int pa = 0) {
    const int N = NT > 0 ? NT : Nrt;
    const int NP = N | 1;
    if (pa == 0) pa = NP;
    for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        double s0 = 0.0, s1 = 0.0;
        int l = 0;
#pragma unroll
        for (; l + 1 < N; l += 2) {
            s0 += (tA ? A[l * pa + i] : A[i * pa + l]) * (tB ? B[j * NP + l] : B[l * NP + j]);
            s1 += (tA ? A[(l + 1) * pa + i] : A[i * pa + l + 1]) * (tB ? B[j * NP + l + 1] : B[(l + 1) * NP + j]);
        }
        if (l < N) s0 += (tA ? A[l * pa + i] : A[i * pa + l]) * (tB ? B[j * NP + l] : B[l * NP + j]);
        double s = (s0 + s1) * alpha;
        if (add_identity && i == j) s += 1.0;
        C[i * NP + j] = accumulate ? C[i * NP + j] + s : s;
    }
}

// dense (N x N) global <-> padded LDS
template <int NT = 0>
__device__ inline void lds_load(double* __restrict__ dst, const double* __restrict__ src, int Nrt) {
    const int N = NT > 0 ? NT : Nrt;
    const int NP = N | 1;
    for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) dst[(idx / N) * NP + idx % N] = src[idx];
}
template <int NT = 0>
__device__ inline void lds_store(double* __restrict__ dst, const double* __restrict__ src, int Nrt) {
    const int N = NT > 0 ? NT : Nrt;
    const int NP = N | 1;
    for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) dst[idx] = src[(idx / N) * NP + idx % N];
}

template <typename T, int NT>
__global__ void __launch_bounds__(1024) expm_fwd_kernel(const T* __restrict__ X, int Nrt, int skew, T* __restrict__ E,
                                                       T* __restrict__ Ec, double* __restrict__ stash) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = NT > 0 ? NT : Nrt;
    const int NN = N * N, NP = N | 1;
    double* A = reinterpret_cast<double*>(smem);
    double* P = A + N * NP;
    double* Q = P + N * NP;
    // A (unscaled) into LDS with one load per entry, branch-free (a per-column loop over global memory with the skew
    // test inside was N dependent round trips: 10 of the forward's 21 us at N = 16)
    for (int idx = threadIdx.x; idx < NN; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        double v;
        if (skew) {
            const double x = (double)X[j > i ? idx : j * N + i];
            v = (j > i) ? x : ((j < i) ? -x : 0.0);
        } else {
            v = (double)X[idx];
        }
        A[i * NP + j] = v;
    }
    __syncthreads();
    // |A|_1 = largest column sum: one thread per column, then thread 0 picks the squaring count
    __shared__ double colsum[64];
    __shared__ int sq_sh;
    if (threadIdx.x < N) {
        double cs = 0.0;
        for (int i = 0; i < N; ++i) cs += fabs(A[i * NP + threadIdx.x]);
        colsum[threadIdx.x] = cs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double nrm = 0.0;
        for (int j = 0; j < N; ++j) nrm = fmax(nrm, colsum[j]);
        int sq = 0;
        if (nrm > 0.25) {                       // nrm / 2^sq <= 1/4  (a NaN norm keeps sq = 0 and propagates)
            int e;
            const double m = frexp(nrm * 4.0, &e);      // nrm * 4 = m 2^e, 0.5 <= m < 1
            sq = (m == 0.5) ? e - 1 : e;
        }
        sq = sq > EXPM_SQ ? EXPM_SQ : sq;
        sq_sh = sq;
        stash[(size_t)EXPM_SLOTS * NN] = (double)sq;
    }
    __syncthreads();
    const int SQ = sq_sh;
    const double scale = ldexp(1.0, -SQ);
    for (int idx = threadIdx.x; idx < NN; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        const double v = A[i * NP + j] * scale;
        A[i * NP + j] = v;
        stash[idx] = v;
        P[i * NP + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int k = EXPM_ORDER; k >= 1; --k) {
        lds_store<NT>(stash + (size_t)k * NN, P, N);                    // P_{k+1}
        mm_small<NT>(A, false, P, false, Q, 1.0 / k, N, true, false);   // P_k = I + A P_{k+1} / k
        __syncthreads();
        double* t = P; P = Q; Q = t;
    }
    for (int i = 0; i < SQ; ++i) {
        lds_store<NT>(stash + (size_t)(EXPM_ORDER + 1 + i) * NN, P, N);  // E_i
        mm_small<NT>(P, false, P, false, Q, 1.0, N, false, false);
        __syncthreads();
        double* t = P; P = Q; Q = t;
    }
    // E: the real matrix; Ec: the same as the complex matrix (re, 0) the per-bin kernels take (no real -> complex pass
    // afterwards); either may be null
    for (int idx = threadIdx.x; idx < NN; idx += blockDim.x) {
        const T v = (T)P[(idx / N) * NP + idx % N];
        if (Ec) {
            Ec[2 * idx] = v;
            Ec[2 * idx + 1] = (T)0;
        }
        if (E) E[idx] = v;
    }
}

template <typename T, int NT>
__global__ void __launch_bounds__(1024) expm_bwd_kernel(const T* __restrict__ gE, const T* __restrict__ gEc, int Nrt, int skew,
                                                       const double* __restrict__ stash, T* __restrict__ gX) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = NT > 0 ? NT : Nrt;
    const int NN = N * N, NP = N | 1;
    double* G = reinterpret_cast<double*>(smem);
    double* Q = G + N * NP;
    double* dA = Q + N * NP;
    double* S0 = dA + N * NP;     // stashed matrix of the current step, staged through LDS
    // A_s: a fifth LDS matrix while it fits in 160 KB (N <= 56), else read from the stash in place
    const bool as_lds = (size_t)5 * N * NP * sizeof(double) <= 160 * 1024;
    const double* S1 = as_lds ? S0 + N * NP : stash;
    const int p1 = as_lds ? NP : N;
    for (int idx = threadIdx.x; idx < NN; idx += blockDim.x) {
        const int o = (idx / N) * NP + idx % N;
        // gradient of the real output plus the real part of the complex output's gradient (either may be null)
        G[o] = (gE ? (double)gE[idx] : 0.0) + (gEc ? (double)gEc[2 * idx] : 0.0);
        dA[o] = 0.0;
    }
    if (as_lds) lds_load<NT>(S0 + N * NP, stash, N);
    const int SQ = (int)stash[(size_t)EXPM_SLOTS * NN];            // the forward pass's squaring count
    // The stashed matrix of step t+1 is requested while step t multiplies (at most 4 entries per thread: N <= 64,
    // 1024 threads): the steps are a dependent chain, and a global round trip in front of each was most of this kernel.
    // step t = 0..SQ-1: E_{SQ-1-t};  t = SQ..SQ+ORDER-1: P_{t-SQ+2}
    const int steps = SQ + EXPM_ORDER;
    auto slot_of = [&](int t) { return t < SQ ? EXPM_ORDER + 1 + (SQ - 1 - t) : t - SQ + 1; };
    double nxt[4];
    auto fetch = [&](int t) {
        const double* src = stash + (size_t)slot_of(t) * NN;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = threadIdx.x + u * blockDim.x;
            nxt[u] = (t < steps && idx < NN) ? src[idx] : 0.0;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = threadIdx.x + u * blockDim.x;
            if (idx < NN) S0[(idx / N) * NP + idx % N] = nxt[u];
        }
    };
    fetch(0);
    for (int t = 0; t < steps; ++t) {
        __syncthreads();           // the previous step's readers of S0 are done
        commit();
        fetch(t + 1);
        __syncthreads();
        if (t < SQ) {                                                  // E_{i+1} = E_i^2
            mm_small<NT>(G, false, S0, true, Q, 1.0, N, false, false);      // G E_i^T
            mm_small<NT>(S0, true, G, false, Q, 1.0, N, false, true);       // + E_i^T G  (a thread re-reads only its own Q entries)
        } else {                                                       // P_k = I + A_s P_{k+1} / k
            const int k = t - SQ + 1;
            mm_small<NT>(G, false, S0, true, dA, 1.0 / k, N, false, true);  // dA_s += G P_{k+1}^T / k
            mm_small<NT>(S1, true, G, false, Q, 1.0 / k, N, false, false, p1);  // G <- A_s^T G / k
        }
        double* tt = G; G = Q; Q = tt;
    }
    __syncthreads();
    const double scale = ldexp(1.0, -SQ);
    for (int idx = threadIdx.x; idx < NN; idx += blockDim.x) {
        const int i = idx / N, j = idx - i * N;
        double v;
        if (skew) v = (j > i) ? (dA[i * NP + j] - dA[j * NP + i]) : 0.0;
        else v = dA[i * NP + j];
        gX[idx] = (T)(v * scale);
    }
}

// ---------------------------------------------------------------- N = 16 on the float64 matrix cores (round 5)
// The 16-channel mixing matrix of the feedback delay networks (configs 3 and 4): ONE wavefront, every matrix in registers, every
// product four v_mfma_f64_16x16x4_f64 -- no LDS, no barrier, no shuffle.  A matrix M lives in the instruction's C/D layout
// (register r of lane l = M[(l >> 4) + 4 r][l & 15]); in that layout register c IS the B operand of K-chunk c
// (B[k = 4 c + (l >> 4)][j = l & 15]), and the A operand of chunk c (A[i = l & 15][k = 4 c + (l >> 4)]) is register c of the
// TRANSPOSE in the same layout.  So the kernels carry every matrix together with its transpose -- (P, P^T), (E, E^T), (G, G^T)
// -- and each step produces both: eight instructions per Horner step or squaring forward, twelve / sixteen backward, all
// chained through registers.  The schedule and the stash (natural row-major order) are those of the LDS kernels above: either
// backward reads either forward's stash.  Forward 12.3 -> 8.7 us, backward 18.0 -> 14.3 us at N = 16 inside the FDN step (rocprofv3;
// one matrix, one wavefront: ~4.5 us of either is the launch, the backward's first three loads are a dependent chain); the
// replayed FDN step 0.382 -> 0.377 ms, the colorless training step 0.344 -> 0.338 ms.
typedef double d4m __attribute__((ext_vector_type(4)));

// acc += L R, Lt = the LEFT factor's transpose in the C/D layout, R = the right factor in the C/D layout
__device__ __forceinline__ d4m mm16(const d4m Lt, const d4m R, d4m acc) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[0], R[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[1], R[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[2], R[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[3], R[3], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ d4m zero16() { return d4m{0.0, 0.0, 0.0, 0.0}; }
// a row-major (16, 16) matrix of doubles into the C/D layout, or its transpose
__device__ __forceinline__ d4m load16(const double* __restrict__ src, int lane, bool transposed) {
    const int j = lane & 15, i0 = lane >> 4;
    d4m m;
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = transposed ? src[j * 16 + i0 + 4 * r] : src[(i0 + 4 * r) * 16 + j];
    return m;
}
__device__ __forceinline__ void store16(double* __restrict__ dst, int lane, const d4m m) {
    const int j = lane & 15, i0 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(i0 + 4 * r) * 16 + j] = m[r];
}
// alpha M + I
__device__ __forceinline__ d4m scale_plus_identity16(const d4m m, double alpha, int lane) {
    const int j = lane & 15, i0 = lane >> 4;
    d4m o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = fma(m[r], alpha, (i0 + 4 * r == j) ? 1.0 : 0.0);
    return o;
}

template <typename T>
__global__ void __launch_bounds__(64) expm16_fwd_kernel(const T* __restrict__ X, int skew, T* __restrict__ E, T* __restrict__ Ec,
                                                       double* __restrict__ stash) {
    constexpr int N = 16, NN = 256;
    const int lane = threadIdx.x, j = lane & 15, i0 = lane >> 4;
    // A[i][jj] and its transpose: the lane's eight loads are requested together, unconditionally (the skew form reads the upper
    // triangle's element for both halves and signs it afterwards: with the load inside the case distinction every one of them
    // was a branch with its own wait -- eight serial round trips in a kernel of 4 us of arithmetic)
    auto index = [&](int i, int jj) { return skew ? (jj > i ? i * N + jj : jj * N + i) : i * N + jj; };
    auto signed_entry = [&](double x, int i, int jj) { return skew ? (jj > i ? x : (jj < i ? -x : 0.0)) : x; };
    T raw[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        raw[r] = X[index(i0 + 4 * r, j)];
        raw[4 + r] = X[index(j, i0 + 4 * r)];
    }
    d4m A, At;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        A[r] = signed_entry((double)raw[r], i0 + 4 * r, j);
        At[r] = signed_entry((double)raw[4 + r], j, i0 + 4 * r);
    }
    // |A|_1 = largest column sum: this lane's four rows of column j, the four lane groups, then the sixteen columns
    double cs = (fabs(A[0]) + fabs(A[1])) + (fabs(A[2]) + fabs(A[3]));
    cs += __shfl_xor(cs, 16, 64);
    cs += __shfl_xor(cs, 32, 64);
    double nrm = cs;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o, 64));
    int sq = 0;
    if (nrm > 0.25) {                            // nrm / 2^sq <= 1/4
        int e;
        const double m = frexp(nrm * 4.0, &e);
        sq = (m == 0.5) ? e - 1 : e;
    }
    sq = sq > EXPM_SQ ? EXPM_SQ : sq;
    sq = __builtin_amdgcn_readfirstlane(sq);
    if (lane == 0) stash[(size_t)EXPM_SLOTS * NN] = (double)sq;
    const double scale = ldexp(1.0, -sq);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        A[r] *= scale;
        At[r] *= scale;
    }
    store16(stash, lane, A);                     // A_s
    d4m P = scale_plus_identity16(zero16(), 0.0, lane), Pt = P;
#pragma unroll 1
    for (int k = EXPM_ORDER; k >= 1; --k) {
        store16(stash + (size_t)k * NN, lane, P);                                        // P_{k+1}
        const d4m q = mm16(At, P, zero16());                                             // A_s P_{k+1}
        const d4m qt = mm16(P, At, zero16());                                            // P_{k+1}^T A_s^T
        P = scale_plus_identity16(q, 1.0 / k, lane);
        Pt = scale_plus_identity16(qt, 1.0 / k, lane);
    }
#pragma unroll 1
    for (int i = 0; i < sq; ++i) {
        store16(stash + (size_t)(EXPM_ORDER + 1 + i) * NN, lane, P);                     // E_i
        const d4m q = mm16(Pt, P, zero16());                                             // E_i E_i
        const d4m qt = mm16(P, Pt, zero16());                                            // E_i^T E_i^T
        P = q;
        Pt = qt;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = (i0 + 4 * r) * N + j;
        const T v = (T)P[r];
        if (Ec) {
            Ec[2 * idx] = v;
            Ec[2 * idx + 1] = (T)0;
        }
        if (E) E[idx] = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(64) expm16_bwd_kernel(const T* __restrict__ gE, const T* __restrict__ gEc, int skew,
                                                       const double* __restrict__ stash, T* __restrict__ gX) {
    constexpr int N = 16, NN = 256;
    __shared__ double dAs[N * (N + 1)];
    const int lane = threadIdx.x, j = lane & 15, i0 = lane >> 4;
    // the cotangent (real form + real part of the complex form, either may be absent): all sixteen loads requested together --
    // an absent form reads the first element of the other with weight zero (no branch, no wait per load)
    const T* ge = gE ? gE : gEc;
    const T* gc = gEc ? gEc : gE;
    const int ge_s = gE ? 1 : 0, gc_s = gEc ? 2 : 0;
    const double ge_w = gE ? 1.0 : 0.0, gc_w = gEc ? 1.0 : 0.0;
    T rawe[8], rawc[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = (i0 + 4 * r) * N + j, b = j * N + i0 + 4 * r;
        rawe[r] = ge[a * ge_s];
        rawc[r] = gc[a * gc_s];
        rawe[4 + r] = ge[b * ge_s];
        rawc[4 + r] = gc[b * gc_s];
    }
    d4m G, Gt, dA = zero16();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        G[r] = ge_w * (double)rawe[r] + gc_w * (double)rawc[r];
        Gt[r] = ge_w * (double)rawe[4 + r] + gc_w * (double)rawc[4 + r];
    }
    const int SQ = __builtin_amdgcn_readfirstlane((int)stash[(size_t)EXPM_SLOTS * NN]);
    const d4m As = load16(stash, lane, false);
    // step t = 0..SQ-1: E_{SQ-1-t};  t = SQ..SQ+ORDER-1: P_{t-SQ+2} -- the stashed matrix of step t + 1 is requested (both
    // layouts) while step t multiplies: the steps are a dependent chain
    const int steps = SQ + EXPM_ORDER;
    auto slot_of = [&](int t) { return t < SQ ? EXPM_ORDER + 1 + (SQ - 1 - t) : t - SQ + 1; };
    d4m S = load16(stash + (size_t)slot_of(0) * NN, lane, false), St = load16(stash + (size_t)slot_of(0) * NN, lane, true);
#pragma unroll 1
    for (int t = 0; t < steps; ++t) {
        const d4m C = S, Ct = St;
        if (t + 1 < steps) {
            S = load16(stash + (size_t)slot_of(t + 1) * NN, lane, false);
            St = load16(stash + (size_t)slot_of(t + 1) * NN, lane, true);
        }
        if (t < SQ) {                                                  // E_{i+1} = E_i^2:  G <- G E^T + E^T G
            d4m g = mm16(Gt, Ct, zero16());                            //   G E^T      (left G: its transpose; right E^T)
            g = mm16(C, G, g);                                         // + E^T G      (left E^T: transpose E; right G)
            d4m gt = mm16(Ct, Gt, zero16());                           //   E G^T
            gt = mm16(G, C, gt);                                       // + G^T E
            G = g;
            Gt = gt;
        } else {                                                       // P_k = I + A_s P_{k+1} / k
            const double ik = 1.0 / (double)(t - SQ + 1);
            const d4m d = mm16(Gt, Ct, zero16());                      // G P_{k+1}^T
#pragma unroll
            for (int r = 0; r < 4; ++r) dA[r] = fma(d[r], ik, dA[r]);  // dA_s += G P_{k+1}^T / k
            const d4m g = mm16(As, G, zero16());                       // A_s^T G      (left A_s^T: transpose A_s)
            const d4m gt = mm16(G, As, zero16());                      // G^T A_s
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G[r] = g[r] * ik;
                Gt[r] = gt[r] * ik;
            }
        }
    }
    // the skew map needs dA - dA^T: the one transposition of the kernel, through 2 KB of LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) dAs[(i0 + 4 * r) * (N + 1) + j] = dA[r];
    __syncthreads();
    const double scale = ldexp(1.0, -SQ);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * r;
        double v = dA[r];
        if (skew) v = (j > i) ? (v - dAs[j * (N + 1) + i]) : 0.0;
        gX[i * N + j] = (T)(v * scale);
    }
}

static int g_expm_mfma = 1;      // test hook: 0 = the LDS kernels also at N = 16

static int expm_threads(int N) {
    int t = (N * N + 63) / 64 * 64;
    return t > 1024 ? 1024 : t;
}

template <typename T>
static int expm_fwd_impl(const void* X, int N, int skew, void* E, void* Ec, void* stash, void* stream) {
    FL_REQUIRE(X && (E || Ec) && stash, "matrix_exp: null pointer");
    FL_REQUIRE(N >= 1 && N <= 64, "matrix_exp: 1 <= N <= 64 (one workgroup, matrices in LDS)");
    const size_t lds = (size_t)3 * N * (N | 1) * sizeof(double);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&expm_fwd_kernel<T, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "matrix_exp LDS size");
        if (rc) return rc;
    }
    if (N == 16 && g_expm_mfma) {
        hipLaunchKernelGGL((expm16_fwd_kernel<T>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const T*)X, skew, (T*)E, (T*)Ec, (double*)stash);
        FL_CHECK_LAUNCH("matrix_exp");
        return FL_OK;
    }
#define FL_EXPM_FWD(NT_)                                                                                                   \
    hipLaunchKernelGGL((expm_fwd_kernel<T, NT_>), dim3(1), dim3(expm_threads(N)), lds, (hipStream_t)stream, (const T*)X, N, \
                       skew, (T*)E, (T*)Ec, (double*)stash)
    switch (N) {        // (the fixed sizes all fit the default 64 KB of dynamic LDS)
        case 4: FL_EXPM_FWD(4); break;
        case 8: FL_EXPM_FWD(8); break;
        case 16: FL_EXPM_FWD(16); break;
        case 32: FL_EXPM_FWD(32); break;
        default: FL_EXPM_FWD(0);
    }
#undef FL_EXPM_FWD
    FL_CHECK_LAUNCH("matrix_exp");
    return FL_OK;
}

template <typename T>
static int expm_bwd_impl(const void* gE, const void* gEc, int N, int skew, const void* stash, void* gX, void* stream) {
    FL_REQUIRE((gE || gEc) && stash && gX, "matrix_exp_bwd: null pointer");
    FL_REQUIRE(N >= 1 && N <= 64, "matrix_exp_bwd: 1 <= N <= 64");
    size_t lds = (size_t)5 * N * (N | 1) * sizeof(double);
    if (lds > 160 * 1024) lds = (size_t)4 * N * (N | 1) * sizeof(double);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&expm_bwd_kernel<T, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "matrix_exp_bwd LDS size");
        if (rc) return rc;
    }
    if (N == 16 && g_expm_mfma) {
        hipLaunchKernelGGL((expm16_bwd_kernel<T>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const T*)gE, (const T*)gEc, skew,
                           (const double*)stash, (T*)gX);
        FL_CHECK_LAUNCH("matrix_exp_bwd");
        return FL_OK;
    }
#define FL_EXPM_BWD(NT_)                                                                                                     \
    hipLaunchKernelGGL((expm_bwd_kernel<T, NT_>), dim3(1), dim3(expm_threads(N)), lds, (hipStream_t)stream, (const T*)gE,    \
                       (const T*)gEc, N, skew, (const double*)stash, (T*)gX)
    switch (N) {
        case 4: FL_EXPM_BWD(4); break;
        case 8: FL_EXPM_BWD(8); break;
        case 16: FL_EXPM_BWD(16); break;
        case 32: FL_EXPM_BWD(32); break;
        default: FL_EXPM_BWD(0);
    }
#undef FL_EXPM_BWD
    FL_CHECK_LAUNCH("matrix_exp_bwd");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_debug_set_expm_mfma(int on) {
    const int prev = g_expm_mfma;
    if (on >= 0) g_expm_mfma = on ? 1 : 0;
    return prev;
}
size_t fl_matrix_exp_stash_elems(int N) { return (size_t)EXPM_SLOTS * N * N + 1; }
int fl_matrix_exp_f32(const void* X, int N, int skew, void* E, void* stash, void* stream) {
    return expm_fwd_impl<float>(X, N, skew, E, nullptr, stash, stream);
}
int fl_matrix_exp_f64(const void* X, int N, int skew, void* E, void* stash, void* stream) {
    return expm_fwd_impl<double>(X, N, skew, E, nullptr, stash, stream);
}
int fl_matrix_exp_bwd_f32(const void* gE, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<float>(gE, nullptr, N, skew, stash, gX, stream);
}
int fl_matrix_exp_bwd_f64(const void* gE, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<double>(gE, nullptr, N, skew, stash, gX, stream);
}
int fl_matrix_exp_cplx_f32(const void* X, int N, int skew, void* E, void* stash, void* stream) {
    return expm_fwd_impl<float>(X, N, skew, nullptr, E, stash, stream);
}
int fl_matrix_exp_cplx_f64(const void* X, int N, int skew, void* E, void* stash, void* stream) {
    return expm_fwd_impl<double>(X, N, skew, nullptr, E, stash, stream);
}
int fl_matrix_exp_bwd_cplx_f32(const void* gE, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<float>(nullptr, gE, N, skew, stash, gX, stream);
}
int fl_matrix_exp_bwd_cplx_f64(const void* gE, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<double>(nullptr, gE, N, skew, stash, gX, stream);
}
int fl_matrix_exp_both_f32(const void* X, int N, int skew, void* E, void* Ec, void* stash, void* stream) {
    return expm_fwd_impl<float>(X, N, skew, E, Ec, stash, stream);
}
int fl_matrix_exp_both_f64(const void* X, int N, int skew, void* E, void* Ec, void* stash, void* stream) {
    return expm_fwd_impl<double>(X, N, skew, E, Ec, stash, stream);
}
int fl_matrix_exp_bwd_both_f32(const void* gE, const void* gEc, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<float>(gE, gEc, N, skew, stash, gX, stream);
}
int fl_matrix_exp_bwd_both_f64(const void* gE, const void* gEc, int N, int skew, const void* stash, void* gX, void* stream) {
    return expm_bwd_impl<double>(gE, gEc, N, skew, stash, gX, stream);
}
}
