// Bin-parallel closed-loop solve for system.Recursion (gfx950 / MI355X).
//
//   per bin f:  A_f = I - P[:,:,f]  (or P, or the conjugate transpose of either)
//               OUT[b,:,k,f] = A_f^{-1} R[b,:,k,f]   for every batch b and trailing column k
//
// replacing torch.linalg.solve(A, B) at flamo/processor/system.py:425.  The reference expands
// the SAME (M,N,N) matrix stack to the batch and LAPACK factors it B times; here each bin is
// factored once (LU, partial pivoting by |re|+|im| like LAPACK's icamax) and the factors are
// applied to all B*K right-hand sides.
//
// Mapping: NMAX = next power of two >= N lanes cooperate on one bin, one matrix ROW per lane
// held in registers (2*NMAX VGPRs for c64), so a 64-wide wavefront factors 64/NMAX bins at
// once.  Row exchanges are implicit (each lane remembers at which step its row was the
// pivot); the pivot row is broadcast with cross-lane shuffles -- no LDS, no atomics.  The
// kernel is vector-ALU bound (8/3 N^3 flop per bin vs ~8 N^2 bytes), the N x N update is a
// rank-1 update per step, not a dense tile contraction, so MFMA does not apply.
#include "common.h"

namespace fl {

template <typename T>
__device__ inline cx<T> shfl_cx(cx<T> v, int src, int width) {
    return cx<T>(__shfl(v.x, src, width), __shfl(v.y, src, width));
}

// Structured loop matrix of a feedback delay network: P[f] = diag(l[f]) U diag(r[f]) with a
// frequency-independent mixing matrix U (N x N) and per-bin (or constant, or absent) diagonal
// factors -- delays, attenuation filters -- on either side.  A = I - P is then built in
// registers from 2N values per bin instead of being streamed as N^2 values per bin.
template <typename T>
struct Dud {
    const cx<T>* l;   // element (n, f) at l[n*l_sn + f*l_sf]; nullptr = ones
    long l_sn, l_sf;
    const cx<T>* U;   // row-major N x N
    const cx<T>* r;
    long r_sn, r_sf;
};

// ---------------------------------------------------------------- DPP exchanges inside a 16-lane row
// The pivot search is a chain of dependent exchanges; through ds_bpermute each one is an LDS round
// trip (~100 cycles the wavefront sits in s_waitcnt: 286 of them were most of this kernel's time).
// Inside a DPP row the same exchanges are register moves at full VALU rate: mirror within 16 lanes,
// mirror within 8, quad permutes.
template <int CTRL>
__device__ inline float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ inline int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_MIRROR = 0x140;     // lane i <- lane 15-i of its 16-lane row
constexpr int DPP_HALF_MIRROR = 0x141;    // lane i <- lane 7-i of its 8-lane half row

template <int CTRL, typename T>
__device__ inline void argmax_dpp(T& bm, int& best) {
    const T om = dpp_mov<CTRL>(bm);
    const int ob = dpp_mov<CTRL>(best);
    // the lowest lane wins ties, as LAPACK's icamax; bitwise logic and selects, no branches
    const bool take = (om > bm) | ((om == bm) & (ob < best));
    bm = take ? om : bm;
    best = take ? ob : best;
}

// (largest |.|, its lane) over the NMAX lanes of a group, result in every lane of the group
template <int NMAX, typename T>
__device__ inline void group_argmax(T& bm, int& best) {
    if constexpr (NMAX <= 16) {
        if constexpr (NMAX >= 16) argmax_dpp<DPP_ROW_MIRROR>(bm, best);
        if constexpr (NMAX >= 8) argmax_dpp<DPP_HALF_MIRROR>(bm, best);
        if constexpr (NMAX >= 4) argmax_dpp<DPP_QUAD_XOR2>(bm, best);
        if constexpr (NMAX >= 2) argmax_dpp<DPP_QUAD_XOR1>(bm, best);
    } else {
#pragma unroll
        for (int off = NMAX / 2; off >= 1; off >>= 1) {
            const T om = __shfl_xor(bm, off, NMAX);
            const int ob = __shfl_xor(best, off, NMAX);
            if (om > bm || (om == bm && ob < best)) {
                bm = om;
                best = ob;
            }
        }
    }
}

template <typename T, int NMAX>
__global__ void __launch_bounds__(256) solve_kernel(
    const cx<T>* __restrict__ P, long p_pitch, Dud<T> dud, int one_minus, int adjoint,
    const cx<T>* __restrict__ R, long rs_b, long rs_n, long rs_k,
    cx<T>* __restrict__ OUT, long os_b, long os_n, long os_k,
    int B, int M, int N, int K) {
    constexpr int BPB = 256 / NMAX;
    const int gi = threadIdx.x % NMAX;
    const int f = blockIdx.x * BPB + threadIdx.x / NMAX;
    if (f >= M) return;  // whole lane group leaves together

    // ---- load (or build) this lane's row of A
    cx<T> row[NMAX];
    if (P) {
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            cx<T> v(0, 0);
            if (gi < N && j < N) {
                v = adjoint ? conj(P[((long)j * N + gi) * p_pitch + f]) : P[((long)gi * N + j) * p_pitch + f];
                if (one_minus) v = cx<T>(-v.x, -v.y);
            }
            if (one_minus ? (j == gi) : (j == gi && gi >= N)) v.x += (T)1;
            row[j] = v;
        }
    } else {
        // A[i][j] = delta_ij - l_i U_ij r_j ;  A^H[i][j] = delta_ij - conj(l_j U_ji r_i)
        const cx<T> one(1, 0);
        cx<T> lv = one, rv = one;
        if (gi < N) {
            if (dud.l) lv = dud.l[(long)gi * dud.l_sn + (long)f * dud.l_sf];
            if (dud.r) rv = dud.r[(long)gi * dud.r_sn + (long)f * dud.r_sf];
        }
        const cx<T> own = adjoint ? rv : lv;       // the factor indexed by this lane's row
        const cx<T> oth = adjoint ? lv : rv;       // the factor indexed by the column (from lane j)
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            const cx<T> oj = shfl_cx(oth, j, NMAX);
            cx<T> v(0, 0);
            if (gi < N && j < N) {
                const cx<T> u = adjoint ? dud.U[(long)j * N + gi] : dud.U[(long)gi * N + j];
                v = own * u * oj;
                if (adjoint) v = conj(v);
                v = cx<T>(-v.x, -v.y);
            }
            if (j == gi) v.x += (T)1;
            row[j] = v;
        }
    }

    // ---- LU with implicit partial pivoting
    int my_step = (gi < N) ? -1 : NMAX + gi;  // step at which this lane's row became the pivot
    int pl[NMAX];                              // pivot lane of each step (uniform in the group)
    cx<T> dinv(0, 0);                          // reciprocal of this lane's pivot
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        pl[k] = 0;
        if (k < N) {
            T bm = (my_step < 0) ? (fabs(row[k].x) + fabs(row[k].y)) : (T)-1;
            int best = gi;
            group_argmax<NMAX>(bm, best);
            pl[k] = best;
            const cx<T> piv = shfl_cx(row[k], best, NMAX);
            const cx<T> inv = crecip(piv);
            const bool elim = (my_step < 0) && (gi != best);
            const cx<T> l = elim ? row[k] * inv : cx<T>(0, 0);
            // no `j < N` guard: padded columns are zero and stay zero, and without per-element branches
            // the NMAX-k-1 broadcasts of the pivot row are issued back to back behind ONE wait
            // (constant trip count: a bound that depends on k is not unrolled)
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                if (j > k) {
                    const cx<T> pr = shfl_cx(row[j], best, NMAX);
                    row[j] = row[j] - l * pr;
                }
            }
            // selects, not branches: a divergent `if` costs an exec-mask save/restore and a branch each
            row[k].x = elim ? l.x : row[k].x;
            row[k].y = elim ? l.y : row[k].y;
            const bool mine = (gi == best);
            my_step = mine ? k : my_step;
            dinv.x = mine ? inv.x : dinv.x;
            dinv.y = mine ? inv.y : dinv.y;
        }
    }

    // ---- apply to every right-hand side
    const int ncols = B * K;
    for (int col = 0; col < ncols; ++col) {
        const int b = col / K, kk = col - b * K;
        cx<T> y(0, 0);
        if (gi < N) y = R[(long)b * rs_b + (long)gi * rs_n + (long)kk * rs_k + f];
        // forward substitution with the stored multipliers
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            if (k < N) {
                const cx<T> yp = shfl_cx(y, pl[k], NMAX);
                const cx<T> t = row[k] * yp;
                const bool on = my_step > k && my_step < NMAX;
                y.x -= on ? t.x : (T)0;
                y.y -= on ? t.y : (T)0;
            }
        }
        // back substitution: lane pl[k] finishes x_k, the earlier pivots subtract U[.,k] x_k
#pragma unroll
        for (int k = NMAX - 1; k >= 0; --k) {
            if (k < N) {
                const cx<T> yd = y * dinv;
                y.x = (my_step == k) ? yd.x : y.x;
                y.y = (my_step == k) ? yd.y : y.y;
                const cx<T> xk = shfl_cx(y, pl[k], NMAX);
                const cx<T> t = row[k] * xk;
                y.x -= (my_step < k) ? t.x : (T)0;
                y.y -= (my_step < k) ? t.y : (T)0;
            }
        }
        if (gi < N) OUT[(long)b * os_b + (long)my_step * os_n + (long)kk * os_k + f] = y;
    }
}

template <typename T, int NMAX>
static int launch_solve(const void* P, long p_pitch, const Dud<T>& dud, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k,
                        void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, hipStream_t st) {
    constexpr int BPB = 256 / NMAX;
    dim3 grid(cdiv_i(M, BPB));
    hipLaunchKernelGGL((solve_kernel<T, NMAX>), grid, dim3(256), 0, st, (const cx<T>*)P, p_pitch, dud, one_minus, adjoint,
                       (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K);
    FL_CHECK_LAUNCH("solve");
    return FL_OK;
}

template <typename T>
static int solve_impl(const void* P, long p_pitch, const Dud<T>& dud, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k,
                      void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE((P || dud.U) && R && OUT, "solve: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0 && (!P || p_pitch >= M), "solve: bad sizes (p_pitch >= M)");
    const int nmax_lim = sizeof(T) == 8 ? 32 : 64;
    if (N > nmax_lim) {
        set_error("solve: N=%d exceeds the register-resident limit (%d) for this precision", N, nmax_lim);
        return FL_ERR_UNSUPPORTED;
    }
    if (B == 0 || M == 0) return FL_OK;
    hipStream_t st = (hipStream_t)stream;
#define FL_SOLVE(NM) return launch_solve<T, NM>(P, p_pitch, dud, one_minus, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, st)
    if (N <= 4) FL_SOLVE(4);
    if (N <= 8) FL_SOLVE(8);
    if (N <= 16) FL_SOLVE(16);
    if (N <= 32) FL_SOLVE(32);
    if constexpr (sizeof(T) == 4) FL_SOLVE(64);
#undef FL_SOLVE
    return FL_ERR_UNSUPPORTED;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_solve_c64(const void* P, long p_pitch, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                 long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    Dud<float> none = {};
    return solve_impl<float>(P, p_pitch, none, one_minus, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_c128(const void* P, long p_pitch, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                  long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    Dud<double> none = {};
    return solve_impl<double>(P, p_pitch, none, one_minus, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_dud_c64(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, int adjoint,
                     const void* R, long rs_b, long rs_n, long rs_k, void* OUT, long os_b, long os_n, long os_k,
                     int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(U, "solve_dud: null mixing matrix");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf};
    return solve_impl<float>(nullptr, 0, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_dud_c128(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, int adjoint,
                      const void* R, long rs_b, long rs_n, long rs_k, void* OUT, long os_b, long os_n, long os_k,
                      int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(U, "solve_dud: null mixing matrix");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf};
    return solve_impl<double>(nullptr, 0, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
}
