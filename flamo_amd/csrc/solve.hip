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

#include <type_traits>

namespace fl {

// kept factors, tiled by the BPB bins of a workgroup: element (s, k) of bin f; pivot row s of bin f
__host__ __device__ inline long kept_lu_index(int f, int s, int k, int N, int BPB) { return (((long)(f / BPB) * N + s) * BPB + f % BPB) * N + k; }
__host__ __device__ inline long kept_piv_index(int f, int s, int N, int BPB) { return ((long)(f / BPB) * N + s) * BPB + f % BPB; }

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
    // A second left factor, l = l (.) l2 (the feedforward path's diagonal -- the delays of a feedback delay network --
    // beside the feedback path's attenuation: their product is never formed), and with rhs_l2 the right-hand side is
    // l2 (.) R as well (R = feedforward(X) is that same diagonal applied to the input): both as loads in the kernel
    // instead of two launches forward and two backward.  Forward system only for rhs_l2 (the adjoint's RHS is a gradient).
    const cx<T>* l2;
    long l2_sn, l2_sf;
    int rhs_l2;
    // Rank-one right-hand side and contracted output (ops.fdn_core; one column per batch item, in-place kernels only):
    // R_i[b][f] = rv_i rs[b][f] (rv conjugated for the adjoint system) -- the input-gain column times the scalar input
    // spectrum, or conj(output-gain row) times the output's gradient -- and, forward system, z[b][f] = sum_i cw_i OUT_i
    // beside OUT (the output-gain row applied in the wavefront).  rv / cw: N values, real (T) or complex.
    const void* rv;
    const cx<T>* rs;
    long rs_sb;
    const void* cw;
    cx<T>* cz;
    long cz_sb;
    int rv_real, cw_real;
    // Kept factors (fl_solve_scaled_keep_*: the shuffle kernel only): the LU factors as they stand in the lanes' registers after
    // the elimination (L below the diagonal, unit diagonal implied, U on and above it) in pivot order, and piv = the row of A that
    // became each pivot row, TILED by the workgroup's bins: element (s, k) of bin f at kept_lu_index = [f / BPB][s][f % BPB][k], a
    // workgroup's BPB = 256 / NMAX bins form one contiguous N^2 BPB block.  The forward kernel's lane stores ITS ROW as one
    // contiguous run (8 N bytes: whole cache lines, 16-byte stores); the adjoint kernel's group of N lanes reads N consecutive
    // values per stored row s -- the transposition the adjoint needs is this choice of the innermost index.  (A planar layout
    // left a workgroup 8 N^2 separate 64-byte pieces: measured 2.2 TB/s on the stores, 1.2 TB/s on the adjoint's loads.)  The backward pass's adjoint system then costs a substitution over N^2 stored values
    // per bin (fl_solve_kept_adjoint_*) instead of a second factorisation: at N = 32 the elimination is 90 % of the solve,
    // and the part has 288 GB to keep 8 N^2 bytes per bin in.
    cx<T>* lu_out;
    int* piv_out;
    // The adjoint system's solution for the output-gain row, w[f] = A[f]^-H cw^H, from the FORWARD launch's factors (float32,
    // 4 < N <= 16: fl_solve_fdn_wadj_c64): the backward pass of a network with ONE output channel has the right-hand side
    // cw^H gy[b][f] -- the same vector times a scalar per (batch item, bin) -- so its solution is w[f] gy[b][f] and the backward
    // needs no solve at all.  The transposed substitutions run as dot products over the lanes (row p of the factors sits in
    // the lane it was eliminated in: column access is a reduction over the group).  Element (n, f) at wadj[n*wadj_sn + f].
    cx<T>* wadj;
    long wadj_sn;
    // ... and as an INPUT of the gradient kernel (fl_solve_dud2_grads_w_*): gR[b][n][f] = wadj[n][f] wgy[b][f] is formed where it
    // is consumed -- the adjoint solution never exists in memory.
    const cx<T>* wgy;
    long wgy_sb;
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

template <typename T>
__device__ inline cx<T> gain_at(const void* v, int is_real, int i) {
    return is_real ? cx<T>(reinterpret_cast<const T*>(v)[i], 0) : reinterpret_cast<const cx<T>*>(v)[i];
}
// sum over the LANES consecutive lanes of a group (4, 8 or 16: inside one DPP row)
template <int LANES, typename T>
__device__ inline T group_sum(T v) {
    v += dpp_mov<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                                   // quad_perm [2,3,0,1]
    if constexpr (LANES >= 8) v += dpp_mov<0x141>(v);        // row_half_mirror
    if constexpr (LANES >= 16) v += dpp_mov<0x140>(v);       // row_mirror
    return v;
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
        // every load unconditional, all in flight together: lanes and columns beyond N read a valid element and discard it (a
        // guard per element made every load a branch with its own wait: N serial round trips per wavefront in front of the
        // elimination -- a quarter of this kernel's time at N = 32, M = 192001)
        const int gr = gi < N ? gi : N - 1;
        const cx<T>* p0 = adjoint ? P + (long)gr * p_pitch + f : P + (long)gr * N * p_pitch + f;
        const long step = adjoint ? (long)N * p_pitch : p_pitch;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) row[j] = p0[(long)(j < N ? j : 0) * step];
        // (uniform) constant row scale of the materialised matrix, A = I - diag(l) P: the lane's own entry, or -- adjoint: the
        // row of P is the column j -- a wavefront-uniform (scalar) load per column
        cx<T> lrow(1, 0);
        if (dud.l && !adjoint) lrow = dud.l[(long)gr * dud.l_sn];
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            cx<T> v = adjoint ? conj(row[j]) : row[j];
            if (dud.l) v = v * (adjoint ? conj(dud.l[(long)(j < N ? j : 0) * dud.l_sn]) : lrow);
            if (one_minus) v = cx<T>(-v.x, -v.y);
            const bool keep = gi < N && j < N;
            v = cx<T>(keep ? v.x : (T)0, keep ? v.y : (T)0);
            if (one_minus ? (j == gi) : (j == gi && gi >= N)) v.x += (T)1;
            row[j] = v;
        }
    } else {
        // A[i][j] = delta_ij - l_i U_ij r_j ;  A^H[i][j] = delta_ij - conj(l_j U_ji r_i)
        const cx<T> one(1, 0);
        cx<T> lv = one, rv = one;
        if (gi < N) {
            if (dud.l) lv = dud.l[(long)gi * dud.l_sn + (long)f * dud.l_sf];
            if (dud.l2) lv = lv * dud.l2[(long)gi * dud.l2_sn + (long)f * dud.l2_sf];
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

    if (dud.lu_out && gi < N) {      // (uniform) the factors and the pivot order are kept for the adjoint system (see Dud)
        cx<T>* lo = dud.lu_out + kept_lu_index(f, my_step, 0, N, BPB);
        if ((N & 1) == 0 || sizeof(T) == 8) {      // (uniform) 16-byte stores: two float values / one double value each
            typedef T v4 __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int PER = 16 / sizeof(cx<T>);
#pragma unroll
            for (int k = 0; k < NMAX; k += PER) {
                if (k < N) {
                    v4 q;
                    if constexpr (PER == 2) q = v4{row[k].x, row[k].y, row[k + 1].x, row[k + 1].y};
                    else q = v4{row[k].x, row[k].y};
                    *reinterpret_cast<v4*>(lo + k) = q;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
                if (k < N) lo[k] = row[k];
        }
        dud.piv_out[kept_piv_index(f, my_step, N, BPB)] = gi;
    }

    // ---- apply to every right-hand side
    const int ncols = B * K;
    for (int col = 0; col < ncols; ++col) {
        const int b = col / K, kk = col - b * K;
        cx<T> y(0, 0);
        if (gi < N) {
            y = R[(long)b * rs_b + (long)gi * rs_n + (long)kk * rs_k + f];
            if (dud.rhs_l2 && !adjoint) y = y * dud.l2[(long)gi * dud.l2_sn + (long)f * dud.l2_sf];
        }
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

// ---------------------------------------------------------------- adjoint system from kept factors
// P A = L U as solve_kernel left it (rows in pivot order: Dud::lu_out, piv[s] = the row of A that is pivot row s).  A^H x = b is
// U^H L^H (P x) = b: lane j of a bin's group holds COLUMN j of the stored factors -- the planar layout makes the transposition a
// choice of plane at load time -- i.e. row j of U^H (lower triangular) and of L^H (unit upper triangular), and both
// substitutions are the outer-product form of the forward kernel: the lane that finishes an unknown broadcasts it, the others
// subtract their stored entry times it.  x[piv[s]] = v[s] on the way out.  HBM-bound: 8 N^2 bytes per bin, once.
template <typename T, int NMAX>
__global__ void __launch_bounds__(256) solve_kept_adjoint_kernel(const cx<T>* __restrict__ LU, const int* __restrict__ piv, int TB,
                                                                 const cx<T>* __restrict__ R, long rs_b, long rs_n,
                                                                 long rs_k, const void* __restrict__ rv, int rv_real,
                                                                 const cx<T>* __restrict__ rsig, long rsig_sb,
                                                                 cx<T>* __restrict__ OUT, long os_b, long os_n, long os_k,
                                                                 int B, int M, int N, int K) {
    // TB: bins per tile of the kernel that stored the factors (its workgroup's bins: kept_lu_index)
    // rv / rsig: rank-one right-hand side conj(rv[j]) rsig[b][f] (the output-gain row times the output's gradient: ops.fdn_core), or R
    constexpr int BPB = 256 / NMAX;
    const int gi = threadIdx.x % NMAX;
    const int f = blockIdx.x * BPB + threadIdx.x / NMAX;
    if (f >= M) return;      // whole lane group leaves together
    const bool on = gi < N;
    cx<T> col[NMAX];         // col[s] = conj(factor[s][gi]): entry (gi, s) of U^H (s <= gi) / L^H (s > gi)
    {   // every load unconditional and in flight together (lanes / rows beyond N read a valid element and discard it: a guard per
        // load made each one a branch with its own wait -- 32 serial round trips per wavefront, 1.06 ms at N = 32, M = 192001)
        const cx<T>* base = LU + kept_lu_index(f, 0, on ? gi : 0, N, TB);
        const long sstride = (long)TB * N;
#pragma unroll
        for (int s = 0; s < NMAX; ++s) col[s] = base[(s < N ? s : 0) * sstride];
#pragma unroll
        for (int s = 0; s < NMAX; ++s) {
            const bool keep = on && s < N;
            col[s] = cx<T>(keep ? col[s].x : (T)0, keep ? -col[s].y : (T)0);
        }
    }
    const int dst = on ? piv[kept_piv_index(f, gi, N, TB)] : 0;
    cx<T> gvc(0, 0);
    if (rv) gvc = conj(gain_at<T>(rv, rv_real, on ? gi : 0));
    cx<T> dinv(0, 0);
#pragma unroll
    for (int s = 0; s < NMAX; ++s) {      // the lane's own diagonal of U^H (a select chain: the register index is the lane)
        dinv.x = (gi == s) ? col[s].x : dinv.x;
        dinv.y = (gi == s) ? col[s].y : dinv.y;
    }
    dinv = on ? crecip(dinv) : cx<T>(0, 0);
    const int ncols = B * K;
    for (int c = 0; c < ncols; ++c) {
        const int b = c / K, kk = c - b * K;
        cx<T> y(0, 0);
        if (rv) y = gvc * rsig[(long)b * rsig_sb + f];
        else y = R[(long)b * rs_b + (long)(on ? gi : 0) * rs_n + (long)kk * rs_k + f];
        y = cx<T>(on ? y.x : (T)0, on ? y.y : (T)0);
        // U^H z = b: z_s = b_s / conj(U[s][s]) in lane s; lanes j > s subtract conj(U[s][j]) z_s
#pragma unroll
        for (int s = 0; s < NMAX; ++s) {
            if (s < N) {
                const cx<T> yd = y * dinv;
                y.x = (gi == s) ? yd.x : y.x;
                y.y = (gi == s) ? yd.y : y.y;
                const cx<T> zs = shfl_cx(y, s, NMAX);
                const cx<T> t = col[s] * zs;
                y.x -= (gi > s) ? t.x : (T)0;
                y.y -= (gi > s) ? t.y : (T)0;
            }
        }
        // L^H v = z (unit diagonal): v_s = z_s in lane s; lanes j < s subtract conj(L[s][j]) v_s
#pragma unroll
        for (int s = NMAX - 1; s >= 1; --s) {
            if (s < N) {
                const cx<T> vs = shfl_cx(y, s, NMAX);
                const cx<T> t = col[s] * vs;
                y.x -= (gi < s) ? t.x : (T)0;
                y.y -= (gi < s) ? t.y : (T)0;
            }
        }
        if (on) OUT[(long)b * os_b + (long)dst * os_n + (long)kk * os_k + f] = y;
    }
}

// ---------------------------------------------------------------- N <= 16: rows in place, DPP broadcasts
// The kernel above is bound by vector-ALU issue (a wave64 instruction occupies its 16-lane SIMD for four
// cycles; ~3500 of them per 4 bins at N = 16), and a third of those instructions only exist because a
// row stays in the lane it was loaded into: the pivot lane is data, so every broadcast of the pivot row is
// a ds_bpermute behind a wait, and the (value, lane) arg-max is five instructions per exchange.  Here rows
// are PHYSICALLY exchanged -- after step k the pivot row sits in lane k -- which makes every broadcast
// source a compile-time lane: one DPP move (row_newbcast:k for 16-lane groups, quad permutes for 4 and 8).
// The exchange itself (2 N bpermutes) is paid only when needed: threshold pivoting keeps the diagonal
// entry whenever it is within a factor 2 of the column maximum (growth is bounded as with partial pivoting,
// by 3^(N-1) instead of 2^(N-1) in theory and indistinguishable in practice), and I - P of a damped loop is
// nowhere near that.  The pivot search is one integer max per exchange: the magnitude's bit pattern with the
// (inverted) lane number in its low bits.
template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// lanes of the 4-lane banks selected by BANKS take the moved value, the others keep their own
template <int CTRL, int BANKS>
__device__ inline int dpp_merge(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, BANKS, false);
}

// the value lane K of each NMAX-lane group holds, in every lane of the group
template <int NMAX, int K>
__device__ inline int group_bcast(int v) {
    static_assert(NMAX == 4 || NMAX == 8 || NMAX == 16, "DPP broadcasts stay inside a 16-lane row");
    if constexpr (NMAX == 16) {
        return dpp_mov<0x150 + K>(v);                          // row_newbcast:K
    } else if constexpr (NMAX == 4) {
        return dpp_mov<K * 0x55>(v);                           // quad_perm:[K,K,K,K]
    } else {
        const int t = dpp_mov<(K & 3) * 0x55>(v);              // every quad: its lane K%4 ...
        if constexpr (K < 4) return dpp_merge<0x114, 0xA>(t);  // ... row_shr:4 into quads 1 and 3
        else return dpp_merge<0x104, 0x5>(t);                  // ... row_shl:4 into quads 0 and 2
    }
}
template <int NMAX, int K>
__device__ inline float group_bcast(float v) {
    return __builtin_bit_cast(float, group_bcast<NMAX, K>(__builtin_bit_cast(int, v)));
}
template <int NMAX, int K>
__device__ inline double group_bcast(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = group_bcast<NMAX, K>((int)(b & 0xffffffffLL));
    const int hi = group_bcast<NMAX, K>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int NMAX, int K, typename T>
__device__ inline cx<T> group_bcast(cx<T> v) {
    return cx<T>(group_bcast<NMAX, K>(v.x), group_bcast<NMAX, K>(v.y));
}

// reciprocals from the hardware estimate (an IEEE division is ~10 instructions, and the pivot reciprocal is on
// the critical path of every elimination step): v_rcp_f32 is accurate to 1 ulp as it is; v_rcp_f64 takes two
// Newton steps
__device__ inline float rcp_est(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double rcp_est(double x) { return __builtin_amdgcn_rcp(x); }
__device__ inline float rcp_full(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double rcp_full(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}
// 1 / b, scaled by ~1/max(|re|,|im|) against overflow of |b|^2; the scale cancels exactly whatever its rounding,
// so the raw estimate serves for it
template <typename T>
__device__ inline cx<T> crecip_fast(cx<T> b) {
    const T is = rcp_est(fmax(fabs(b.x), fabs(b.y)));
    const T x = b.x * is, y = b.y * is;
    const T d = is * rcp_full(x * x + y * y);
    return cx<T>(x * d, -y * d);
}

// acc - l * p.  float: two packed FMAs, (-l.x,-l.x)*(p.x,p.y) and (l.y,-l.y)*(p.y,p.x) -- the broadcast and the swap
// are operand selectors of v_pk_fma_f32, the two multiplier pairs are formed once per elimination step
typedef float f2v __attribute__((ext_vector_type(2)));
template <typename T> struct CMul { T lx, ly; };
template <> struct CMul<float> { f2v nlxx, lyny; };
__device__ inline CMul<float> cmul_of(cx<float> l) { return {f2v{-l.x, -l.x}, f2v{l.y, -l.y}}; }
__device__ inline CMul<double> cmul_of(cx<double> l) { return {l.x, l.y}; }
__device__ inline cx<float> cfnma(cx<float> acc, const CMul<float>& m, cx<float> p) {
    f2v a = {acc.x, acc.y};
    a = __builtin_elementwise_fma(m.nlxx, f2v{p.x, p.y}, a);
    a = __builtin_elementwise_fma(m.lyny, f2v{p.y, p.x}, a);
    return cx<float>(a.x, a.y);
}
__device__ inline cx<double> cfnma(cx<double> acc, const CMul<double>& m, cx<double> p) {
    return cx<double>(fma(m.ly, p.y, fma(-m.lx, p.x, acc.x)), fma(-m.ly, p.x, fma(-m.lx, p.y, acc.y)));
}
// compiler barrier on a register-resident complex value, as ONE 64-bit operand per component pair (separate
// 32-bit operands would split the register pairs the packed instructions need)
__device__ inline void opaque(cx<float>& v) {
    unsigned long long t = __builtin_bit_cast(unsigned long long, v);
    asm("" : "+v"(t));
    v = __builtin_bit_cast(cx<float>, t);
}
__device__ inline void opaque(cx<double>& v) { asm("" : "+v"(v.x), "+v"(v.y)); }

// the constant 1 in device memory: what an absent diagonal factor is read as (one load path, no branch per factor)
__device__ double kOneRe[2] = {1.0, 0.0};
__device__ float kOneRef[2] = {1.f, 0.f};
template <typename T> __device__ inline const cx<T>* one_ptr() {
    if constexpr (sizeof(T) == 8) return reinterpret_cast<const cx<T>*>(kOneRe);
    else return reinterpret_cast<const cx<T>*>(kOneRef);
}

// LANES lanes per bin, RPL rows per lane: row s*LANES + g of A lives in slot s of lane g, NMAX = LANES * RPL.
// RPL = 1 is the layout described above (N <= 16).  RPL = 2 (16 < N <= 32, float) keeps the 16-lane DPP row as the
// group -- a 32-lane group has no one-instruction broadcast -- and every broadcast of a pivot-row element now
// feeds two row updates; which slot holds the pivot (K / LANES) and which slots still have rows below it are
// known at compile time, so finished slots drop out of the update loops altogether.
template <typename T, int LANES, int RPL, bool WADJ = false>
__global__ void __launch_bounds__(256) solve_inplace_kernel(
    const cx<T>* __restrict__ P, long p_pitch, Dud<T> dud, int one_minus, int adjoint,
    const cx<T>* __restrict__ R, long rs_b, long rs_n, long rs_k,
    cx<T>* __restrict__ OUT, long os_b, long os_n, long os_k,
    int B, int M, int N, int K) {
    constexpr int NMAX = LANES * RPL, BPB = 256 / LANES;
    const int thr_steps = (adjoint >> 8) & 0xFF;     // tuning: exponent steps of the pivot threshold (host passes >= 1)
    const bool no_prefetch = (adjoint >> 16) & 1;    // tuning (fl_debug_set_solve_variant(5)): the right-hand side is fetched behind the elimination
    adjoint &= 1;
    const int gi = threadIdx.x % LANES;
    const int f = blockIdx.x * BPB + threadIdx.x / LANES;
    // the frequency-independent mixing matrix, zero-padded to NMAX x NMAX (and transposed for the adjoint
    // system), staged once per workgroup: the row build below reads it with compile-time offsets, no guards
    __shared__ cx<T> Us[NMAX * NMAX];
    // ---- factored loop: EVERYTHING this lane reads from global memory is requested here, unconditionally and together -- the two
    // diagonal factors of its rows, the first right-hand side, the output-gain entries -- in front of the mixing matrix's staging
    // barrier.  Lanes beyond M and rows beyond N read a valid element and discard it; the right-hand side is fetched for the row
    // the slot holds BEFORE pivoting (threshold pivoting keeps the diagonal for every damped loop: a wavefront in which a row did
    // move fetches again, below).  With a guard per element every load was a branch with its own wait: ~11 serial round trips
    // per wavefront against ~4 us of arithmetic, two wavefronts per SIMD to hide them.
    // (double: the right-hand side and the output gains stay behind the elimination -- their registers would take the kernel from
    // two wavefronts per SIMD to one: 0.41 -> 0.49 ms on the float64 FDN step)
    constexpr bool kPrefetchRhs = sizeof(T) == 4;
    cx<T> pre_l[RPL], pre_l2[RPL], pre_r[RPL], pre_y[RPL], pre_cw[RPL];
    if (!P) {
        const cx<T>* one = one_ptr<T>();
        const int fc = f < M ? f : M - 1;
        const cx<T>* lp = dud.l ? dud.l : one;
        const long l_sn = dud.l ? dud.l_sn : 0, l_sf = dud.l ? dud.l_sf : 0;
        const cx<T>* l2p = dud.l2 ? dud.l2 : one;
        const long l2_sn = dud.l2 ? dud.l2_sn : 0, l2_sf = dud.l2 ? dud.l2_sf : 0;
        const cx<T>* rp = dud.r ? dud.r : one;
        const long r_sn = dud.r ? dud.r_sn : 0, r_sf = dud.r ? dud.r_sf : 0;
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int ri = s * LANES + gi, rc = ri < N ? ri : N - 1;
            pre_l[s] = lp[(long)rc * l_sn + (long)fc * l_sf];
            pre_l2[s] = l2p[(long)rc * l2_sn + (long)fc * l2_sf];
            pre_r[s] = rp[(long)rc * r_sn + (long)fc * r_sf];
            if constexpr (kPrefetchRhs) {
                if (dud.rv) {        // (uniform) rank-one right-hand side: the gain entry; the scalar signal's value is lane-independent
                    pre_y[s] = gain_at<T>(dud.rv, dud.rv_real, rc);
                } else {
                    pre_y[s] = R[(long)rc * rs_n + fc];                      // column 0: b = 0, kk = 0
                }
                pre_cw[s] = (dud.cz || dud.wadj) ? gain_at<T>(dud.cw, dud.cw_real, rc) : cx<T>(0, 0);
            }
        }
    }
    cx<T> pre_rs(0, 0);
    if (kPrefetchRhs && !P && dud.rv) pre_rs = dud.rs[f < M ? f : M - 1];                // b = 0
    if (!P) {
        for (int e = threadIdx.x; e < NMAX * NMAX; e += 256) {
            const int i = e / NMAX, j = e % NMAX;
            cx<T> u(0, 0);
            if (i < N && j < N) u = adjoint ? conj(dud.U[(long)j * N + i]) : dud.U[(long)i * N + j];
            Us[e] = u;
        }
        __syncthreads();
    }
    cx<T> row[RPL][NMAX];
    if constexpr (LANES == 16) {
        if (P) {
            // A materialised P is read through LDS: with a row per lane a load instruction touches 16 planes x 4
            // bins (32-byte pieces of planes ~1 MB apart).  Here the 256 threads fetch CR matrix rows (columns for
            // the adjoint system) of the block's 16 bins with 16 consecutive threads on one plane (128 bytes), park
            // them in LDS and the four lanes that own those rows pick them up.  Every thread stays for the barriers.
            constexpr int CR = 4, NLD = CR * NMAX / 16;
            __shared__ cx<T> Pt[CR][NMAX][17];
            const int lb = threadIdx.x % 16;
            const int fl = min(blockIdx.x * BPB + lb, M - 1);
            const T sgn = one_minus ? (T)-1 : (T)1;
            // every load of the block is requested before the first is consumed (the row registers are still empty, so
            // there is room): one memory latency per workgroup instead of one per pass
            cx<T> pre[NMAX / CR][NLD];
            static_for<0, NMAX / CR>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = threadIdx.x / 16 + 16 * i;          // (row in chunk, column)
                    const int pr = e / NMAX, pj = e % NMAX, ri = c * CR + pr;
                    cx<T> x(0, 0);
                    if (ri < N && pj < N) {
                        x = P[(adjoint ? (long)pj * N + ri : (long)ri * N + pj) * p_pitch + fl];
                        if (dud.l) x = x * dud.l[(long)(adjoint ? pj : ri) * dud.l_sn];      // row scale (row of P)
                        x = cx<T>(sgn * x.x, (adjoint ? -sgn : sgn) * x.y);
                    }
                    if (one_minus ? (pj == ri) : (pj == ri && ri >= N)) x.x += (T)1;
                    pre[c][i] = x;
                }
            });
            static_for<0, NMAX / CR>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = threadIdx.x / 16 + 16 * i;
                    Pt[e / NMAX][e % NMAX][lb] = pre[c][i];
                }
                __syncthreads();
                constexpr int S = (c * CR) / LANES;                   // the slot these rows live in
                if ((gi >> 2) == ((c * CR) % LANES) / 4) {
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) row[S][j] = Pt[gi & 3][j][threadIdx.x / 16];
                }
                __syncthreads();
            });
        }
    }
    if (f >= M) return;  // whole lane group leaves together; no exchange below crosses a group

    // ---- load (or build) this lane's rows of A; rows and columns >= N are those of the identity
    if (P) {
      if constexpr (LANES != 16) {
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int ri = s * LANES + gi;
            const int gr = min(ri, N - 1);                   // padded rows read a valid row and discard it
            const cx<T>* p = adjoint ? P + (long)gr * p_pitch + f : P + (long)gr * N * p_pitch + f;
            const long step = adjoint ? (long)N * p_pitch : p_pitch;
            const T sgn = one_minus ? (T)-1 : (T)1;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                cx<T> v(0, 0);
                if (j < N) {                                 // uniform
                    v = p[(long)j * step];
                    if (dud.l) v = v * dud.l[(long)(adjoint ? j : gr) * dud.l_sn];           // row scale (row of P)
                    v = cx<T>(sgn * v.x, (adjoint ? -sgn : sgn) * v.y);
                    v.x = ri < N ? v.x : (T)0;
                    v.y = ri < N ? v.y : (T)0;
                }
                if (one_minus ? (j == ri) : (j == ri && ri >= N)) v.x += (T)1;
                row[s][j] = v;
            }
        }
      }
    } else {
        // A[i][j] = delta_ij - l_i U_ij r_j ;  A^H[i][j] = delta_ij - conj(r_i) conj(U_ji) conj(l_j)
        cx<T> own[RPL], oth[RPL];
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const bool real_row = s * LANES + gi < N;
            cx<T> lv = pre_l[s] * pre_l2[s], rv = pre_r[s];        // (absent factors were read as ones)
            lv = cx<T>(real_row ? lv.x : (T)1, real_row ? lv.y : (T)0);
            rv = cx<T>(real_row ? rv.x : (T)1, real_row ? rv.y : (T)0);
            own[s] = adjoint ? conj(rv) : lv;
            oth[s] = adjoint ? conj(lv) : rv;
        }
        static_for<0, NMAX>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const cx<T> oj = group_bcast<LANES, J % LANES>(oth[J / LANES]);
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const int ri = s * LANES + gi;
                const cx<T> v = own[s] * Us[ri * NMAX + J] * oj;
                row[s][J] = cx<T>((J == ri ? (T)1 : (T)0) - v.x, -v.y);
            }
        });
    }

    // ---- LU, threshold partial pivoting, rows exchanged in place
    int orig[RPL];          // which row of A (= which entry of the right-hand side) each slot holds
    cx<T> dinv[RPL];        // reciprocal of each slot's pivot
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        orig[s] = s * LANES + gi;
        dinv[s] = cx<T>(0, 0);
    }
    static_for<0, NMAX>([&](auto kc) {
        constexpr int KK = decltype(kc)::value;
        constexpr int SK = KK / LANES, LK = KK % LANES;       // slot and lane of the diagonal
        // steps >= N act on identity rows: skipped (uniform branch).  For RPL > 1 the branch also keeps the compiler
        // from interleaving neighbouring steps, which is what drove that kernel's register count
        if (RPL > 1 && KK >= N) return;
        if constexpr (KK < NMAX - 1) {
            // non-negative floats order like their bit patterns; the low bits carry NMAX-1-row so that the lowest
            // row wins among (nearly) equal magnitudes, as icamax would pick.  Rows above K are no candidates:
            // whole slots below SK, and the lanes below LK of slot SK.
            int dkey = 0, kmax = (int)0x80000000;
#pragma unroll
            for (int s = SK; s < RPL; ++s) {
                const int ri = s * LANES + gi;
                float mag = (float)(fabs(row[s][KK].x) + fabs(row[s][KK].y));
                if (s == SK) mag = (gi >= LK) ? mag : -1.0f;
                const int key = (__builtin_bit_cast(int, mag) & ~(NMAX - 1)) | (NMAX - 1 - ri);
                if (s == SK) dkey = key;
                kmax = max(kmax, key);
            }
            if constexpr (LANES >= 16) kmax = max(kmax, dpp_mov<DPP_ROW_MIRROR>(kmax));
            if constexpr (LANES >= 8) kmax = max(kmax, dpp_mov<DPP_HALF_MIRROR>(kmax));
            kmax = max(kmax, dpp_mov<DPP_QUAD_XOR2>(kmax));
            kmax = max(kmax, dpp_mov<DPP_QUAD_XOR1>(kmax));
            // keep the diagonal unless it is more than 2x smaller than the column maximum: a factor 2 is one
            // exponent step, i.e. 1 << 23 on the bit pattern (denormal magnitudes compare conservatively)
            const bool exchange = group_bcast<LANES, LK>(dkey) + (thr_steps << 23) < kmax;
            if (__builtin_expect(__any(exchange), 0)) {   // uniform over the wavefront (and rare: laid out off the hot path);
                                                          // groups that keep their diagonal map to themselves
                const int best = NMAX - 1 - (kmax & (NMAX - 1));
                const int lb = best % LANES, sb = best / LANES;
                const int src = exchange ? (gi == LK ? lb : (gi == lb ? LK : gi)) : gi;
                const bool atk = exchange & (gi == LK), atb = exchange & (gi == lb);
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    cx<T> q[RPL];
#pragma unroll
                    for (int s = 0; s < RPL; ++s) q[s] = shfl_cx(row[s][j], src, LANES);
                    cx<T> from_b = q[0];
#pragma unroll
                    for (int s = 1; s < RPL; ++s) {
                        from_b.x = (sb == s) ? q[s].x : from_b.x;
                        from_b.y = (sb == s) ? q[s].y : from_b.y;
                    }
#pragma unroll
                    for (int s = 0; s < RPL; ++s) {
                        cx<T> v = row[s][j];
                        if (s == SK) {
                            v.x = atk ? from_b.x : v.x;
                            v.y = atk ? from_b.y : v.y;
                        }
                        const bool tb = atb & (sb == s);
                        v.x = tb ? q[SK].x : v.x;
                        v.y = tb ? q[SK].y : v.y;
                        row[s][j] = v;
                    }
                }
                {
                    int q[RPL];
#pragma unroll
                    for (int s = 0; s < RPL; ++s) q[s] = __shfl(orig[s], src, LANES);
                    int from_b = q[0];
#pragma unroll
                    for (int s = 1; s < RPL; ++s) from_b = (sb == s) ? q[s] : from_b;
#pragma unroll
                    for (int s = 0; s < RPL; ++s) {
                        int v = orig[s];
                        if (s == SK) v = atk ? from_b : v;
                        v = (atb & (sb == s)) ? q[SK] : v;
                        orig[s] = v;
                    }
                }
            }
            // values merged from the two paths are opaque from here on: InstCombine otherwise walks the chain of
            // two-way merges per register recursively (compile time doubles with every step)
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
#pragma unroll
                for (int j = 0; j < NMAX; ++j) opaque(row[s][j]);
                asm("" : "+v"(orig[s]));
            }
        }
        const cx<T> inv = crecip_fast(group_bcast<LANES, LK>(row[SK][KK]));
        // rows below K: the lanes above LK of slot SK, and every later slot
        cx<T> l[RPL];
#pragma unroll
        for (int s = SK; s < RPL; ++s) {
            l[s] = row[s][KK] * inv;
            if (s == SK) {
                l[s].x = (gi > LK) ? l[s].x : (T)0;
                l[s].y = (gi > LK) ? l[s].y : (T)0;
            }
        }
        static_for<KK + 1, NMAX>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const cx<T> pr = group_bcast<LANES, LK>(row[SK][J]);
#pragma unroll
            for (int s = SK; s < RPL; ++s) row[s][J] = cfnma(row[s][J], cmul_of(l[s]), pr);
        });
#pragma unroll
        for (int s = SK; s < RPL; ++s) {
            if (s == SK) {
                row[s][KK].x = (gi > LK) ? l[s].x : row[s][KK].x;
                row[s][KK].y = (gi > LK) ? l[s].y : row[s][KK].y;
            } else {
                row[s][KK] = l[s];
            }
        }
        dinv[SK].x = (gi == LK) ? inv.x : dinv[SK].x;
        dinv[SK].y = (gi == LK) ? inv.y : dinv[SK].y;
    });

    if (dud.lu_out) {      // (uniform) the factors and the pivot order are kept for the adjoint system (see Dud): rows are in pivot order already
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int pr = s * LANES + gi;
            if (pr < N) {
                cx<T>* lo = dud.lu_out + kept_lu_index(f, pr, 0, N, BPB);
                if ((N & 1) == 0 || sizeof(T) == 8) {
                    typedef T v4 __attribute__((ext_vector_type(16 / sizeof(T))));
                    constexpr int PER = 16 / sizeof(cx<T>);
#pragma unroll
                    for (int k = 0; k < NMAX; k += PER) {
                        if (k < N) {
                            v4 q;
                            if constexpr (PER == 2) q = v4{row[s][k].x, row[s][k].y, row[s][k + 1].x, row[s][k + 1].y};
                            else q = v4{row[s][k].x, row[s][k].y};
                            *reinterpret_cast<v4*>(lo + k) = q;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NMAX; ++k)
                        if (k < N) lo[k] = row[s][k];
                }
                dud.piv_out[kept_piv_index(f, pr, N, BPB)] = orig[s];
            }
        }
    }

    // ---- w = A^-H cw^H from the factors just formed (see Dud::wadj).  P A = L U with the rows in pivot order: A^H = U^H L^H P, so
    // U^H z = cw^H, L^H u = z, w[orig[p]] = u[p].  Lane (slot s, lane g) holds row p = s LANES + g of the factors, i.e. COLUMN p of
    // U^H and L^H: unknown k is a dot product over the lanes -- sum_p conj(row_p[k]) z_p, the not-yet-known entries still zero, so
    // no masks -- one DPP reduction per unknown.
    // (an instantiation of its own, WADJ: its extra registers take the float64 kernel from two wavefronts per SIMD to one, which the
    // launches that do not ask for w must not pay)
    constexpr bool kWadj = WADJ && (LANES == 8 || LANES == 4) && RPL == 2;      // 4 < N <= 16
    if constexpr (kWadj) {
        if (dud.wadj) {       // (uniform)
            cx<T> v[RPL], z[RPL], u[RPL];
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const bool real_row = s * LANES + gi < N;
                const cx<T> cwv = kPrefetchRhs ? pre_cw[s] : gain_at<T>(dud.cw, dud.cw_real, real_row ? s * LANES + gi : 0);
                v[s] = real_row ? conj(cwv) : cx<T>(0, 0);
                z[s] = u[s] = cx<T>(0, 0);
            }
            static_for<0, NMAX>([&](auto kc) {          // U^H z = cw^H, ascending
                constexpr int KK = decltype(kc)::value;
                constexpr int SK = KK / LANES, LK = KK % LANES;
                if (KK >= N) return;
                cx<T> part(0, 0);
#pragma unroll
                for (int s = 0; s <= SK; ++s) fma_cxc(part, z[s], row[s][KK]);       // z_p conj(U[p][k]); rows of later slots are below the diagonal
                part.x = group_sum<LANES>(part.x);
                part.y = group_sum<LANES>(part.y);
                const cx<T> zk = (v[SK] - part) * conj(dinv[SK]);
                z[SK].x = (gi == LK) ? zk.x : z[SK].x;
                z[SK].y = (gi == LK) ? zk.y : z[SK].y;
            });
            static_for<0, NMAX>([&](auto kc) {          // L^H u = z (unit diagonal), descending
                constexpr int KK = NMAX - 1 - decltype(kc)::value;
                constexpr int SK = KK / LANES, LK = KK % LANES;
                if (KK >= N) return;
                cx<T> part(0, 0);
#pragma unroll
                for (int s = SK; s < RPL; ++s) fma_cxc(part, u[s], row[s][KK]);      // u_p conj(L[p][k]), p > k
                part.x = group_sum<LANES>(part.x);
                part.y = group_sum<LANES>(part.y);
                const cx<T> uk = z[SK] - part;
                u[SK].x = (gi == LK) ? uk.x : u[SK].x;
                u[SK].y = (gi == LK) ? uk.y : u[SK].y;
            });
#pragma unroll
            for (int s = 0; s < RPL; ++s)
                if (s * LANES + gi < N) dud.wadj[(long)orig[s] * dud.wadj_sn + f] = u[s];
        }
    }

    // ---- apply to every right-hand side: slot s of lane g ends up with x_{s*LANES+g}
    const int ncols = B * K;
    // did any row of this wavefront change its slot?  (uniform; almost never: the first column's right-hand side is then the
    // one requested at the top of the kernel)
    bool moved = false;
#pragma unroll
    for (int s = 0; s < RPL; ++s) moved |= orig[s] != s * LANES + gi;
    const bool refetch = !kPrefetchRhs || P != nullptr || no_prefetch || __any(moved);
    for (int col = 0; col < ncols; ++col) {
        const int b = col / K, kk = col - b * K;
        cx<T> y[RPL];
        if (col == 0 && !refetch) {
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                cx<T> v = pre_y[s];
                if (dud.rv) v = (adjoint ? conj(v) : v) * pre_rs;
                if (dud.rhs_l2 && !(adjoint & 1)) v = v * pre_l2[s];
                const bool real_row = s * LANES + gi < N;
                y[s] = cx<T>(real_row ? v.x : (T)0, real_row ? v.y : (T)0);
            }
        } else {
            // all of the column's loads together, unconditionally (rows beyond N read row N - 1 and discard it)
            cx<T> raw[RPL], l2v[RPL], rsv(0, 0);
            if (dud.rv) rsv = dud.rs[(long)b * dud.rs_sb + f];
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                const int oc = orig[s] < N ? orig[s] : N - 1;
                if (dud.rv) raw[s] = gain_at<T>(dud.rv, dud.rv_real, oc);      // rank-one right-hand side (K == 1)
                else raw[s] = R[(long)b * rs_b + (long)oc * rs_n + (long)kk * rs_k + f];
                l2v[s] = (dud.rhs_l2 && !(adjoint & 1)) ? dud.l2[(long)oc * dud.l2_sn + (long)f * dud.l2_sf] : cx<T>(1, 0);
            }
#pragma unroll
            for (int s = 0; s < RPL; ++s) {
                cx<T> v = raw[s];
                if (dud.rv) v = (adjoint ? conj(v) : v) * rsv;
                if (dud.rhs_l2 && !(adjoint & 1)) v = v * l2v[s];
                const bool real_row = s * LANES + gi < N;
                y[s] = cx<T>(real_row ? v.x : (T)0, real_row ? v.y : (T)0);
            }
        }
        static_for<0, NMAX - 1>([&](auto kc) {          // forward: y_i -= L[i][k] y_k, i > k
            constexpr int KK = decltype(kc)::value;
            constexpr int SK = KK / LANES, LK = KK % LANES;
            const cx<T> yk = group_bcast<LANES, LK>(y[SK]);
#pragma unroll
            for (int s = SK; s < RPL; ++s) {
                cx<T> m = row[s][KK];
                if (s == SK) m = cx<T>((gi > LK) ? m.x : (T)0, (gi > LK) ? m.y : (T)0);
                y[s] = cfnma(y[s], cmul_of(m), yk);
            }
        });
        static_for<0, NMAX>([&](auto kc) {              // back: x_k = y_k / U[k][k];  y_i -= U[i][k] x_k, i < k
            constexpr int KK = NMAX - 1 - decltype(kc)::value;
            constexpr int SK = KK / LANES, LK = KK % LANES;
            const cx<T> yd = y[SK] * dinv[SK];
            y[SK].x = (gi == LK) ? yd.x : y[SK].x;
            y[SK].y = (gi == LK) ? yd.y : y[SK].y;
            if constexpr (KK > 0) {
                const cx<T> xk = group_bcast<LANES, LK>(y[SK]);
#pragma unroll
                for (int s = 0; s <= SK; ++s) {
                    cx<T> m = row[s][KK];
                    if (s == SK) m = cx<T>((gi < LK) ? m.x : (T)0, (gi < LK) ? m.y : (T)0);
                    y[s] = cfnma(y[s], cmul_of(m), xk);
                }
            }
        });
        cx<T> z(0, 0);
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const int ri = s * LANES + gi;
            if (ri < N) {
                OUT[(long)b * os_b + (long)ri * os_n + (long)kk * os_k + f] = y[s];
                if (dud.cz) fma_cx(z, (P || !kPrefetchRhs) ? gain_at<T>(dud.cw, dud.cw_real, ri) : pre_cw[s], y[s]);
            }
        }
        if (dud.cz && !adjoint) {      // uniform: the output-gain row applied in the wavefront
            z.x = group_sum<LANES>(z.x);
            z.y = group_sum<LANES>(z.y);
            if (gi == 0) dud.cz[(long)b * dud.cz_sb + f] = z;
        }
    }
}

static int g_solve_rpl2_16 = 0;   // factored loop, N in (4, 16]: 0 = two rows per lane (default), 1 = one row per lane (variant 4)
static int g_solve_rpl2_p = 0;    // tuning: variant 3 = two-rows-per-lane kernel also for a materialised P
static int g_solve_thr = 1;       // pivot threshold 2^-thr (tuning: variant 10 + thr)
static int g_solve_noprefetch = 0;   // tuning: variant 5 = the in-place kernels fetch the right-hand side behind the elimination
static int g_solve_variant = 0;   // tuning hook: 1 forces the shuffle kernel for every N

template <typename T, int NMAX>
static int launch_solve(const void* P, long p_pitch, const Dud<T>& dud, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k,
                        void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, hipStream_t st) {
    constexpr int BPB = 256 / NMAX;
    dim3 grid(cdiv_i(M, BPB));
    if (g_solve_variant == 0) {
        if constexpr (NMAX == 16) {
            // N in (8, 16], factored loop: 8 lanes x 2 rows per lane -- 8 bins per wavefront, every broadcast of a pivot-row
            // element feeds two row updates: 86 -> 73 us at N = 16 (c64), 131 -> 108 us (c128), 82 -> 63 us at N = 9
            // (tools/dbg/archive/solve_rpl2_16.py; 4 lanes x 4 rows: 80 us, and it spills).  fl_debug_set_solve_variant(4): the
            // one-row-per-lane kernels.
            if (!P && g_solve_rpl2_16 != 1 && dud.wadj) {
                hipLaunchKernelGGL((solve_inplace_kernel<T, 8, 2, true>), dim3(cdiv_i(M, 32)), dim3(256), 0, st, (const cx<T>*)P, p_pitch,
                                   dud, one_minus, adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT,
                                   os_b, os_n, os_k, B, M, N, K);
                FL_CHECK_LAUNCH("solve");
                return FL_OK;
            }
            if (!P && g_solve_rpl2_16 != 1) {
                hipLaunchKernelGGL((solve_inplace_kernel<T, 8, 2>), dim3(cdiv_i(M, 32)), dim3(256), 0, st, (const cx<T>*)P, p_pitch,
                                   dud, one_minus, adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT,
                                   os_b, os_n, os_k, B, M, N, K);
                FL_CHECK_LAUNCH("solve");
                return FL_OK;
            }
        }
        if constexpr (NMAX == 8) {
            if (!P && g_solve_rpl2_16 != 1 && dud.wadj) {
                hipLaunchKernelGGL((solve_inplace_kernel<T, 4, 2, true>), dim3(cdiv_i(M, 64)), dim3(256), 0, st, (const cx<T>*)P, p_pitch,
                                   dud, one_minus, adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT,
                                   os_b, os_n, os_k, B, M, N, K);
                FL_CHECK_LAUNCH("solve");
                return FL_OK;
            }
            if (!P && g_solve_rpl2_16 != 1) {      // N in (4, 8] on 4 lanes x 2 rows (16 bins per wavefront): 29 -> 23 us at N = 8
                hipLaunchKernelGGL((solve_inplace_kernel<T, 4, 2>), dim3(cdiv_i(M, 64)), dim3(256), 0, st, (const cx<T>*)P, p_pitch,
                                   dud, one_minus, adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT,
                                   os_b, os_n, os_k, B, M, N, K);
                FL_CHECK_LAUNCH("solve");
                return FL_OK;
            }
        }
        if constexpr (NMAX <= 16) {
            hipLaunchKernelGGL((solve_inplace_kernel<T, NMAX, 1>), grid, dim3(256), 0, st, (const cx<T>*)P, p_pitch, dud, one_minus,
                               adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K);
            FL_CHECK_LAUNCH("solve");
            return FL_OK;
        } else if constexpr (NMAX == 32 && sizeof(T) == 4) {    // two rows per lane, 16 lanes per bin
            // (a materialised P keeps the shuffle kernel for now: its row-per-lane loads are 32-byte pieces here)
            if (!P || g_solve_rpl2_p) {
                hipLaunchKernelGGL((solve_inplace_kernel<T, 16, 2>), dim3(cdiv_i(M, 16)), dim3(256), 0, st, (const cx<T>*)P, p_pitch,
                                   dud, one_minus, adjoint | (g_solve_thr << 8) | (g_solve_noprefetch << 16), (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b,
                                   os_n, os_k, B, M, N, K);
                FL_CHECK_LAUNCH("solve");
                return FL_OK;
            }
        }
    }
    hipLaunchKernelGGL((solve_kernel<T, NMAX>), grid, dim3(256), 0, st, (const cx<T>*)P, p_pitch, dud, one_minus, adjoint,
                       (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K);
    FL_CHECK_LAUNCH("solve");
    return FL_OK;
}


// ---------------------------------------------------------------- loops beyond the register-resident sizes
// N > 64 (float) / 32 (double): one workgroup per bin, the matrix in LDS (pitch N + 1), LU with partial pivoting by the
// whole workgroup (pivot search = a block reduction of |re| + |im| as in the register kernels, row exchange, multipliers,
// rank-one update of the trailing block), then the right-hand sides in blocks of CB columns (forward and back substitution
// with the columns of a block side by side).  A correctness path for sizes the reference's torch.linalg.solve
// (system.py:425) accepts and feedback delay networks rarely use: ~4 N + 2 N (B K / CB) barriers per bin, not tuned.
// WS (loops beyond what the LDS holds: N > 138 / 97): the same algorithm with the matrix in a caller-owned workspace in global
// memory -- one N x (N + 1) slot per workgroup, a workgroup walks the bins f = blockIdx.x, + gridDim.x, ... -- and only the
// right-hand-side block and the pivot list in LDS.  __syncthreads() orders a workgroup's global accesses (workgroup-scope
// fence; its wavefronts share the CU's vector cache).  torch.linalg.solve (system.py:425) has no size bound: neither has this.
template <typename T, bool WS>
__global__ void __launch_bounds__(256) solve_lds_kernel(const cx<T>* __restrict__ P, long p_pitch, int one_minus, int adjoint,
                                                        const cx<T>* __restrict__ R, long rs_b, long rs_n, long rs_k,
                                                        cx<T>* __restrict__ OUT, long os_b, long os_n, long os_k,
                                                        int B, int M, int N, int K, int CB, cx<T>* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = N + 1;
    cx<T>* Am = WS ? ws + (size_t)blockIdx.x * N * NP : reinterpret_cast<cx<T>*>(smem);             // [N][NP]
    cx<T>* Y = WS ? reinterpret_cast<cx<T>*>(smem) : Am + (size_t)N * NP;                            // [N][CB]
    int* piv = reinterpret_cast<int*>(Y + (size_t)N * CB);   // [N]
    __shared__ T red_v[256];
    __shared__ int red_i[256];
    const int tid = threadIdx.x;
  for (int f = blockIdx.x; f < M; f += gridDim.x) {
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e - i * N;
        cx<T> v = adjoint ? conj(P[(size_t)(j * N + i) * p_pitch + f]) : P[(size_t)(i * N + j) * p_pitch + f];
        if (one_minus) v = cx<T>((i == j ? (T)1 : (T)0) - v.x, -v.y);
        Am[i * NP + j] = v;
    }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        // pivot: largest |re| + |im| in column k at or below the diagonal (LAPACK icamax)
        T best = (T)-1;
        int bi = k;
        for (int i = k + tid; i < N; i += 256) {
            const cx<T> v = Am[i * NP + k];
            const T m = fabs(v.x) + fabs(v.y);
            if (m > best) { best = m; bi = i; }
        }
        red_v[tid] = best;
        red_i[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s && (red_v[tid + s] > red_v[tid] || (red_v[tid + s] == red_v[tid] && red_i[tid + s] < red_i[tid]))) {
                red_v[tid] = red_v[tid + s];
                red_i[tid] = red_i[tid + s];
            }
            __syncthreads();
        }
        const int p = red_i[0];
        if (tid == 0) piv[k] = p;
        if (p != k)
            for (int j = tid; j < N; j += 256) {
                const cx<T> t = Am[k * NP + j];
                Am[k * NP + j] = Am[p * NP + j];
                Am[p * NP + j] = t;
            }
        __syncthreads();
        const cx<T> ip = crecip(Am[k * NP + k]);
        for (int i = k + 1 + tid; i < N; i += 256) Am[i * NP + k] = Am[i * NP + k] * ip;
        __syncthreads();
        const int nt = N - k - 1;
        for (int e = tid; e < nt * nt; e += 256) {
            const int i = k + 1 + e / nt, j = k + 1 + e % nt;
            const cx<T> l = Am[i * NP + k], u = Am[k * NP + j];
            cx<T> a = Am[i * NP + j];
            a.x -= l.x * u.x - l.y * u.y;
            a.y -= l.x * u.y + l.y * u.x;
            Am[i * NP + j] = a;
        }
        __syncthreads();
    }
    // right-hand sides, CB columns (b, kk) at a time
    const int ncol = B * K;
    for (int c0 = 0; c0 < ncol; c0 += CB) {
        const int nc = min(CB, ncol - c0);
        for (int e = tid; e < N * nc; e += 256) {
            const int i = e / nc, c = e - i * nc, col = c0 + c;
            const int b = col / K, kk = col - b * K;
            Y[i * CB + c] = R[(size_t)b * rs_b + (size_t)i * rs_n + (size_t)kk * rs_k + f];
        }
        __syncthreads();
        for (int k = 0; k < N; ++k) {           // the factorisation's row exchanges, in order
            const int p = piv[k];
            if (p != k && tid < nc) {
                const cx<T> t = Y[k * CB + tid];
                Y[k * CB + tid] = Y[p * CB + tid];
                Y[p * CB + tid] = t;
            }
            __syncthreads();
        }
        for (int k = 0; k < N; ++k) {           // L y = b (unit lower)
            const int nr = N - k - 1;
            for (int e = tid; e < nr * nc; e += 256) {
                const int i = k + 1 + e / nc, c = e % nc;
                const cx<T> l = Am[i * NP + k], y = Y[k * CB + c];
                cx<T> a = Y[i * CB + c];
                a.x -= l.x * y.x - l.y * y.y;
                a.y -= l.x * y.y + l.y * y.x;
                Y[i * CB + c] = a;
            }
            __syncthreads();
        }
        for (int k = N - 1; k >= 0; --k) {      // U x = y
            const cx<T> ip = crecip(Am[k * NP + k]);
            if (tid < nc) Y[k * CB + tid] = Y[k * CB + tid] * ip;
            __syncthreads();
            for (int e = tid; e < k * nc; e += 256) {
                const int i = e / nc, c = e % nc;
                const cx<T> u = Am[i * NP + k], x = Y[k * CB + c];
                cx<T> a = Y[i * CB + c];
                a.x -= u.x * x.x - u.y * x.y;
                a.y -= u.x * x.y + u.y * x.x;
                Y[i * CB + c] = a;
            }
            __syncthreads();
        }
        for (int e = tid; e < N * nc; e += 256) {
            const int i = e / nc, c = e - i * nc, col = c0 + c;
            const int b = col / K, kk = col - b * K;
            OUT[(size_t)b * os_b + (size_t)i * os_n + (size_t)kk * os_k + f] = Y[i * CB + c];
        }
        __syncthreads();
    }
  }
}

// LDS one workgroup of the current device may ask for (160 KB on MI355X), less 4 KB for the kernel's static arrays
static size_t solve_lds_budget() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
        v = 160 * 1024;      // no device to ask (host-side construction checks): the part this library is built for
    return (size_t)v - 4096;
}
// largest loop the LDS kernel takes: the matrix (pitch N + 1) plus at least four right-hand-side columns in that budget
template <typename T>
static int solve_lds_max_n() {
    const size_t budget = solve_lds_budget();
    int n = 0;
    while (((size_t)(n + 1) * (n + 2) + 4 * (size_t)(n + 1)) * sizeof(cx<T>) + (size_t)(n + 1) * sizeof(int) <= budget) ++n;
    return n;
}

// the workspace form (N beyond solve_lds_max_n, to kSolveWsMaxN): slots of N x (N + 1) values, one per workgroup; as many
// workgroups as two per CU, the bins, and 1 GB of workspace allow
constexpr int kSolveWsMaxN = 1024;
template <typename T>
static int solve_ws_slots(int N, int M) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    long slots = 2L * cus;
    const long cap = (1L << 30) / ((long)N * (N + 1) * (long)sizeof(cx<T>));
    if (slots > cap) slots = cap;
    if (slots > M) slots = M;
    return slots < 1 ? 1 : (int)slots;
}

template <typename T>
static int solve_impl(const void* P, long p_pitch, const Dud<T>& dud, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k,
                      void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream, void* ws = nullptr,
                      size_t ws_bytes = 0) {
    FL_REQUIRE((P || dud.U) && (R || dud.rv) && OUT, "solve: null pointer");
    if (dud.rv || dud.cz) {
        FL_REQUIRE(dud.rv && dud.rs && K == 1 && !P, "solve: the rank-one right-hand side needs its scalar signal, one column per batch item and the factored loop");
        if (g_solve_variant != 0 || N > (sizeof(T) == 4 ? 32 : 16)) {
            set_error("solve: the rank-one right-hand side / contracted output exist in the in-place kernels only (N <= %d here)",
                      sizeof(T) == 4 ? 32 : 16);
            return FL_ERR_UNSUPPORTED;
        }
    }
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0 && (!P || p_pitch >= M), "solve: bad sizes (p_pitch >= M)");
    const int nmax_lim = sizeof(T) == 8 ? 32 : 64;
    if (N > nmax_lim) {
        const int big = solve_lds_max_n<T>();
        if (P && N > big && N <= kSolveWsMaxN && ws) {
            if (B == 0 || M == 0) return FL_OK;
            const int slots = solve_ws_slots<T>(N, M);
            FL_REQUIRE(ws_bytes >= (size_t)slots * N * (N + 1) * sizeof(cx<T>), "solve: workspace too small (fl_solve_ws_bytes)");
            const size_t budget = solve_lds_budget();
            int cb = 16;
            while (cb > 1 && (size_t)N * cb * sizeof(cx<T>) + (size_t)N * sizeof(int) > budget) cb >>= 1;
            const size_t lds = (size_t)N * cb * sizeof(cx<T>) + (size_t)N * sizeof(int);
            static bool attr_ws[64] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
            if (!attr_ws[dev] && lds > 64 * 1024) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(solve_lds_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
                if (e != hipSuccess) {
                    set_error("solve: %zu bytes of LDS per workgroup are not available on device %d (%s)", budget, dev, hipGetErrorString(e));
                    return FL_ERR_UNSUPPORTED;
                }
                attr_ws[dev] = true;
            }
            hipLaunchKernelGGL((solve_lds_kernel<T, true>), dim3(slots), dim3(256), lds, (hipStream_t)stream, (const cx<T>*)P, p_pitch, one_minus,
                               adjoint, (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K, cb, (cx<T>*)ws);
            FL_CHECK_LAUNCH("solve_ws");
            return FL_OK;
        }
        if (!P || N > big) {
            set_error("solve: N=%d exceeds %s (%d for this precision)", N,
                      P ? (N > kSolveWsMaxN ? "the workspace form's limit" : "what one workgroup's LDS holds (pass a workspace: fl_solve_ws_*)")
                        : "the register-resident limit of the factored forms",
                      P ? (N > kSolveWsMaxN ? kSolveWsMaxN : big) : nmax_lim);
            return FL_ERR_UNSUPPORTED;
        }
        if (B == 0 || M == 0) return FL_OK;
        const size_t budget = solve_lds_budget();
        int cb = 16;
        while (cb > 4 && ((size_t)N * (N + 1) + (size_t)N * cb) * sizeof(cx<T>) + (size_t)N * sizeof(int) > budget) cb >>= 1;
        const size_t lds = ((size_t)N * (N + 1) + (size_t)N * cb) * sizeof(cx<T>) + (size_t)N * sizeof(int);
        // the attribute is per device: set (and checked) once on each device this process uses
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(solve_lds_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
            if (e != hipSuccess) {
                set_error("solve: %zu bytes of LDS per workgroup are not available on device %d (%s)", budget, dev, hipGetErrorString(e));
                return FL_ERR_UNSUPPORTED;
            }
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL((solve_lds_kernel<T, false>), dim3(M), dim3(256), lds, (hipStream_t)stream, (const cx<T>*)P, p_pitch, one_minus, adjoint,
                           (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K, cb, (cx<T>*)nullptr);
        FL_CHECK_LAUNCH("solve_lds");
        return FL_OK;
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


// ---------------------------------------------------------------- gradients of the factored loop matrix
// Backward of OUT = (I - diag(l) U diag(r))^-1 R, after the adjoint solve gR = A^-H g.  With P = diag(l) U diag(r) and
// dP_ij = sum gR_i conj(out_j) (sums over batch b and signal column k), one pass over (gR, OUT, l, r) gives all three:
//   gU_ij   = sum_{f,b,k} conj(l_i) gR_i conj(r_j out_j)
//   gl_i[f] = sum_{b,k} gR_i conj((U (r . out))_i)
//   gr_j[f] = sum_{b,k} (U^H (conj(l) . gR))_j conj(out_j)
// (the layered form is five launches -- two diagonal products, the outer-product reduction and its final, a matrix
// product, a per-bin reduction -- each re-reading gR / OUT).  Thread (i, bl): row i of bin f0 + bl, BPI = 256 / NP
// bins per iteration, bl fastest so the loads of a row run along f.  r . out and conj(l) . gR of the iteration's bins go
// through LDS; a lane keeps row i of the gU sum in registers across its block's bins; the lanes of a row are combined
// with a fixed butterfly and the per-block partials summed by a final launch (deterministic).
template <typename T> struct alignas(2 * sizeof(cx<T>)) cx2 { cx<T> a, b; };

// Side reductions of the one-pass backward for a loop that sits between an input-gain column b (R0 = b x, x a scalar
// spectrum) and an output-gain row c (y = c . OUT): with sx = x and sy = gy (element (b, f) at b * s_b + f, one column),
//   g_b[i] = sum_{f,b} conj(l2_i) gR_i conj(x)      g_c[i] = sum_{f,b} gy conj(out_i)
// -- the two gains' gradients, otherwise two bin-reduction launches with a final each.  part: (blocks, 2, N).
template <typename T>
struct DudSide {
    const cx<T>* sx;
    const cx<T>* sy;
    long sx_b, sy_b;
    int on;          // partials then have N*N + 2N entries per block: [gU | g_b | g_c]
    T* real_out;     // non-null: Re g_b, Re g_c as a real (2N) array (real gain vectors) instead of the tail of gU
};
// 1 + 0i in memory, deliberately not const: a constant the compiler can see through turns the loads of an absent factor
// back into a branch

template <typename T, int NP, bool WR>
__global__ void __launch_bounds__(256, (sizeof(T) == 4 && NP <= 16) ? 3 : 1) dud_grads_kernel(Dud<T> d, const cx<T>* __restrict__ gR, const cx<T>* __restrict__ OUT, long s_b,
                                                        long s_n, long s_k, int B, int M, int N, int K, int bins_per_block,
                                                        cx<T>* __restrict__ gl, long gl_sn, cx<T>* __restrict__ gr, long gr_sn,
                                                        cx<T>* __restrict__ partU, cx<T>* __restrict__ gR0, DudSide<T> side) {
    constexpr int BPI = 256 / NP, LDT = NP + 2;      // rows 16-byte aligned and conflict-free for 16-byte reads
    extern __shared__ __attribute__((aligned(32))) char smem_dg[];
    cx<T>* Us = reinterpret_cast<cx<T>*>(smem_dg);   // [NP][LDT]
    cx<T>* t1s = Us + NP * LDT;                        // [2][BPI][LDT]: r . out
    cx<T>* t2s = t1s + 2 * BPI * LDT;                  // [2][BPI][LDT]: conj(l) . gR
    const int tid = threadIdx.x, i = tid / BPI, bl = tid % BPI;
    const bool row = i < N;
    for (int e = tid; e < NP * NP; e += 256) {
        const int a = e / NP, b = e % NP;
        Us[a * LDT + b] = (a < N && b < N) ? d.U[a * N + b] : cx<T>(0, 0);
    }
    cx<T> accU[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) accU[j] = cx<T>(0, 0);
    const int f_begin = blockIdx.x * bins_per_block;
    const int f_end = min(M, f_begin + bins_per_block);
    // rounds q = (iteration, b, k) flattened, k fastest.  The chain of a wavefront is what bounds this kernel (every
    // workgroup is resident at once): the round is kept short -- uniform counters instead of divisions, per-lane base
    // pointers, 16-byte LDS reads -- and PF rounds of operands are in flight per thread.
    const int BK = B * K;
    const int rounds = f_begin < f_end ? ((f_end - f_begin + BPI - 1) / BPI) * BK : 0;
    struct Ops { cx<T> g, o, l, r, l2, sx, sy; };
    // Loads are unconditional, from clamped addresses (lanes outside the problem re-read the last valid element and are
    // zeroed when consumed; an absent factor reads a constant 1 with stride 0): loads under divergent control flow get an
    // s_waitcnt vmcnt(0) at the join, which would serialise every round on the memory latency.
    const int ic = min(i, N - 1), span = f_end - f_begin - 1;
    const long lane_off = (long)ic * s_n + f_begin;
    // the adjoint solution: gR[b][i][f] as a tensor, or W[i][f] gy[b][f] formed here (Dud::wgy) -- ONE load path for both (base and
    // strides selected once; the tensor form multiplies by a constant 1): a branch around the loads would put a wait at its join
    const bool wmode = d.wadj != nullptr;
    const cx<T>* gRp = wmode ? d.wadj + (long)ic * d.wadj_sn + f_begin : gR + lane_off;
    const long g_sb = wmode ? 0 : s_b, g_sk = wmode ? 0 : s_k;
    const cx<T>* one_g = one_ptr<T>();
    const cx<T>* gyp = wmode ? d.wgy + f_begin : one_g;
    const long gy_sb = wmode ? d.wgy_sb : 0, gy_sf = wmode ? 1 : 0;
    const cx<T>* OUTp = OUT + lane_off;
    const cx<T>* one = one_ptr<T>();
    const long l_sf = d.l ? d.l_sf : 0, r_sf = d.r ? d.r_sf : 0;
    const cx<T>* lp = d.l ? d.l + (long)ic * d.l_sn + (long)f_begin * d.l_sf : one;
    const cx<T>* rp = d.r ? d.r + (long)ic * d.r_sn + (long)f_begin * d.r_sf : one;
    // second left factor (Dud::l2): l = l (.) l2; then gl is the gradient of the FIRST factor, gl (.) conj(l2), and gR0
    // (when asked for) the gradient of the unscaled right-hand side, conj(l2) (.) gR
    const long l2_sf = d.l2 ? d.l2_sf : 0;
    const cx<T>* l2p = d.l2 ? d.l2 + (long)ic * d.l2_sn + (long)f_begin * d.l2_sf : one;
    int fit = 0, fb = 0, fk = 0;               // fetch side (uniform)
    auto fetch = [&]() {
        Ops x;
        const int fl = min(fit * BPI + bl, span);
        const long off = (long)fb * s_b + (long)fk * s_k + fl;
        x.g = gRp[(long)fb * g_sb + (long)fk * g_sk + fl] * gyp[(long)fb * gy_sb + (long)fl * gy_sf];
        x.o = OUTp[off];
        x.l = lp[(long)fl * l_sf];
        x.r = rp[(long)fl * r_sf];
        x.l2 = l2p[(long)fl * l2_sf];
        if (side.on) {                                     // (K == 1: host-checked)
            x.sx = side.sx[(long)fb * side.sx_b + f_begin + fl];
            x.sy = side.sy[(long)fb * side.sy_b + f_begin + fl];
        }
        if (++fk == K) {
            fk = 0;
            if (++fb == B) {
                fb = 0;
                ++fit;
            }
        }
        return x;
    };
    constexpr int PF = 2;
    Ops ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = fetch();
    cx<T> gl_acc(0, 0), gr_acc(0, 0), sb_acc(0, 0), sc_acc(0, 0);
    int cit = 0, cbk = 0, cb = 0, ck = 0;        // consume side (uniform)
    __syncthreads();                             // U is in LDS
    // (rounds past the end run on zeros -- no early exit inside the unrolled body, whose phi moves cost more than the
    // arithmetic they skip)
    for (int q0 = 0; q0 < rounds; q0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int q = q0 + u;
            Ops x = ring[u];
            ring[u] = fetch();
            if (!(row && cit * BPI + bl <= span)) x.g = x.o = cx<T>(0, 0);
            cx<T>* t1b = t1s + (q & 1) * BPI * LDT;     // double-buffered: one barrier per round
            cx<T>* t2b = t2s + (q & 1) * BPI * LDT;
            if (gR0 && row && cit * BPI + bl <= span)
                gR0[(long)i * s_n + f_begin + cit * BPI + bl + (long)cb * s_b + (long)ck * s_k] = mulc(x.g, x.l2);
            if (++ck == K) {
                ck = 0;
                if (++cb == B) cb = 0;
            }
            if (side.on) {
                fma_cxc(sb_acc, mulc(x.g, x.l2), x.sx);  // conj(l2) gR conj(x)     (x.g, x.o are zero outside the problem)
                fma_cxc(sc_acc, x.sy, x.o);              // gy conj(out)
            }
            const cx<T> t2 = mulc(x.g, x.l * x.l2);     // conj(l) gR
            t1b[bl * LDT + i] = x.r * x.o;
            if (WR) t2b[bl * LDT + i] = t2;
            __syncthreads();
            cx<T> v(0, 0);
            const cx2<T>* t1v = reinterpret_cast<const cx2<T>*>(t1b + bl * LDT);
            const cx2<T>* urow = reinterpret_cast<const cx2<T>*>(Us + i * LDT);
#pragma unroll
            for (int j = 0; j < NP; j += 2) {
                const cx2<T> tt = t1v[j / 2], uu = urow[j / 2];
                fma_cxc(accU[j], t2, tt.a);
                fma_cxc(accU[j + 1], t2, tt.b);
                fma_cx(v, uu.a, tt.a);
                fma_cx(v, uu.b, tt.b);
            }
            fma_cxc(gl_acc, x.g, v);
            if (WR) {
                cx<T> w(0, 0);
#pragma unroll
                for (int j = 0; j < NP; ++j) {          // w_i += conj(U_ji) t2_j
                    const cx<T> uji = Us[j * LDT + i], t2j = t2b[bl * LDT + j];
                    w.x += uji.x * t2j.x + uji.y * t2j.y;
                    w.y += uji.x * t2j.y - uji.y * t2j.x;
                }
                fma_cxc(gr_acc, w, x.o);
            }
            if (++cbk == BK) {
                const int f = f_begin + cit * BPI + bl;
                if (row && f < f_end) {
                    if (gl) gl[(long)i * gl_sn + f] = mulc(gl_acc, x.l2);
                    if (WR) gr[(long)i * gr_sn + f] = gr_acc;
                }
                gl_acc = cx<T>(0, 0);
                gr_acc = cx<T>(0, 0);
                cbk = 0;
                ++cit;
            }
        }
    }
    if (!partU) return;
    // the BPI lanes of a row are consecutive threads: butterfly inside the wavefront (BPI <= 64)
    auto row_sum = [&](T& vr, T& vi) {
        if constexpr (BPI <= 16) {      // the lanes of a row sit inside one DPP row: register moves, no LDS round trips
            vr += dpp_mov<0xB1>(vr);    // quad_perm [1,0,3,2]
            vi += dpp_mov<0xB1>(vi);
            vr += dpp_mov<0x4E>(vr);    // quad_perm [2,3,0,1]
            vi += dpp_mov<0x4E>(vi);
            if constexpr (BPI >= 8) {
                vr += dpp_mov<0x141>(vr);   // row_half_mirror
                vi += dpp_mov<0x141>(vi);
            }
            if constexpr (BPI >= 16) {
                vr += dpp_mov<0x140>(vr);   // row_mirror
                vi += dpp_mov<0x140>(vi);
            }
        } else {
#pragma unroll
            for (int off = BPI / 2; off >= 1; off >>= 1) {
                vr += __shfl_xor(vr, off, 64);
                vi += __shfl_xor(vi, off, 64);
            }
        }
    };
    const size_t pstride = (size_t)N * N + (side.on ? 2 * (size_t)N : 0);
    cx<T>* pblk = partU + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        T vr = accU[j].x, vi = accU[j].y;
        row_sum(vr, vi);
        if (bl == 0 && row && j < N) pblk[(size_t)i * N + j] = cx<T>(vr, vi);
    }
    if (side.on) {
        T br = sb_acc.x, bi = sb_acc.y, cr = sc_acc.x, ci = sc_acc.y;
        row_sum(br, bi);
        row_sum(cr, ci);
        if (bl == 0 && row) {
            pblk[(size_t)N * N + i] = cx<T>(br, bi);
            pblk[(size_t)N * N + N + i] = cx<T>(cr, ci);
        }
    }
}

// per-block partials -> gU.  Thread (pg, e): element e0 + e, partials pg, pg + 16, ...: 128-byte coalesced reads, 16
// independent chains per element, combined in a fixed order
template <typename T>
__global__ void __launch_bounds__(256) dud_grads_final_kernel(const cx<T>* __restrict__ part, int nblk, int count, cx<T>* __restrict__ gU,
                                                              int tail_from, T* __restrict__ tail_real) {
    __shared__ cx<T> red[16][17];
    const int e = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int el = blockIdx.x * 16 + e;
    T vr = 0, vi = 0;
    if (el < count) {
#pragma unroll 8
        for (int b = pg; b < nblk; b += 16) {
            const cx<T> v = part[(size_t)b * count + el];
            vr += v.x;
            vi += v.y;
        }
    }
    red[pg][e] = cx<T>(vr, vi);
    __syncthreads();
    if (pg == 0 && el < count) {
        cx<T> a = red[0][e];
#pragma unroll
        for (int p = 1; p < 16; ++p) a = a + red[p][e];
        // (tail_real: the entries from tail_from on are gradients of REAL gain vectors: their real parts, as a real array)
        if (tail_real && el >= tail_from) tail_real[el - tail_from] = a.x;
        else gU[el] = a;
    }
}

static int dud_grads_blocks(int M, int N) {
    int np = 4;
    while (np < N) np *= 2;
    const int bpi = 256 / np;
    int nblk = cdiv_i(M, bpi);
    return nblk > 768 ? 768 : nblk;      // three resident workgroups per CU: one wave of workgroups
}

template <typename T>
static int dud_grads_impl(const Dud<T>& d, const void* gR, const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N, int K,
                          void* gl, long gl_sn, void* gr, long gr_sn, void* partU, void* gU, void* stream, void* gR0 = nullptr,
                          DudSide<T> side = DudSide<T>{nullptr, nullptr, 0, 0, 0, nullptr}) {
    FL_REQUIRE(d.U && (gR || (d.wadj && d.wgy)) && OUT, "solve_dud_grads: null pointer");
    FL_REQUIRE(!d.wadj || K == 1, "solve_dud_grads: the rank-one adjoint solution has one column per batch item");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0, "solve_dud_grads: bad sizes");
    FL_REQUIRE((partU == nullptr) == (gU == nullptr), "solve_dud_grads: partU and gU go together");
    FL_REQUIRE((!gl || (d.l && d.l_sf)) && (!gr || (d.r && d.r_sf)), "solve_dud_grads: gl / gr are per-bin gradients of per-bin factors");
    const int nmax_lim = sizeof(T) == 8 ? 32 : 64;
    if (N > nmax_lim) {
        set_error("solve_dud_grads: N=%d exceeds the register-resident limit (%d) for this precision", N, nmax_lim);
        return FL_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nblk = dud_grads_blocks(M, N);
    if (B == 0 || M == 0) {
        if (gU) {
            int rc = check_hip(hipMemsetAsync(gU, 0, sizeof(cx<T>) * (N * N + (side.on ? 2 * N : 0)), st), "solve_dud_grads: memset");
            if (rc) return rc;
        }
        return FL_OK;
    }
#define FL_DG(NP_)                                                                                                              \
    {                                                                                                                           \
        constexpr int BPI = 256 / NP_;                                                                                          \
        const int per = cdiv_i(cdiv_i(M, nblk), BPI) * BPI;                                                                     \
        const size_t lds = ((size_t)NP_ * (NP_ + 2) + 4 * (size_t)BPI * (NP_ + 2)) * sizeof(cx<T>);                             \
        if (gr)                                                                                                                 \
            hipLaunchKernelGGL((dud_grads_kernel<T, NP_, true>), dim3(nblk), dim3(256), lds, st, d, (const cx<T>*)gR,           \
                               (const cx<T>*)OUT, s_b, s_n, s_k, B, M, N, K, per, (cx<T>*)gl, gl_sn, (cx<T>*)gr, gr_sn,         \
                               (cx<T>*)partU, (cx<T>*)gR0, side);                                                               \
        else                                                                                                                    \
            hipLaunchKernelGGL((dud_grads_kernel<T, NP_, false>), dim3(nblk), dim3(256), lds, st, d, (const cx<T>*)gR,          \
                               (const cx<T>*)OUT, s_b, s_n, s_k, B, M, N, K, per, (cx<T>*)gl, gl_sn, (cx<T>*)gr, gr_sn,         \
                               (cx<T>*)partU, (cx<T>*)gR0, side);                                                               \
    }
    if (N <= 4) FL_DG(4) else if (N <= 8) FL_DG(8) else if (N <= 16) FL_DG(16) else if (N <= 32) FL_DG(32) else {
        if constexpr (sizeof(T) == 4) FL_DG(64) else return FL_ERR_UNSUPPORTED;
    }
#undef FL_DG
    FL_CHECK_LAUNCH("solve_dud_grads");
    if (gU) {
        const int count = N * N + (side.on ? 2 * N : 0);       // gU is then (N*N + 2N): [gU | g_b | g_c]
        hipLaunchKernelGGL((dud_grads_final_kernel<T>), dim3(cdiv_i(count, 16)), dim3(256), 0, st, (const cx<T>*)partU, nblk, count,
                           (cx<T>*)gU, N * N, side.on ? side.real_out : (T*)nullptr);
        FL_CHECK_LAUNCH("solve_dud_grads_final");
    }
    return FL_OK;
}

}  // namespace fl

using namespace fl;

// forward system of the scaled loop with its factors kept (shuffle kernel: N <= 64 / 32), and the adjoint system from them
template <typename T>
static int solve_scaled_keep_impl(const void* P, long p_pitch, const void* l, long l_sn, const void* R, long rs_b, long rs_n, long rs_k,
                                  void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* LU, void* piv, void* stream) {
    FL_REQUIRE(P && l && R && OUT && LU && piv, "solve_scaled_keep: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0 && p_pitch >= M, "solve_scaled_keep: bad sizes (p_pitch >= M)");
    FL_REQUIRE(N <= (sizeof(T) == 8 ? 32 : 64), "solve_scaled_keep: N exceeds the register-resident kernels (%d)", sizeof(T) == 8 ? 32 : 64);
    if (B == 0 || M == 0) return FL_OK;
    Dud<T> d = {(const cx<T>*)l, l_sn, 0, nullptr, nullptr, 0, 0};
    d.lu_out = (cx<T>*)LU; d.piv_out = (int*)piv;
    hipStream_t st = (hipStream_t)stream;
#define FL_KEEP(NM)                                                                                                                  \
    {                                                                                                                                \
        hipLaunchKernelGGL((solve_kernel<T, NM>), dim3(cdiv_i(M, 256 / NM)), dim3(256), 0, st, (const cx<T>*)P, p_pitch, d, 1, 0,    \
                           (const cx<T>*)R, rs_b, rs_n, rs_k, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K);                            \
        FL_CHECK_LAUNCH("solve_scaled_keep");                                                                                        \
        return FL_OK;                                                                                                                \
    }
    if (N <= 4) FL_KEEP(4)
    if (N <= 8) FL_KEEP(8)
    if (N <= 16) FL_KEEP(16)
    if (N <= 32) FL_KEEP(32)
    if constexpr (sizeof(T) == 4) FL_KEEP(64)
#undef FL_KEEP
    return FL_ERR_UNSUPPORTED;
}
template <typename T>
static int solve_kept_adjoint_impl(const void* LU, const void* piv, const void* R, long rs_b, long rs_n,
                                   long rs_k, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream,
                                   int tile_bins = 0, const void* rv = nullptr, int rv_real = 0, const void* rsig = nullptr,
                                   long rsig_sb = 0) {
    FL_REQUIRE(LU && piv && (R || (rv && rsig)) && OUT, "solve_kept_adjoint: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && N > 0 && K > 0, "solve_kept_adjoint: bad sizes");
    FL_REQUIRE(!rv || K == 1, "solve_kept_adjoint: the rank-one right-hand side has one column per batch item");
    FL_REQUIRE(N <= (sizeof(T) == 8 ? 32 : 64), "solve_kept_adjoint: N exceeds the register-resident kernels (%d)", sizeof(T) == 8 ? 32 : 64);
    if (B == 0 || M == 0) return FL_OK;
    hipStream_t st = (hipStream_t)stream;
#define FL_KEPT(NM)                                                                                                                  \
    {                                                                                                                                \
        hipLaunchKernelGGL((solve_kept_adjoint_kernel<T, NM>), dim3(cdiv_i(M, 256 / NM)), dim3(256), 0, st, (const cx<T>*)LU,       \
                           (const int*)piv, tile_bins > 0 ? tile_bins : 256 / NM, (const cx<T>*)R, rs_b, rs_n, rs_k, rv, rv_real,    \
                           (const cx<T>*)rsig, rsig_sb, (cx<T>*)OUT, os_b, os_n, os_k, B, M, N, K);                                  \
        FL_CHECK_LAUNCH("solve_kept_adjoint");                                                                                       \
        return FL_OK;                                                                                                                \
    }
    if (N <= 4) FL_KEPT(4)
    if (N <= 8) FL_KEPT(8)
    if (N <= 16) FL_KEPT(16)
    if (N <= 32) FL_KEPT(32)
    if constexpr (sizeof(T) == 4) FL_KEPT(64)
#undef FL_KEPT
    return FL_ERR_UNSUPPORTED;
}
extern "C" {
int fl_solve_max_n(int f64) { return f64 ? solve_lds_max_n<double>() : solve_lds_max_n<float>(); }

int fl_debug_set_solve_variant(int variant) {
    g_solve_thr = 1;
    g_solve_rpl2_p = variant == 3;
    g_solve_rpl2_16 = variant == 4 ? 1 : 0;
    g_solve_noprefetch = variant == 5 ? 1 : 0;
    if (variant == 3 || variant == 4 || variant == 5) variant = 0;
    if (variant >= 10) {          // 10 + t: in-place kernels with pivot threshold 2^-t
        g_solve_thr = variant - 10;
        variant = 0;
    }
    g_solve_variant = variant;
    return FL_OK;
}
int fl_solve_ws_max_n(void) { return kSolveWsMaxN; }
long fl_solve_ws_bytes(int N, int M, int f64) {
    if (N <= (f64 ? solve_lds_max_n<double>() : solve_lds_max_n<float>()) || N > kSolveWsMaxN || M <= 0) return 0;
    return f64 ? (long)solve_ws_slots<double>(N, M) * N * (N + 1) * (long)sizeof(cx<double>)
               : (long)solve_ws_slots<float>(N, M) * N * (N + 1) * (long)sizeof(cx<float>);
}
int fl_solve_ws_c64(const void* P, long p_pitch, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                    long os_b, long os_n, long os_k, int B, int M, int N, int K, void* ws, long ws_bytes, void* stream) {
    Dud<float> none = {};
    return solve_impl<float>(P, p_pitch, none, one_minus, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream, ws,
                             ws_bytes > 0 ? (size_t)ws_bytes : 0);
}
int fl_solve_ws_c128(const void* P, long p_pitch, int one_minus, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                     long os_b, long os_n, long os_k, int B, int M, int N, int K, void* ws, long ws_bytes, void* stream) {
    Dud<double> none = {};
    return solve_impl<double>(P, p_pitch, none, one_minus, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream, ws,
                              ws_bytes > 0 ? (size_t)ws_bytes : 0);
}
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
int fl_solve_scaled_c64(const void* P, long p_pitch, const void* l, long l_sn, int adjoint, const void* R, long rs_b, long rs_n,
                        long rs_k, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(P && l, "solve_scaled: null pointer");
    Dud<float> d = {(const cx<float>*)l, l_sn, 0, nullptr, nullptr, 0, 0};
    return solve_impl<float>(P, p_pitch, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
// elements of the kept-factor arrays for N channels and M bins (both tiled by the workgroup's bins: see Dud)
static int kept_bpb(int N, bool f64) { return 256 / (N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64); }
size_t fl_solve_kept_lu_elems(int N, int M, int f64) {
    if (N <= 0 || M <= 0 || N > (f64 ? 32 : 64)) return 0;
    const int bpb = kept_bpb(N, f64);
    return (size_t)cdiv_i(M, bpb) * N * N * bpb;
}
size_t fl_solve_kept_piv_elems(int N, int M, int f64) {
    if (N <= 0 || M <= 0 || N > (f64 ? 32 : 64)) return 0;
    const int bpb = kept_bpb(N, f64);
    return (size_t)cdiv_i(M, bpb) * N * bpb;
}
// the FDN form, forward system, with w = A^-H cw^H beside OUT and cz (Dud::wadj): float32, 4 < N <= 16
int fl_solve_fdn_wadj_supported(int N) { return (N > 4 && N <= 16 && g_solve_variant == 0 && g_solve_rpl2_16 != 1) ? 1 : 0; }
int fl_solve_fdn_wadj_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                          long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                          void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* wadj, long wadj_sn,
                          void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && cw && wadj, "solve_fdn_wadj: null pointer");
    FL_REQUIRE(fl_solve_fdn_wadj_supported(N), "solve_fdn_wadj: 4 < N <= 16 on the default kernels (fl_solve_fdn_wadj_supported)");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, 1, rv, (const cx<float>*)rs, rs_sb, cw, (cx<float>*)cz, cz_sb, rv_real, cw_real};
    d.wadj = (cx<float>*)wadj; d.wadj_sn = wadj_sn;
    return solve_impl<float>(nullptr, 0, d, 1, 0, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
int fl_solve_fdn_wadj_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                           long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                           void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* wadj, long wadj_sn,
                           void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && cw && wadj, "solve_fdn_wadj: null pointer");
    FL_REQUIRE(fl_solve_fdn_wadj_supported(N), "solve_fdn_wadj: 4 < N <= 16 on the default kernels (fl_solve_fdn_wadj_supported)");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, 1, rv, (const cx<double>*)rs, rs_sb, cw, (cx<double>*)cz, cz_sb, rv_real, cw_real};
    d.wadj = (cx<double>*)wadj; d.wadj_sn = wadj_sn;
    return solve_impl<double>(nullptr, 0, d, 1, 0, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
int fl_solve_dud2_grads_w_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                               long r_sn, long r_sf, const void* W, long w_sn, const void* gy, long gy_sb, const void* OUT, long s_b,
                               long s_n, long s_k, int B, int M, int N, void* gl, long gl_sn, void* gr, long gr_sn, void* partU,
                               void* gU, void* gR0, const void* sx, long sx_b, const void* sy, long sy_b, void* g_side_real,
                               void* stream) {
    FL_REQUIRE(l2 && W && gy, "solve_dud2_grads_w: null pointer");
    FL_REQUIRE((sx == nullptr) == (sy == nullptr) && (!sx || (partU && gU)), "solve_dud2_grads_w: side reductions need sx, sy and the partial buffers");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, 0};
    d.wadj = (cx<double>*)const_cast<void*>(W); d.wadj_sn = w_sn; d.wgy = (const cx<double>*)gy; d.wgy_sb = gy_sb;
    DudSide<double> side = {(const cx<double>*)sx, (const cx<double>*)sy, sx_b, sy_b, sx ? 1 : 0, (double*)g_side_real};
    return dud_grads_impl<double>(d, nullptr, OUT, s_b, s_n, s_k, B, M, N, 1, gl, gl_sn, gr, gr_sn, partU, gU, stream, gR0, side);
}
// the FDN form (fl_solve_fdn_*) with kept factors: 8 < N <= 16 on the two-rows-per-lane kernel (its workgroup's 32 bins are a tile)
int fl_solve_fdn_keep_tile(int N, int f64) {
    (void)f64;
    return (N > 8 && N <= 16 && g_solve_variant == 0 && g_solve_rpl2_16 != 1) ? 32 : 0;
}
int fl_solve_fdn_keep_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                          long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                          void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* LU, void* piv,
                          void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && (!cz || cw) && LU && piv, "solve_fdn_keep: null pointer");
    FL_REQUIRE(fl_solve_fdn_keep_tile(N, 0) > 0, "solve_fdn_keep: 8 < N <= 16 on the default kernels (fl_solve_fdn_keep_tile)");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, 1, rv, (const cx<float>*)rs, rs_sb, cw, (cx<float>*)cz, cz_sb, rv_real, cw_real};
    d.lu_out = (cx<float>*)LU; d.piv_out = (int*)piv;
    return solve_impl<float>(nullptr, 0, d, 1, 0, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
int fl_solve_fdn_keep_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                           long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                           void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* LU, void* piv,
                           void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && (!cz || cw) && LU && piv, "solve_fdn_keep: null pointer");
    FL_REQUIRE(fl_solve_fdn_keep_tile(N, 1) > 0, "solve_fdn_keep: 8 < N <= 16 on the default kernels (fl_solve_fdn_keep_tile)");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, 1, rv, (const cx<double>*)rs, rs_sb, cw, (cx<double>*)cz, cz_sb, rv_real, cw_real};
    d.lu_out = (cx<double>*)LU; d.piv_out = (int*)piv;
    return solve_impl<double>(nullptr, 0, d, 1, 0, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
// A^-H (conj(rv) . rs) from factors kept by fl_solve_fdn_keep_* (tile_bins = fl_solve_fdn_keep_tile)
int fl_solve_kept_adjoint_rank1_c64(const void* LU, const void* piv, int tile_bins, const void* rv, int rv_real, const void* rs, long rs_sb,
                                    void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream) {
    FL_REQUIRE(rv && rs && tile_bins > 0, "solve_kept_adjoint_rank1: null pointer / tile");
    return solve_kept_adjoint_impl<float>(LU, piv, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream, tile_bins, rv, rv_real, rs, rs_sb);
}
int fl_solve_kept_adjoint_rank1_c128(const void* LU, const void* piv, int tile_bins, const void* rv, int rv_real, const void* rs, long rs_sb,
                                     void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream) {
    FL_REQUIRE(rv && rs && tile_bins > 0, "solve_kept_adjoint_rank1: null pointer / tile");
    return solve_kept_adjoint_impl<double>(LU, piv, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream, tile_bins, rv, rv_real, rs, rs_sb);
}
int fl_solve_scaled_keep_c64(const void* P, long p_pitch, const void* l, long l_sn, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                             long os_b, long os_n, long os_k, int B, int M, int N, int K, void* LU, void* piv, void* stream) {
    return solve_scaled_keep_impl<float>(P, p_pitch, l, l_sn, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, LU, piv, stream);
}
int fl_solve_scaled_keep_c128(const void* P, long p_pitch, const void* l, long l_sn, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                              long os_b, long os_n, long os_k, int B, int M, int N, int K, void* LU, void* piv, void* stream) {
    return solve_scaled_keep_impl<double>(P, p_pitch, l, l_sn, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, LU, piv, stream);
}
int fl_solve_kept_adjoint_c64(const void* LU, const void* piv, const void* R, long rs_b, long rs_n, long rs_k,
                              void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    return solve_kept_adjoint_impl<float>(LU, piv, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_kept_adjoint_c128(const void* LU, const void* piv, const void* R, long rs_b, long rs_n, long rs_k,
                               void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    return solve_kept_adjoint_impl<double>(LU, piv, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_scaled_c128(const void* P, long p_pitch, const void* l, long l_sn, int adjoint, const void* R, long rs_b, long rs_n,
                         long rs_k, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(P && l, "solve_scaled: null pointer");
    Dud<double> d = {(const cx<double>*)l, l_sn, 0, nullptr, nullptr, 0, 0};
    return solve_impl<double>(P, p_pitch, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
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
int fl_solve_dud_grads_blocks(int M, int N) { return dud_grads_blocks(M, N); }
int fl_solve_dud_grads_c64(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, const void* gR,
                           const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N, int K, void* gl, long gl_sn, void* gr,
                           long gr_sn, void* partU, void* gU, void* stream) {
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf};
    return dud_grads_impl<float>(d, gR, OUT, s_b, s_n, s_k, B, M, N, K, gl, gl_sn, gr, gr_sn, partU, gU, stream);
}
int fl_solve_dud_grads_c128(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, const void* gR,
                            const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N, int K, void* gl, long gl_sn, void* gr,
                            long gr_sn, void* partU, void* gU, void* stream) {
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf};
    return dud_grads_impl<double>(d, gR, OUT, s_b, s_n, s_k, B, M, N, K, gl, gl_sn, gr, gr_sn, partU, gU, stream);
}
/* two left factors: A_f = I - diag(l (.) l2) U diag(r); rhs_l2: the right-hand side is l2 (.) R (forward system) */
int fl_solve_dud2_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, int rhs_l2, const void* U,
                      const void* r, long r_sn, long r_sf, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                      long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(U && l2, "solve_dud2: null pointer");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, rhs_l2};
    return solve_impl<float>(nullptr, 0, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_dud2_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, int rhs_l2, const void* U,
                       const void* r, long r_sn, long r_sf, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                       long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream) {
    FL_REQUIRE(U && l2, "solve_dud2: null pointer");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, rhs_l2};
    return solve_impl<double>(nullptr, 0, d, 1, adjoint, R, rs_b, rs_n, rs_k, OUT, os_b, os_n, os_k, B, M, N, K, stream);
}
int fl_solve_dud2_grads_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                            long r_sn, long r_sf, const void* gR, const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N,
                            int K, void* gl, long gl_sn, void* gr, long gr_sn, void* partU, void* gU, void* gR0, const void* sx,
                            long sx_b, const void* sy, long sy_b, void* g_side_real, void* stream) {
    FL_REQUIRE(l2, "solve_dud2_grads: null pointer");
    FL_REQUIRE((sx == nullptr) == (sy == nullptr) && (!sx || (K == 1 && partU && gU)), "solve_dud2_grads: side reductions need sx, sy, one column per batch item and the partial buffers");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, 0};
    DudSide<float> side = {(const cx<float>*)sx, (const cx<float>*)sy, sx_b, sy_b, sx ? 1 : 0, (float*)g_side_real};
    return dud_grads_impl<float>(d, gR, OUT, s_b, s_n, s_k, B, M, N, K, gl, gl_sn, gr, gr_sn, partU, gU, stream, gR0, side);
}
int fl_solve_dud2_grads_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                             long r_sn, long r_sf, const void* gR, const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N,
                             int K, void* gl, long gl_sn, void* gr, long gr_sn, void* partU, void* gU, void* gR0, const void* sx,
                             long sx_b, const void* sy, long sy_b, void* g_side_real, void* stream) {
    FL_REQUIRE(l2, "solve_dud2_grads: null pointer");
    FL_REQUIRE((sx == nullptr) == (sy == nullptr) && (!sx || (K == 1 && partU && gU)), "solve_dud2_grads: side reductions need sx, sy, one column per batch item and the partial buffers");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, 0};
    DudSide<double> side = {(const cx<double>*)sx, (const cx<double>*)sy, sx_b, sy_b, sx ? 1 : 0, (double*)g_side_real};
    return dud_grads_impl<double>(d, gR, OUT, s_b, s_n, s_k, B, M, N, K, gl, gl_sn, gr, gr_sn, partU, gU, stream, gR0, side);
}
// the same with the adjoint solution given as gR[b][n][f] = W[n][f] gy[b][f] (W from fl_solve_fdn_wadj_c64), formed in the kernel
int fl_solve_dud2_grads_w_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                              long r_sn, long r_sf, const void* W, long w_sn, const void* gy, long gy_sb, const void* OUT, long s_b,
                              long s_n, long s_k, int B, int M, int N, void* gl, long gl_sn, void* gr, long gr_sn, void* partU,
                              void* gU, void* gR0, const void* sx, long sx_b, const void* sy, long sy_b, void* g_side_real,
                              void* stream) {
    FL_REQUIRE(l2 && W && gy, "solve_dud2_grads_w: null pointer");
    FL_REQUIRE((sx == nullptr) == (sy == nullptr) && (!sx || (partU && gU)), "solve_dud2_grads_w: side reductions need sx, sy and the partial buffers");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, 0};
    d.wadj = (cx<float>*)const_cast<void*>(W); d.wadj_sn = w_sn; d.wgy = (const cx<float>*)gy; d.wgy_sb = gy_sb;
    DudSide<float> side = {(const cx<float>*)sx, (const cx<float>*)sy, sx_b, sy_b, sx ? 1 : 0, (float*)g_side_real};
    return dud_grads_impl<float>(d, nullptr, OUT, s_b, s_n, s_k, B, M, N, 1, gl, gl_sn, gr, gr_sn, partU, gU, stream, gR0, side);
}
/* fl_solve_dud2 with the right-hand side built in the kernel, R_i = rv_i rs (rv conjugated for the adjoint system; scaled by
 * l2 for the forward one), and -- forward system, cz non-NULL -- the contracted output z = sum_i cw_i OUT_i beside OUT */
int fl_solve_fdn_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                     long r_sn, long r_sf, int adjoint, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw,
                     int cw_real, void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && (!cz || cw), "solve_fdn: null pointer");
    Dud<float> d = {(const cx<float>*)l, l_sn, l_sf, (const cx<float>*)U, (const cx<float>*)r, r_sn, r_sf,
                    (const cx<float>*)l2, l2_sn, l2_sf, adjoint ? 0 : 1, rv, (const cx<float>*)rs, rs_sb, cw, (cx<float>*)cz, cz_sb,
                    rv_real, cw_real};
    return solve_impl<float>(nullptr, 0, d, 1, adjoint, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
int fl_solve_fdn_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                      long r_sn, long r_sf, int adjoint, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw,
                      int cw_real, void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream) {
    FL_REQUIRE(U && l2 && rv && rs && (!cz || cw), "solve_fdn: null pointer");
    Dud<double> d = {(const cx<double>*)l, l_sn, l_sf, (const cx<double>*)U, (const cx<double>*)r, r_sn, r_sf,
                     (const cx<double>*)l2, l2_sn, l2_sf, adjoint ? 0 : 1, rv, (const cx<double>*)rs, rs_sb, cw, (cx<double>*)cz, cz_sb,
                     rv_real, cw_real};
    return solve_impl<double>(nullptr, 0, d, 1, adjoint, nullptr, 0, 0, 0, OUT, os_b, os_n, os_k, B, M, N, 1, stream);
}
}
