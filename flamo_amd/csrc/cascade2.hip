// Second-order-section cascades of graphic equalisers, second generation (gfx950 / MI355X): the kernels that carry the
// Matrix-then-GEQ pair of BASELINE configs[1] (flamo/processor/dsp.py:2563-2593 over flamo/auxiliary/eq.py:57-111, the
// cascade tail dsp.py:1520-1526 and the einsum dsp.py:922-924 of the composition with dsp.py:466-468).
//
//  * backward, "one lane per (channel pair, section)" (sos_bwd_lanes_kernel).  The first-generation kernel
//    (response.hip: sos_response_bwd_mixed_kernel) gives a lane one BIN and walks the 12 sections of a cascade with 72 running
//    sums per lane: 207 registers, two wavefronts per SIMD, table reads, a 72-value reduce-scatter per workgroup -- 363
//    instructions per bin and cascade of which 202 are arithmetic, at 2.6x its arithmetic bound.  Here a lane owns ONE section
//    of ONE channel pair for the whole launch and walks the bins: its section's coefficients live in registers, its sums are
//    four floats, there is no cross-lane reduction at all, and everything that depends on the bin only (1 -+ cos, sin) or on
//    (pair, bin) only (the cotangent times the saved response) is produced once per tile by a short lane-per-bin phase and read
//    from LDS as broadcasts.  Per lane and bin: 11 packed instructions + 2 reciprocals.
//  * numerator and denominator polynomial of a section share a packed instruction ((B, A) in the two halves), not two
//    sections: a graphic equaliser's band 0 is a pure gain (eq.py:91-94) whose gradient is sum Re(q) / b0 in closed form, so
//    11 lanes per pair, not 12.
//  * two running sums per polynomial instead of three: with t = q / P~, P~ = (S cos + T) + i D sin the section polynomial
//    turned by half a sample (response_common.h), the three tap gradients need sum Re t, sum (1 - cos) Re t, sum sin Im t; but
//    sum Re(t P~) = sum Re(q) =: Q holds identically, i.e. (S + T) G0 - S G1 - D G2 = Q, and Q is shared by all sections of
//    the pair.  G1 is recovered from it in double.  Well conditioned for equaliser sections (S = b0 + g^2 b2 ~ 2 sqrt(gain);
//    S + T is the SMALL coefficient at low frequency, so the large sum G0 enters scaled down): the generic SOS classes
//    (Biquad, SVF, PEQ, user maps; S may vanish) keep the first-generation kernels.
//  * the constant factor's gradient dL/dW[j][n] = sum_f Re(conj(G[m][j]) dL/dH[m][n]) rides in the lane-per-bin work as packed
//    multiply-adds (an 8 x 8 contraction per bin: tried on v_mfma_f32_16x16x4_f32 -- a quarter of each tile, two bins per
//    instruction, operands through LDS -- it cost more than the whole lane-per-section work; north_star's "MFMA only where the
//    channel tile is a genuine dense contraction" decides against it here).
//  * the per-block partial sums (float) are reduced, completed (G1, band 0) and pushed through the design's backward by ONE
//    small launch, one wavefront per (band, pair): geq_bwd_lanes_kernel.
#include "common.h"
#include "response_common.h"
#include "rc_ba_body.h"
#include "fusedfwd.h"

namespace fl {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
// the packed pair / quadruple of a scalar type: float pairs are the operands of v_pk_*_f32; double "pairs" are two registers
// pairs and two v_fma_f64 -- the SAME issue time (a packed float instruction occupies the SIMD for two passes), which is why
// the float64 kernel below runs at the float one's speed where the lane-per-bin kernel of response.hip takes three times as long
template <typename T> struct Vec;
template <> struct Vec<float> { typedef f2 v2; typedef f4 v4; };
template <> struct Vec<double> { typedef d2 v2; typedef d4 v4; };
__device__ __forceinline__ float rcp_of(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_of(double x) {      // hardware estimate + two Newton steps (an IEEE division is ~15 instructions)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}

template <typename T>
struct LanesArgs {
    const double* b;       // (3, S, C) taps (designed by the forward launch)
    const double* a;
    int S, C;
    double g;              // anti-aliasing radius (gamma)
    const cx<double>* Wd;  // float64 master twiddles
    int nfft, bin0, m_local;
    const cx<T>* gH;       // cotangent planes: rc mode (row * NIW + n), plain mode c
    long g_pitch;
    const cx<T>* G;        // saved response, planes c
    long h_pitch;
    const T* Wr;           // (PPR, NIW) constant factor (rc mode)
    T* psum;               // (S * C, nbx, 4) partial sums (G0 of B, G0 of A, G2 of B, G2 of A)
    T* pq;                 // (C, nbx) partial sums of Re(q)
    T* partW;              // (PPR * NIW, nbx * pair groups) partial sums of the constant factor's gradient (rc mode)
    // "outer" mode: dL/dG[m][n] = sum_b gY[b][m] conj(X[b][n]) formed from the two signals
    const cx<T>* oG;
    const cx<T>* oX;
    long o_gb, o_gn, o_xb, o_xn;
    int oB;
    // geometry (lanes_plan)
    int npb, rows, seff, s_first, tb, tbp, ntiles, tph, half, L1, nlow, tiles_low, nbx;
    int lds1;              // bytes of ONE set of tile buffers (the kernel takes two)
    long long* stamps;     // tuning: per (block, wavefront) cycles spent in {lane-per-bin phase, first barrier, MFMA, lane-per-section phase, second barrier, whole kernel}
    int skip;              // tuning: 1 skips the lane-per-bin phase's work, 2 the lane-per-section phase's, 4: one order of the two in every wavefront
    unsigned pol;          // cache policy of the operand loads (common.h: POL_LANES_*)
};

// one (lane, bin): (B, A) of the lane's section in the halves of the packed values.  11 packed + 1 multiply + 1 reciprocal
// (a transcendental occupies the SIMD for four passes: ONE reciprocal of |B~|^2 |A~|^2 serves both polynomials).
// x = (1 -+ cos, sin); qa = (q.x, q.y), qb = (sin q.y, -sin q.x) as the lane-per-bin work left them: every scalar factor is a
// half of an aligned register pair, i.e. an operand selector
template <int NSUM, typename P2, typename T>
__device__ __forceinline__ void lane_bin(P2 C0, P2 C1, P2 C2, P2 x, T uu, P2 qa, P2 qb, P2& t0, P2& t1, P2& t2) {
    const P2 R = C1 * x.x + C0;
    const P2 I = C2 * x.y;
    P2 n = R * R;
    n = I * I + n;
    const T r = rcp_of(n.x * n.y);
    const P2 inv = P2{n.y, n.x} * r;                     // (1 / |B~|^2, 1 / |A~|^2)
    const P2 uR = R * inv, uI = I * inv;                 // P~ / |P~|^2
    // (one fused multiply-add per statement)
    t0 = uR * qa.x + t0;                                 // Re(q / P~)
    t0 = uI * qa.y + t0;
    t2 = uR * qb.x + t2;                                 // sin Im(q / P~)
    t2 = uI * qb.y + t2;
    if constexpr (NSUM == 3) {
        const P2 uq = qa * uu;
        t1 = uR * uq.x + t1;                             // (1 - cos) Re(q / P~)
        t1 = uI * uq.y + t1;
    }
}

// NIW > 0: constant-factor mode (rows of PPR pairs share NIW cotangent planes); NIW == 0, !OUTER: plain (PPR = 1);
// OUTER: rows of PPR pairs (m, n), cotangent formed from gY and X.
template <typename RT, int NIW, int PPR, int NSUM, bool OUTER>
__global__ void __launch_bounds__(sizeof(RT) == 8 ? 384 : 768, sizeof(RT) == 8 ? 2 : 3) sos_bwd_lanes_kernel(const LanesArgs<RT> A) {
    typedef typename Vec<RT>::v2 P2;
    typedef typename Vec<RT>::v4 P4;
    typedef cx<RT> cT;
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bx = blockIdx.x, cg = blockIdx.y;
    const int npb = A.npb, tbp = A.tbp, S = A.S, C = A.C;
    // two sets of tile buffers: the lane-per-bin work of tile t + 1 writes one while the lane-per-section work of tile t reads the other
    struct Bufs {
        P4* qs;             // [npb][tbp]  (q, sin * (-i q)), q = conj(gG) G; zero where nothing is due
        P2* xs;             // [tbp]       (1 -+ cos, sin)
        RT* us;          // [tbp]       1 - cos      (NSUM == 3)
    };
    auto bufs_at = [&](int which) {
        Bufs B;
        char* p = smem + (size_t)which * A.lds1;
        B.qs = reinterpret_cast<P4*>(p);
        B.xs = reinterpret_cast<P2*>(B.qs + (size_t)npb * tbp);
        B.us = reinterpret_cast<RT*>(B.xs + tbp);
        return B;
    };

    // ---- the lane's section: (B, A) polynomial pair turned by half a sample, about omega = 0 and about pi
    const int nl = npb * A.seff;
    const bool l_on = t < nl;
    const int sl = A.s_first + (l_on ? t / npb : 0), pl = l_on ? t % npb : 0;
    const int c = cg * npb + pl;
    P2 C0lo = {(RT)1.0, (RT)1.0}, C1lo = {(RT)0.0, (RT)0.0}, C0hi = {(RT)1.0, (RT)1.0}, C2 = {(RT)0.0, (RT)0.0};      // (C1hi = -C1lo)
    if (l_on) {
        const double g2 = A.g * A.g;
        const double b0 = A.b[(size_t)sl * C + c], b1 = A.b[(size_t)(S + sl) * C + c], b2 = A.b[(size_t)(2 * S + sl) * C + c];
        const double a0 = A.a[(size_t)sl * C + c], a1 = A.a[(size_t)(S + sl) * C + c], a2 = A.a[(size_t)(2 * S + sl) * C + c];
        const double SB = b0 + g2 * b2, TB = A.g * b1, DB = b0 - g2 * b2;
        const double SA = a0 + g2 * a2, TA = A.g * a1, DA = a0 - g2 * a2;
        C0lo = P2{(RT)(SB + TB), (RT)(SA + TA)};
        C1lo = P2{(RT)(-SB), (RT)(-SA)};
        C0hi = P2{(RT)(TB - SB), (RT)(TA - SA)};
        C2 = P2{(RT)DB, (RT)DA};
    }
    const bool wave_on = wave * 64 < nl;

    P2 acc0 = {(RT)0.0, (RT)0.0}, acc1 = {(RT)0.0, (RT)0.0}, acc2 = {(RT)0.0, (RT)0.0};
    // ---- tiles: elements f0 .. f0 + n - 1, one expansion point, bin of element f0 + i = kbase + i * kstep
    struct Tile {
        int f0, n, kbase, kstep;
        bool low;
    };
    auto tile_of = [&](int tile) {
        Tile T;
        if (A.bin0 < 0) {      // row-major bin order (spectral.hip): tiles never straddle a half row
            if (tile == A.ntiles - 1) {
                T.f0 = A.nfft >> 1; T.n = 1; T.low = false; T.kbase = A.nfft >> 1; T.kstep = 0;
            } else {
                const int hr = tile / A.tph, sub = tile - hr * A.tph;
                T.f0 = hr * A.half + sub * A.tb;
                T.n = A.tb;
                T.low = (hr & 1) == 0;
                T.kbase = (hr >> 1) + A.L1 * ((hr & 1) * A.half + sub * A.tb);
                T.kstep = A.L1;
            }
        } else {               // contiguous bins: tiles never straddle nfft / 4
            if (tile < A.tiles_low) {
                T.f0 = tile * A.tb; T.n = min(A.tb, A.nlow - T.f0); T.low = true;
            } else {
                T.f0 = A.nlow + (tile - A.tiles_low) * A.tb; T.n = min(A.tb, A.m_local - T.f0); T.low = false;
            }
            T.kbase = A.bin0 + T.f0;
            T.kstep = 1;
        }
        return T;
    };

    // ---- the lane-per-bin work: item (element i of the tile, row r, group jg of JPT pairs of the row), operands requested one
    // tile ahead (they arrive under the lane-per-section work in between).  Plane bases are wavefront-uniform, the lane's part
    // is ONE 32-bit byte offset per tensor (saddr + voffset loads: a 64-bit address per plane would be 32 more registers)
    constexpr bool RC = NIW > 0 && !OUTER;
    // pairs of the row per item (double: one -- two pairs' worth of constant-factor sums and cotangent rows in doubles do not fit the registers)
    constexpr int JPT = RC ? ((NIW >= 16 || sizeof(RT) == 8) ? 1 : (PPR >= 2 ? 2 : PPR)) : (OUTER ? PPR : 1);
    constexpr int JG = (RC || OUTER) ? PPR / JPT : 1;
    constexpr int NG = OUTER ? 2 : (NIW > 0 ? NIW : 1);        // cotangent values per item
    constexpr int TRIPS = (RC || OUTER) ? 1 : 4;               // items per thread and tile (plain mode: one per pair)
    struct Item {
        cT gv[NG];
        cT hv[JPT];
        cx<double> w1;
    };
    Item pre[TRIPS];
    const int nitems = A.tb * A.rows * JG;
    int it_i[TRIPS], it_r[TRIPS], it_j[TRIPS];
#pragma unroll
    for (int q = 0; q < TRIPS; ++q) {
        const int it = t + q * (int)blockDim.x;
        const int ir = it / JG;
        it_j[q] = (it - ir * JG) * JPT;
        it_i[q] = ir % A.tb;
        it_r[q] = it < nitems ? ir / A.tb : -1;
    }
    // constant-factor mode: the item's rows of W in registers, and its share of dL/dW[j][n] = sum Re(conj(G[row][j]) dL/dH[row][n]):
    // (re, im) products in the halves, summed over the items at the end.  (The contraction is 8 x 8 per bin: on
    // v_mfma_f32_16x16x4_f32 a quarter of a tile, two bins per instruction -- measured: 44 k cycles per workgroup against the
    // ~2 k of these packed multiply-adds.)
    constexpr int NWJ = RC ? JPT : 1, NWN = RC ? NIW : 1;
    RT wrow[NWJ][NWN];
    P2 aw[NWJ][NWN];
#pragma unroll
    for (int j = 0; j < NWJ; ++j)
#pragma unroll
        for (int nn = 0; nn < NWN; ++nn) {
            wrow[j][nn] = RC ? A.Wr[(it_j[0] + j) * NIW + nn] : (RT)0.0;
            aw[j][nn] = P2{(RT)0.0, (RT)0.0};
        }
    auto at = [](const cT* base, unsigned byte_off) {
        return *reinterpret_cast<const cT*>(reinterpret_cast<const char*>(base) + byte_off);
    };
    auto at_nt = [](const cT* base, unsigned byte_off) {
        typedef RT rv2 __attribute__((ext_vector_type(2)));
        const rv2 v = __builtin_nontemporal_load(reinterpret_cast<const rv2*>(reinterpret_cast<const char*>(base) + byte_off));
        return cT(v.x, v.y);
    };
    auto request = [&](const Tile& T) {
        const int np = (T.n + 1) & ~1;
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int i = it_i[q], r = it_r[q];
            if (r < 0 || i >= np) continue;
            const int ii = i < T.n ? i : T.n - 1;
            const int f = T.f0 + ii, k = T.kbase + ii * T.kstep;
            pre[q].w1 = A.Wd[k];                                     // (k <= nfft / 2)
            const int row = cg * A.rows + r;                         // global row
            if constexpr (OUTER) {
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    pre[q].gv[bb] = bb < A.oB ? A.oG[(size_t)bb * A.o_gb + (size_t)row * A.o_gn + f] : cT((RT)0.0, (RT)0.0);
                const unsigned hoff = (unsigned)(((size_t)(row * PPR) * A.h_pitch + f) * sizeof(cT));
#pragma unroll
                for (int j = 0; j < JPT; ++j) pre[q].hv[j] = at(A.G + (size_t)j * A.h_pitch, hoff);
            } else if constexpr (RC) {
                const unsigned goff = (unsigned)(((size_t)(row * NIW) * A.g_pitch + f) * sizeof(cT));
                const unsigned hoff = (unsigned)(((size_t)(row * PPR + it_j[q]) * A.h_pitch + f) * sizeof(cT));
                if (A.pol & POL_LANES_GH_NT) {
#pragma unroll
                    for (int nn = 0; nn < NG; ++nn) pre[q].gv[nn] = at_nt(A.gH + (size_t)nn * A.g_pitch, goff);
                } else {
#pragma unroll
                    for (int nn = 0; nn < NG; ++nn) pre[q].gv[nn] = at(A.gH + (size_t)nn * A.g_pitch, goff);
                }
                if (A.pol & POL_LANES_G_NT) {
#pragma unroll
                    for (int j = 0; j < JPT; ++j) pre[q].hv[j] = at_nt(A.G + (size_t)j * A.h_pitch, hoff);
                } else {
#pragma unroll
                    for (int j = 0; j < JPT; ++j) pre[q].hv[j] = at(A.G + (size_t)j * A.h_pitch, hoff);
                }
            } else {
                pre[q].gv[0] = A.gH[(size_t)row * A.g_pitch + f];
                pre[q].hv[0] = A.G[(size_t)row * A.h_pitch + f];
            }
        }
    };
    // the items' products into the tile buffers: q (with its quarter-turned, sine-weighted copy), the bins' (x, sin)
    RT qsum[TRIPS][JPT];
#pragma unroll
    for (int q = 0; q < TRIPS; ++q)
#pragma unroll
        for (int j = 0; j < JPT; ++j) qsum[q][j] = (RT)0.0;
    auto bin_work = [&](const Tile& T, const Bufs& B) {
        const int n = T.n, np = (n + 1) & ~1;       // an odd tail is padded with a copy of the last bin and a zero cotangent
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int i = it_i[q], r = it_r[q], j0 = it_j[q];
            if (r < 0 || i >= np) continue;
            const bool valid = i < n;
            const Item& P = pre[q];
            const RT sn = (RT)(-P.w1.y);      // sin(omega)
            if (r == 0 && j0 == 0) {
                B.xs[i] = P2{(RT)(T.low ? 1.0 - P.w1.x : 1.0 + P.w1.x), sn};
                if constexpr (NSUM == 3) B.us[i] = (RT)(1.0 - P.w1.x);
            }
            if constexpr (RC) {
#pragma unroll
                for (int j = 0; j < JPT; ++j) {
                    P2 gi = {(RT)0.0, (RT)0.0};
#pragma unroll
                    for (int nn = 0; nn < NIW; ++nn) gi = P2{P.gv[nn].x, P.gv[nn].y} * wrow[j][nn] + gi;
                    const cT h = valid ? P.hv[j] : cT((RT)0.0, (RT)0.0);
                    const bool live = valid && !(h.x == eps_of<RT>() && h.y == (RT)0.0);
                    const cT qv(gi.x * h.x + gi.y * h.y, gi.x * h.y - gi.y * h.x);      // conj(gi) h
                    const cT qz = live ? qv : cT((RT)0.0, (RT)0.0);
                    B.qs[(size_t)(r * PPR + j0 + j) * tbp + i] = P4{qz.x, qz.y, sn * qz.y, -(sn * qz.x)};
                    qsum[q][j] += qz.x;
#pragma unroll
                    for (int nn = 0; nn < NIW; ++nn) aw[j][nn] = P2{h.x, h.y} * P2{P.gv[nn].x, P.gv[nn].y} + aw[j][nn];
                }
            } else if constexpr (OUTER) {
                // row = output channel m, the PPR pairs of the row are its input channels n: dL/dG[m][n] = sum_b gY[b][m] conj(X[b][n])
                const int f = T.f0 + (valid ? i : n - 1);
#pragma unroll 4
                for (int j = 0; j < PPR; ++j) {
                    cT gi((RT)0.0, (RT)0.0);
                    for (int bb = 0; bb < A.oB && bb < 2; ++bb) fma_cxc(gi, P.gv[bb], A.oX[(size_t)bb * A.o_xb + (size_t)j * A.o_xn + f]);
                    const cT h = P.hv[j];
                    const bool live = valid && !(h.x == eps_of<RT>() && h.y == (RT)0.0);
                    const cT qv(gi.x * h.x + gi.y * h.y, gi.x * h.y - gi.y * h.x);
                    const cT qz = live ? qv : cT((RT)0.0, (RT)0.0);
                    B.qs[(size_t)(r * PPR + j) * tbp + i] = P4{qz.x, qz.y, sn * qz.y, -(sn * qz.x)};
                    qsum[q][j] += qz.x;
                }
            } else {
                const cT gi = P.gv[0], h = P.hv[0];
                const bool live = valid && !(h.x == eps_of<RT>() && h.y == (RT)0.0);
                const cT qv(gi.x * h.x + gi.y * h.y, gi.x * h.y - gi.y * h.x);
                const cT qz = live ? qv : cT((RT)0.0, (RT)0.0);
                B.qs[(size_t)r * tbp + i] = P4{qz.x, qz.y, sn * qz.y, -(sn * qz.x)};
                qsum[q][0] += qz.x;
            }
        }
    };

    const int t_begin = (int)((long)bx * A.ntiles / A.nbx), t_end = (int)((long)(bx + 1) * A.ntiles / A.nbx);
    long long st[6] = {0, 0, 0, 0, 0, 0};
    const long long k_start = A.stamps ? (long long)__builtin_readcyclecounter() : 0;
    const long long r_start = A.stamps ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#define FL_STAMP(slot, t_prev)                                                  \
    if (A.stamps) {                                                             \
        const long long now_ = (long long)__builtin_readcyclecounter();         \
        st[slot] += now_ - t_prev;                                              \
        t_prev = now_;                                                          \
    }
    // pipeline fill: the first tile's buffers, the second tile's operands on their way
    Tile T = tile_of(t_begin < t_end ? t_begin : 0), Tn = T;
    if (t_begin < t_end) {
        request(T);
        bin_work(T, bufs_at(0));
        if (t_begin + 1 < t_end) {
            Tn = tile_of(t_begin + 1);
            request(Tn);
        }
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        long long tp = A.stamps ? (long long)__builtin_readcyclecounter() : 0;
        const Bufs B = bufs_at((tile - t_begin) & 1);
        const int np = (T.n + 1) & ~1;
        const bool low = T.low;
        // The two halves of a trip are independent (the bin work fills the OTHER set of buffers): the wavefronts of a SIMD take
        // them in opposite orders -- wavefronts 4..7 the sections first -- so that one wavefront's packed arithmetic issues while
        // its neighbour on the SIMD waits for the bin work's operands and LDS stores (in one order everywhere the whole SIMD
        // stalls together behind the barrier).
        auto section_work = [&]() {
            if (!(wave_on && !(A.skip & 2))) return;
            const P2 C0 = low ? C0lo : C0hi, C1 = low ? C1lo : -C1lo;
            P2 t0 = {(RT)0.0, (RT)0.0}, t1 = {(RT)0.0, (RT)0.0}, t2 = {(RT)0.0, (RT)0.0};
            const P4* qrow = B.qs + (size_t)pl * tbp;
#pragma unroll 2
            for (int i = 0; i < np; i += 2) {
                P4 x4;
                if constexpr (sizeof(RT) == 4) {
                    x4 = *reinterpret_cast<const P4*>(B.xs + i);
                } else {
                    const P2 xa = B.xs[i], xb = B.xs[i + 1];
                    x4 = P4{xa.x, xa.y, xb.x, xb.y};
                }
                const P4 qA = qrow[i], qB = qrow[i + 1];
                RT u0 = (RT)0.0, u1 = (RT)0.0;
                if constexpr (NSUM == 3) {
                    const P2 u2 = *reinterpret_cast<const P2*>(B.us + i);
                    u0 = u2.x;
                    u1 = u2.y;
                }
                lane_bin<NSUM, P2, RT>(C0, C1, C2, P2{x4.x, x4.y}, u0, P2{qA.x, qA.y}, P2{qA.z, qA.w}, t0, t1, t2);
                lane_bin<NSUM, P2, RT>(C0, C1, C2, P2{x4.z, x4.w}, u1, P2{qB.x, qB.y}, P2{qB.z, qB.w}, t0, t1, t2);
            }
            // (no range test: an equaliser section has no zero on the sampling circle for finite positive gains -- B(1) = sqrt(g) (2 - 2 cos wc),
            // B(-1) = sqrt(g) (2 + 2 cos wc), the imaginary part g t sin(omega) -- and for a gain of exactly 0 the reference's own
            // gradient is not finite either (d sqrt(g) / dg); a vanishing norm shows as a non-finite gradient, as there)
            acc0 += t0;
            acc2 += t2;
            if constexpr (NSUM == 3) acc1 += t1;
        };
        const bool sections_first = !(A.skip & 4) && ((wave >> 2) & 1);
        if (sections_first) section_work();
        // ---- the NEXT tile's lane-per-bin work (its operands were requested a tile ago), then the request for the tile after it
        if (tile + 1 < t_end && !(A.skip & 1)) {
            bin_work(Tn, bufs_at((tile + 1 - t_begin) & 1));
        }
        T = Tn;
        if (tile + 2 < t_end) {
            Tn = tile_of(tile + 2);
            request(Tn);
        }
        FL_STAMP(0, tp)

        FL_STAMP(2, tp)

        // ---- lane-per-section work
        if (!sections_first) section_work();
        FL_STAMP(3, tp)
        __syncthreads();
        FL_STAMP(1, tp)
    }
    if (A.stamps && lane == 0) {
        st[5] = (long long)__builtin_readcyclecounter() - k_start;
        st[4] = (long long)__builtin_amdgcn_s_memrealtime() - r_start;      // (100 MHz)
        for (int v = 0; v < 6; ++v) A.stamps[((size_t)(bx * gridDim.y + cg) * (blockDim.x >> 6) + wave) * 6 + v] = st[v];
    }
#undef FL_STAMP

    // ---- partial sums of this block: [section * C + pair][block][4 sums] -- one 16-byte store per lane, and the reduction
    // reads an entry's partials as one run
    (void)acc1;
    if (l_on) {
        static_assert(NSUM == 2, "the partials are one 16-byte record per (entry, block)");
        reinterpret_cast<P4*>(A.psum)[((size_t)sl * C + c) * A.nbx + bx] = P4{acc0.x, acc0.y, acc2.x, acc2.y};
    }
    {   // sum Re(q) per pair: the items' sums through LDS (the tile buffers are done with), one thread per pair adds its tb items
        RT* qred = reinterpret_cast<RT*>(smem);      // [items][JPT]
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TRIPS; ++q)
            if (it_r[q] >= 0) {
                const int it = t + q * (int)blockDim.x;
#pragma unroll
                for (int j = 0; j < JPT; ++j) qred[(size_t)it * JPT + j] = qsum[q][j];
            }
        __syncthreads();
        if (t < npb) {
            const int r = t / (JPT * JG), jr = t - r * (JPT * JG), jg = jr / JPT, jj = jr - jg * JPT;
            RT tot = (RT)0.0;
            for (int i = 0; i < A.tb; ++i) tot += qred[(size_t)((r * A.tb + i) * JG + jg) * JPT + jj];
            A.pq[(size_t)(cg * npb + t) * A.nbx + bx] = tot;
        }
    }
    if constexpr (RC) {
        // dL/dW partials: every item thread's JPT x NIW sums through LDS, then one thread per entry adds the items of its pair
        // group in a fixed order: one (PPR, NIW) matrix per workgroup
        RT* wred = reinterpret_cast<RT*>(smem);      // [threads][JPT * NIW]  (the tile buffers are done with: reuse)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JPT; ++j)
#pragma unroll
            for (int nn = 0; nn < NIW; ++nn) wred[(size_t)t * (JPT * NIW) + j * NIW + nn] = it_r[0] >= 0 ? aw[j][nn].x + aw[j][nn].y : (RT)0.0;
        __syncthreads();
        if (t < PPR * NIW) {
            const int j = t / NIW, nn = t - j * NIW, jg = j / JPT, jj = j - jg * JPT;
            RT tot = (RT)0.0;
            const int nit = A.tb * A.rows;      // items of one pair group
            for (int u = 0; u < nit; ++u) tot += wred[(size_t)(jg + JG * u) * (JPT * NIW) + jj * NIW + nn];
            A.partW[(size_t)t * (A.nbx * gridDim.y) + (size_t)bx * gridDim.y + cg] = tot;
        }
    }
}

// ---------------------------------------------------------------- reduction + completion + design backward
// One wavefront per (band, pair) entry, its lanes striding over the nbx block partials (double sums, fixed butterfly:
// deterministic); the first lane recovers G1 from sum Re(t P~) = Q, forms the six tap gradients as the first-generation kernel's epilogue does
// (d/db0 = G0 - G1 - G2, d/db1 = g G0, d/db2 = g^2 (G0 - G1 + G2)) and runs the design's backward (geq_design_bwd).  Band 0
// is the pure gain: d/db0 = Q / b0.  Tail blocks: gW[e] = sum of the constant factor's partials, as geq_sections_bwd_kernel.
template <typename T>
__global__ void __launch_bounds__(256) geq_bwd_lanes_kernel(const void* __restrict__ gain, int in_kind,
                                                            const T* __restrict__ psum, const T* __restrict__ pq, int nbx,
                                                            const double* __restrict__ b, const double* __restrict__ a, double gam,
                                                            int nb, int C, const double* __restrict__ k, void* __restrict__ ggain,
                                                            int main_blocks, int epb, const T* __restrict__ partW, int wrows,
                                                            int wn, T* __restrict__ gW) {
    typedef typename Vec<T>::v4 P4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if ((int)blockIdx.x >= main_blocks) {
        // gW[e] = sum_r partW[e][r]: one wavefront per entry, the lanes stride over the (contiguous) rows, fixed butterfly
        const int e = ((int)blockIdx.x - main_blocks) * 4 + wave;
        if (e >= wn) return;
        T v = (T)0;
#pragma unroll 4
        for (int r = lane; r < wrows; r += 64) v += partW[(size_t)e * wrows + r];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) gW[e] = v;
        return;
    }
    // W wavefronts per (band, pair) entry (W = 4 / entries per workgroup): the entry's nbx partial records are ONE contiguous run
    // (the cascade kernel writes them transposed), each wavefront takes a quarter / half / all of it with every load issued
    // before the first is needed; the entry's own operands (gain, taps) are requested first, so that their round trips run
    // under the partials' ones
    __shared__ double red[4][5];
    const int W = 4 / epb, part = wave % W;
    const int idx0 = blockIdx.x * epb + wave / W;
    const size_t SC = (size_t)nb * C;
    const bool on = idx0 < nb * C;
    const int idx = on ? idx0 : 0;
    const int band = idx / C, c = idx - band * C;
    double raw;
    const double g = geq_linear_gain(gain, in_kind, idx, &raw);
    double tap[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        tap[0][p] = b[p * SC + idx];
        tap[1][p] = a[p * SC + idx];
    }
    double v[5] = {0, 0, 0, 0, 0};
    const int per = (nbx + W - 1) / W, b0 = part * per, b1 = min(nbx, b0 + per);
    const P4* ps4 = reinterpret_cast<const P4*>(psum) + (size_t)idx * nbx;
    if (on) {
        // four records per lane and trip, all eight loads requested before the first is added (a record beyond the range reads
        // the range's first one and is masked: with the guard around the load every trip was a round trip of its own -- four in
        // a row in a kernel that is nothing but latency)
        const T* pqc = pq + (size_t)c * nbx;
        for (int base = b0; base < b1; base += 256) {
            P4 r4[4];
            T q1[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bx = base + lane + 64 * u;
                ok[u] = bx < b1;
                const int bc = ok[u] ? bx : b0;
                r4[u] = ps4[bc];
                q1[u] = pqc[bc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double m = ok[u] ? 1.0 : 0.0, mb = (ok[u] && band > 0) ? 1.0 : 0.0;
                v[0] += mb * (double)r4[u].x; v[1] += mb * (double)r4[u].y; v[2] += mb * (double)r4[u].z; v[3] += mb * (double)r4[u].w;
                v[4] += m * (double)q1[u];
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int p = 0; p < 5; ++p) v[p] += __shfl_xor(v[p], o, 64);
    if (W > 1) {
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < 5; ++p) red[wave][p] = v[p];
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                double tot = red[wave][p];
                for (int w2 = 1; w2 < W; ++w2) tot += red[wave + w2][p];
                v[p] = tot;
            }
        }
    }
    if (lane != 0 || part != 0 || !on) return;
    const double Q = v[4];
    double out[2][3] = {{0, 0, 0}, {0, 0, 0}};
    // the closed forms divide by band 0's gain and by S = b0 + g^2 b2: a gain of exactly zero (|x| map at x = 0, an underflowed
    // sigmoid) would store 0/0 = NaN into the parameter's gradient where the product-rule gradient is finite -- such a tap
    // takes a zero gradient, as the first-generation epilogue gives it
    constexpr double kTiny = 1e-290;
    if (band == 0) {
        out[0][0] = fabs(tap[0][0]) > kTiny ? Q / tap[0][0] : 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double t0 = tap[i][0], t1 = tap[i][1], t2 = tap[i][2];
            const double Sg = t0 + gam * gam * t2, Tg = gam * t1, D = t0 - gam * gam * t2;
            const double sgn = i ? -1.0 : 1.0;
            const double G0 = sgn * v[i], G2 = sgn * v[2 + i];
            const double G1 = fabs(Sg) > kTiny ? ((Sg + Tg) * G0 - D * G2 - sgn * Q) / Sg : 0.0;
            out[i][0] = G0 - G1 - G2;
            out[i][1] = gam * G0;
            out[i][2] = gam * gam * (G0 - G1 + G2);
        }
    }
    const double dg = geq_design_bwd(band, nb, g, k, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2]);
    geq_store_gain_grad(ggain, in_kind, idx, dg, g, raw);
}

static int g_lanes = 1;        // 0: the first-generation kernels everywhere (test hook)
static int g_lanes_fwd = 1;    // ... of the forward kernel alone

// ---------------------------------------------------------------- forward: cascade response times a constant matrix (body: rc_ba_body.h)
template <int NIW>
__global__ void __launch_bounds__(256) sos_response_rc_ba_kernel(RcBaArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* dbg = A.dbg ? A.dbg + 8 * ((size_t)blockIdx.y * 4096 + blockIdx.x) : nullptr;      // (tuning: start / end of the workgroup)
    if (dbg && threadIdx.x == 0) dbg[4] = (long long)__builtin_amdgcn_s_memrealtime();
    rc_ba_body<NIW>(A, (int)blockIdx.x, (int)blockIdx.y, smem);
    if (dbg && threadIdx.x == 0) dbg[5] = (long long)__builtin_amdgcn_s_memrealtime();
}

// host side of the kernel above (called by response.hip: rc_impl); returns FL_ERR_UNSUPPORTED (no error text) when the shape is
// not taken, so that the caller falls back to the first generation
int rc_ba_launch_now(const PendingRc& p0, hipStream_t st) {
    PendingRc p = p0;
    p.args.dbg = pair_dbg() && p.gx <= 4096 ? pair_dbg() + 4 * 8192 : nullptr;
    const dim3 grid(p.gx, p.gy);
    if (p.niw == 2) hipLaunchKernelGGL((sos_response_rc_ba_kernel<2>), grid, dim3(256), p.lds, st, p.args);
    else if (p.niw == 4) hipLaunchKernelGGL((sos_response_rc_ba_kernel<4>), grid, dim3(256), p.lds, st, p.args);
    else if (p.niw == 8) hipLaunchKernelGGL((sos_response_rc_ba_kernel<8>), grid, dim3(256), p.lds, st, p.args);
    else hipLaunchKernelGGL((sos_response_rc_ba_kernel<16>), grid, dim3(256), p.lds, st, p.args);
    FL_CHECK_LAUNCH("sos_response_rc_ba");
    return FL_OK;
}

int rc_ba_launch(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd,
                 int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch, void* stream, GeqDesign gd) {
    if (!g_lanes_fwd || S < 2 || !(Ni == 2 || Ni == 4 || Ni == 8 || Ni == 16)) return FL_ERR_UNSUPPORTED;
    const int Seff = S - (gd.gain ? 1 : 0);
    PendingRc p;
    p.lds = (size_t)Nmid * 2 * Seff * 32 + (size_t)((Nmid * Ni + 3) & ~3) * 4 + (size_t)Nmid * S * 6 * 8;
    if (p.lds > 64 * 1024) return FL_ERR_UNSUPPORTED;
    const int npairs = bin0 >= 0 ? cdiv_i(m_local, 2) : (((nfft / 2 / (-bin0)) + 1) / 2) * (-bin0) + 1;
    p.gx = cdiv_i(npairs, 256);
    p.gy = No;
    p.niw = Ni;
    p.args = RcBaArgs{(const double*)b, (const double*)a, S, No * Nmid, Nmid, (const float*)Wr, gamma, (const cx<double>*)Wd, nfft, bin0,
                      m_local, (cx<float>*)G, g_pitch, (cx<float>*)H, h_pitch, gd, stream_policy()};
    if (pair_mode()) {        // recorded: the next forward column pass carries it (fusedfwd.h)
        pending_rc_put(p);
        return FL_OK;
    }
    return rc_ba_launch_now(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------- host: geometry
struct LanesPlan {
    int ok;
    int npb, ng, rows, seff, s_first, threads, tb, tbp, ntiles, tph, half, L1, nlow, tiles_low, nbx, tpw;
    size_t lds1;      // one set of tile buffers (the kernel takes two)
};

static int g_lanes_bpc = 0;    // resident workgroups per CU the grid is sized for (0: as many as 12 wavefronts per CU allow)
static int g_lanes_tb = 0;     // > 0: forced tile length
static long long* g_lanes_stamps = nullptr;
static int g_lanes_skip = 0;

static size_t lanes_lds_limit() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
    return (size_t)v;
}
// more dynamic LDS than the default 64 KB: the attribute is per function AND per device, set (and checked) once per pair
static int lanes_ensure_lds(const void* kern, size_t lds, bool* done /* [64] */) {
    if (lds <= 64 * 1024) return FL_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done[dev]) return FL_OK;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lanes_lds_limit());
    if (e != hipSuccess) {
        set_error("lanes cascade backward: %zu bytes of LDS per workgroup are not available on device %d (%s)", lds, dev, hipGetErrorString(e));
        return FL_ERR_UNSUPPORTED;
    }
    done[dev] = true;
    return FL_OK;
}
static int lanes_cus() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    return 256;
}

// mode: 0 plain (ppr = 1), 1 constant factor (ppr = N_mid, niw columns), 2 outer (ppr = N_in)
// esz: bytes of the kernels' real type (4: float, 8: double -- half the lanes per workgroup, twice the LDS per value)
static LanesPlan lanes_plan(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw, int mode, int esz = 4) {
    LanesPlan P{};
    static const int env_maxt = [] { const char* e = getenv("FLAMO_LANES_MAXT"); return e ? atoi(e) : 0; }();      // tuning: lanes per workgroup
    const int max_threads = esz == 8 ? 384 : (env_maxt >= 64 && env_maxt <= 768 ? env_maxt : 768), sc = esz / 4;
    if (!g_lanes || S < 4 || S > 64 || C < 1 || m_local < 1 || ppr < 1 || C % ppr) return P;
    if (mode == 1 && !((niw == 8 && ppr == 8) || (niw == 4 && ppr == 4) || (niw == 2 && ppr == 2) || (niw == 16 && ppr == 16))) return P;
    if (mode == 2 && !(ppr == 8 || ppr == 16 || ppr == 32)) return P;
    if (esz == 8 && (mode == 2 || (mode == 1 && niw >= 16))) return P;      // (double: sixteen cotangent rows per item spill)
    if (bin0 < 0 && ((-bin0) % 2 || (nfft / 2) % (-bin0) || m_local != nfft / 2 + 1)) return P;
    P.seff = S - 1;
    P.s_first = 1;
    // pairs per block: whole rows, a divisor of C, as many as 768 lanes take
    int npb = 0;
    for (int cand = ppr; cand <= C && cand * P.seff <= max_threads; cand += ppr)
        if (C % cand == 0) npb = cand;
    if (!npb || npb * P.seff < 48) return P;      // (tiny cascades: a lane per bin serves them better)
    P.npb = npb;
    P.ng = C / npb;
    P.rows = npb / ppr;
    P.threads = ((npb * P.seff + 63) / 64) * 64;
    const int nw = P.threads / 64;
    P.tpw = 1;
    // items of the lane-per-bin work per tile element: rows x groups of pairs -- at most one item per thread (plain mode: four)
    const int jpt = mode == 1 ? ((niw >= 16 || esz == 8) ? 1 : (ppr >= 2 ? 2 : ppr)) : ppr;
    const int per_elem = mode == 0 ? P.rows : P.rows * (ppr / jpt);
    const int tb_max = (mode == 0 ? 4 * P.threads : P.threads) / per_elem;
    if (tb_max < 8) return P;
    int bpc = g_lanes_bpc > 0 ? g_lanes_bpc : 12 / nw;      // (the kernel is built for three wavefronts per SIMD: 12 per CU)
    if (bpc * nw > 12) bpc = 12 / nw;
    // double: 213 registers with eight cotangent rows, i.e. two wavefronts per SIMD and ONE six-wavefront workgroup per CU (93 us at
    // config 2 against 131 for the lane-per-bin kernel; built for three per SIMD it spills 53 registers: 213 us)
    if (esz == 8) bpc = 1;
    if (bpc < 1) bpc = 1;
    const int slots = lanes_cus() * bpc;
    int nbx_target = slots / P.ng;
    if (nbx_target < 1) nbx_target = 1;
    if (bin0 < 0) {
        const int L2 = -bin0;
        P.half = L2 / 2;
        P.L1 = (nfft / 2) / L2;
        int best = 0;
        double best_score = -1;
        for (int d = 8; d <= 64 && d <= P.half && d <= tb_max; ++d) {
            if (P.half % d) continue;
            if (g_lanes_tb > 0 && d != g_lanes_tb) continue;
            const long nt = 2L * P.L1 * (P.half / d) + 1;
            const long nbx = nt < nbx_target ? nt : nbx_target;
            const long per = (nt + nbx - 1) / nbx;
            const double score = (double)nt / (double)(nbx * per) * d / (d + 6.0);
            if (score > best_score) {
                best_score = score;
                best = d;
            }
        }
        if (!best) return P;
        P.tb = best;
        P.tph = P.half / best;
        P.ntiles = 2 * P.L1 * P.tph + 1;
    } else {
        P.tb = g_lanes_tb > 0 ? g_lanes_tb : 32;
        if (P.tb > tb_max) P.tb = tb_max & ~1;
        const long q4 = ((long)nfft + 3) / 4;      // first bin with 4 k >= nfft
        long nlow = q4 - bin0;
        if (nlow < 0) nlow = 0;
        if (nlow > m_local) nlow = m_local;
        P.nlow = (int)nlow;
        P.tiles_low = (P.nlow + P.tb - 1) / P.tb;
        P.ntiles = P.tiles_low + (m_local - P.nlow + P.tb - 1) / P.tb;
    }
    P.nbx = P.ntiles < nbx_target ? P.ntiles : nbx_target;
    const int npad = (P.tb + 1) & ~1;
    P.tbp = npad + 1;      // pitch = 16 bytes x odd: the lanes' rows fall on different banks
    P.lds1 = ((size_t)P.npb * P.tbp * 16 + (size_t)(P.tbp + 1) * 8 + (size_t)((P.tbp + 3) & ~3) * 4) * sc;
    P.lds1 = (P.lds1 + 31) & ~(size_t)31;
    {
        const size_t items = mode == 0 ? (size_t)P.tb * P.rows : (size_t)P.threads;
        const size_t need = items * (mode == 1 ? jpt * (niw > 1 ? niw : 1) : jpt) * 4 * sc;      // (the epilogue's reductions reuse the buffers)
        if (2 * P.lds1 < need) P.lds1 = (need + 63) / 64 * 32;
    }
    if (2 * P.lds1 > lanes_lds_limit()) return P;
    if ((size_t)C * (size_t)m_local * 8 * sc >= (1ull << 32)) return P;      // (32-bit byte offsets into the response / cotangent planes)
    P.ok = 1;
    return P;
}

template <typename T>
static void lanes_fill(LanesArgs<T>& A, const LanesPlan& P) {
    A.npb = P.npb; A.rows = P.rows; A.seff = P.seff; A.s_first = P.s_first; A.tb = P.tb; A.tbp = P.tbp; A.ntiles = P.ntiles;
    A.tph = P.tph; A.half = P.half; A.L1 = P.L1; A.nlow = P.nlow; A.tiles_low = P.tiles_low; A.nbx = P.nbx;
    A.lds1 = (int)P.lds1;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_debug_set_cascade_lanes(int on, int blocks_per_cu, int tile_bins) {
    const int prev = g_lanes;
    if (on >= 0) {
        g_lanes = on & 1;
        g_lanes_fwd = (on & 1) && !(on & 2);      // (on = 3: second-generation backward, first-generation forward)
    }
    if (blocks_per_cu >= 0) g_lanes_bpc = blocks_per_cu;
    if (tile_bins >= 0) g_lanes_tb = tile_bins;
    return prev;
}

int fl_debug_set_cascade_stamps(void* device_buffer, int skip) {
    g_lanes_stamps = (long long*)device_buffer;
    g_lanes_skip = skip;
    return 0;
}

// rows of partW: one (Nmid, Ni) matrix per workgroup
int fl_geq_bwd_lanes_wrows(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw) {
    const LanesPlan P = lanes_plan(m_local, C, S, nfft, bin0, ppr, niw, 1);
    return P.ok ? P.nbx * P.ng : 0;
}
int fl_geq_bwd_lanes_wrows_f64(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw) {
    const LanesPlan P = lanes_plan(m_local, C, S, nfft, bin0, ppr, niw, 1, 8);
    return P.ok ? P.nbx * P.ng : 0;
}

int fl_geq_bwd_lanes_blocks(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw, int mode) {
    const LanesPlan P = lanes_plan(m_local, C, S, nfft, bin0, ppr, niw, mode);
    return P.ok ? P.nbx : 0;
}
int fl_geq_bwd_lanes_blocks_f64(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw, int mode) {
    const LanesPlan P = lanes_plan(m_local, C, S, nfft, bin0, ppr, niw, mode, 8);
    return P.ok ? P.nbx : 0;
}
}  // extern "C"

// mode 0: gH planes c (C = channel pairs); mode 1: gH planes (m * Ni + n), G planes (m * Nmid + j), Wr (Nmid, Ni), partW out
template <typename T>
static int lanes_bwd_impl(int mode, const void* gH, long g_pitch, const void* G, long h_pitch, const void* b, const void* a, int S, int No,
                          int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* psum,
                          void* pq, void* partW, void* stream) {
    FL_REQUIRE(mode == 0 || mode == 1, "geq_response_bwd_lanes: mode 0 (plain) or 1 (constant factor)");
    FL_REQUIRE(gH && G && b && a && Wd && psum && pq, "geq_response_bwd_lanes: null pointer");
    FL_REQUIRE(mode == 0 || (Wr && partW), "geq_response_bwd_lanes: the constant-factor mode needs Wr and partW");
    FL_REQUIRE(g_pitch >= m_local && h_pitch >= m_local, "geq_response_bwd_lanes: pitches must be >= m_local");
    const int C = No * Nmid, ppr = mode == 1 ? Nmid : 1;
    const LanesPlan P = lanes_plan(m_local, C, S, nfft, bin0, ppr, Ni, mode, (int)sizeof(T));
    if (!P.ok) {
        set_error("geq_response_bwd_lanes: unsupported shape (ask fl_geq_bwd_lanes_blocks first)");
        return FL_ERR_UNSUPPORTED;
    }
    LanesArgs<T> A{};
    A.b = (const double*)b; A.a = (const double*)a; A.S = S; A.C = C; A.g = gamma; A.Wd = (const cx<double>*)Wd;
    A.nfft = nfft; A.bin0 = bin0; A.m_local = m_local; A.gH = (const cx<T>*)gH; A.g_pitch = g_pitch;
    A.G = (const cx<T>*)G; A.h_pitch = h_pitch; A.Wr = (const T*)Wr; A.psum = (T*)psum; A.pq = (T*)pq;
    A.partW = (T*)partW;
    A.stamps = g_lanes_stamps;
    A.skip = g_lanes_skip;
    A.pol = stream_policy();
    lanes_fill(A, P);
    const dim3 grid(P.nbx, P.ng), block(P.threads);
    const size_t lds = 2 * P.lds1;
#define FL_LANES(NIW_, PPR_)                                                                                                     \
    {                                                                                                                            \
        static bool done[64] = {};                                                                                               \
        auto kern = sos_bwd_lanes_kernel<T, NIW_, PPR_, 2, false>;                                                               \
        const int rc_ = lanes_ensure_lds(reinterpret_cast<const void*>(kern), lds, done);                                        \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, A);                                                      \
    }
    if (mode == 0) FL_LANES(0, 1)
    else if (Ni == 8) FL_LANES(8, 8)
    else if (Ni == 4) FL_LANES(4, 4)
    else if (Ni == 2) FL_LANES(2, 2)
    else FL_LANES(16, 16)
#undef FL_LANES
    FL_CHECK_LAUNCH("geq_response_bwd_lanes");
    return FL_OK;
}

// psum / pq as left by the kernel above (nbx = fl_geq_bwd_lanes_blocks), b / a the designed taps -> ggain in the
// parameter's dtype (in_kind); wn > 0: gW[e] = sum_r partW[r * wn + e], r < wrows
template <typename T>
static int lanes_sections_bwd_impl(const void* gain, int in_kind, const void* psum, const void* pq, int nbx, const void* b, const void* a,
                                   double gamma, int nb, int C, const void* consts, void* ggain, const void* partW, int wrows, int wn,
                                   void* gW, void* stream) {
    FL_REQUIRE(gain && psum && pq && b && a && consts && ggain, "geq_sections_bwd_lanes: null pointer");
    FL_REQUIRE(in_kind >= 0 && in_kind <= 4 && nb >= 4 && C > 0 && nbx > 0, "geq_sections_bwd_lanes: bad sizes");
    FL_REQUIRE(wn == 0 || (partW && gW && wrows > 0), "geq_sections_bwd_lanes: bad constant-factor partials");
    const int epb = nbx > 512 ? 1 : nbx > 256 ? 2 : 4;      // entries per workgroup: 4 / 2 / 1 wavefronts per entry
    const int main_blocks = cdiv_i((long)nb * C, epb);
    hipLaunchKernelGGL(geq_bwd_lanes_kernel<T>, dim3(main_blocks + cdiv_i(wn, 4)), dim3(256), 0, (hipStream_t)stream, gain, in_kind,
                       (const T*)psum, (const T*)pq, nbx, (const double*)b, (const double*)a, gamma, nb, C,
                       (const double*)consts, ggain, main_blocks, epb, (const T*)partW, wrows, wn, (T*)gW);
    FL_CHECK_LAUNCH("geq_sections_bwd_lanes");
    return FL_OK;
}

extern "C" {
int fl_geq_response_bwd_lanes_c64(int mode, const void* gH, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                  int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                  int bin0, int m_local, void* psum, void* pq, void* partW, void* stream) {
    return lanes_bwd_impl<float>(mode, gH, g_pitch, G, h_pitch, b, a, S, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, psum, pq, partW, stream);
}
int fl_geq_response_bwd_lanes_c128(int mode, const void* gH, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                   int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                   int bin0, int m_local, void* psum, void* pq, void* partW, void* stream) {
    return lanes_bwd_impl<double>(mode, gH, g_pitch, G, h_pitch, b, a, S, No, Nmid, Ni, Wr, gamma, Wd, nfft, bin0, m_local, psum, pq, partW, stream);
}
int fl_geq_sections_bwd_lanes(const void* gain, int in_kind, const void* psum, const void* pq, int nbx, const void* b, const void* a,
                              double gamma, int nb, int C, const void* consts, void* ggain, const void* partW, int wrows, int wn,
                              void* gW, void* stream) {
    return lanes_sections_bwd_impl<float>(gain, in_kind, psum, pq, nbx, b, a, gamma, nb, C, consts, ggain, partW, wrows, wn, gW, stream);
}
int fl_geq_sections_bwd_lanes_f64(const void* gain, int in_kind, const void* psum, const void* pq, int nbx, const void* b, const void* a,
                                  double gamma, int nb, int C, const void* consts, void* ggain, const void* partW, int wrows, int wn,
                                  void* gW, void* stream) {
    return lanes_sections_bwd_impl<double>(gain, in_kind, psum, pq, nbx, b, a, gamma, nb, C, consts, ggain, partW, wrows, wn, gW, stream);
}
}  // extern "C"
