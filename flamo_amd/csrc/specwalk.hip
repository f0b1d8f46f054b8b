// Row kernels of the fused Shell pipeline that WALK THE BATCH (gfx950, MI355X).
//
// spectral.hip's spec_mid gives one workgroup one (row pair, batch item): the 2*L2*NO*NI response values of the row
// pair (245 KB at 8x8 channels, L2 = 240) then cross the L2 -> L1 path once per batch item -- 8x the signal's bytes,
// the largest single cost of that kernel.  Here ONE workgroup per CU is resident for the whole launch and
// owns a contiguous range of (row pair, batch item) units; the response of its current row pair sits in REGISTERS
// (thread (pair p, ms) holds H[2 bins][NO/MS][NI]: 128 VGPRs at 8x8 with MS = 2 -- half of the CU's register file holds
// the slice) and is fetched once per row pair and workgroup.  A round takes BG = 2 units through the five phases of
// spec_mid (row FFTs, split step, per-bin product, Hermitian pre-step, inverse row FFTs); the scratch rows of the NEXT
// round are requested right after the current round's have been consumed and stay in flight under phases 2-5, so the
// HBM round trip that is 20 % of a spec_mid workgroup's life is hidden.  Two LDS buffers (123 KB) alternate between
// the phases, which leaves four barriers per round and none between rounds.
//
//   forward  (spec_mid_walk):   S (B, L1, L2, NI) -> Y[f] = op(H[f]) X[f] -> S2 (B, L1, L2, NO).  If the backward pass will
//                               need the spectrum it is kept PAIR-MAJOR, Xp[unit][k | L-k][n/2][pair][n%2]: exactly what the
//                               product's thread holds, written and read back as whole 512-byte runs
//   backward (spec_gradh_walk): Sg, Xp -> dL/dH[m][n][f] = sum_b gY[b,m,f] conj(X[b,n,f]) accumulated in registers over the
//                               workgroup's batch slice: the gradient's row FFTs + split step happen in the kernel (its
//                               spectrum never exists in HBM) and the separate mimo_gradh pass (1.47x over-fetch) is
//                               gone.  Batch slices of a row pair are summed from per-slice partial planes
//                               (deterministic: no atomics).
// All of these kernels are bound by instruction issue, not by HBM (PMC: VALU busy 45 %, 36 % of wavefront cycles parked at
// barriers / waits, HBM at a third of its rate): the spectrum is kept rather than recomputed because eight row FFTs per
// unit cost more issue slots than 31 KB of coalesced traffic.
#include "spectral_common.h"

namespace fl {
using namespace sp32;      // these kernels exist in float32 only (spectral_common.h)

struct WalkArgs {
    const cf* S;          // (Bn, L1, L2, NI)
    cf* S2;               // (Bn, L1, L2, NO)
    const cf* H;          // H[m*hs_m + n*hs_n + i], row-major bin order
    long hs_m, hs_n;
    int conj_h;
    const cf* W;
    int n, L, L1, L2, Bn;
    float spec_scale;     // scale of the forward transform
    int spec_interior2;   // double the interior bins of the spectrum (irfft backward)
    int pre_half;         // halve the interior bins in front of the inverse transform (rfft backward)
    const int* bounds;    // work partition: workgroup w takes units [bounds[w], bounds[w+1]) (fl_spec_walk_partition), or null: equal counts
    cf* Xp;               // spectrum out, pair-major: Xp[(u*2 + e)*NI*LEN + ((n/2)*LEN + p)*2 + n%2], u = r*Bn + b, e = 0: bin k, 1: bin L-k; or null
    long long* dbg_times; // tuning: per-workgroup cycle stamps, or null
    unsigned pol;         // cache policy of the streams (common.h: POL_WALK_*)
    long long* stamps;    // measurement, or null: stamps[2 w], stamps[2 w + 1] = s_memrealtime when workgroup w started / finished --
                          // the launch's duration INSIDE a replayed graph (events cannot be recorded there), bench.py's roofline
};

// a copy of a per-lane value the optimiser cannot see through: what is derived from it is recomputed where it is used
// instead of being hoisted out of the unit loop and kept in registers across all five phases (the response slice takes
// half of the register budget; per-thread LDS addresses of five phases would take a fifth of the rest)
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// One complex value from LDS as ONE ds_read_b64.  hipcc merges two 8-byte reads at constant offsets into a ds_read2_b64, which
// the LDS services at half the rate (8 array cycles per wave-instruction for 1 KB against 2 for 512 B) and under the narrow
// banking rule -- and the start of every step is exactly an LDS read burst of all eight wavefronts (a third of step C).  The
// read is issued as inline assembly, so the compiler neither merges it nor knows it is in flight: lds_reads_done() waits for
// the LDS counter and pins the values behind the wait.
__device__ __forceinline__ f2 lds_rd(unsigned byte_addr, int off) {
    f2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "i"(off));
    return v;
}
// (the raw register pairs are pinned: any use of them ahead of the wait -- even a move into another register -- reads garbage)
template <int N>
__device__ __forceinline__ void lds_reads_done(f2 (&r)[N], cf (&v)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        asm volatile("" : "+v"(r[i]));
        v[i] = c2(r[i]);
    }
}
#ifndef FL_XP_PAIRS
#define FL_XP_PAIRS 1      // the kept spectrum in channel pairs (16-byte stores / LDS reads); 0: one channel plane per 8-byte access
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt16(cf* base, unsigned byte_off, f2 lo, f2 hi) {
    const f4 q = {lo.x, lo.y, hi.x, hi.y};
    __builtin_nontemporal_store(q, reinterpret_cast<f4*>(reinterpret_cast<char*>(base) + byte_off));
}

__device__ __forceinline__ void st_pl16(cf* base, unsigned byte_off, f2 lo, f2 hi) {
    const f4 q = {lo.x, lo.y, hi.x, hi.y};
    *reinterpret_cast<f4*>(reinterpret_cast<char*>(base) + byte_off) = q;
}

// ---------------------------------------------------------------- packed complex pieces (two floats per lane and instruction)
// Written out with 2-vectors: every line below is ONE v_pk_* instruction with the swaps / sign flips of complex arithmetic
// as operand selectors.  The kernels of this file are bound by instruction issue (a wavefront issues one instruction per
// four cycles whatever it is), so the count is what is optimised.
__device__ __forceinline__ f2 cj(f2 a) { return f2{a.x, -a.y}; }
__device__ __forceinline__ f2 rot_i(f2 a) { return f2{-a.y, a.x}; }       // i a
__device__ __forceinline__ f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 cmul2(f2 a, f2 b) { return pfma(f2{a.y, a.y}, rot_i(b), f2{a.x, a.x} * b); }      // a b
// acc + a b, with ib = i b at hand (shared by every a that multiplies this b)
__device__ __forceinline__ f2 cmac2(f2 acc, f2 a, f2 b, f2 ib) { return pfma(f2{a.y, a.y}, ib, pfma(f2{a.x, a.x}, b, acc)); }

// Real-FFT split step of the bin pair (k, L-k):  zk = Z[k], zm = Z[L-k] of the packed half-length transform, wk = W_n^k.
//   X[k] = sc (P - i t),  X[L-k] = sc conj(P + i t),  P = zk + conj(zm),  t = wk (zk - conj(zm))
// For the pair (0, L) pass zm = zk, wk = 1: the same lines give X[0] = 2 sc (re + im), X[L] = 2 sc (re - im).
// The constant factor is taken as the two vectors iw = i wk and niw = i iw = -wk, the variable one by broadcast halves: no
// rotation of a variable (a swap AND a sign flip of one half is not something the compiler folds into operand selectors).
__device__ __forceinline__ f2 cmulc(f2 b, f2 a, f2 ia) { return pfma(f2{b.y, b.y}, ia, f2{b.x, b.x} * a); }      // a b, ia = i a
// packed adds with per-half signs as operand modifiers (one instruction each; written out, "(a.x + b.x, a.y - b.y)" costs two
// adds and two moves)
#if FL_PK_DEV
#define FL_PK_ADD(name, mods, plain)                                                 \
    __device__ __forceinline__ f2 name(f2 a, f2 b) {                                 \
        f2 r;                                                                        \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));             \
        return r;                                                                    \
    }
#else
#define FL_PK_ADD(name, mods, plain) \
    __device__ __forceinline__ f2 name(f2 a, f2 b) { return plain; }
#endif
FL_PK_ADD(add_pm, "neg_hi:[0,1]", (f2{a.x + b.x, a.y - b.y}))                    // a + conj(b)
FL_PK_ADD(add_mp, "neg_lo:[0,1]", (f2{a.x - b.x, a.y + b.y}))                    // a - conj(b)
FL_PK_ADD(add_cj, "neg_hi:[1,1]", (f2{a.x + b.x, -a.y - b.y}))                   // conj(a + b)
FL_PK_ADD(sub_cj, "neg_lo:[0,1] neg_hi:[1,0]", (f2{a.x - b.x, b.y - a.y}))       // conj(a - b)
#undef FL_PK_ADD
// acc + x.y (-i g) = (acc.x + x.y g.y, acc.y - x.y g.x)
__device__ __forceinline__ f2 fma_mi(f2 x, f2 g, f2 acc) {
#if FL_PK_DEV
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(g));
    return acc;
#else
    return f2{acc.x + x.y * g.y, acc.y - x.y * g.x};
#endif
}
// iws = sc i wk, niws = -sc wk (the scale rides in the constant factor), sck = (sc, sc):  xk = sc (P - i t), xm = sc conj(P + i t)
__device__ __forceinline__ void split_pair(f2 zk, f2 zm, f2 iws, f2 niws, f2 sck, f2& xk, f2& xm) {
    const f2 Pp = sck * add_pm(zk, zm), D = add_mp(zk, zm);
    const f2 it = cmulc(D, iws, niws);          // sc i wk D
    xk = Pp - it;
    xm = add_cj(Pp, it);
}
// Hermitian pre-step of the inverse real FFT for the pair (k, L-k): from Y[k], Y[L-k] to Zf[k], Zf[L-k];
// cwk = conj(W_n^k).  For the pair (0, L) pass yk, ym with zeroed imaginary parts (cwk = 1 there).
__device__ __forceinline__ void pre_pair(f2 yk, f2 ym, f2 icw, f2 nicw, f2& zk, f2& zm) {      // icw = i conj(wk), nicw = -conj(wk)
    const f2 s_ = add_pm(yk, ym), d_ = add_mp(yk, ym);
    const f2 t_ = cmulc(d_, icw, nicw);         // i conj(wk) d
    zk = s_ + t_;
    zm = sub_cj(s_, t_);                        // conj(s - t)
}

// ---------------------------------------------------------------- forward: rows + split + product + pre-step + inverse rows
// One unit = rows r and L1 - r of all channels of one batch item.  512 threads in two groups of four wavefronts:
//   step A  all:  P3(u)   split step, product with the registers' response, Hermitian pre-step        Y  -> XI
//   step B  G0:   P4(u)   first stage of the inverse row FFTs, in place in XI
//           G1:   P1(u+1) first stage of the NEXT unit's forward row FFTs                           stage -> XF
//   step C  G0:   P5(u)   second stage, twiddle, store                                               XI -> S2
//           G1:   P2(u+1) second stage of the next unit's forward rows                               XF -> Y
// with one barrier behind each step; the scratch rows of unit u+2 are requested by LDS-DMA at the start of step C (the
// staging buffer is free once P1(u+1) has read it) and waited for at the end of step A.  Every SIMD always has one
// wavefront of each group: the FFT stages (work for 240..256 threads only) of two different units run side by side
// instead of leaving half of the wavefronts idle.  The pipeline drains at a row-pair boundary (new response).
// Row pitch of the LDS row buffers, in complex elements: 4 times an odd number (mod 32), so that the FFT stages' lane
// patterns (8 channels x 4 consecutive columns, or 8 channels x 4 column groups 15 apart) fall on 32 different 8-byte banks
// first-stage twiddle table [tb][ka] with an odd pitch: the four tb of a 32-lane group then sit on four different banks
constexpr int tw_pitch(int A) { return A | 1; }
constexpr int walk_pitch(int len) {
    int p = len;
    while (p % 8 != 4) ++p;
    return p;
}

template <int A, int B, int NI, int NO, int OCC, bool DBG = false>
__global__ void __launch_bounds__(512, OCC) spec_mid_walk(WalkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = walk_pitch(LEN), NCH = NI > NO ? NI : NO, MS = 2;
    constexpr int MO = NO / MS;                    // output channels per thread
    constexpr int UB = 2 * NCH * LENP;             // one row buffer: [2][NCH][LENP]
    constexpr int SB = 4 * A * 64;                 // staging: [forward-side wavefront][ta][lane]: see fetch()
    constexpr int NI1 = 2 * B * NI, NI4 = 2 * B * NO, NI2 = 2 * NI * A, NI5 = 2 * NO * A;
    static_assert(LEN <= 256, "one bin pair per thread");
    static_assert(NO % MS == 0, "output channels split over two threads");
    static_assert(NI % 2 == 0 && A % 2 == 0, "16-byte DMA granules are channel pairs; a 1-KB piece is two first-stage inputs of 64 items");
    static_assert(NI1 <= 256 && NI2 <= 256 && NI4 <= 256 && NI5 <= 256, "an FFT stage of one unit fits one group");
    cf* XF = reinterpret_cast<cf*>(smem);
    cf* Yb = XF + UB;
    cf* XI = Yb + UB;
    cf* stage = XI + UB;                           // [SB]
    constexpr int TWP = tw_pitch(A), TWL = (B * TWP + 1) & ~1;
    cf* tw = stage + SB;                           // W_LEN^(ka tb) at [tb][ka], pitch TWP
    cf* ws = tw + TWL;                             // W_n^(L1*k2)
    cf* wtab = ws + LEN;                           // [2][LEN]: conj(W_L^(row * c)) of the current row pair
    cf* dummy = wtab + 2 * LEN;                    // [64]: where lanes without a partner bin store
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = wave >> 2;                     // 0: inverse side, 1: forward side; also ms of the product
    const int P = a.L1 / 2 + 1;
    const int Utot = P * a.Bn;
    // XCD-aware order: logically consecutive workgroups (which share a row pair's response) sit on one XCD (block q runs
    // on XCD q % 8), so the slice comes from HBM once and from that L2 afterwards
    const int G = gridDim.x;
    const int w = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int u = a.bounds ? a.bounds[w] : (int)((long)Utot * w / G);
    const int u_hi = a.bounds ? a.bounds[w + 1] : (int)((long)Utot * (w + 1) / G);
    if (a.stamps && tid == 0) a.stamps[2 * blockIdx.x] = a.stamps[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    if (u >= u_hi) return;
    const size_t bstride_i = (size_t)a.L1 * a.L2 * NI, bstride_o = (size_t)a.L1 * a.L2 * NO;
    const unsigned stage_lds = lds_addr_of(stage);

    // The scratch rows of unit uu -> staging, by the forward-side wavefronts, each for ITSELF: wavefront wv fetches
    // exactly the A x 64 values its own 64 first-stage items will read, laid out [ta][lane], so that no other wavefront ever
    // touches its region: the transfer is issued as soon as the wavefront's own reads of the previous unit have returned
    // and awaited (vmcnt) just before its next reads -- no barrier is involved, and the rows have a whole pipeline cycle
    // to arrive.  One 1-KB piece = first-stage inputs ta = 2q, 2q+1 of the 64 items; a lane's 16 bytes = channels (n, n+1)
    // of one column.
    // Addressing: everything per-lane -- (row r or L1 - r, column, channel pair) -- is one 32-bit byte offset, rebuilt when the
    // row pair changes (fetch_row); the batch item enters through the wavefront-uniform base, and the fetches go through the
    // units in order, so (row pair, batch item) of the next fetch are two scalars that count (no division per fetch).
    unsigned fetch_voff = 0;
    int f_r = -1, f_b = 0;
    auto fetch_row = [&](int r) {
        const int rm = (a.L1 - r) % a.L1;
        const int wv = wave & 3, hi = lane >> 5;
        int item0 = 64 * wv + 2 * (lane & 31);
        if (item0 >= NI1 || (item0 >= B * NI && rm == r)) item0 = 0;          // no such item: any valid address
        const int nn0 = item0 % NI, tb = (item0 / NI) % B, slot = item0 / (NI * B);
        fetch_voff = 8u * ((unsigned)(slot ? rm : r) * (unsigned)(a.L2 * NI) + (unsigned)((hi * B + tb) * NI + nn0));
    };
    auto fetch = [&](int uu) {           // called for consecutive units uu
        if (f_r < 0) {
            f_r = uu / a.Bn;
            f_b = uu - f_r * a.Bn;
            fetch_row(f_r);
        }
        const char* base = reinterpret_cast<const char*>(a.S + (size_t)f_b * bstride_i);
        const int wv = wave & 3;
        if (a.pol & POL_WALK_S_NT) {
#pragma unroll
            for (int q = 0; q < A / 2; ++q)
                dma16sp<true>(base + (size_t)q * (2 * B * NI * 8), fetch_voff, stage_lds + (unsigned)(((wv * A + 2 * q) * 64) * 8));
        } else {
#pragma unroll
            for (int q = 0; q < A / 2; ++q)
                dma16sp<false>(base + (size_t)q * (2 * B * NI * 8), fetch_voff, stage_lds + (unsigned)(((wv * A + 2 * q) * 64) * 8));
        }
        if (++f_b == a.Bn) {             // the next fetch is for the next row pair
            f_b = 0;
            ++f_r;
            fetch_row(f_r < a.L1 / 2 + 1 ? f_r : 0);
        }
    };
    // ---- the FFT stages, each for the 256 threads of one group
    auto P1 = [&](bool selfm) {          // first stage of the forward rows, staging -> XF.  item = (slot, tb, n), n fastest
        const int item = opaque(tid) & 255;
        const int nn = item % NI, tb = (item / NI) % B, slot = item / (NI * B);
        const bool have = item < NI1 && !(slot && selfm);
        cf v[A], t[A];
        wait_vm0();                              // this wavefront's own transfer (fetch) has landed
        const cf* sp = stage + (item >> 6) * (A * 64) + (item & 63);
#pragma unroll
        for (int ta = 0; ta < A; ++ta) v[ta] = sp[ta * 64];
        // the twiddles are read with the data: behind the first LDS store the compiler cannot move a read any more (it
        // cannot tell the table from the row buffers), and 15 dependent read -> multiply -> store round trips would follow
#pragma unroll
        for (int ka = 1; ka < A; ++ka) t[ka] = tw[(have ? tb : 0) * TWP + ka];
        if (!have) return;
        RegFFT<float, A, false>::run(v);
        cf* uu = XF + (slot * NCH + nn) * LENP + tb;
        uu[0] = v[0];
#pragma unroll
        for (int ka = 1; ka < A; ++ka) uu[ka * B] = v[ka] * t[ka];
    };
    auto P2 = [&](bool selfm) {          // second stage XF -> Y, natural order.  item = (n fastest, ka, slot)
        const int item = opaque(tid) & 255;
        const int nn = item % NI, ka = (item / NI) % A, slot = item / (NI * A);
        if (item >= NI2 || (slot && selfm)) return;
        cf v[B];
        const cf* xr = XF + (slot * NCH + nn) * LENP + ka * B;
#pragma unroll
        for (int tb = 0; tb < B; ++tb) v[tb] = xr[tb];
        RegFFT<float, B, false>::run(v);
        cf* yr = Yb + (slot * NCH + nn) * LENP + ka;
#pragma unroll
        for (int kb = 0; kb < B; ++kb) yr[A * kb] = v[kb];
    };
    auto P4 = [&](bool selfm) {          // first stage of the inverse rows, in place in XI
        const int item = opaque(tid) & 255;
        const int m = item % NO, tb = (item / NO) % B, slot = item / (NO * B);
        if (item >= NI4 || (slot && selfm)) return;
        cf* uu = XI + (slot * NCH + m) * LENP + tb;
        cf v[A], t[A];
#pragma unroll
        for (int ta = 0; ta < A; ++ta) v[ta] = uu[ta * B];
#pragma unroll
        for (int ka = 1; ka < A; ++ka) t[ka] = tw[tb * TWP + ka];
        RegFFT<float, A, true>::run(v);
        uu[0] = v[0];
#pragma unroll
        for (int ka = 1; ka < A; ++ka) uu[ka * B] = mulc(v[ka], t[ka]);
    };
    long long q_ph[3] = {0, 0, 0}, q_t0 = 0;
    auto P5 = [&](bool selfm, int r, int rm, int b) {     // second stage, twiddle conj(W_L^(row*c)), store
        if (DBG) q_t0 = clock64();
        // item = (m fastest, ka, slot): the lanes of a store cover runs of (c = ka + A kb, m): 512 contiguous bytes per kb
        const int item = opaque(tid) & 255;
        const int m = item % NO, ka = (item / NO) % A, slot = item / (NO * A);
        if (item >= NI5 || (slot && selfm)) return;
        cf v[B], t[B];
        const unsigned uu = lds_addr_of(XI + (slot * NCH + m) * LENP + ka * B);
        const unsigned wt = lds_addr_of(wtab + slot * LEN + ka);
        f2 rv[B], rt[B];
#pragma unroll
        for (int tb = 0; tb < B; ++tb) rv[tb] = lds_rd(uu, tb * 8);
#pragma unroll
        for (int kb = 0; kb < B; ++kb) rt[kb] = lds_rd(wt, A * kb * 8);
        lds_reads_done(rv, v);
        lds_reads_done(rt, t);
        long long tq0 = 0;
        if (DBG) tq0 = clock64();
        RegFFT<float, B, true>::run(v);
        if (DBG) {
            const long long tq1 = clock64();
            q_ph[0] += tq0 - q_t0;       // reads
            q_ph[1] += tq1 - tq0;        // butterflies
            q_t0 = tq1;
        }
        cf* S2b = a.S2 + (size_t)b * bstride_o;
        const unsigned dst0 = (unsigned)(slot ? rm : r) * (unsigned)a.L2 * NO + (unsigned)(ka * NO + m);
        // (16-byte stores of channel pairs -- four DPP moves per column pair to buy one store instruction -- measured no faster:
        // 70.7-71.3 against 69.4-69.8 us in the step; the moves and their hazard slots cost what the stores saved)
        if (a.pol & POL_WALK_S2_PLAIN) {
#pragma unroll
            for (int kb = 0; kb < B; ++kb) at(S2b, 8u * (dst0 + (unsigned)(A * kb * NO))) = v[kb] * t[kb];
        } else {
#pragma unroll
            for (int kb = 0; kb < B; ++kb) st_nt(S2b, 8u * (dst0 + (unsigned)(A * kb * NO)), v[kb] * t[kb]);
        }
        if (DBG) q_ph[2] += clock64() - q_t0;     // twiddles + stores
    };

    if (grp == 1) fetch(u);
    for (int j = tid; j < LEN; j += 512) {         // contiguous copies behind the master table (fl_spec_aux_fill_f32)
        tw[(j / A) * TWP + j % A] = a.W[a.n + a.L1 + (j / A) * (j % A)];      // a thread's A first-stage twiddles are one run
        ws[j] = a.W[a.n + a.L1 + a.L2 + j];
    }
    const float hs = 0.5f * a.spec_scale, wi = a.spec_interior2 ? 2.f : 1.f;
    const float ph = a.pre_half ? 0.5f : 1.f;
    long long t_ph[4] = {0, 0, 0, 0}, t_last = 0, w_ph[3] = {0, 0, 0};
    if (DBG) t_last = clock64();
    const long long t_begin = t_last;
#define FL_STAMP(i)                         \
    if (DBG) {                              \
        const long long t_now = clock64();  \
        t_ph[i] += t_now - t_last;          \
        t_last = t_now;                     \
    }
    lds_barrier();                                  // the tables are visible

    while (u < u_hi) {
        // ================= a row pair: its response into registers, its tables into LDS, the pipeline filled
        const int r = u / a.Bn;
        const int rm = (a.L1 - r) % a.L1;
        const bool selfm = rm == r;
        const int seg_end = (r + 1) * a.Bn < u_hi ? (r + 1) * a.Bn : u_hi;
        // the product's thread: bin pair p of the row pair, output channels [grp MO, (grp + 1) MO)
        const int p = tid & 255;
        int slotB = 0, colB = 0;
        bool dc = false;
        const bool valid = p < LEN && pair_of(r, selfm, p, LEN, slotB, colB, dc);
        const bool twin = valid && !dc && !(slotB == 0 && colB == p);    // the partner bin has a place of its own in the rows
        const int pc = p < LEN ? p : 0;
        f2 h[2][MO][NI];
        {
            const unsigned ik = (unsigned)r * LEN + (unsigned)pc;
            unsigned im = dc ? (unsigned)a.L : (unsigned)(slotB ? rm : r) * LEN + colB;
            if (!valid) im = ik;
#pragma unroll
            for (int m2 = 0; m2 < MO; ++m2) {
                // one base + 32-bit lane offsets stepped from plane to plane (a scalar base or offset per plane is loop
                // invariant: the compiler would pin one or two SGPRs per plane for the whole kernel)
                unsigned ok = 8u * ((unsigned)(grp * MO + m2) * (unsigned)a.hs_m + ik), om = ok + 8u * (im - ik);
                const unsigned step = 8u * (unsigned)a.hs_n;
#pragma unroll
                for (int nn = 0; nn < NI; ++nn) {
                    h[0][m2][nn] = v2(at(a.H, ok));
                    h[1][m2][nn] = v2(at(a.H, om));
                    ok += step;
                    om += step;
                }
            }
        }
        // conj(W_L^(row c)) = conj(W_n^(2 row c)), 2 row c < n: row r from group 0's threads, row L1 - r from group 1's
        // (the previous pair's last P5 read the table in its last step, a barrier ago)
        const cf wt_mine = conj(a.W[2 * (grp ? rm : r) * pc]);
        const f2 wk = v2(a.W[r] * ws[pc]);                      // W_n^k of the pair's first bin, k = r + L1 p
        const f2 icw = f2{wk.y, wk.x}, nicw = f2{-wk.x, wk.y};
        const float sc = dc ? hs : hs * wi;                     // the pair (0, L) is not interior
        const f2 sck = f2{sc, sc}, iw = sc * rot_i(wk), niw = f2{-sc * wk.x, -sc * wk.y};      // the split step's scale rides in its twiddle
        const f2 phv = dc ? f2{1.f, 0.f} : f2{ph, ph};          // ... and only its real parts enter the inverse transform
        const int yk_o = pc, ym_o = slotB * NCH * LENP + colB;  // where the pair's bins sit in a row buffer
        cf* zm_dst = twin ? XI + ym_o : dummy + lane;           // lanes without a partner bin of their own store aside
        if (grp == 1) P1(selfm);
        if (p < LEN) wtab[grp * LEN + p] = wt_mine;
        lds_barrier();
        if (grp == 1) P2(selfm);
        if (a.conj_h) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int m2 = 0; m2 < MO; ++m2)
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) h[e][m2][nn].y = -h[e][m2][nn].y;
        }
        // every load above has landed: the compiler's own waits end here, outside the unit loop.  The next unit's rows are
        // requested only now: waiting for the response must not mean waiting for them
        wait_vm0();
        if (grp == 1 && u + 1 < u_hi) fetch(u + 1);
        lds_barrier();
        FL_STAMP(0)

#pragma unroll 1
        for (; u < seg_end; ++u) {
            const bool next = u + 1 < seg_end;      // the forward side works on the next unit of this row pair
            // ---- step A / P3: split step, product, Hermitian pre-step: Y -> XI
            // (the response registers are made opaque once per unit: whatever the product derives from them -- the broadcast
            // halves a packed multiply takes as operand selectors for free -- is then formed here, per use, instead of being
            // hoisted out of the unit loop as a second, twice as large, copy of the slice)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int m2 = 0; m2 < MO; ++m2)
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) asm volatile("" : "+v"(h[e][m2][nn]));
            if (valid) {
                const cf* yk_p = Yb + yk_o;
                const cf* ym_p = Yb + ym_o;
                f2 xk[NI], xm[NI];
#pragma unroll
                for (int nn = 0; nn < NI; ++nn) split_pair(v2(yk_p[nn * LENP]), v2(ym_p[nn * LENP]), iw, niw, sck, xk[nn], xm[nn]);
                if (a.Xp) {      // group 0 keeps the bins k, group 1 the bins L-k (both hold both); grp is wavefront-uniform: a branch
                    // channel PAIRS are the unit of the layout: one 16-byte store per pair (8-byte stores are bound by their issue
                    // rate, not by bytes: half as many instructions for the same data), and the backward kernel reads them back
                    // from its staging buffer as one ds_read_b128
#if FL_XP_PAIRS
                    cf* xo = a.Xp + ((size_t)u * 2 + grp) * (NI * LEN) + 2 * p;
                    if (a.pol & POL_WALK_XP_PLAIN) {
                        if (grp) {
#pragma unroll
                            for (int j = 0; j < NI / 2; ++j) st_pl16(xo, 16u * (unsigned)(j * LEN), xm[2 * j], xm[2 * j + 1]);
                        } else {
#pragma unroll
                            for (int j = 0; j < NI / 2; ++j) st_pl16(xo, 16u * (unsigned)(j * LEN), xk[2 * j], xk[2 * j + 1]);
                        }
                    } else if (grp) {
#pragma unroll
                        for (int j = 0; j < NI / 2; ++j) st_nt16(xo, 16u * (unsigned)(j * LEN), xm[2 * j], xm[2 * j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NI / 2; ++j) st_nt16(xo, 16u * (unsigned)(j * LEN), xk[2 * j], xk[2 * j + 1]);
                    }
#else
                    cf* xo = a.Xp + ((size_t)u * 2 + grp) * (NI * LEN) + p;
                    if (grp) {
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) st_nt(xo, 8u * (unsigned)(nn * LEN), c2(xm[nn]));
                    } else {
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) st_nt(xo, 8u * (unsigned)(nn * LEN), c2(xk[nn]));
                    }
#endif
                }
                cf* zk_p = XI + yk_o + grp * MO * LENP;
                cf* zm_p = zm_dst + (twin ? grp * MO * LENP : 0);
#pragma unroll
                for (int m2 = 0; m2 < MO; ++m2) {
                    // sum_n h x = sum_n Re(h) x + i sum_n Im(h) x: two plain packed chains per bin, one rotation at the end
                    f2 ka_ = {0.f, 0.f}, kb_ = {0.f, 0.f}, ma_ = {0.f, 0.f}, mb_ = {0.f, 0.f};
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        ka_ = pfma(f2{h[0][m2][nn].x, h[0][m2][nn].x}, xk[nn], ka_);
                        kb_ = pfma(f2{h[0][m2][nn].y, h[0][m2][nn].y}, xk[nn], kb_);
                        ma_ = pfma(f2{h[1][m2][nn].x, h[1][m2][nn].x}, xm[nn], ma_);
                        mb_ = pfma(f2{h[1][m2][nn].y, h[1][m2][nn].y}, xm[nn], mb_);
                    }
                    const f2 yk = pk_add_i(ka_, kb_), ym = pk_add_i(ma_, mb_);
                    f2 zk, zm;
                    pre_pair(phv * yk, phv * ym, icw, nicw, zk, zm);
                    zk_p[m2 * LENP] = c2(zk);
                    zm_p[twin ? m2 * LENP : 0] = c2(zm);
                }
            }
            if (DBG) w_ph[0] += clock64() - t_last;     // own work of this wavefront in the step (before it waits at the barrier)
            lds_barrier();
            FL_STAMP(1)
            // ---- step B
            if (grp == 0) P4(selfm);
            else if (next) P1(selfm);
            if (DBG) w_ph[1] += clock64() - t_last;
            lds_barrier();
            FL_STAMP(2)
            // ---- step C.  The forward side's own staging regions are free (its reads of step B are behind the barrier): the rows
            // of unit u+2 are requested here, beside the LIGHTER of the two second stages (the transfer's issue costs ~100
            // cycles per 1-KB piece: in step B it made the forward side the slower half)
            if (grp == 0) P5(selfm, r, rm, u - r * a.Bn);
            else if (next) {
                if (u + 2 < u_hi) fetch(u + 2);
                P2(selfm);
            }
            if (DBG) w_ph[2] += clock64() - t_last;
            lds_barrier();
            FL_STAMP(3)
        }
    }
    if (a.stamps && tid == 0) {      // (behind this wavefront's last stores: the kernel's end as the memory system sees it)
        wait_vm0();
        a.stamps[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (DBG && a.dbg_times && tid == 0) {
        long long* o = a.dbg_times + (size_t)blockIdx.x * 8;
        o[0] = t_begin;
        o[1] = clock64();
#pragma unroll
        for (int i = 0; i < 4; ++i) o[2 + i] = t_ph[i];
    }
    // own work per step of one wavefront of each group, three 21-bit fields (units of 16 cycles): A | B << 21 | C << 42
    if (DBG && a.dbg_times && (tid == 0 || tid == 256))
        a.dbg_times[(size_t)blockIdx.x * 8 + 6 + grp] = grp == 0 && a.spec_interior2 == 7
            ? (q_ph[0] >> 4) | ((q_ph[1] >> 4) << 21) | ((q_ph[2] >> 4) << 42)       // tuning: P5's reads | butterflies | twiddles + stores
            : (w_ph[0] >> 4) | ((w_ph[1] >> 4) << 21) | ((w_ph[2] >> 4) << 42);
#undef FL_STAMP
}

// ---------------------------------------------------------------- backward: dL/dH accumulated over a batch slice
// dH[m][n][f] = sum_b gY[b,m,f] conj(X[b,n,f]),  gY = weighted spectrum of the rows in Sg (K1 of the output's gradient),
// X = the spectrum the forward kernel kept pair-major.  Workgroup (row pair r, batch slice s) walks its items; thread
// (pair p, e) owns ONE bin -- k of the pair if e = 0 (group 0), L-k if e = 1 (group 1) -- and keeps dH[NO][NI] of it in
// registers, written once into partial plane set s.  Software pipeline over the items, two barriers per item:
//   step 1  all:  P3(b-2)  split step of the gradient (the thread's side of the pair), X from staging, accumulate
//   step 2  G0:   P1(b)    first stage of the gradient's row FFTs                                  stage_g -> XF[b & 1]
//           G1:   P2(b-1)  second stage                                                            XF[(b-1) & 1] -> Y
// Staging is owner-wave (see spec_mid_walk's fetch): a group-0 wavefront transfers the gradient rows its own P1 items
// read, every wavefront the 4 KB of spectrum its own P3 threads read (with one bin per thread the two groups read
// different halves of a unit's block) -- each transfer issued right after the owner's reads of the item that used the
// region and awaited just before its next ones, a full pipeline cycle later (DEPTH 1: every shape but 8 -> 8 channels).
// DEPTH 2 doubles the staging regions and moves the transfers' issue to the wavefronts that have no FFT stage in step 2
// (8 -> 8 channels since late round 4: 54.2 against 56.5 us; in round 3, with both FFT stages still on SIMD 0 and 1, it
// measured no faster -- 69.5 against 68.7 us, and 66.8 us with the barrier between the two steps removed altogether).
struct GradhArgs {
    const cf* Sg;         // (Bn, L1, L2, NO)
    const cf* Xp;         // pair-major spectrum of the forward kernel
    cf* dH;               // dH[s*ds_s + m*ds_m + n*ds_n + i], row-major bin order, s < NS
    long ds_s, ds_m, ds_n;
    const cf* W;
    int n, L, L1, L2, Bn, NS;
    float scale_g;        // scale of the gradient's forward transform (the inverse transform's scale)
    const float* out_scale;   // device scalar multiplied into dH on the way out, or null (the objective's 2 g / N: see ops.mean_square)
    int interior2_g;      // double its interior bins (irfft backward)
    long long* dbg_times; // tuning: 8 int64 per workgroup (begin, end, cycles in step 1, in step 2, per wavefront 0 / 1 / 4 / 7 of step 2)
    unsigned pol;         // cache policy of the streams (common.h: POL_GRADH_*)
};

template <int A, int B, int NI, int NO, int NSC, int OCC, int DEPTH, bool DBG = false>
__global__ void __launch_bounds__(512, OCC) spec_gradh_walk(GradhArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = walk_pitch(LEN);
    constexpr int NOL = NO / NSC;                  // gradient channels of this workgroup: [mo, mo + NOL)
    constexpr int UB = 2 * NOL * LENP;             // a row buffer of the gradient: [2][NOL][LENP]
    constexpr int NI1 = 2 * B * NOL, NI2 = 2 * NOL * A;
    constexpr int NW1 = (NI1 + 63) / 64;           // group-0 wavefronts that have first-stage items
    constexpr int SBG = NW1 * A * 64;              // gradient staging: [group-0 wavefront][ta][lane]
    constexpr int SBX = 8 * NI * 64;               // spectrum staging: [wavefront][n][lane]
    static_assert(LEN <= 256, "one bin pair per thread pair");
    static_assert(NO % NSC == 0 && NOL % 2 == 0 && NI % 2 == 0 && A % 2 == 0, "16-byte DMA granules");
    static_assert(NI1 <= 256 && NI2 <= 256, "an FFT stage of one unit fits one group");
    static_assert(DEPTH == 1 || DEPTH == 2, "one or two items in flight");
    constexpr int GA = A / 2, XA = NI / 2;         // transfers (instructions) of one item's rows / spectrum block per wavefront
    constexpr int P2_OFF = 256 - ((NI2 + 63) / 64) * 64;       // second-stage items on the last wavefronts of group 1
    cf* XF = reinterpret_cast<cf*>(smem);          // [2][UB]
    cf* Yb = XF + 2 * UB;
    cf* stage_g = Yb + UB;                         // [DEPTH][SBG]
    cf* stage_x = stage_g + DEPTH * SBG;           // [DEPTH][SBX]
    constexpr int TWP = tw_pitch(A), TWL = (B * TWP + 1) & ~1;
    cf* tw = stage_x + DEPTH * SBX;                // W_LEN^(ka tb) at [tb][ka], pitch TWP
    cf* ws = tw + TWL;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = wave >> 2;
    const int P = a.L1 / 2 + 1;
    // XCD-aware order: the slices of a row pair on one XCD (it keeps a pair's partial planes' lines in one L2)
    const int xcd = blockIdx.x & 7, q_ = blockIdx.x >> 3;
    const int r = (q_ / (a.NS * NSC)) * 8 + xcd, sl = (q_ / NSC) % a.NS, mo = (q_ % NSC) * NOL;
    if (r >= P) return;
    const int rm = (a.L1 - r) % a.L1;
    const bool selfm = rm == r;
    const int b_lo = (int)((long)a.Bn * sl / a.NS), n_it = (int)((long)a.Bn * (sl + 1) / a.NS) - b_lo;
    const size_t bstride_g = (size_t)a.L1 * a.L2 * NO;
    const unsigned sg_lds = lds_addr_of(stage_g), sx_lds = lds_addr_of(stage_x);

    // cw: the CONSUMER wavefront whose staging region is filled (the wavefront itself when it transfers for itself)
    auto fetch_g = [&](int i, int cw) {  // the A x 64 first-stage inputs of group-0 wavefront cw's own items, item i of the slice
        const int b = b_lo + ((a.pol & POL_GRADH_FORWARD) ? i : n_it - 1 - i);      // (last to first: common.h)
        const unsigned sg_buf = sg_lds + (unsigned)((i % DEPTH) * SBG * 8);
        const int hi = lane >> 5;
        int item0 = 64 * cw + 2 * (lane & 31);
        if (item0 >= NI1 || (item0 >= B * NOL && selfm)) item0 = 0;
        const int m0 = item0 % NOL, tb = (item0 / NOL) % B, slot = item0 / (NOL * B);
        const cf* src = a.Sg + (size_t)b * bstride_g + (size_t)(slot ? rm : r) * (a.L2 * NO) + ((hi * B + tb) * NO + mo + m0);
        if (a.pol & POL_GRADH_SG_NT) {
#pragma unroll
            for (int q = 0; q < A / 2; ++q) dma16p<true>(src + q * (2 * B * NO), sg_buf + (unsigned)(((cw * A + 2 * q) * 64) * 8));
        } else {
#pragma unroll
            for (int q = 0; q < A / 2; ++q) dma16p<false>(src + q * (2 * B * NO), sg_buf + (unsigned)(((cw * A + 2 * q) * 64) * 8));
        }
    };
    auto fetch_x = [&](int i, int cw) {  // X[n][the 64 pairs of wavefront cw] of its side of the unit's block
        const int b = b_lo + ((a.pol & POL_GRADH_FORWARD) ? i : n_it - 1 - i);      // (last to first: common.h)
        const unsigned sx_buf = sx_lds + (unsigned)((i % DEPTH) * SBX * 8);
#if FL_XP_PAIRS
        int pp = 64 * (cw & 3) + lane;           // a lane's 16 bytes: channels (2q, 2q + 1) of one bin pair
        if (pp > LEN - 1) pp = LEN - 1;
        const cf* src = a.Xp + (((size_t)r * a.Bn + b) * 2 + (cw >> 2)) * (NI * LEN) + 2 * pp;
#else
        int pp = 64 * (cw & 3) + 2 * (lane & 31);
        if (pp > LEN - 2) pp = LEN - 2;
        const cf* src = a.Xp + (((size_t)r * a.Bn + b) * 2 + (cw >> 2)) * (NI * LEN) + (lane >> 5) * LEN + pp;
#endif
        if (a.pol & POL_GRADH_XP_NT) {
#pragma unroll
            for (int q = 0; q < NI / 2; ++q) dma16p<true>(src + q * (2 * LEN), sx_buf + (unsigned)(((cw * NI + 2 * q) * 64) * 8));
        } else {
#pragma unroll
            for (int q = 0; q < NI / 2; ++q) dma16p<false>(src + q * (2 * LEN), sx_buf + (unsigned)(((cw * NI + 2 * q) * 64) * 8));
        }
    };

    // the product's thread
    const int p = tid & 255;
    int slotB = 0, colB = 0;
    bool dc = false;
    const bool valid = p < LEN && pair_of(r, selfm, p, LEN, slotB, colB, dc);
    const int pc = p < LEN ? p : 0;
    f2 acc[NOL][NI];
#pragma unroll
    for (int m = 0; m < NOL; ++m)
#pragma unroll
        for (int nn = 0; nn < NI; ++nn) acc[m][nn] = f2{0.f, 0.f};
    const bool has_g = grp == 0 && (wave & 3) < NW1;             // this wavefront has first-stage items
    // DEPTH 2: the transfers are issued by the wavefronts that have NO stage of their own in step 2 (2 .. 5: the first stage
    // sits on wavefronts 0 .. NW1-1, the second on the last ones), for the others as well as for themselves -- issuing a
    // 1-KB piece costs a wavefront ~100 cycles, and with every wavefront fetching for itself that sat on the two longest
    // paths: 4 pieces per wavefront in step 1, 8 more in front of the first stage's arithmetic.  Fetcher f = wave - 2 fills
    // the rows of first-stage wavefront f (if there is one) and the spectrum blocks of wavefronts XLO[f] .. XLO[f+1]-1.
    const bool fetcher = DEPTH == 2 && wave >= 2 && wave <= 5;
    const int fi = wave - 2;
    const int xlo = NW1 >= 2 ? (fi == 0 ? 0 : fi == 1 ? 1 : fi == 2 ? 2 : 5) : (fi == 0 ? 0 : fi == 1 ? 1 : fi == 2 ? 3 : 6);
    const int xhi = NW1 >= 2 ? (fi == 0 ? 1 : fi == 1 ? 2 : fi == 2 ? 5 : 8) : (fi == 0 ? 1 : fi == 1 ? 3 : fi == 2 ? 6 : 8);
    int x_issued = 0;                                            // spectrum pieces this fetcher issued in the last step 2
    if (DEPTH == 2) {
        if (fetcher && fi < NW1 && n_it > 0) fetch_g(0, fi);
    } else if (has_g && n_it > 0) {
        fetch_g(0, wave & 3);
    }
    for (int j = tid; j < LEN; j += 512) {
        tw[(j / A) * TWP + j % A] = a.W[a.n + a.L1 + (j / A) * (j % A)];      // a thread's A first-stage twiddles are one run
        ws[j] = a.W[a.n + a.L1 + a.L2 + j];
    }
    const cf wr = a.W[r];
    lds_barrier();
    const f2 wk = v2(wr * ws[pc]);
    // i wk D enters with + for the bin L-k, with - for the bin k; the bin L-k takes the conjugate (see split_pair)
    const f2 iws = grp ? rot_i(wk) : f2{wk.y, -wk.x}, niws = grp ? f2{-wk.x, -wk.y} : wk;
    const float hg = 0.5f * a.scale_g;
    const float sc = dc ? hg : hg * (a.interior2_g ? 2.f : 1.f);
    const f2 scv = grp ? f2{sc, -sc} : f2{sc, sc};
    const int yk_o = pc, ym_o = slotB * NOL * LENP + colB;

    long long g_ph[2] = {0, 0}, g_last = 0, g_begin = 0;
    if (DBG) g_begin = g_last = clock64();
#define FL_GSTAMP(i)                        \
    if (DBG) {                              \
        const long long t_now = clock64();  \
        g_ph[i] += t_now - g_last;          \
        g_last = t_now;                     \
    }
#pragma unroll 1
    for (int t = 0; t < n_it + 2; ++t) {
        // DEPTH 1 (owner-wave transfers): step 1 of cycle t requests the spectrum block of item t-1, P1(t) the rows of item
        // t+1; a wait names how many NEWER transfers of the wavefront may stay in flight.
        const bool x_now = DEPTH == 1 && t >= 1 && t <= n_it;
        // ---- step 1
        if (t >= 2) {
#pragma unroll
            for (int m = 0; m < NOL; ++m)
#pragma unroll
                for (int nn = 0; nn < NI; ++nn) asm volatile("" : "+v"(acc[m][nn]));
            // DEPTH 1: this wavefront's spectrum block of item t-2 has landed (behind it: its own newer row pieces)
            if (DEPTH == 1) wait_vm_le((has_g && t < n_it) ? GA : 0);
        }
        f2 x[NI];
        {
#if FL_XP_PAIRS
            const f4* xs = reinterpret_cast<const f4*>(stage_x + (t % DEPTH) * SBX + (wave * NI) * 64) + lane;      // item t-2's region: [n/2][lane][n%2]
#pragma unroll
            for (int j = 0; j < NI / 2; ++j) {
                const f4 q = xs[j * 64];
                x[2 * j] = f2{q.x, q.y};
                x[2 * j + 1] = f2{q.z, q.w};
            }
#else
            const cf* xs = stage_x + (t % DEPTH) * SBX + (wave * NI) * 64 + lane;      // item t-2's region
#pragma unroll
            for (int nn = 0; nn < NI; ++nn) x[nn] = v2(xs[nn * 64]);
#endif
        }
        if (x_now) {
            wait_lgkm0();                                        // the reads above have returned: the region may be refilled
            fetch_x(t - 1, wave);
        }
        if (t >= 2 && valid) {
#pragma unroll
            for (int m = 0; m < NOL; ++m) {
                const f2 zk = v2(Yb[yk_o + m * LENP]), zm = v2(Yb[ym_o + m * LENP]);
                const f2 Pp = add_pm(zk, zm), D = add_mp(zk, zm);
                const f2 g = scv * (Pp + cmulc(D, iws, niws));
                // g conj(x) = Re(x) g + Im(x) (-i g): the quarter turn of g is the second multiply-add's operand selectors
#pragma unroll
                for (int nn = 0; nn < NI; ++nn) acc[m][nn] = fma_mi(x[nn], g, pfma(f2{x[nn].x, x[nn].x}, g, acc[m][nn]));
            }
        }
        // DEPTH 2: the rows of item t (requested in the last step 2, ahead of that step's spectrum pieces) have landed
        if (fetcher) wait_vm_le(x_issued);
        lds_barrier();
        FL_GSTAMP(0)
        // ---- step 2
        if (fetcher) {
            wait_vm0();                  // the spectrum blocks of item t-1 have landed: a whole cycle in flight
            if (fi < NW1 && t + 1 < n_it) fetch_g(t + 1, fi);
            x_issued = 0;
            if (t < n_it) {
                for (int cw = xlo; cw < xhi; ++cw) fetch_x(t, cw);
                x_issued = XA * (xhi - xlo);
            }
        } else if (grp == 0) {
            if (t < n_it && has_g) {     // P1(t): first stage of the gradient rows, staging -> XF[t & 1].  item = (m fastest, tb, slot)
                const int item = opaque(tid) & 255;
                const int m = item % NOL, tb = (item / NOL) % B, slot = item / (NOL * B);
                const bool have = item < NI1 && !(slot && selfm);
                cf v[A], tt[A];
                // DEPTH 1: this wavefront's rows of item t have landed (behind them: the spectrum pieces just requested)
                if (DEPTH == 1) wait_vm_le(x_now ? XA : 0);
                const cf* sp = stage_g + (t % DEPTH) * SBG + (item >> 6) * (A * 64) + (item & 63);
#pragma unroll
                for (int ta = 0; ta < A; ++ta) v[ta] = sp[ta * 64];
#pragma unroll
                for (int ka = 1; ka < A; ++ka) tt[ka] = tw[(have ? tb : 0) * TWP + ka];
                if (DEPTH == 1 && t + 1 < n_it) {
                    wait_lgkm0();
                    fetch_g(t + 1, wave & 3);
                }
                if (have) {
                    RegFFT<float, A, false>::run(v);
                    cf* uu = XF + (t & 1) * UB + (slot * NOL + m) * LENP + tb;
                    uu[0] = v[0];
#pragma unroll
                    for (int ka = 1; ka < A; ++ka) uu[ka * B] = v[ka] * tt[ka];
                }
            }
        } else if (t >= 1 && t <= n_it) {     // P2(t-1): second stage XF[(t-1) & 1] -> Y.  item = (m fastest, ka, slot)
            // The items sit on the group's LAST wavefronts: wavefront w runs on SIMD w % 4, the first stage's items are on
            // wavefronts 0.. of group 0 -- with 2 x 64 items each (four output channels) that is SIMD 0, 1 for P1 and SIMD 2, 3
            // for P2; on the group's first wavefronts both stages shared SIMD 0 and 1 while the other two idled through the step
            // (own work 52 k / 61 k cycles per workgroup against 7 k on wavefront 7).  (DEPTH 2: wavefronts 4 and 5 issue transfers.)
            const int item = (opaque(tid) & 255) - P2_OFF;
            const int m = item % NOL, ka = (item / NOL) % A, slot = item / (NOL * A);
            if (item >= 0 && item < NI2 && !(slot && selfm)) {
                cf v[B];
                const cf* xr = XF + ((t - 1) & 1) * UB + (slot * NOL + m) * LENP + ka * B;
#pragma unroll
                for (int tb = 0; tb < B; ++tb) v[tb] = xr[tb];
                RegFFT<float, B, false>::run(v);
                cf* yr = Yb + (slot * NOL + m) * LENP + ka;
#pragma unroll
                for (int kb = 0; kb < B; ++kb) yr[A * kb] = v[kb];
            }
        }
        long long t_work = 0;
        if (DBG) t_work = clock64() - g_last;        // this wavefront's own work in step 2, before it waits at the barrier
        lds_barrier();
        FL_GSTAMP(1)
        if (DBG && a.dbg_times && lane == 0 && (wave == 0 || wave == 1 || wave == 4 || wave == 7))
            atomicAdd((unsigned long long*)(a.dbg_times + (size_t)blockIdx.x * 8 + (wave == 0 ? 4 : wave == 1 ? 5 : wave == 4 ? 6 : 7)), (unsigned long long)t_work);
    }
#undef FL_GSTAMP
    if (DBG && a.dbg_times && tid == 0) {
        long long* o = a.dbg_times + (size_t)blockIdx.x * 8;
        o[0] = g_begin;
        o[1] = clock64();
        o[2] = g_ph[0];
        o[3] = g_ph[1];
    }
    // ---- the slice's sums -> partial plane set sl
    {
        const unsigned ik = (unsigned)r * LEN + p;
        const unsigned im = dc ? (unsigned)a.L : (unsigned)(slotB ? rm : r) * LEN + colB;
        if (valid && !(grp && im == ik)) {
            cf* out = a.dH + (size_t)sl * a.ds_s;
            const unsigned bin = grp ? im : ik;
            if (a.out_scale) {
                const float os = *a.out_scale;
#pragma unroll
                for (int m = 0; m < NOL; ++m)
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) acc[m][nn] *= os;
            }
#pragma unroll
            for (int m = 0; m < NOL; ++m) {
                unsigned o = 8u * ((unsigned)(mo + m) * (unsigned)a.ds_m + bin);
                const unsigned step = 8u * (unsigned)a.ds_n;
                if (a.pol & POL_GRADH_DH_NT) {
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        st_nt(out, o, c2(acc[m][nn]));
                        o += step;
                    }
                } else {
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        at(out, o) = c2(acc[m][nn]);
                        o += step;
                    }
                }
            }
        }
    }
}

// out[j] = sum_s parts[s*stride + j]  (the batch slices' partial planes; fixed order: deterministic)
__global__ void __launch_bounds__(256) sum_parts_kernel(const float4* __restrict__ parts, long stride4, int ns, float4* __restrict__ out, long n4) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n4) return;
    float4 s = parts[j];
    for (int k = 1; k < ns; ++k) {
        const float4 v = parts[(long)k * stride4 + j];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[j] = s;
}

constexpr int kMaxDevices = 64;
constexpr int kMaxWalkWgs = 1024;     // stamps buffer: [2 * kMaxWalkWgs] per-workgroup (start, end) + [1] start of the next launch
static int g_walk = 1;            // 0: off (spec_mid only); 1: on for the shapes below
static int g_walk_wgs = 0;        // workgroups of the forward walking kernel (0: one per CU)
static int g_walk_slices = 0;     // batch slices of the backward walking kernel (0: CUs / row pairs)
static long long* g_walk_times = nullptr;
static long long* g_walk_stamps = nullptr;      // fl_debug_set_walk_stamps
static int g_walk_nsc = 2;        // output-channel groups of the backward kernel (tuning: fl_debug_set_walk mode 14 -> 4)
static int g_walk_fc = [] { const char* e = getenv("FLAMO_WALK_FC"); return e ? atoi(e) : 38; }();      // cost of a workgroup's first row pair, in twentieths of a unit
static int g_walk_fc2 = [] { const char* e = getenv("FLAMO_WALK_FC2"); return e ? atoi(e) : 38; }();    // ... of changing to the next one
static int g_walk_us = [] { const char* e = getenv("FLAMO_WALK_US"); return e ? atoi(e) : 16; }();      // a unit of a self-mirrored row pair

long long* walk_successor_stamp() { return g_walk_stamps ? g_walk_stamps + 2 * kMaxWalkWgs : nullptr; }

static int device_cus() {       // of the current device, cached per device
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

// workgroups of the forward walking kernel: one per CU (a multiple of 8: XCD-aware order), never more than units
static long walk_wgs(int L1, int Bn) {
    const long units = (long)(L1 / 2 + 1) * Bn;
    long g = g_walk_wgs > 0 ? g_walk_wgs : device_cus();
    if (g > units) g = units;
    if (g >= 8) g -= g % 8;
    return g;
}

// The kernels of this file ask for more dynamic LDS than the default 64 KB: the attribute is per function AND per device, so
// it is set once per (kernel, device) -- a second GPU used by the same process gets its own -- and its result is checked
// (a part with less LDS per workgroup fails here with a message, not at the launch).
static int ensure_lds(const void* kern, size_t lds, bool* done /* [kMaxDevices] */) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (done[dev]) return FL_OK;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("walking kernels: %zu bytes of LDS per workgroup are not available on device %d (%s)", lds, dev, hipGetErrorString(e));
        return FL_ERR_UNSUPPORTED;
    }
    done[dev] = true;
    return FL_OK;
}
static size_t device_lds_limit() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
    return (size_t)v;
}

template <int A, int B, int NI, int NO>
constexpr size_t walk_lds_bytes() {
    constexpr int LEN = A * B, LENP = walk_pitch(LEN), NCH = NI > NO ? NI : NO;
    return ((size_t)3 * 2 * NCH * LENP + 4 * A * 64 + ((B * tw_pitch(A) + 1) & ~1) + 3 * LEN + 64) * sizeof(cf);
}

template <int A, int B, int NI, int NO, int OCC>
static int launch_walk(const WalkArgs& a, hipStream_t st) {
    constexpr size_t lds = walk_lds_bytes<A, B, NI, NO>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool done[kMaxDevices] = {}, done_dbg[kMaxDevices] = {};
    auto kern = spec_mid_walk<A, B, NI, NO, OCC>;
    auto kern_dbg = spec_mid_walk<A, B, NI, NO, OCC, true>;
    int rc = ensure_lds(reinterpret_cast<const void*>(kern), lds, done);
    if (rc) return rc;
    if (a.dbg_times && (rc = ensure_lds(reinterpret_cast<const void*>(kern_dbg), lds, done_dbg))) return rc;
    const long g = walk_wgs(a.L1, a.Bn);
    if (a.dbg_times) hipLaunchKernelGGL(kern_dbg, dim3((unsigned)g), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(512), lds, st, a);
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {

int fl_debug_set_walk_stamps(void* buf) {
    g_walk_stamps = (long long*)buf;
    return FL_OK;
}

int fl_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    return khz;
}

int fl_debug_set_walk(int mode, int wgs, int slices, void* times) {
    g_walk_nsc = mode == 14 ? 4 : mode == 15 ? 1 : 2;
    g_walk = mode;
    g_walk_wgs = wgs;
    g_walk_slices = slices;
    g_walk_times = (long long*)times;
    return FL_OK;
}

// Shapes of the walking kernels: one bin pair per thread pair (row length <= 256), every FFT stage of a unit within one group
// of 256 threads (2 B N <= 256 and 2 A N <= 256), the row pair's response slice 2 L2 N_out N_in values in half of the CU's
// register file (<= 128 registers per thread: N_out N_in <= 64).  Instantiated: the 16 x 15 rows (nfft = 96000 and every planned
// length with 240-bin rows) and the 16 x 16 rows (65536, 131072, 64000, 128000 ...), 2 / 4 / 8 channels on either side.  16
// channels do not fit (the slice alone would be 512 registers); row lengths 320 / 480 (nfft = 192000 / 384000) would need two
// bin pairs per thread pair: those stay with spec_mid.
static bool walk_shape(int l2, int n_in, int n_out) {
    auto ch = [](int c) { return c == 2 || c == 4 || c == 8; };
    return (l2 == 240 || l2 == 256) && ch(n_in) && ch(n_out);
}
int fl_spec_walk_supports(int nfft, int n_in, int n_out) {
    int l1, l2;
    if (!g_walk || spec_plan(nfft, l1, l2) != FL_OK) return 0;
    if (!walk_shape(l2, n_in, n_out)) return 0;
    return walk_lds_bytes<16, 16, 8, 8>() <= device_lds_limit();
}

int fl_spec_gradh_slices(int nfft, int Bn) {
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK || Bn <= 0) return 0;
    if (g_walk_slices > 0) return g_walk_slices < Bn ? g_walk_slices : Bn;
    // the gradient kernel splits a row pair's work over the output channels first (two workgroups per row pair, no
    // partial sums); batch slices only beyond that
    const int P = l1 / 2 + 1;
    int ns = device_cus() / (2 * P);
    if (g_walk_nsc == 4) ns = 1;
    if (g_walk_nsc == 1) ns = device_cus() / P;
    if (ns < 1) ns = 1;
    return ns < Bn ? ns : Bn;
}

static int gradh_walk_impl(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g, const float* out_scale, void* stream);
int fl_spec_gradh_walk_f32(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g, void* stream) {
    return gradh_walk_impl(Sg, Xp, dH_parts, ds_s, ds_m, ds_n, n_slices, W, nfft, Bn, NI, NO, scale_g, interior2_g, nullptr, stream);
}
int fl_spec_gradh_walk_scaled_f32(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices, const void* W,
                                  int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g, const void* out_scale, void* stream) {
    return gradh_walk_impl(Sg, Xp, dH_parts, ds_s, ds_m, ds_n, n_slices, W, nfft, Bn, NI, NO, scale_g, interior2_g, (const float*)out_scale, stream);
}
}  // extern "C"
static int gradh_walk_impl(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g, const float* out_scale, void* stream) {
    FL_REQUIRE(Sg && Xp && dH_parts && W, "spec_gradh_walk: null pointer");
    FL_REQUIRE(n_slices >= 1 && n_slices <= (Bn > 0 ? Bn : 1), "spec_gradh_walk: slices must be in [1, batch]");
    FL_REQUIRE(reinterpret_cast<uintptr_t>(Sg) % 16 == 0 && reinterpret_cast<uintptr_t>(Xp) % 16 == 0, "spec_gradh_walk: scratch arrays must be 16-byte aligned");
    if (Bn == 0) return FL_OK;
    GradhArgs a = {};
    int rc = spec_plan(nfft, a.L1, a.L2);
    if (rc) return rc;
    a.Sg = (const cf*)Sg; a.Xp = (const cf*)Xp; a.dH = (cf*)dH_parts; a.ds_s = ds_s; a.ds_m = ds_m; a.ds_n = ds_n;
    a.W = (const cf*)W; a.n = nfft; a.L = nfft / 2; a.Bn = Bn; a.NS = n_slices;
    a.scale_g = (float)scale_g; a.interior2_g = interior2_g; a.dbg_times = g_walk_times; a.out_scale = out_scale;
    a.pol = stream_policy();
    FL_REQUIRE((size_t)NO * (size_t)ds_m * 8ull < (1ull << 32), "spec_gradh_walk: a partial plane set exceeds 32-bit offsets");
    const int P = a.L1 / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    int lrc = FL_ERR_UNSUPPORTED;
#define FL_GRADH(A_, B_, NI_, NO_, NSC_, OCC_, DEPTH_)                                                                            \
    {                                                                                                                            \
        constexpr int LEN = A_ * B_, LENP = walk_pitch(LEN), NOL = NO_ / NSC_;                                                   \
        constexpr size_t lds = ((size_t)3 * 2 * NOL * LENP + DEPTH_ * (((2 * B_ * NOL + 63) / 64) * A_ * 64 + 8 * NI_ * 64) +    \
                                ((B_ * tw_pitch(A_) + 1) & ~1) + LEN) * sizeof(cf);                                              \
        static_assert(lds <= 160 * 1024, "LDS budget");                                                                          \
        const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * n_slices * NSC_);                                                    \
        auto kern = spec_gradh_walk<A_, B_, NI_, NO_, NSC_, OCC_, DEPTH_>;                                                       \
        auto kern_dbg = spec_gradh_walk<A_, B_, NI_, NO_, NSC_, OCC_, DEPTH_, true>;                                             \
        static bool done[kMaxDevices] = {}, done_dbg[kMaxDevices] = {};                                                          \
        lrc = ensure_lds(reinterpret_cast<const void*>(kern), lds, done);                                                        \
        if (!lrc && a.dbg_times) lrc = ensure_lds(reinterpret_cast<const void*>(kern_dbg), lds, done_dbg);                       \
        if (!lrc) {                                                                                                              \
            if (a.dbg_times) hipLaunchKernelGGL(kern_dbg, dim3(nblk), dim3(512), lds, st, a);                                    \
            else hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, st, a);                                                    \
        }                                                                                                                        \
    }
    // output channels of a row pair over 2 workgroups (one per CU) -- or one for 2 output channels; tuning: over 4 (two per CU);
    // 8 -> 8 channels: two staging regions per kind, transfers issued by the wavefronts without a stage of their own (DEPTH 2).
    // Round 3 measured that form no faster than owner-wave transfers (69.5 against 68.7 us); since the second stage moved to
    // SIMD 2 and 3 the first stage is the longest path of the FFT step and the eight transfer issues in front of its
    // arithmetic count: 54.2 against 56.5 us replayed at config 2.  (Mode 17, tuning: the owner-wave form, DEPTH 1.)
#define FL_GRADH_ROWS(A_, B_)                                                                                                    \
    if (NI == 8 && NO == 8) {                                                                                                    \
        if (A_ == 16 && B_ == 15 && g_walk_nsc == 4) FL_GRADH(16, 15, 8, 8, 4, 4, 1)                                             \
        else if (A_ == 16 && B_ == 15 && g_walk_nsc == 1) FL_GRADH(16, 15, 8, 8, 1, 2, 1)                                        \
        else if (A_ == 16 && B_ == 15 && g_walk == 17) FL_GRADH(16, 15, 8, 8, 2, 2, 1)                                           \
        else FL_GRADH(A_, B_, 8, 8, 2, 2, 2)                                                                                     \
    }                                                                                                                            \
    else if (NI == 4 && NO == 8) FL_GRADH(A_, B_, 4, 8, 2, 2, 1)                                                                 \
    else if (NI == 2 && NO == 8) FL_GRADH(A_, B_, 2, 8, 2, 2, 1)                                                                 \
    else if (NI == 8 && NO == 4) FL_GRADH(A_, B_, 8, 4, 2, 2, 1)                                                                 \
    else if (NI == 4 && NO == 4) FL_GRADH(A_, B_, 4, 4, 2, 2, 1)                                                                 \
    else if (NI == 2 && NO == 4) FL_GRADH(A_, B_, 2, 4, 2, 2, 1)                                                                 \
    else if (NI == 8 && NO == 2) FL_GRADH(A_, B_, 8, 2, 1, 2, 1)                                                                 \
    else if (NI == 4 && NO == 2) FL_GRADH(A_, B_, 4, 2, 1, 2, 1)                                                                 \
    else if (NI == 2 && NO == 2) FL_GRADH(A_, B_, 2, 2, 1, 2, 1)
    if (a.L2 == 240) { FL_GRADH_ROWS(16, 15) }
    else if (a.L2 == 256) { FL_GRADH_ROWS(16, 16) }
#undef FL_GRADH_ROWS
#undef FL_GRADH
    if (lrc == FL_ERR_UNSUPPORTED && !walk_shape(a.L2, NI, NO)) {
        set_error("spec_gradh_walk: no kernel for nfft=%d, %d -> %d channels", nfft, NI, NO);
        return FL_ERR_UNSUPPORTED;
    }
    if (lrc) return lrc;
    FL_CHECK_LAUNCH("spec_gradh_walk");
    return FL_OK;
}
extern "C" {

int fl_sum_parts_c64(const void* parts, long part_stride, int n_parts, void* out, long n, void* stream) {
    FL_REQUIRE(parts && out && n_parts >= 1 && n >= 0, "sum_parts: bad arguments");
    FL_REQUIRE(n % 2 == 0 && part_stride % 2 == 0 && reinterpret_cast<uintptr_t>(parts) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
               "sum_parts: 16-byte granularity");
    if (n == 0) return FL_OK;
    const long n4 = n / 2;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)parts,
                       part_stride / 2, n_parts, (float4*)out, n4);
    FL_CHECK_LAUNCH("sum_parts");
    return FL_OK;
}

int fl_spec_walk_workgroups(int nfft, int Bn) {
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK || Bn <= 0) return 0;
    return (int)walk_wgs(l1, Bn);
}

int fl_spec_walk_partition(int nfft, int Bn, int n_wg, int* bounds) {
    FL_REQUIRE(bounds && n_wg > 0 && Bn > 0, "spec_walk_partition: bad arguments");
    int L1, L2;
    int rc = spec_plan(nfft, L1, L2);
    if (rc) return rc;
    // Contiguous unit ranges of minimal maximum COST.  A unit of a self-mirrored row pair (k1 = 0, and L1/2 for even L1) has
    // one row instead of two; entering a row pair costs a fill (the response slice into registers, tables, the pipeline's
    // first two stages with nothing beside them).  Fitted to per-workgroup cycle counts (tools/dbg/walk_bench.py's table by
    // (units, entries)): 11.0 k cycles per unit, 20.7 k per row pair entered, 8.7 k per self-mirrored unit (refitted late in
    // round 3: with the earlier 1.55 units per entry the workgroups that enter two row pairs finished 6 % behind the rest).
    // Equal unit COUNTS left the slowest workgroup 13-19 % above the mean.  Costs in twentieths of a unit; the smallest cap for which a greedy sweep needs no more than
    // n_wg ranges (binary search), then the sweep's cuts.
    const int P = L1 / 2 + 1, U = P * Bn, UC = 20, US = g_walk_us, FC = g_walk_fc, FC2 = g_walk_fc2;
    auto ucost = [&](int u) { const int r = u / Bn; return (r == 0 || 2 * r == L1) ? US : UC; };
    auto sweep = [&](long cap, int* out) {       // number of ranges used; out[i] = first unit of range i
        int n = 0, u = 0;
        while (u < U) {
            if (out) out[n] = u;
            ++n;
            long c = FC + ucost(u);
            int last_r = u / Bn;
            ++u;
            while (u < U) {
                const int r = u / Bn;
                const long add = ucost(u) + (r != last_r ? FC2 : 0);
                if (c + add > cap) break;
                c += add;
                last_r = r;
                ++u;
            }
        }
        return n;
    };
    long lo = FC + UC, hi = (long)U * UC + (long)P * FC;
    while (lo < hi) {
        const long mid = (lo + hi) / 2;
        if (sweep(mid, nullptr) <= n_wg) hi = mid;
        else lo = mid + 1;
    }
    const int used = sweep(lo, bounds);
    for (int i = used; i <= n_wg; ++i) bounds[i] = U;      // surplus workgroups get empty ranges
    return FL_OK;
}

size_t fl_spec_walk_spectrum_elems(int nfft, int Bn, int NI) {
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK || Bn < 0) return 0;
    return (size_t)(l1 / 2 + 1) * Bn * 2 * NI * l2;
}

int fl_spec_mid_walk_f32(const void* S, void* S2, void* Xp, const void* H, long hs_m, long hs_n, int conj_h, const void* W, int nfft, int Bn,
                         int NI, int NO, double spec_scale, int spec_interior2, int pre_half, const void* bounds, void* stream) {
    FL_REQUIRE(S && S2 && H && W, "spec_mid_walk: null pointer");
    FL_REQUIRE(S != S2, "spec_mid_walk: not an in-place kernel (the rows of a round are read while earlier rounds' are stored)");
    if (Bn == 0) return FL_OK;
    WalkArgs a = {};
    int rc = spec_plan(nfft, a.L1, a.L2);
    if (rc) return rc;
    a.S = (const cf*)S; a.S2 = (cf*)S2; a.H = (const cf*)H; a.hs_m = hs_m; a.hs_n = hs_n; a.conj_h = conj_h;
    a.W = (const cf*)W; a.n = nfft; a.L = nfft / 2; a.Bn = Bn;
    a.spec_scale = (float)spec_scale; a.spec_interior2 = spec_interior2; a.pre_half = pre_half; a.dbg_times = g_walk_times;
    a.Xp = (cf*)Xp;
    a.bounds = (const int*)bounds;
    a.pol = stream_policy();
    a.stamps = walk_wgs(a.L1, Bn) <= kMaxWalkWgs ? g_walk_stamps : nullptr;
    FL_REQUIRE((size_t)a.L1 * a.L2 * (NI > NO ? NI : NO) * 8ull < (1ull << 32), "spec_mid_walk: a batch item exceeds 32-bit offsets");
    FL_REQUIRE(reinterpret_cast<uintptr_t>(S) % 16 == 0, "spec_mid_walk: S must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    rc = FL_ERR_UNSUPPORTED;
#define FL_WALK_ROWS(A_, B_)                                                  \
    if (NI == 8 && NO == 8) rc = launch_walk<A_, B_, 8, 8, 2>(a, st);         \
    else if (NI == 4 && NO == 8) rc = launch_walk<A_, B_, 4, 8, 2>(a, st);    \
    else if (NI == 2 && NO == 8) rc = launch_walk<A_, B_, 2, 8, 2>(a, st);    \
    else if (NI == 8 && NO == 4) rc = launch_walk<A_, B_, 8, 4, 2>(a, st);    \
    else if (NI == 4 && NO == 4) rc = launch_walk<A_, B_, 4, 4, 2>(a, st);    \
    else if (NI == 2 && NO == 4) rc = launch_walk<A_, B_, 2, 4, 2>(a, st);    \
    else if (NI == 8 && NO == 2) rc = launch_walk<A_, B_, 8, 2, 2>(a, st);    \
    else if (NI == 4 && NO == 2) rc = launch_walk<A_, B_, 4, 2, 2>(a, st);    \
    else if (NI == 2 && NO == 2) rc = launch_walk<A_, B_, 2, 2, 2>(a, st);
    if (a.L2 == 240) { FL_WALK_ROWS(16, 15) }
    else if (a.L2 == 256) { FL_WALK_ROWS(16, 16) }
#undef FL_WALK_ROWS
    if (rc == FL_ERR_UNSUPPORTED && !walk_shape(a.L2, NI, NO)) {
        set_error("spec_mid_walk: no kernel for nfft=%d, %d -> %d channels", nfft, NI, NO);
        return FL_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    FL_CHECK_LAUNCH("spec_mid_walk");
    return FL_OK;
}

}  // extern "C"
