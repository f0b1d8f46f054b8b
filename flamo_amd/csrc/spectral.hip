// Fused frequency-sampling pipeline for gfx950 (MI355X):   y = irfft( H[f] . rfft(x) )   in three launches.
//
// flamo's Shell applies FFT -> per-bin products -> iFFT (system.py:839-855 with dsp.py:88, dsp.py:922-924,
// dsp.py:114).  When the core is a chain of per-bin products the whole Shell is ONE linear operator per batch
// item and the spectrum in between is private to it, so it is kept in whatever layout suits the hardware:
//
//   time domain      x, y : (B, T, G)  channel-innermost, exactly as the user hands it in / gets it back
//   column pass  (K1)     : L1-point FFTs over t1 of the packed sequence z[c + L2 t1] = x[2j] + i x[2j+1] for a
//                           tile of CT columns x ALL G channels (CT*G*8 B = 256-byte runs of x per row: whole
//                           lines, no layout-conversion pass), inter-pass twiddle, scratch S[b][k1][c][g]
//   row pass     (mid)    : a workgroup owns rows k1 = r and L1-r of ALL channels of one batch item: L2-point
//                           FFTs over c, real-FFT split step (couples bin k with L-k = the mirror row), the
//                           per-bin product Y[f] = H[f] X[f] on the 2 L2 bins it holds, Hermitian pre-step of
//                           the inverse transform, L2-point inverse FFTs, twiddle, scratch S2[b][k1][c][g].
//                           The spectrum leaves only if the backward pass needs it, in "row-major bin order"
//                           i = k1*L2 + k2 for bin k = k1 + L1*k2 (contiguous 8*L2-byte runs; Nyquist at i = L).
//   column pass  (K3)     : L1-point inverse FFTs over k1 for a tile of columns x all channels, scale, anti-alias
//                           envelope, y written channel-innermost.
//
// Against the layered route (layout conversion, two FFT passes, product, two inverse passes) the (B, M, N)
// spectrum is never written and re-read between the transforms and the product: 6 passes over the data
// instead of 11.  The same three kernels run the backward pass (irfft' = weighted rfft, rfft' = weighted irfft).
#include "spectral_common.h"
#include "spec_cols_body.h"
#include "fusedfwd.h"

namespace fl {
namespace FL_SPEC_NS {

// ---------------------------------------------------------------- K1: forward column pass (body: spec_cols_body.h)
template <int A, int B, int VT, int RG, bool PLAIN>
__global__ void __launch_bounds__(256, sizeof(real_t) == 8 ? 2 : 3) spec_cols_fwd(ColsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    spec_cols_fwd_body<A, B, VT, RG, PLAIN>(a, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------- K3: inverse column pass
// PLAIN: no envelope and every sample inside the output (t_lim >= n): the guards are not compiled at all -- left to a
// run-time flag the compiler evaluates the envelope for every output and selects (25 x (int -> double, double multiply, exp,
// ldexp, four selects) per item, a third of the kernel's instructions, and 112 registers against 9x).
// FUSE (PLAIN only): the launch also leaves a.Sg = the forward column pass of the tile it stores (fl_spec_cols_fwd of y, the first
// pass of the gradient's transform when the objective's g_y is a multiple of y: ops.mean_square).  A workgroup's inverse
// column transforms produce exactly the samples its forward column transforms consume -- the tile goes back through the row
// buffer and the two forward stages of spec_cols_body.h with the same operations in the same order (the stored float32 sample
// IS the register value): the pass that re-reads y (a sixth of the step's streaming bytes) and its launch are not run.
template <int A, int B, int VT, int RG, bool PLAIN, bool FUSE = false>
__global__ void __launch_bounds__(256) spec_cols_inv(ColsArgs a) {
    static_assert(!FUSE || PLAIN, "the fused gradient pass exists for the plain shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = LEN | 1;
    cf* U = reinterpret_cast<cf*>(smem);   // [VT][LENP]
    cf* tw = U + VT * LENP;                 // W_LEN^m
    cf* t2 = tw + LEN;                      // [CT][B]: W_L^(c * A * kb)   (FUSE)
    int blk = blockIdx.x;
    const int ct = blk % a.nct; blk /= a.nct;
    const int gt = blk % a.ngt;
    const int b = blk / a.ngt;
    const int CG = 1 << a.cgs;
    const int c0 = ct * a.CT, g0 = gt * CG;
    constexpr int NIT = B * VT, NR = (NIT + 255) / 256;
    const cf* in = a.S + (size_t)b * a.L1 * a.L2 * a.G + g0;
    if (a.stamp && blockIdx.x == 0 && threadIdx.x == 0) *a.stamp = (long long)__builtin_amdgcn_s_memrealtime();
    cf v[RG][A];
    const bool ld_nt_pol = a.pol & 1u, st_nt_pol = a.pol & 2u;      // workgroup-uniform: one branch around each group of accesses
    auto load_group_p = [&](int r0, auto nt_tag) {
        constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            const int item = threadIdx.x + (r0 + rr) * 256;
            if (r0 + rr < NR && item < NIT) {
                const int tb = item / VT, vv = item % VT;
                const int cl = vv >> a.cgs, gl = vv & (CG - 1);
#pragma unroll
                for (int ta = 0; ta < A; ++ta) {
                    const v2f q = ldv<NT>(in, ESZ * (((unsigned)(ta * B + tb) * (unsigned)a.L2 + (unsigned)(c0 + cl)) * (unsigned)a.G + (unsigned)gl));
                    v[rr][ta] = cf(q.x, q.y);
                }
            }
        }
    };
    auto load_group = [&](int r0) {
        if (ld_nt_pol) load_group_p(r0, std::true_type{});
        else load_group_p(r0, std::false_type{});
    };
    load_group(0);                                 // data first, tables behind it (see the forward pass)
    for (int j = threadIdx.x; j < LEN; j += 256) tw[j] = a.W[a.n + j];
    if constexpr (FUSE) {
        for (int j = threadIdx.x; j < a.CT * B; j += 256) {
            const int cl = j / B, kb = j - cl * B;
            t2[j] = a.W[2 * (c0 + cl) * A * kb];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int r0 = 0; r0 < NR; r0 += RG) {
        if (r0 > 0) load_group(r0);
#pragma unroll
        for (int rr = 0; rr < RG; ++rr) {
            const int item = threadIdx.x + (r0 + rr) * 256;
            if (r0 + rr < NR && item < NIT) {
                const int tb = item / VT, vv = item % VT;
                RegFFT<real_t, A, true, false>::run(v[rr]);
                cf* u = U + vv * LENP + tb;
                u[0] = v[rr][0];
#pragma unroll
                for (int ka = 1; ka < A; ++ka) u[ka * B] = mul_plain(v[rr][ka], conj(tw[ka * tb]));
            }
        }
    }
    __syncthreads();
    real_t* yb = a.y + (size_t)b * a.t_len * a.G + g0;
    const real_t scale0 = a.dev_scale ? a.scale * *a.dev_scale : a.scale;
    real_t sq = 0;
    static_assert(!FUSE || A * VT <= 256, "fused gradient pass: one second-stage item per thread");
    auto second_stage = [&](int ka, int vv, cf (&v)[B], auto fused_tag) {
        constexpr bool FU = decltype(fused_tag)::value;
        const int cl = vv >> a.cgs, gl = vv & (CG - 1);
        const int par = gl & 1;
        RegFFT<real_t, B, true, false>::run(v);
        cf* zu = U + vv * LENP + ka;      // FU: the scaled tile value z[t1 = ka + A kb] (per channel: (y[2j], y[2j+1])) goes back to the row buffer
#pragma unroll
        for (int kb = 0; kb < B; ++kb) {
            const int t1 = ka + A * kb;
            const int t = 2 * (c0 + cl + a.L2 * t1) + par;
            // (re_g, im_g) per lane -> even lane (re_g, re_g+1) at sample 2j, odd lane (im_g-1, im_g) at 2j+1
            const real_t got = swap1(par ? v[kb].x : v[kb].y);
            real2 q = par ? make_real2(got, v[kb].y) : make_real2(v[kb].x, got);
            real_t s = scale0;
            if (!PLAIN && a.env_log2 != 0.0) s *= env_at(a.env_log2, t);
            q.x *= s;
            q.y *= s;
            if constexpr (FU) zu[A * kb] = cf(v[kb].x * s, v[kb].y * s);      // (the values the forward pass would load and un-swap)
            if (PLAIN || t < a.t_lim) {
                const unsigned yo = RSZ * ((unsigned)t * (unsigned)a.G + (unsigned)(gl - par));
                if (st_nt_pol) stv<true>(yb, yo, v2f{q.x, q.y});
                else stv<false>(yb, yo, v2f{q.x, q.y});
                sq += q.x * q.x + q.y * q.y;
            }
        }
    };
    if constexpr (!FUSE) {
        for (int item = threadIdx.x; item < A * VT; item += 256) {
            const int ka = item / VT, vv = item % VT;
            cf v[B];
            const cf* u = U + vv * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
            second_stage(ka, vv, v, std::false_type{});
        }
    } else {
        const bool act = threadIdx.x < A * VT;
        const int ka = threadIdx.x / VT, vv = threadIdx.x % VT;
        cf v[B];
        if (act) {
            const cf* u = U + vv * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
        }
        __syncthreads();                           // every second-stage read of the row buffer is done: the tile goes back into it
        if (act) second_stage(ka, vv, v, std::true_type{});
        // ---- the forward column pass of the same tile (spec_cols_body.h, its loads replaced by the row buffer)
        __syncthreads();
#pragma unroll 1
        for (int r0 = 0; r0 < NR; ++r0) {
            const int item = threadIdx.x + r0 * 256;
            if (item < NIT) {
                const int tb = item / VT, vv = item % VT;
                cf w[A];
                cf* u = U + vv * LENP + tb;
#pragma unroll
                for (int ta = 0; ta < A; ++ta) w[ta] = u[ta * B];
                RegFFT<real_t, A, false>::run(w);
                u[0] = w[0];                       // (in place: the item's own A slots)
#pragma unroll
                for (int ka = 1; ka < A; ++ka) u[ka * B] = w[ka] * tw[ka * tb];
            }
        }
        __syncthreads();
        cf* out = a.Sg + (size_t)b * a.L1 * a.L2 * a.G + g0;
        if (threadIdx.x < A * VT) {
            const int ka = threadIdx.x / VT, vv = threadIdx.x % VT;
            const int cl = vv >> a.cgs, gl = vv & (CG - 1);
            const int c = c0 + cl;
            cf w[B];
            const cf* u = U + vv * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) w[tb] = u[tb];
            RegFFT<real_t, B, false>::run(w);
            const cf w1 = a.W[2 * c * ka];
            const cf* w2 = t2 + cl * B;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) {
                const int k1 = ka + A * kb;
                const cf r = w[kb] * (w1 * w2[kb]);
                const unsigned so = ESZ * (((unsigned)k1 * (unsigned)a.L2 + (unsigned)c) * (unsigned)a.G + (unsigned)gl);
                if (a.pol & 4u) stv<true>(out, so, v2f{r.x, r.y});
                else stv<false>(out, so, v2f{r.x, r.y});
            }
        }
    }
    if (a.sumsq) {        // (workgroup-uniform) per-workgroup partial of sum y^2, combined in a fixed order by fl_mean_square_final_*
        double d = (double)sq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
        __syncthreads();                          // the row buffer is free: its first words carry the four wavefront sums
        double* red = reinterpret_cast<double*>(smem);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) a.sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------- mid: rows + split + product + pre-step + inverse rows
struct MidArgs {
    const cf* S;          // (Bn, L1, L2, NI)
    cf* S2;               // (Bn, L1, L2, NO)                      [DO_INV]
    cf* Xs;               // spectrum out, row-major bin order: Xs[b*xs_b + n*xs_n + i], or null
    long xs_b, xs_n;
    const cf* H;          // H[m*hs_m + n*hs_n + i], row-major bin order  [HAS_H]
    long hs_m, hs_n;
    int conj_h;
    const cf* W;
    int n, L, L1, L2, Bn;
    real_t spec_scale;    // scale of the forward transform
    int spec_interior2;   // double the interior bins of the spectrum (irfft backward)
    int pre_half;         // halve the interior bins in front of the inverse transform (rfft backward)
    int dbg_hfake;        // tuning: every bin reads the first 64 bins' response (cache-resident) -- isolates the fetch cost
    long long* dbg_times; // tuning: per-workgroup cycle stamps at the phase boundaries (8 per workgroup), or null
};

// BG batch items per workgroup (256*BG threads): a response row fetched for a bin pair is applied to the BG spectra in
// registers, so the response's trips through the TA/L2 shrink by BG (at one item per workgroup they are 8x the
// signal's bytes).  Streaming data (scratch, stored spectrum) is non-temporal so that it does not evict the response
// slice that the batch items of a row pair share in their XCD's L2.
template <int A, int B, int NI, int NO, bool HAS_H, bool DO_INV, int BG, int MS, int PFD = 0, int NTH = MS, int P3V = 0>
__global__ void __launch_bounds__(256 * NTH, ((NTH == 2 && MS == 1) || P3V == 1 ? 4 : 1)) spec_mid(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LEN = A * B, LENP = LEN | 1, NCH = NI > NO ? NI : NO, NT = 256 * NTH;
    constexpr int RPR = NT / A;              // rows per stage-2 round (every k_a of a row in the same round)
    // MS: the product of a bin pair is split over MS threads (output channels); 256*NTH threads in the workgroup
    // (NTH > MS: the extra threads take part in the FFT phases only)
    static_assert(NT % A == 0 && NO % MS == 0 && NI % MS == 0, "tile shape");
    cf* U = reinterpret_cast<cf*>(smem);     // [BG][2][NCH][LENP]
    cf* tw = U + BG * 2 * NCH * LENP;        // W_LEN^m
    cf* ws = tw + LEN;                       // W_n^(L1*k2)
    cf* wi2 = ws + LEN;                      // [2][B]: W_L^(row * A * kb)
    cf* nyq = wi2 + 2 * B;                   // [BG][NCH]: the Nyquist bin of row pair 0 (P3V == 1)
    // XCD-aware order: the batch groups of one row pair run back to back on the same XCD (block q runs on XCD
    // q % 8), so that pair's slice of H is fetched from HBM once and from that L2 afterwards
    const int P = a.L1 / 2 + 1;
    const int nbq = (a.Bn + BG - 1) / BG;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int r = (q / nbq) * 8 + xcd, b0 = (q % nbq) * BG;
    if (r >= P) return;
    const int nb = min(BG, a.Bn - b0);
    const int rm = (a.L1 - r) % a.L1;
    const bool selfm = rm == r;
    const int tid = threadIdx.x;
    const unsigned bstride_i = (unsigned)a.L1 * (unsigned)a.L2 * NI, bstride_o = (unsigned)a.L1 * (unsigned)a.L2 * NO;
    // ---- P1: load + first stage of the forward row FFTs.  item = (batch item, slot, tb, n), n fastest: the loads of
    // a wavefront cover contiguous (tb, n) runs of a scratch row.  The first item's samples are requested before the
    // twiddle tables (whose round trip would otherwise sit in front of the data's).
    const cf* Sb = a.S + (size_t)b0 * bstride_i;
    cf v0[A];
    auto p1_load = [&](int item, cf* v) {
        const int nn = item % NI, tb = (item / NI) % B, bs = item / (NI * B);
        const int slot = bs & 1, bb = bs >> 1;
        if (item >= BG * 2 * B * NI || (slot && selfm) || bb >= nb) return false;
        const unsigned src0 = (unsigned)bb * bstride_i + (unsigned)(slot ? rm : r) * (unsigned)a.L2 * NI + nn;
#pragma unroll
        for (int ta = 0; ta < A; ++ta) v[ta] = ld_nt(Sb, ESZ * (src0 + (unsigned)(ta * B + tb) * NI));
        return true;
    };
    const bool have0 = p1_load(tid, v0);
    for (int j = tid; j < LEN; j += NT) {       // contiguous copies behind the master table (fl_spec_aux_fill_f32)
        tw[j] = a.W[a.n + a.L1 + j];
        ws[j] = a.W[a.n + a.L1 + a.L2 + j];
    }
    if (DO_INV && tid < 2 * B) {
        const int slot = tid / B, kb = tid - slot * B;
        wi2[tid] = a.W[2 * (slot ? rm : r) * A * kb];
    }
    __syncthreads();
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 0] = clock64();
    {
        for (int item = tid; item < BG * 2 * B * NI; item += NT) {
            const int nn = item % NI, tb = (item / NI) % B, bs = item / (NI * B);
            cf v[A];
            bool have;
            if (item == tid) {
                have = have0;
#pragma unroll
                for (int ta = 0; ta < A; ++ta) v[ta] = v0[ta];
            } else {
                have = p1_load(item, v);
            }
            if (!have) continue;
            RegFFT<real_t, A, false>::run(v);
            cf* u = U + (bs * NCH + nn) * LENP + tb;
            u[0] = v[0];
#pragma unroll
            for (int ka = 1; ka < A; ++ka) u[ka * B] = v[ka] * tw[ka * tb];
        }
    }
    __syncthreads();
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 1] = clock64();
    // ---- P2: second stage, natural order written back in place (read all, barrier, write all per round)
    for (int row0 = 0; row0 < BG * 2 * NI; row0 += RPR) {
        const int rl = row0 + tid % RPR, ka = tid / RPR;
        const int bs = rl / NI, nn = rl % NI;
        const bool act = rl < BG * 2 * NI && !((bs & 1) && selfm) && (bs >> 1) < nb;
        cf v[B];
        cf* urow = U + (bs * NCH + nn) * LENP;
        if (act) {
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = urow[ka * B + tb];
            RegFFT<real_t, B, false>::run(v);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int kb = 0; kb < B; ++kb) urow[ka + A * kb] = v[kb];
        }
        __syncthreads();
    }
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 2] = clock64();
    if constexpr (P3V == 0) {
    // ---- P3: split step, spectrum store, product, Hermitian pre-step.  Thread (p, ms): bin pair (k, L-k) number p, output
    // channels [ms NO/MS, (ms+1) NO/MS), all BG batch items.  The MS threads of a pair read the same two columns of U
    // (all input channels) and write their own output channels back into them: one barrier between the two.
    {
        const cf wr = a.W[r];
        const real_t hs = (real_t)0.5 * a.spec_scale, wi = a.spec_interior2 ? (real_t)2 : (real_t)1;
        const real_t ph = a.pre_half ? (real_t)0.5 : (real_t)1;
        const int ms = tid / 256;
        for (int p0 = 0; p0 < LEN; p0 += 256) {
            const int p = p0 + (tid & 255);
            int slotB = 0, colB = 0;
            bool dc = false;
            const bool valid = ms < MS && p < LEN && pair_of(r, selfm, p, LEN, slotB, colB, dc);
            const unsigned ik = (unsigned)r * LEN + p;
            const unsigned im = dc ? (unsigned)a.L : (unsigned)(slotB ? rm : r) * LEN + colB;
            cf xk[BG][NI], xm[BG][NI];
            cf wk(1, 0);
            if (valid) {
                wk = wr * ws[p];
#pragma unroll
                for (int bb = 0; bb < BG; ++bb) {
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        const cf zk = U[((2 * bb) * NCH + nn) * LENP + p];
                        const cf zm = U[((2 * bb + slotB) * NCH + nn) * LENP + colB];
                        if (dc) {
                            xk[bb][nn] = cf(a.spec_scale * (zk.x + zk.y), 0);     // X[0]
                            xm[bb][nn] = cf(a.spec_scale * (zk.x - zk.y), 0);     // X[L]
                        } else {
                            const cf pk = zk + conj(zm), dk = zk - conj(zm);
                            const cf ok = pk + mul_mi(wk * dk);
                            const cf pm = zm + conj(zk), dm = zm - conj(zk);
                            const cf wm(-wk.x, wk.y);                               // W_n^(L-k) = -conj(W_n^k)
                            const cf om = pm + mul_mi(wm * dm);
                            xk[bb][nn] = cf(hs * wi * ok.x, hs * wi * ok.y);
                            xm[bb][nn] = cf(hs * wi * om.x, hs * wi * om.y);
                        }
                    }
                }
                if (a.Xs) {     // the MS threads of a pair share the stores: input channels [ms NI/MS, (ms+1) NI/MS)
#pragma unroll
                    for (int bb = 0; bb < BG; ++bb) {
                        if (bb >= nb) break;
                        cf* xo = a.Xs + (size_t)(b0 + bb) * a.xs_b;
#pragma unroll
                        for (int n2 = 0; n2 < NI / MS; ++n2) {
                            const int nn = ms * (NI / MS) + n2;
                            cf vk = xk[bb][0], vm = xm[bb][0];
#pragma unroll
                            for (int e = 1; e < NI; ++e)
                                if (e == nn) { vk = xk[bb][e]; vm = xm[bb][e]; }
                            st_nt(xo, ESZ * ((unsigned)nn * (unsigned)a.xs_n + ik), vk);
                            if (im != ik) st_nt(xo, ESZ * ((unsigned)nn * (unsigned)a.xs_n + im), vm);
                        }
                    }
                }
            }
            if (!DO_INV) continue;
            if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 6] = clock64();
            if (MS > 1) __syncthreads();
            if (!valid) continue;
            const cf cwk = conj(wk);
            // response rows: row m2 + PFD is requested before row m2 is used (PFD + 1 rows of 2 NI values in registers);
            // plane (m, n) is a workgroup-uniform base (scalar arithmetic), the bin the lane's 32-bit offset
            constexpr int NB = PFD + 1;
            cf hrows[NB][2 * NI];
            const unsigned hoff_k = a.dbg_hfake ? ESZ * (ik & 63u) : ESZ * ik, hoff_m = a.dbg_hfake ? ESZ * (im & 63u) : ESZ * im;
            auto load_row = [&](int m2, cf* dst) {
                const cf* Hm = a.H + (size_t)__builtin_amdgcn_readfirstlane(ms * (NO / MS) + m2) * a.hs_m;
#pragma unroll
                for (int nn = 0; nn < NI; ++nn) {
                    dst[nn] = at(Hm + (size_t)nn * a.hs_n, hoff_k);
                    dst[NI + nn] = at(Hm + (size_t)nn * a.hs_n, hoff_m);
                }
            };
            if (HAS_H) {
#pragma unroll
                for (int i = 0; i < PFD; ++i)
                    if (i < NO / MS) load_row(i, hrows[i]);
            }
#pragma unroll
            for (int m2 = 0; m2 < NO / MS; ++m2) {
                const int m = ms * (NO / MS) + m2;
                cf hkv[NI], hmv[NI];
                if (HAS_H) {
                    if (m2 + PFD < NO / MS) load_row(m2 + PFD, hrows[(m2 + PFD) % NB]);
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn) {
                        hkv[nn] = hrows[m2 % NB][nn];
                        hmv[nn] = hrows[m2 % NB][NI + nn];
                        if (a.conj_h) {
                            hkv[nn].y = -hkv[nn].y;
                            hmv[nn].y = -hmv[nn].y;
                        }
                    }
                }
#pragma unroll
                for (int bb = 0; bb < BG; ++bb) {
                    cf yk, ym;
                    if (HAS_H) {
                        yk = cf(0, 0);
                        ym = cf(0, 0);
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) {
                            fma_cx(yk, hkv[nn], xk[bb][nn]);
                            fma_cx(ym, hmv[nn], xm[bb][nn]);
                        }
                    } else {
                        yk = xk[bb][0];
                        ym = xm[bb][0];
#pragma unroll
                        for (int e = 1; e < NI; ++e)
                            if (e == m) { yk = xk[bb][e]; ym = xm[bb][e]; }
                    }
                    cf zk, zm;
                    if (dc) {      // Zf[0] from the real parts of Y[0] and Y[L] (C2R semantics)
                        zk = cf(yk.x + ym.x, yk.x - ym.x);
                        zm = zk;
                    } else {
                        const cf xa(ph * yk.x, ph * yk.y), xb(ph * ym.x, ph * ym.y);
                        const cf s_ = xa + conj(xb), t_ = mul_i(cwk * (xa - conj(xb)));
                        zk = s_ + t_;
                        zm = conj(s_ - t_);
                    }
                    U[((2 * bb) * NCH + m) * LENP + p] = zk;
                    if (im != ik && !dc) U[((2 * bb + slotB) * NCH + m) * LENP + colB] = zm;
                }
            }
        }
    }
    }
    if constexpr (P3V == 1) {
        // ---- P3 in three sweeps (BG batch items per workgroup, 256*BG threads), so that a response row fetched for a
        // bin serves the BG spectra the workgroup holds -- the response's trips through the L2 -> L1 path (measured
        // 31 B/clk/CU for 8-byte loads: 245 KB per row pair = 8k cycles, the largest single cost of the one-item form)
        // shrink by BG, and the product runs one thread per BIN (not per pair) with 2 NI + 2 NI values in registers:
        //   a) split step per (item, pair): spectrum written back in place (and to global memory for the backward pass)
        //   b) product per bin: Y[item][:, bin] = H[:, :, bin] X[item][:, bin], in place
        //   c) Hermitian pre-step per (item, pair), in place
        static_assert(HAS_H && DO_INV && NTH == BG, "three-sweep product: response + inverse half, 256 threads per item");
        const cf wr = a.W[r];
        const real_t hs = (real_t)0.5 * a.spec_scale, wi = a.spec_interior2 ? (real_t)2 : (real_t)1;
        const real_t ph = a.pre_half ? (real_t)0.5 : (real_t)1;
        const int bbt = tid / 256;                      // the batch item this thread serves in sweeps a and c
        for (int p0 = 0; p0 < LEN; p0 += 256) {        // ---- a
            const int p = p0 + (tid & 255);
            int slotB = 0, colB = 0;
            bool dc = false;
            if (!(p < LEN && bbt < nb && pair_of(r, selfm, p, LEN, slotB, colB, dc))) continue;
            const unsigned ik = (unsigned)r * LEN + p;
            const unsigned im = dc ? (unsigned)a.L : (unsigned)(slotB ? rm : r) * LEN + colB;
            const cf wk = wr * ws[p];
            cf* xo = a.Xs ? a.Xs + (size_t)(b0 + bbt) * a.xs_b : nullptr;
#pragma unroll
            for (int nn = 0; nn < NI; ++nn) {
                cf* pk = U + ((2 * bbt) * NCH + nn) * LENP + p;
                cf* pm = U + ((2 * bbt + slotB) * NCH + nn) * LENP + colB;
                const cf zk = *pk, zm = *pm;
                cf xk, xm;
                if (dc) {
                    xk = cf(a.spec_scale * (zk.x + zk.y), 0);     // X[0]
                    xm = cf(a.spec_scale * (zk.x - zk.y), 0);     // X[L]
                    nyq[bbt * NCH + nn] = xm;
                } else {
                    const cf pk_ = zk + conj(zm), dk = zk - conj(zm);
                    const cf ok = pk_ + mul_mi(wk * dk);
                    const cf pm_ = zm + conj(zk), dm = zm - conj(zk);
                    const cf wm(-wk.x, wk.y);
                    const cf om = pm_ + mul_mi(wm * dm);
                    xk = cf(hs * wi * ok.x, hs * wi * ok.y);
                    xm = cf(hs * wi * om.x, hs * wi * om.y);
                    if (im != ik) *pm = xm;
                }
                *pk = xk;
                if (xo) {
                    st_nt(xo, ESZ * ((unsigned)nn * (unsigned)a.xs_n + ik), xk);
                    if (im != ik) st_nt(xo, ESZ * ((unsigned)nn * (unsigned)a.xs_n + im), xm);
                }
            }
        }
        __syncthreads();
        {                                                // ---- b: one thread per bin held by the workgroup
            const int nbins = (selfm ? 1 : 2) * LEN + (r == 0 ? 1 : 0);
            for (int j = tid; j < nbins; j += NT) {
                const bool isnyq = j == (selfm ? 1 : 2) * LEN;
                const int slot = isnyq ? 0 : j / LEN, col = isnyq ? 0 : j - slot * LEN;
                const unsigned ib = isnyq ? (unsigned)a.L : (unsigned)(slot ? rm : r) * LEN + col;
                cf x[BG][NI];
#pragma unroll
                for (int bb = 0; bb < BG; ++bb)
#pragma unroll
                    for (int nn = 0; nn < NI; ++nn)
                        x[bb][nn] = isnyq ? nyq[bb * NCH + nn] : U[((2 * bb + slot) * NCH + nn) * LENP + col];
                const unsigned hoff = ESZ * ib;
                // MC response rows requested together (MC*NI loads in flight per thread): the sweep is a chain of
                // NO/MC load round trips, each ~3.5k cycles under the kernel's own traffic -- not NO of them
                // (MC = 4 rows in flight spills under the 128-register cap of this form and measured slower: 109 us against 95)
                constexpr int MC = 1;
#pragma unroll
                for (int m0 = 0; m0 < NO; m0 += MC) {
                    cf h[MC][NI];
#pragma unroll
                    for (int mc = 0; mc < MC; ++mc) {
                        const cf* Hm = a.H + (size_t)(m0 + mc) * a.hs_m;
#pragma unroll
                        for (int nn = 0; nn < NI; ++nn) h[mc][nn] = at(Hm + (size_t)nn * a.hs_n, hoff);
                    }
#pragma unroll
                    for (int mc = 0; mc < MC; ++mc) {
                        if (a.conj_h) {
#pragma unroll
                            for (int nn = 0; nn < NI; ++nn) h[mc][nn].y = -h[mc][nn].y;
                        }
#pragma unroll
                        for (int bb = 0; bb < BG; ++bb) {
                            cf y(0, 0);
#pragma unroll
                            for (int nn = 0; nn < NI; ++nn) fma_cx(y, h[mc][nn], x[bb][nn]);
                            if (isnyq) nyq[bb * NCH + m0 + mc] = y;
                            else U[((2 * bb + slot) * NCH + m0 + mc) * LENP + col] = y;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int p0 = 0; p0 < LEN; p0 += 256) {        // ---- c
            const int p = p0 + (tid & 255);
            int slotB = 0, colB = 0;
            bool dc = false;
            if (!(p < LEN && bbt < nb && pair_of(r, selfm, p, LEN, slotB, colB, dc))) continue;
            const bool self = !dc && slotB == 0 && colB == p;
            const cf cwk = conj(wr * ws[p]);
#pragma unroll
            for (int m = 0; m < NO; ++m) {
                cf* pk = U + ((2 * bbt) * NCH + m) * LENP + p;
                cf* pm = U + ((2 * bbt + slotB) * NCH + m) * LENP + colB;
                const cf yk = *pk, ym = dc ? nyq[bbt * NCH + m] : *pm;
                if (dc) {      // Zf[0] from the real parts of Y[0] and Y[L] (C2R semantics)
                    *pk = cf(yk.x + ym.x, yk.x - ym.x);
                } else {
                    const cf xa(ph * yk.x, ph * yk.y), xb(ph * ym.x, ph * ym.y);
                    const cf s_ = xa + conj(xb), t_ = mul_i(cwk * (xa - conj(xb)));
                    *pk = s_ + t_;
                    if (!self) *pm = conj(s_ - t_);
                }
            }
        }
    }
    if (!DO_INV) return;
    __syncthreads();
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 3] = clock64();
    // ---- P4: first stage of the inverse row FFTs, in place (a thread owns positions tb + B*i of its row)
    for (int item = tid; item < BG * 2 * B * NO; item += NT) {
        const int m = item % NO, tb = (item / NO) % B, bs = item / (NO * B);
        if (((bs & 1) && selfm) || (bs >> 1) >= nb) continue;
        cf* u = U + (bs * NCH + m) * LENP + tb;
        cf v[A];
#pragma unroll
        for (int ta = 0; ta < A; ++ta) v[ta] = u[ta * B];
        RegFFT<real_t, A, true>::run(v);
        u[0] = v[0];
#pragma unroll
        for (int ka = 1; ka < A; ++ka) u[ka * B] = v[ka] * conj(tw[ka * tb]);
    }
    __syncthreads();
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 4] = clock64();
    // ---- P5: second stage, inter-pass twiddle conj(W_L^(row*c)), store
    {
        cf* S2b = a.S2 + (size_t)b0 * bstride_o;
        for (int row0 = 0; row0 < BG * 2 * NO; row0 += RPR) {
            const int rl = row0 + tid % RPR, ka = tid / RPR;
            const int bs = rl / NO, m = rl % NO;
            const int slot = bs & 1, bb = bs >> 1;
            if (rl >= BG * 2 * NO || (slot && selfm) || bb >= nb) continue;
            const int row = slot ? rm : r;
            cf v[B];
            const cf* u = U + (bs * NCH + m) * LENP + ka * B;
#pragma unroll
            for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
            RegFFT<real_t, B, true>::run(v);
            const cf w1 = a.W[2 * row * ka];
            const unsigned dst0 = (unsigned)bb * bstride_o + (unsigned)row * (unsigned)a.L2 * NO + m;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) {
                const int c = ka + A * kb;
                st_nt(S2b, ESZ * (dst0 + (unsigned)c * NO), v[kb] * conj(w1 * wi2[slot * B + kb]));
            }
        }
    }
    if (a.dbg_times && tid == 0) a.dbg_times[(size_t)blockIdx.x * 8 + 5] = clock64();
}

// ---------------------------------------------------------------- backward: dL/dH without the gradient's spectrum in HBM
// dH[m][n][i] = sum_b gY[b][m][i] conj(X[b][n][i]) for the shapes and precisions the batch-walking kernels of specwalk.hip do not
// take (float64 -- the reference examples' default dtype --, fewer than four batch items).  The layered form is two launches:
// spec_mid without a response turns the gradient's scratch rows into its spectrum (one read + one write of the signal), then
// fl_mimo_gradh reads that spectrum and the kept one (two reads): 4.25 signal passes.  Here a workgroup owns (row pair,
// output-channel group) and WALKS THE BATCH: per item the row FFTs of its NOL gradient channels (P1, P2 as in spec_mid), the
// split step in registers, and the outer product with the kept spectrum accumulated in registers -- thread (bin pair p, half
// ms) holds dH[2 bins][NOL / 2][NI]; 2.25 passes, no atomics, the sum over the batch in a fixed order.
// Both operands of an item arrive by LDS-DMA (global_load_lds_dwordx4, no register round trip: in double the accumulators are
// half of the register file and a register prefetch of the rows spilled -- 1.0 ms): the gradient rows of item b+1 are in flight
// while item b runs its second stage and its products, the kept spectrum of item b+1 while item b+1 runs its two FFT stages.
// One residue class of a size-R transform, out[k] = X[S k + E], from all R inputs (decimation in frequency by S: S threads share
// a transform and each keeps R / S values -- a quarter of the registers of the whole transform in one thread, which is what lets
// the double-precision accumulators below stay in registers)
template <typename T, int R, int S, int E, typename LD>
__device__ __forceinline__ void fft_residue(LD ld, cx<T> (&y)[R / S]) {
    constexpr int Q = R / S;
    constexpr TwTab<R> tw = TwTab<R>();
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        cx<T> acc = ld(j);
#pragma unroll
        for (int s = 1; s < S; ++s) {
            const int m = ((s * E) % S) * Q;                    // W_S^(sE) = W_R^(Q s E)
            const cx<T> x = ld(j + s * Q);
            if (m == 0) acc = acc + x;
            else if (2 * m == R) acc = acc - x;
            else if (4 * m == R) acc = acc + mul_mi(x);
            else if (4 * m == 3 * R) acc = acc + mul_i(x);
            else fma_cx(acc, x, cx<T>((T)tw.re[m], (T)tw.im[m]));
        }
        const int mj = (j * E) % R;
        y[j] = mj == 0 ? acc : mul_plain(acc, cx<T>((T)tw.re[mj], (T)tw.im[mj]));
    }
    RegFFT<T, Q, false>::run(y);
}

struct GradLoopArgs {
    const cf* Sg;         // (Bn, L1, L2, NO): column pass of the output's gradient
    const cf* Xs;         // kept spectrum, row-major bin order: Xs[b*xs_b + n*xs_n + i]
    long xs_b, xs_n;
    cf* dH;               // dH[m*ds_m + n*ds_n + i], row-major bin order
    long ds_m, ds_n;
    const cf* W;
    int n, L, L1, L2, Bn;
    real_t scale_g;       // scale of the gradient's forward transform
    int interior2;        // double its interior bins (irfft backward)
    real_t out_scale;     // host factor on dH
    const real_t* dev_scale;   // device scalar multiplied into dH, or null
    long long* dbg;       // tuning: cycle sums per phase (8 per workgroup), or null
};

template <int A, int B, int NI, int NO, int NSC>
struct GradLoopShape {
    static constexpr int LEN = A * B, LENP = LEN | 1, NOL = NO / NSC, NT = 512, NW = NT / 64;
    static constexpr int PC = 16 / (int)sizeof(cf);                 // complex values per 16-byte piece
    static constexpr int CP = NOL / PC;                              // pieces per time step of the gradient rows
    static constexpr int GP = 2 * LEN * CP, XP = 2 * NI * LEN / PC;  // pieces per item: gradient rows, kept spectrum
    static constexpr int GI = (GP + 63) / 64, XI = (XP + 63) / 64;   // wavefront transfers per item
    static constexpr int GW = (GI + NW - 1) / NW, XW = (XI + NW - 1) / NW;   // ... per wavefront (the last ones repeated)
    static constexpr size_t lds_bytes = ((size_t)2 * LEN * NOL + (size_t)2 * NI * LEN + (size_t)2 * NOL * LENP + 2 * LEN) * sizeof(cf);
    static_assert(NO % NSC == 0 && NOL % 2 == 0 && NOL % PC == 0 && LEN % PC == 0 && LEN <= 256, "tile shape");
};

template <int A, int B, int NI, int NO, int NSC>
__global__ void __launch_bounds__(512) spec_gradh_loop(GradLoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using SH = GradLoopShape<A, B, NI, NO, NSC>;
    constexpr int LEN = SH::LEN, LENP = SH::LENP, NOL = SH::NOL, MPT = NOL / 2, NT = SH::NT, NW = SH::NW;
    constexpr int PC = SH::PC, CP = SH::CP, GP = SH::GP, XP = SH::XP, GI = SH::GI, XI = SH::XI, GW = SH::GW, XW = SH::XW;
    cf* G = reinterpret_cast<cf*>(smem);     // [2][LEN][NOL]   gradient rows of the item, as they lie in HBM
    cf* X = G + 2 * LEN * NOL;               // [2][NI][LEN]    kept spectrum of the item
    cf* U = X + 2 * NI * LEN;                // [2][NOL][LENP]  the rows between the stages, then their spectrum
    cf* tw = U + 2 * NOL * LENP;             // W_LEN^m
    cf* ws = tw + LEN;                       // W_n^(L1*k2)
    const int P = a.L1 / 2 + 1;
    // XCD-aware order: the channel groups of a row pair on one XCD (they read the same block of the kept spectrum)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int r = (q / NSC) * 8 + xcd, mo = (q % NSC) * NOL;
    if (r >= P) return;
    const int rm = (a.L1 - r) % a.L1;
    const bool selfm = rm == r;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const size_t bstride = (size_t)a.L1 * (size_t)a.L2 * NO;
    for (int j = tid; j < LEN; j += NT) {
        tw[j] = a.W[a.n + a.L1 + j];
        ws[j] = a.W[a.n + a.L1 + a.L2 + j];
    }
    // ---- the transfers of this wavefront: byte offsets inside an item, LDS bases (a wavefront whose turn lies behind the last
    // transfer repeats the last one: every wavefront issues the same count, which is what the counter waits below rely on)
    unsigned goff[GW], xoff[XW], glds[GW], xlds[XW];
    bool gon[GW], xon[XW];
#pragma unroll
    for (int k = 0; k < GW; ++k) {
        int ins = k * NW + wv;
        if (ins >= GI) ins = GI - 1;
        const int g = ins * 64 + lane;
        gon[k] = g < GP;
        const int gg = gon[k] ? g : 0;
        const int cc = gg % CP, t = (gg / CP) % LEN, slot = gg / (CP * LEN);
        goff[k] = ESZ * (((unsigned)(slot ? rm : r) * LEN + t) * NO + mo + cc * PC);
        glds[k] = lds_addr_of(G) + 1024u * ins;
    }
#pragma unroll
    for (int k = 0; k < XW; ++k) {
        int ins = k * NW + wv;
        if (ins >= XI) ins = XI - 1;
        const int x = ins * 64 + lane;
        xon[k] = x < XP;
        const int xx = xon[k] ? x : 0;
        const int jc = xx % (LEN / PC), nn = (xx / (LEN / PC)) % NI, slot = xx / (LEN / PC * NI);
        xoff[k] = ESZ * ((unsigned)nn * (unsigned)a.xs_n + (unsigned)(slot ? rm : r) * LEN + jc * PC);
        xlds[k] = lds_addr_of(X) + 1024u * ins;
    }
    auto issue_g = [&](int b) {
        const cf* Sb = a.Sg + (size_t)b * bstride;
#pragma unroll
        for (int k = 0; k < GW; ++k)
            if (gon[k]) dma16s(Sb, goff[k], glds[k]);
    };
    auto issue_x = [&](int b) {
        const cf* Xb = a.Xs + (size_t)b * a.xs_b;
#pragma unroll
        for (int k = 0; k < XW; ++k)
            if (xon[k]) dma16s(Xb, xoff[k], xlds[k]);
    };
    // the product's thread: bin pair p, output channels [mo + ms MPT, + MPT)
    const int p = tid & 255, ms = tid >> 8;
    int slotB = 0, colB = 0;
    bool dc = false;
    const bool valid = p < LEN && pair_of(r, selfm, p, LEN, slotB, colB, dc);
    const unsigned ik = (unsigned)r * LEN + (p < LEN ? p : 0);
    const unsigned im = dc ? (unsigned)a.L : (unsigned)(slotB ? rm : r) * LEN + colB;
    const bool two = im != ik;
    cf acck[MPT][NI], accm[MPT][NI];
#pragma unroll
    for (int m = 0; m < MPT; ++m)
#pragma unroll
        for (int nn = 0; nn < NI; ++nn) acck[m][nn] = accm[m][nn] = cf(0, 0);
    __syncthreads();
    const cf wk = a.W[r] * ws[p < LEN ? p : 0];
    const cf wm(-wk.x, wk.y);                                   // W_n^(L-k) = -conj(W_n^k)
    const real_t hs = (real_t)0.5 * a.scale_g * (a.interior2 ? (real_t)2 : (real_t)1);
    // Both FFT stages are shared by S threads per transform (fft_residue): thread (e = tid / 128, item = tid % 128), e uniform
    // over a wavefront.  P1 item: (slot, tb, nn), nn fastest; P2 item: (row rl = slot * NOL + nn, ka), rl fastest.
    constexpr int S1 = 2, S2 = B % 3 == 0 ? 3 : 4;
    static_assert(2 * B * NOL <= 128 && 2 * NOL * A <= 128 && A % S1 == 0 && B % S2 == 0, "128 items per residue class");
    const int e12 = __builtin_amdgcn_readfirstlane(tid >> 7), it = tid & 127;
    const int nn1 = it % NOL, tb1 = (it / NOL) % B, slot1 = it / (NOL * B);
    const bool have1 = e12 < S1 && it < 2 * B * NOL && !(slot1 && selfm);
    const int rl = it % (2 * NOL), ka = it / (2 * NOL);
    const bool act2 = e12 < S2 && it < 2 * NOL * A && !((rl / NOL) && selfm);
    if (a.Bn > 0) {
        issue_g(0);
        issue_x(0);
        wait_vm<XW>();                       // the rows have landed (this wavefront's share; the spectrum may stay in flight)
    }
    lds_barrier();
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tq = 0;
#define FL_STAMP(i) if (a.dbg) { const long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tq; tq = t_; }
    if (a.dbg) tq = __builtin_readcyclecounter();
#pragma unroll 1
    for (int b = 0; b < a.Bn; ++b) {
        const bool more = b + 1 < a.Bn;
        // ---- P1: first stage of the gradient rows
        if (have1) {
            const cf* g = G + ((slot1 * LEN + tb1) * NOL + nn1);
            auto ld = [&](int ta) { return g[ta * B * NOL]; };
            cf v1[A / S1];
            if (e12 == 0) fft_residue<real_t, A, S1, 0>(ld, v1); else fft_residue<real_t, A, S1, 1>(ld, v1);
            cf* u = U + (slot1 * NOL + nn1) * LENP + tb1;
#pragma unroll
            for (int k = 0; k < A / S1; ++k) {
                const int kk = S1 * k + e12;
                u[kk * B] = v1[k] * tw[kk * tb1];
            }
        }
        FL_STAMP(0)
        lds_barrier();
        FL_STAMP(1)
        if (more) issue_g(b + 1);            // the next item's rows: in flight through P2 and P3
        // ---- P2: second stage, natural order in place
        {
            cf v[B / S2];
            cf* urow = U + rl * LENP;
            if (act2) {
                auto ld = [&](int tb) { return urow[ka * B + tb]; };
                if (e12 == 0) fft_residue<real_t, B, S2, 0>(ld, v);
                else if (e12 == 1) fft_residue<real_t, B, S2, 1>(ld, v);
                else if (e12 == 2) fft_residue<real_t, B, S2, 2>(ld, v);
                else fft_residue<real_t, B, S2, S2 - 1>(ld, v);
            }
            lds_barrier();
            if (act2) {
#pragma unroll
                for (int k = 0; k < B / S2; ++k) urow[ka + A * (S2 * k + e12)] = v[k];
            }
        }
        FL_STAMP(2)
        if (more) wait_vm<GW>(); else wait_vm<0>();      // this item's kept spectrum has landed
        lds_barrier();
        FL_STAMP(3)
        // ---- P3: split step of this thread's channels, outer product with the kept spectrum
        if (valid) {
            cf gk[MPT], gm[MPT];
#pragma unroll
            for (int m2 = 0; m2 < MPT; ++m2) {
                const int m = ms * MPT + m2;
                const cf zk = U[m * LENP + p];
                const cf zm = U[(slotB * NOL + m) * LENP + colB];
                if (dc) {
                    gk[m2] = cf(a.scale_g * (zk.x + zk.y), 0);      // gY[0]
                    gm[m2] = cf(a.scale_g * (zk.x - zk.y), 0);      // gY[L]
                } else {
                    const cf pk = zk + conj(zm), dk = zk - conj(zm);
                    const cf ok = pk + mul_mi(wk * dk);
                    const cf pm = zm + conj(zk), dm = zm - conj(zk);
                    const cf om = pm + mul_mi(wm * dm);
                    gk[m2] = cf(hs * ok.x, hs * ok.y);
                    gm[m2] = cf(hs * om.x, hs * om.y);
                }
            }
            const cf* xr = X + p;
            const cf* xq = X + slotB * NI * LEN + colB;
#pragma unroll
            for (int nn = 0; nn < NI; ++nn) {
                const cf xk = xr[nn * LEN];
                // the Nyquist bin is not part of any row: the one thread that owns it reads it where it lies
                const cf xm = dc ? a.Xs[(size_t)b * a.xs_b + (size_t)nn * a.xs_n + a.L] : xq[nn * LEN];
#pragma unroll
                for (int m2 = 0; m2 < MPT; ++m2) {
                    fma_cxc(acck[m2][nn], gk[m2], xk);           // += g conj(x)
                    fma_cxc(accm[m2][nn], gm[m2], xm);
                }
            }
        }
        FL_STAMP(4)
        wait_vm<0>();                        // the next item's rows have landed (a second stage and a product in flight)
        lds_barrier();                       // ... everybody's, and the spectrum's region is free
        if (more) issue_x(b + 1);            // in flight through the next item's two FFT stages
        FL_STAMP(5)
    }
#undef FL_STAMP
    if (a.dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = ph[i];
    }
    if (valid) {
        real_t os = a.out_scale;
        if (a.dev_scale) os *= *a.dev_scale;
#pragma unroll
        for (int m2 = 0; m2 < MPT; ++m2) {
            const int m = mo + ms * MPT + m2;
#pragma unroll
            for (int nn = 0; nn < NI; ++nn) {
                cf* o = a.dH + (size_t)m * a.ds_m + (size_t)nn * a.ds_n;
                at(o, ESZ * ik) = cf(os * acck[m2][nn].x, os * acck[m2][nn].y);
                if (two) at(o, ESZ * im) = cf(os * accm[m2][nn].x, os * accm[m2][nn].y);
            }
        }
    }
}


// ---------------------------------------------------------------- contiguous twiddle copies
// W[n + j] = W_L1^j (j < L1), W[n + L1 + j] = W_L2^j (j < L2), W[n + L1 + L2 + j] = W_n^(L1 j) (j < L2): what every
// workgroup of the three kernels stages into LDS -- as contiguous runs instead of L1 + 2 L2 gathers from the master table
// (each gather a cache line of its own, and a dependent round trip in front of the first barrier)
__global__ void __launch_bounds__(256) spec_aux_fill_kernel(cf* W, int n, int L1, int L2) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < L1) W[n + j] = W[j * (n / L1)];
    if (j < L2) {
        W[n + L1 + j] = W[j * (n / L2)];
        W[n + L1 + L2 + j] = W[L1 * j];
    }
}

// ---------------------------------------------------------------- bin order conversion
// natural bin order k <-> row-major order i = (k % L1) * L2 + k / L1 (Nyquist bin L stays at L), per plane
__global__ void __launch_bounds__(256) permute_bins_kernel(const cf* __restrict__ src, long sp, cf* __restrict__ dst, long dp,
                                                           int L1, int L2, int inverse) {
    const int L = L1 * L2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > L) return;
    const cf* s = src + (size_t)blockIdx.y * sp;
    cf* d = dst + (size_t)blockIdx.y * dp;
    const int k = (i == L) ? L : (i / L2) + L1 * (i % L2);
    if (inverse) d[k] = s[i];      // row-major -> natural
    else d[i] = s[k];              // natural -> row-major (coalesced stores)
}

// ---------------------------------------------------------------- host side
// (nfft, L1, L2): nfft/2 = L1 L2, column length x row length, both products of two in-register FFT sizes with factors 2, 3,
// 5 (launch_cols / launch_mid below), L2 >= L1 so that a row pair of all channels stays small.  The three BASELINE lengths
// carry every channel combination and the tuning variants; the others -- the reference's default 2^11 (dsp.py:84), the
// powers of two up to 2^17, 48000 and 144000 (one and three seconds at 48 kHz) -- the equal-channel kernels.
}  // namespace FL_SPEC_NS
#ifndef FL_F64
struct PlanEntry { int nfft, L1, L2, lean; };
// the tuned lengths: every channel combination and the tuning variants exist for the three BASELINE lengths
static const PlanEntry kPlans[] = {{96000, 200, 240, 0}, {192000, 400, 240, 0}, {384000, 800, 240, 0},
                                   {2048, 32, 32, 1},    {4096, 32, 64, 1},     {8192, 64, 64, 1},     {16384, 64, 128, 1},
                                   {32768, 128, 128, 1}, {65536, 128, 256, 1},  {131072, 256, 256, 1}, {48000, 150, 160, 1},
                                   {144000, 225, 320, 1}};
// Every other length is PLANNED from the factorisation: nfft/2 = L1 L2 with L1 among the column lengths and L2 among the row
// lengths that are instantiated (cols_launch / fl_spec_mid below; each a product of two in-register FFT sizes) -- 32000
// (examples/e4_recursion_nn.py:349: 50 x 320), 44100 (441 x 50), 88200 (441 x 100), 64000, 24000, 16000, 160000, 256000 ...
// (a row length is 2^k B, k <= 5: the row kernel's threads divide evenly over the k_a of its second stage)
// Preference: the row length nearest to 256 (a row pair of all channels in LDS, 240..256 bin pairs for a workgroup's threads),
// rows not shorter than columns.  Planned lengths take the equal-channel kernels.
static const int kColLens[] = {32, 50, 64, 75, 100, 125, 128, 150, 200, 225, 250, 256, 300, 400, 441, 800};
static const int kRowLens[] = {32, 50, 64, 100, 128, 160, 200, 240, 256, 320, 400, 480};
static const PlanEntry* plan_of(int nfft) {
    for (const PlanEntry& p : kPlans)
        if (p.nfft == nfft) return &p;
    return nullptr;
}

static bool plan_search(int nfft, int& L1, int& L2) {
    if (nfft < 2 || (nfft & 1)) return false;
    const int L = nfft / 2;
    double best = 1e30;
    bool found = false;
    for (int l2 : kRowLens) {
        if (L % l2) continue;
        const int l1 = L / l2;
        bool have = false;
        for (int c : kColLens) have = have || c == l1;
        if (!have) continue;
        double cost = fabs(log2((double)l2 / 256.0)) + (l2 < l1 ? 1.0 : 0.0);
        if (cost < best) {
            best = cost;
            L1 = l1;
            L2 = l2;
            found = true;
        }
    }
    return found;
}

int spec_plan(int nfft, int& L1, int& L2) {
    if (const PlanEntry* p = plan_of(nfft)) {
        L1 = p->L1;
        L2 = p->L2;
        return FL_OK;
    }
    if (plan_search(nfft, L1, L2)) return FL_OK;
    set_error("spectral: nfft=%d has no fused plan (nfft/2 is not a product of an instantiated column and row length)", nfft);
    return FL_ERR_UNSUPPORTED;
}

int spec_plan_lean(int nfft) {
    const PlanEntry* p = plan_of(nfft);
    return p ? p->lean : 1;
}
#endif
namespace FL_SPEC_NS {

static int g_spec_vt = 0, g_spec_rg = 0;      // 0 = pick per shape (below); fl_debug_set_spec overrides
static size_t g_cols_inv_min_lds = [] { const char* e = getenv("FLAMO_COLS_INV_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();

// Column-pass tile: virtual columns per workgroup and loads in flight per thread group. Measured under graph replay on
// the whole training step (tools/dbg/graph_ab.py): 16 virtual columns (27 KB of LDS, five workgroups per CU) win for
// up to 8 channels at every plan length (3 % at nfft=96000, 11 % at 192000); with 16 channels a 16-wide tile would be a
// single column of 128-byte segments and the 32-wide tile with all loads in flight is ahead.
// (float64: always the 16-wide tile -- its LDS is twice the float32 tile's)
// (the 32-wide tile of a long column pass does not fit the LDS: 32 x 801 values at L1 = 800 -- the 16-wide tile then)
static int cols_vt(int G, int L1 = 0) {
    int vt = sizeof(real_t) == 8 ? 16 : g_spec_vt ? g_spec_vt : (G <= 8 ? 16 : 32);
    while (vt > 8 && ((size_t)vt * (L1 | 1) + L1 + vt * 25) * sizeof(cf) > 150 * 1024) vt >>= 1;      // (8: float64 at L1 = 800)
    return vt;
}
static int cols_vt_of(int nfft, int G) {
    int l1 = 0, l2 = 0;
    if (spec_plan(nfft, l1, l2) != FL_OK) l1 = 0;
    return cols_vt(G, l1);
}
static int cols_rg(int vt, int L1) { return sizeof(real_t) == 8 ? 1 : g_spec_rg ? g_spec_rg : (vt == 32 ? 4 : (L1 <= 200 ? 1 : 2)); }

static int cols_setup(ColsArgs& a, int nfft, int Bn, int t_len, int t_lim, int G, const void* W, int vt) {
    int L1, L2;
    int rc = spec_plan(nfft, L1, L2);
    if (rc) return rc;
    FL_REQUIRE(Bn > 0 && G >= 2 && (G & 1) == 0 && t_len >= 0 && W, "spectral cols: bad arguments (even channel count >= 2)");
    int cg = G < vt ? G : vt;
    FL_REQUIRE((cg & (cg - 1)) == 0 && G % cg == 0, "spectral cols: channel count must be a power of two or a multiple of %d", vt);
    int cgs = 0;
    while ((1 << cgs) < cg) ++cgs;
    a.n = nfft; a.L = nfft / 2; a.L1 = L1; a.L2 = L2; a.G = G; a.cgs = cgs; a.CT = vt / cg;
    FL_REQUIRE(L2 % a.CT == 0, "spectral cols: column tile does not divide the row length");
    a.nct = L2 / a.CT; a.ngt = G / cg;
    a.t_len = t_len; a.t_lim = t_lim < nfft ? t_lim : nfft;
    a.W = (const cf*)W;
    return FL_OK;
}

template <int A, int B, bool LEAN = false>
static void launch_cols(bool inverse, const ColsArgs& a, unsigned nblk, hipStream_t st) {
    constexpr int LEN = A * B, LENP = LEN | 1;
#define FL_COLS(VT_, RG_)                                                                                        \
    {                                                                                                            \
        size_t lds = ((size_t)VT_ * LENP + LEN + (size_t)VT_ * B) * sizeof(cf);                                  \
        if (inverse && g_cols_inv_min_lds > lds) lds = g_cols_inv_min_lds;                                       \
        if constexpr (A * VT_ <= 256) {                                                                          \
            if (inverse && a.Sg) {                                                                               \
                hipLaunchKernelGGL((spec_cols_inv<A, B, VT_, RG_, true, true>), dim3(nblk), dim3(256), lds, st, a); \
                return;                                                                                          \
            }                                                                                                    \
        }                                                                                                        \
        if (inverse && a.env_log2 == 0.0 && a.t_lim >= a.n)                                                      \
            hipLaunchKernelGGL((spec_cols_inv<A, B, VT_, RG_, true>), dim3(nblk), dim3(256), lds, st, a);        \
        else if (inverse) hipLaunchKernelGGL((spec_cols_inv<A, B, VT_, RG_, false>), dim3(nblk), dim3(256), lds, st, a); \
        else if (a.env_log2 == 0.0 && a.t_lim >= a.n)                                                            \
            hipLaunchKernelGGL((spec_cols_fwd<A, B, VT_, RG_, true>), dim3(nblk), dim3(256), lds, st, a);        \
        else hipLaunchKernelGGL((spec_cols_fwd<A, B, VT_, RG_, false>), dim3(nblk), dim3(256), lds, st, a);      \
    }
    const int vt = a.CT << a.cgs;
    if constexpr (sizeof(real_t) == 8) {      // float64: the 16-wide tile, one load group (its values are four registers each)
        if constexpr (A * B >= 800) {
            if (vt == 8) FL_COLS(8, 1) else FL_COLS(16, 1)      // (the longest column pass: a 16-wide tile of doubles exceeds the LDS)
        } else
        FL_COLS(16, 1)
    } else if constexpr (LEAN) {              // one load-group choice per tile width
        if (vt == 32) FL_COLS(32, 2) else FL_COLS(16, 2)
    } else {
        if (vt == 32) {
            const int rg = cols_rg(32, a.L1);
            if (rg == 1) FL_COLS(32, 1) else if (rg == 4) FL_COLS(32, 4) else FL_COLS(32, 2)
        } else {
            if (cols_rg(16, a.L1) == 1) FL_COLS(16, 1) else FL_COLS(16, 2)
        }
    }
#undef FL_COLS
}

static int cols_launch(bool inverse, const ColsArgs& a, int Bn, hipStream_t st) {
    const size_t nblk = (size_t)Bn * a.nct * a.ngt;
    FL_REQUIRE(nblk < (1ull << 31), "spectral cols: grid too large");
    {   // a cascade-response launch recorded by this thread (fusedfwd.h): beside the float32 forward pass in one grid when that
        // has the shape for it, otherwise in front of this launch on its own
        PendingRc rc;
        if (pending_rc_take(rc)) {
            int r = FL_ERR_UNSUPPORTED;
#ifndef FL_F64
            if (!inverse) r = fused_cols_rc_launch(a, (unsigned)nblk, rc, st);
#endif
            if (r != FL_ERR_UNSUPPORTED) return r;
            r = rc_ba_launch_now(rc, st);
            if (r) return r;
        }
    }
    switch (a.L1) {
        case 200: launch_cols<8, 25>(inverse, a, (unsigned)nblk, st); break;
        case 300: launch_cols<12, 25>(inverse, a, (unsigned)nblk, st); break;
        case 400: launch_cols<16, 25>(inverse, a, (unsigned)nblk, st); break;
        case 32: launch_cols<8, 4, true>(inverse, a, (unsigned)nblk, st); break;
        case 64: launch_cols<8, 8, true>(inverse, a, (unsigned)nblk, st); break;
        case 128: launch_cols<16, 8, true>(inverse, a, (unsigned)nblk, st); break;
        case 150: launch_cols<10, 15, true>(inverse, a, (unsigned)nblk, st); break;
        case 225: launch_cols<15, 15, true>(inverse, a, (unsigned)nblk, st); break;
        case 256: launch_cols<16, 16, true>(inverse, a, (unsigned)nblk, st); break;
        case 50: launch_cols<2, 25, true>(inverse, a, (unsigned)nblk, st); break;
        case 75: launch_cols<5, 15, true>(inverse, a, (unsigned)nblk, st); break;
        case 100: launch_cols<4, 25, true>(inverse, a, (unsigned)nblk, st); break;
        case 125: launch_cols<5, 25, true>(inverse, a, (unsigned)nblk, st); break;
        case 250: launch_cols<10, 25, true>(inverse, a, (unsigned)nblk, st); break;
        case 441: launch_cols<21, 21, true>(inverse, a, (unsigned)nblk, st); break;
        case 800: launch_cols<32, 25, true>(inverse, a, (unsigned)nblk, st); break;
        default: set_error("spectral cols: unsupported column length %d", a.L1); return FL_ERR_UNSUPPORTED;
    }
    FL_CHECK_LAUNCH(inverse ? "spec_cols_inv" : "spec_cols_fwd");
    return FL_OK;
}

static int g_mid_bg = 1, g_mid_hfake = 0, g_mid_pfd = 0;
static long long* g_mid_times = nullptr;
}  // namespace FL_SPEC_NS
// tuning: cycle sums per phase of spec_gradh_loop (8 per workgroup), shared by the float32 and float64 builds
#ifndef FL_F64
long long* g_gradloop_times = nullptr;
#else
extern long long* g_gradloop_times;
#endif
namespace FL_SPEC_NS {

template <int A, int B, int NI, int NO, int BG, int MS>
static void launch_mid_bg(const MidArgs& a, hipStream_t st) {
    constexpr int LEN = A * B, LENP = LEN | 1, NCH = NI > NO ? NI : NO;
    const size_t lds = ((size_t)BG * 2 * NCH * LENP + 2 * LEN + 2 * B + BG * NCH) * sizeof(cf);
    const int P = a.L1 / 2 + 1;
    const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * cdiv_i(a.Bn, BG));
    if (a.S2) {
        if (a.H) {
            if (g_mid_pfd == 1) hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, BG, MS, 1>), dim3(nblk), dim3(256 * MS), lds, st, a);
            else if (g_mid_pfd == 2) hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, BG, MS, 2>), dim3(nblk), dim3(256 * MS), lds, st, a);
            else hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, BG, MS, 0>), dim3(nblk), dim3(256 * MS), lds, st, a);
        }
        else if constexpr (NI == NO) hipLaunchKernelGGL((spec_mid<A, B, NI, NO, false, true, BG, MS>), dim3(nblk), dim3(256 * MS), lds, st, a);
    } else {
        if constexpr (NI == NO) hipLaunchKernelGGL((spec_mid<A, B, NI, NO, false, false, BG, MS>), dim3(nblk), dim3(256 * MS), lds, st, a);
    }
}

template <int A, int B, int NI, int NO>
static void launch_mid_n(const MidArgs& a, unsigned, hipStream_t st) {
    // (two batch items per workgroup -- the response row applied to two spectra -- measured slower at config 2:
    // 104 us with 256 threads, 118 us with 512, against 98 us; the kernel keeps the BG/MS parameters for that experiment)
    // response present + inverse half: BG batch items per workgroup with the three-sweep product (the response row of a bin
    // is fetched once per BG items)
    if (a.H && a.S2 && g_mid_bg == 5) {       // one item per workgroup, three-sweep product
        constexpr int LEN = A * B, LENP = LEN | 1, NCH = NI > NO ? NI : NO;
        const int P = a.L1 / 2 + 1;
        const size_t lds = ((size_t)2 * NCH * LENP + 2 * LEN + 2 * B + NCH) * sizeof(cf);
        const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * a.Bn);
        hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, 1, 1, 0, 1, 1>), dim3(nblk), dim3(256), lds, st, a);
        return;
    }
    if (a.H && a.S2 && a.Bn > 1 && g_mid_bg >= 2) {
        constexpr int LEN = A * B, LENP = LEN | 1, NCH = NI > NO ? NI : NO;
        const int P = a.L1 / 2 + 1;
        if constexpr ((size_t)2 * 2 * NCH * LENP * sizeof(cf) <= 70 * 1024) {
            if (g_mid_bg == 2 || a.Bn < 4 || (size_t)4 * 2 * NCH * LENP * sizeof(cf) > 150 * 1024) {
                const size_t lds = ((size_t)2 * 2 * NCH * LENP + 2 * LEN + 2 * B + 2 * NCH) * sizeof(cf);
                const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * cdiv_i(a.Bn, 2));
                hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, 2, 1, 0, 2, 1>), dim3(nblk), dim3(512), lds, st, a);
                return;
            }
        }
        if constexpr ((size_t)4 * 2 * NCH * LENP * sizeof(cf) <= 150 * 1024) {
            if (g_mid_bg == 4 && a.Bn >= 4) {
                const size_t lds = ((size_t)4 * 2 * NCH * LENP + 2 * LEN + 2 * B + 4 * NCH) * sizeof(cf);
                const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * cdiv_i(a.Bn, 4));
                hipLaunchKernelGGL((spec_mid<A, B, NI, NO, true, true, 4, 1, 0, 4, 1>), dim3(nblk), dim3(1024), lds, st, a);
                return;
            }
        }
    }
    launch_mid_bg<A, B, NI, NO, 1, 1>(a, st);
}

// equal channel counts, one batch item per workgroup, no tuning variants (the lengths beyond the three BASELINE ones)
template <int A, int B, int N>
static void launch_mid_lean_n(const MidArgs& a, hipStream_t st) {
    constexpr int LEN = A * B, LENP = LEN | 1;
    const size_t lds = ((size_t)2 * N * LENP + 2 * LEN + 2 * B + N) * sizeof(cf);
    const int P = a.L1 / 2 + 1;
    const unsigned nblk = (unsigned)(cdiv_i(P, 8) * 8 * a.Bn);
    if (a.S2) {
        if (a.H) hipLaunchKernelGGL((spec_mid<A, B, N, N, true, true, 1, 1, 0>), dim3(nblk), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((spec_mid<A, B, N, N, false, true, 1, 1>), dim3(nblk), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL((spec_mid<A, B, N, N, false, false, 1, 1>), dim3(nblk), dim3(256), lds, st, a);
    }
}

template <int A, int B>
static int launch_mid_lean(const MidArgs& a, int NI, int NO, hipStream_t st) {
    if (NI == NO) {
        switch (NI) {
            case 2: launch_mid_lean_n<A, B, 2>(a, st); return FL_OK;
            case 4: launch_mid_lean_n<A, B, 4>(a, st); return FL_OK;
            case 8: launch_mid_lean_n<A, B, 8>(a, st); return FL_OK;
            case 16: launch_mid_lean_n<A, B, 16>(a, st); return FL_OK;
        }
    }
    set_error("spectral mid: no kernel for %d -> %d channels at this transform length (2, 4, 8 or 16 channels, equal in and out)", NI, NO);
    return FL_ERR_UNSUPPORTED;
}

template <int A, int B>
static int launch_mid(const MidArgs& a, int NI, int NO, unsigned nblk, hipStream_t st) {
#define FL_MID(NI_, NO_)                              \
    if (NI == NI_ && NO == NO_) {                     \
        launch_mid_n<A, B, NI_, NO_>(a, nblk, st);    \
        return FL_OK;                                 \
    }
    FL_MID(2, 2) FL_MID(4, 4) FL_MID(8, 8) FL_MID(16, 16)
    FL_MID(2, 4) FL_MID(4, 2) FL_MID(2, 8) FL_MID(8, 2) FL_MID(4, 8) FL_MID(8, 4)
#undef FL_MID
    set_error("spectral mid: no kernel for %d -> %d channels", NI, NO);
    return FL_ERR_UNSUPPORTED;
}

}  // namespace FL_SPEC_NS
}  // namespace fl

using namespace fl;
using namespace fl::FL_SPEC_NS;

#ifdef FL_F64
#define FL_SPEC_FN(base) base##_f64
#define FL_SPEC_CFN(base) base##_c128
#else
#define FL_SPEC_FN(base) base##_f32
#define FL_SPEC_CFN(base) base##_c64
#endif

// LDS of the column kernels for G channels: (VT (L1 | 1) + L1 + VT B) complex values, B <= 25
static size_t cols_lds_need(int L1, int G) {
    const int vt = cols_vt(G, L1);
    return ((size_t)vt * (L1 | 1) + L1 + (size_t)vt * 25) * sizeof(cf);
}

extern "C" {

#ifndef FL_F64
int fl_spec_plan(int nfft, int* L1, int* L2) {
    int l1 = 0, l2 = 0;
    int rc = spec_plan(nfft, l1, l2);
    if (rc) return rc;
    if (L1) *L1 = l1;
    if (L2) *L2 = l2;
    return FL_OK;
}

size_t fl_spec_aux_elems(int nfft) {
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK) return 0;
    return (size_t)l1 + 2 * (size_t)l2;
}

#endif

// the batch-walking gradient kernel takes 240- and 256-bin rows, equal channel counts of 2 / 4 / 8
static bool gradh_loop_shape(int l2, int NI, int NO) { return (l2 == 240 || l2 == 256) && NI == NO && (NI == 2 || NI == 4 || NI == 8); }

int FL_SPEC_FN(fl_spec_gradh_loop_supports)(int nfft, int NI, int NO) {
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK) return 0;
    return gradh_loop_shape(l2, NI, NO) ? 1 : 0;
}

extern "C++" {
template <int A, int B, int N>
static int launch_gradh_loop(const GradLoopArgs& a, hipStream_t st) {
    constexpr int NSC = N >= 4 ? 2 : 1;
    using SH = GradLoopShape<A, B, N, N, NSC>;
    // more dynamic LDS than the default 64 KB: the attribute is per function and per device, set once each and checked
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!done[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spec_gradh_loop<A, B, N, N, NSC>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SH::lds_bytes);
        if (e != hipSuccess) {
            set_error("spec_gradh_loop: %zu bytes of LDS per workgroup are not available on device %d (%s)", SH::lds_bytes, dev, hipGetErrorString(e));
            return FL_ERR_UNSUPPORTED;
        }
        done[dev] = true;
    }
    const int P = a.L1 / 2 + 1;
    hipLaunchKernelGGL((spec_gradh_loop<A, B, N, N, NSC>), dim3((unsigned)(cdiv_i(P, 8) * 8 * NSC)), dim3(512), SH::lds_bytes, st, a);
    return FL_OK;
}
}

int FL_SPEC_FN(fl_spec_gradh_loop)(const void* Sg, const void* Xs, long xs_b, long xs_n, void* dH, long ds_m, long ds_n, const void* W,
                                   int nfft, int Bn, int NI, int NO, double scale_g, int interior2, double out_scale,
                                   const void* dev_scale, void* stream) {
    FL_REQUIRE(Sg && Xs && dH && W, "spec_gradh_loop: null pointer");
    GradLoopArgs a = {};
    int rc = spec_plan(nfft, a.L1, a.L2);
    if (rc) return rc;
    if (!gradh_loop_shape(a.L2, NI, NO)) {
        set_error("spec_gradh_loop: shape not taken (ask fl_spec_gradh_loop_supports): rows of %d bins, %d -> %d channels", a.L2, NI, NO);
        return FL_ERR_UNSUPPORTED;
    }
    FL_REQUIRE(Bn >= 0 && (size_t)a.L1 * a.L2 * NO * sizeof(cf) * (size_t)(Bn > 0 ? 1 : 0) < (1ull << 32), "spec_gradh_loop: bad sizes");
    a.Sg = (const cf*)Sg; a.Xs = (const cf*)Xs; a.xs_b = xs_b; a.xs_n = xs_n; a.dH = (cf*)dH; a.ds_m = ds_m; a.ds_n = ds_n;
    a.W = (const cf*)W; a.n = nfft; a.L = nfft / 2; a.Bn = Bn;
    a.scale_g = (real_t)scale_g; a.interior2 = interior2; a.out_scale = (real_t)out_scale; a.dev_scale = (const real_t*)dev_scale;
    a.dbg = g_gradloop_times;
    hipStream_t st = (hipStream_t)stream;
    if (a.L2 == 240) {
        if (NI == 8) rc = launch_gradh_loop<16, 15, 8>(a, st);
        else if (NI == 4) rc = launch_gradh_loop<16, 15, 4>(a, st);
        else rc = launch_gradh_loop<16, 15, 2>(a, st);
    } else {
        if (NI == 8) rc = launch_gradh_loop<16, 16, 8>(a, st);
        else if (NI == 4) rc = launch_gradh_loop<16, 16, 4>(a, st);
        else rc = launch_gradh_loop<16, 16, 2>(a, st);
    }
    if (rc) return rc;
    FL_CHECK_LAUNCH("spec_gradh_loop");
    return FL_OK;
}

int FL_SPEC_FN(fl_spec_aux_fill)(void* W, int nfft, void* stream) {
    FL_REQUIRE(W, "spec_aux_fill: null pointer");
    int l1, l2;
    int rc = spec_plan(nfft, l1, l2);
    if (rc) return rc;
    hipLaunchKernelGGL(spec_aux_fill_kernel, dim3(cdiv_i(l2 > l1 ? l2 : l1, 256)), dim3(256), 0, (hipStream_t)stream, (cf*)W, nfft, l1, l2);
    FL_CHECK_LAUNCH("spec_aux_fill");
    return FL_OK;
}

#ifdef FL_F64
int fl_spec_supports_f64(int nfft, int n_in, int n_out) {
#else
int fl_spec_supports(int nfft, int n_in, int n_out) {
#endif
    int l1, l2;
    if (spec_plan(nfft, l1, l2) != FL_OK) return 0;
    auto ok = [](int c) { return c == 2 || c == 4 || c == 8 || c == 16; };
    if (!ok(n_in) || !ok(n_out)) return 0;
    if (n_in != n_out && (n_in > 8 || n_out > 8)) return 0;
    if (n_in != n_out && (spec_plan_lean(nfft) || sizeof(real_t) == 8)) return 0;
    // the row kernel holds a row pair of all channels in LDS: (2 max(n_in, n_out) (L2 | 1) + 2 L2 + ...) complex values --
    // 131 KB at 16 channels, nfft = 384000; a part with less LDS per workgroup than that takes the layered route
    static int lds_limit = 0;
    if (!lds_limit) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
            v = 64 * 1024;
        lds_limit = v;
    }
    // a column-pass workgroup owns CT = VT / min(G, VT) columns of all G channels: CT must divide the row length
    for (int G : {n_in, n_out}) {
        const int vt = cols_vt(G, l1), ct = vt / (G < vt ? G : vt);
        if (l2 % ct) return 0;
    }
    const int nch = n_in > n_out ? n_in : n_out;
    const size_t need = ((size_t)2 * nch * (l2 | 1) + 2 * (size_t)l2 + 64 + nch) * sizeof(cf);
    if (need > (size_t)lds_limit) return 0;
    return cols_lds_need(l1, n_in) <= (size_t)lds_limit && cols_lds_need(l1, n_out) <= (size_t)lds_limit ? 1 : 0;
}

#ifndef FL_F64
int fl_debug_set_spec_times(void* buf) {
    g_mid_times = (long long*)buf;
    g_gradloop_times = (long long*)buf;
    return FL_OK;
}

int fl_debug_set_spec(int vt, int rg) {
    g_spec_vt = (vt == 16 || vt == 32) ? vt : 0;
    g_spec_rg = (rg % 100 == 1 || rg % 100 == 4 || rg % 100 == 2) ? rg % 100 : 0;
    g_mid_bg = (rg >= 100) ? (rg / 100) % 10 : 1;
    g_mid_hfake = (rg / 1000) % 10 == 1;
    g_mid_pfd = (rg / 10000) % 10;       // rg = 10000*prefetch depth + 1000*fake + 100*bg + load group
    return FL_OK;
}
#endif

int FL_SPEC_FN(fl_spec_cols_fwd)(const void* x, int Bn, int t_len, int G, void* S, const void* W, int nfft, double env_log2,
                         void* stream) {
    FL_REQUIRE(x && S, "spec_cols_fwd: null pointer");
    FL_REQUIRE(reinterpret_cast<uintptr_t>(x) % (2 * RSZ) == 0, "spec_cols_fwd: x must be aligned to two samples");
    if (Bn == 0) return FL_OK;
    ColsArgs a = {};
    int rc = cols_setup(a, nfft, Bn, t_len, t_len, G, W, cols_vt_of(nfft, G));
    if (rc) return rc;
    a.x = (const real_t*)x;
    a.S = (cf*)S;
    a.env_log2 = env_log2;
    a.pol = (stream_policy() >> (stream_site() ? POL_SITE1_SHIFT : 0)) & 3u;
    return cols_launch(false, a, Bn, (hipStream_t)stream);
}

static int cols_inv_impl(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                         double env_log2, double* sumsq, void* stream, const real_t* dev_scale = nullptr, void* Sg = nullptr);
// the fused gradient pass (spec_cols_inv<..., FUSE>) exists where one thread takes one second-stage item of the tile (first radix
// of the column plan x tile width <= 256: every plan but the 441- and 800-point columns and the 32-wide tile of 16-point radices)
static int cols_first_radix(int l1) {      // A of launch_cols<A, B> (cols_launch)
    switch (l1) {
        case 200: return 8; case 300: return 12; case 400: return 16; case 32: return 8; case 64: return 8; case 128: return 16;
        case 150: return 10; case 225: return 15; case 256: return 16; case 50: return 2; case 75: return 5; case 100: return 4;
        case 125: return 5; case 250: return 10; case 441: return 21; case 800: return 32;
        default: return 0;
    }
}
static bool cols_inv_grad_ok(int nfft, int G) {
    int l1 = 0, l2 = 0;
    if (G < 2 || (G & 1) || spec_plan(nfft, l1, l2) != FL_OK) return false;
    const int a_len = cols_first_radix(l1);
    return a_len && a_len * cols_vt(G, l1) <= 256;      // (cols_vt: the tile width cols_launch takes)
}
int FL_SPEC_FN(fl_spec_cols_inv)(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                         double env_log2, void* stream) {
    return cols_inv_impl(S2, y, Bn, t_len, t_out, G, W, nfft, scale, env_log2, nullptr, stream);
}
int FL_SPEC_FN(fl_spec_cols_inv_sumsq)(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                               double env_log2, void* sumsq_parts, void* stream) {
    FL_REQUIRE(sumsq_parts, "spec_cols_inv_sumsq: null pointer");
    return cols_inv_impl(S2, y, Bn, t_len, t_out, G, W, nfft, scale, env_log2, (double*)sumsq_parts, stream);
}
/* 1: the inverse column pass may write y over its own input (y = S2's storage read as real (Bn, nfft, G)): a workgroup's tile of
 * samples occupies the bytes of the tile of column values it has read in full before its first store -- when one tile carries
 * all G channels (G <= the tile width) and every sample is stored (t_len = t_out = nfft) */
int FL_SPEC_FN(fl_spec_cols_inv_inplace_ok)(int nfft, int G) {
    int l1 = 0, l2 = 0;
    if (G < 2 || (G & 1) || spec_plan(nfft, l1, l2) != FL_OK) return 0;
    return G <= cols_vt(G, l1) ? 1 : 0;
}
int FL_SPEC_FN(fl_spec_cols_inv_grad_supported)(int nfft, int G) { return cols_inv_grad_ok(nfft, G) ? 1 : 0; }
int FL_SPEC_FN(fl_spec_cols_inv_sumsq_grad)(const void* S2, void* y, void* Sg, int Bn, int G, const void* W, int nfft, double scale,
                                    void* sumsq_parts, void* stream) {
    FL_REQUIRE(sumsq_parts && Sg, "spec_cols_inv_sumsq_grad: null pointer");
    FL_REQUIRE(cols_inv_grad_ok(nfft, G), "spec_cols_inv_sumsq_grad: shape not taken (fl_spec_cols_inv_grad_supported)");
    return cols_inv_impl(S2, y, Bn, nfft, nfft, G, W, nfft, scale, 0.0, (double*)sumsq_parts, stream, nullptr, Sg);
}
int FL_SPEC_FN(fl_spec_cols_inv_scaled)(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                                const void* dev_scale, double env_log2, void* stream) {
    FL_REQUIRE(dev_scale, "spec_cols_inv_scaled: null pointer");
    return cols_inv_impl(S2, y, Bn, t_len, t_out, G, W, nfft, scale, env_log2, nullptr, stream, (const real_t*)dev_scale);
}
int FL_SPEC_FN(fl_spec_cols_blocks)(int nfft, int Bn, int G) {      // workgroups of a column pass = entries of sumsq_parts
    ColsArgs a = {};
    static const real_t dummy = 0;
    if (Bn <= 0) return 0;
    if (cols_setup(a, nfft, Bn, 0, 0, G, &dummy, cols_vt_of(nfft, G))) return -1;
    return Bn * a.nct * a.ngt;
}
}  // extern "C"
static int cols_inv_impl(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                         double env_log2, double* sumsq, void* stream, const real_t* dev_scale, void* Sg) {
    FL_REQUIRE(S2 && y, "spec_cols_inv: null pointer");
    FL_REQUIRE(reinterpret_cast<uintptr_t>(y) % (2 * RSZ) == 0, "spec_cols_inv: y must be aligned to two samples");
    FL_REQUIRE(t_out >= 0 && t_out <= t_len, "spec_cols_inv: t_out must be in [0, t_len]");
    if (Bn == 0) return FL_OK;
    ColsArgs a = {};
    int rc = cols_setup(a, nfft, Bn, t_len, t_out, G, W, cols_vt_of(nfft, G));
    if (rc) return rc;
    a.y = (real_t*)y;
    a.S = (cf*)S2;
    a.scale = (real_t)scale;
    a.env_log2 = env_log2;
    a.sumsq = sumsq;
    a.dev_scale = dev_scale;
    a.Sg = (cf*)Sg;
    a.pol = ((stream_policy() >> 2) & 3u) | ((stream_policy() & POL_INV_SG_NT) ? 4u : 0u);      // POL_INV_LD_NT, POL_INV_ST_NT, POL_INV_SG_NT
    a.stamp = walk_successor_stamp();
    return cols_launch(true, a, Bn, (hipStream_t)stream);
}
extern "C" {

int FL_SPEC_FN(fl_spec_mid)(const void* S, void* S2, void* Xs, long xs_b, long xs_n, const void* H, long hs_m, long hs_n, int conj_h,
                    const void* W, int nfft, int Bn, int NI, int NO, double spec_scale, int spec_interior2, int pre_half,
                    void* stream) {
    FL_REQUIRE(S && W, "spec_mid: null pointer");
    FL_REQUIRE(S2 || Xs, "spec_mid: nothing to produce");
    FL_REQUIRE(H == nullptr || S2 != nullptr, "spec_mid: a product without the inverse half is not a mode");
    FL_REQUIRE(H != nullptr || NI == NO, "spec_mid: channel counts differ without a response");
    if (Bn == 0) return FL_OK;
    MidArgs a = {};
    int rc = spec_plan(nfft, a.L1, a.L2);
    if (rc) return rc;
    a.S = (const cf*)S; a.S2 = (cf*)S2; a.Xs = (cf*)Xs; a.xs_b = xs_b; a.xs_n = xs_n;
    a.H = (const cf*)H; a.hs_m = hs_m; a.hs_n = hs_n; a.conj_h = conj_h;
    a.W = (const cf*)W; a.n = nfft; a.L = nfft / 2; a.Bn = Bn;
    a.spec_scale = (real_t)spec_scale; a.spec_interior2 = spec_interior2; a.pre_half = pre_half; a.dbg_hfake = g_mid_hfake; a.dbg_times = g_mid_times;
    const int P = a.L1 / 2 + 1;
    const size_t nblk = (size_t)cdiv_i(P, 8) * 8 * Bn;
    FL_REQUIRE(nblk < (1ull << 31), "spec_mid: grid too large");
    hipStream_t st = (hipStream_t)stream;
    switch (a.L2) {
#ifdef FL_F64       // float64: one workgroup per (row pair, batch item), equal channel counts, at every plan length
        case 240: rc = launch_mid_lean<16, 15>(a, NI, NO, st); break;
        case 320: rc = launch_mid_lean<16, 20>(a, NI, NO, st); break;
        case 480: rc = launch_mid_lean<32, 15>(a, NI, NO, st); break;
#else
        case 240: rc = launch_mid<16, 15>(a, NI, NO, (unsigned)nblk, st); break;
        case 320: rc = launch_mid<16, 20>(a, NI, NO, (unsigned)nblk, st); break;
        case 480: rc = launch_mid<32, 15>(a, NI, NO, (unsigned)nblk, st); break;
#endif
        case 32: rc = launch_mid_lean<8, 4>(a, NI, NO, st); break;
        case 64: rc = launch_mid_lean<8, 8>(a, NI, NO, st); break;
        case 128: rc = launch_mid_lean<16, 8>(a, NI, NO, st); break;
        case 160: rc = launch_mid_lean<16, 10>(a, NI, NO, st); break;
        case 256: rc = launch_mid_lean<16, 16>(a, NI, NO, st); break;
        case 50: rc = launch_mid_lean<2, 25>(a, NI, NO, st); break;
        case 100: rc = launch_mid_lean<4, 25>(a, NI, NO, st); break;
        case 200: rc = launch_mid_lean<8, 25>(a, NI, NO, st); break;
        case 400: rc = launch_mid_lean<16, 25>(a, NI, NO, st); break;
        default: set_error("spec_mid: unsupported row length %d", a.L2); return FL_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    FL_CHECK_LAUNCH("spec_mid");
    return FL_OK;
}

int FL_SPEC_CFN(fl_permute_bins)(const void* src, long src_pitch, void* dst, long dst_pitch, int nplanes, int nfft, int inverse,
                        void* stream) {
    FL_REQUIRE(src && dst && nplanes >= 0 && nplanes <= 65535, "permute_bins: bad arguments");
    int L1, L2;
    int rc = spec_plan(nfft, L1, L2);
    if (rc) return rc;
    FL_REQUIRE(src_pitch > L1 * L2 && dst_pitch > L1 * L2, "permute_bins: pitch must be >= nfft/2+1");
    if (nplanes == 0) return FL_OK;
    hipLaunchKernelGGL(permute_bins_kernel, dim3(cdiv_i(L1 * L2 + 1, 256), nplanes), dim3(256), 0, (hipStream_t)stream,
                       (const cf*)src, src_pitch, (cf*)dst, dst_pitch, L1, L2, inverse);
    FL_CHECK_LAUNCH("permute_bins");
    return FL_OK;
}

}  // extern "C"
