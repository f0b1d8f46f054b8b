// Forward column pass of the fused Shell pipeline (K1 of spectral.hip) as a device function; gfx950 only.
#pragma once
#include "spectral_common.h"

namespace fl {
namespace FL_SPEC_NS {

// ---------------------------------------------------------------- K1: forward column pass
// Workgroup = (batch item, tile of CT columns, tile of CG channels): VT = CT*CG "virtual columns" v = cl*CG + gl,
// lanes run over v, so every global access of a wavefront is VT*8 contiguous bytes per row.
// The (re, im) pair of z[j] sits G floats apart in x: the even lane of a channel pair loads x[2j][g..g+1], the
// odd one x[2j+1][g..g+1] (8-byte loads, the VT lanes cover one contiguous run of both sample rows), and one
// DPP exchange turns that into (re, im) per channel.
// (the body is a device function of the workgroup's number `blk` and its LDS: the plain kernel of spectral.hip and the
// launch that carries the cascade response's workgroups beside it, fusedfwd.hip, both run it)
template <int A, int B, int VT, int RG, bool PLAIN>
__device__ __forceinline__ void spec_cols_fwd_body(const ColsArgs& a, int blk, char* smem) {
    constexpr int LEN = A * B, LENP = LEN | 1;
    cf* U = reinterpret_cast<cf*>(smem);   // [VT][LENP]
    cf* tw = U + VT * LENP;                 // W_LEN^m
    cf* t2 = tw + LEN;                      // [CT][B]: W_L^(c * A * kb)
    const int ct = blk % a.nct; blk /= a.nct;
    const int gt = blk % a.ngt;
    const int b = blk / a.ngt;
    const int CG = 1 << a.cgs;
    const int c0 = ct * a.CT, g0 = gt * CG;
    constexpr int NIT = B * VT, NR = (NIT + 255) / 256;
    const real_t* xb = a.x + (size_t)b * a.t_len * a.G + g0;
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
                const int par = gl & 1;
#pragma unroll
                for (int ta = 0; ta < A; ++ta) {
                    const int j = c0 + cl + a.L2 * (ta * B + tb);
                    const int t = 2 * j + par;
                    v2f q = {0, 0};
                    if (PLAIN || t < a.t_lim) q = ldv<NT>(xb, RSZ * ((unsigned)t * (unsigned)a.G + (unsigned)(gl - par)));
                    v[rr][ta] = cf(q.x, q.y);
                }
            }
        }
    };
    auto load_group = [&](int r0) {
        if (ld_nt_pol) load_group_p(r0, std::true_type{});
        else load_group_p(r0, std::false_type{});
    };
    // the first group's samples are requested BEFORE the twiddle tables are fetched: the tables' latency (a dependent
    // global round trip in front of the barrier) then overlaps the data's instead of preceding it
    load_group(0);
    const cf* aux = a.W + a.n;                     // contiguous copies: W_L1^j (L1), W_L2^j (L2), W_n^(L1 j) (L2)
    for (int j = threadIdx.x; j < LEN; j += 256) tw[j] = aux[j];
    for (int j = threadIdx.x; j < a.CT * B; j += 256) {
        const int cl = j / B, kb = j - cl * B;
        t2[j] = a.W[2 * (c0 + cl) * A * kb];
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
                const int cl = vv >> a.cgs, gl = vv & (CG - 1);
                const int par = gl & 1;
#pragma unroll
                for (int ta = 0; ta < A; ++ta) {
                    cf q = v[rr][ta];
                    if (!PLAIN && a.env_log2 != 0.0) {
                        const int t = 2 * (c0 + cl + a.L2 * (ta * B + tb)) + par;
                        const real_t e = env_at(a.env_log2, t);
                        q.x *= e;
                        q.y *= e;
                    }
                    // even lane holds (re_g, re_g+1), odd lane (im_g-1, im_g)
                    const real_t got = swap1(par ? q.x : q.y);
                    v[rr][ta] = par ? cf(got, q.y) : cf(q.x, got);
                }
                RegFFT<real_t, A, false>::run(v[rr]);
                cf* u = U + vv * LENP + tb;
                u[0] = v[rr][0];
#pragma unroll
                for (int ka = 1; ka < A; ++ka) u[ka * B] = v[rr][ka] * tw[ka * tb];
            }
        }
    }
    __syncthreads();
    cf* out = a.S + (size_t)b * a.L1 * a.L2 * a.G + g0;
    for (int item = threadIdx.x; item < A * VT; item += 256) {
        const int ka = item / VT, vv = item % VT;
        const int cl = vv >> a.cgs, gl = vv & (CG - 1);
        const int c = c0 + cl;
        cf v[B];
        const cf* u = U + vv * LENP + ka * B;
#pragma unroll
        for (int tb = 0; tb < B; ++tb) v[tb] = u[tb];
        RegFFT<real_t, B, false>::run(v);
        const cf w1 = a.W[2 * c * ka];
        const cf* w2 = t2 + cl * B;
        auto store_all = [&](auto nt_tag) {
            constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
            for (int kb = 0; kb < B; ++kb) {
                const int k1 = ka + A * kb;
                const cf r = v[kb] * (w1 * w2[kb]);
                stv<NT>(out, ESZ * (((unsigned)k1 * (unsigned)a.L2 + (unsigned)c) * (unsigned)a.G + (unsigned)gl), v2f{r.x, r.y});
            }
        };
        if (st_nt_pol) store_all(std::true_type{});
        else store_all(std::false_type{});
    }
}

}  // namespace FL_SPEC_NS
}  // namespace fl
