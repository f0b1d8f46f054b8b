// Matrix-then-cascade response (forward, second generation) as a device function; gfx950 only.  See cascade2.hip.
#pragma once
#include "common.h"
#include "response_common.h"

namespace fl {
typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- forward: cascade response times a constant matrix
// H[m][n] = sum_j G[m][j] W[j][n], G[m][j] = prod_s B_s / prod_s A_s (response.hip: sos_response_rc_fast_kernel is the first
// generation: two sections per packed instruction, two running products per polynomial, merged at the end).  Here numerator and
// denominator of ONE section share the packed halves: the running product (prod B, prod A) is one packed complex value, a
// section costs six packed instructions per bin (value 2, product 4), there is no merge of chains, and -- graphic equaliser --
// the pure-gain band 0 (eq.py:91-94) is folded into band 1's numerator table: 11 sections per cascade, not 12.  One table
// entry = two 16-byte reads: (c0B, c0A, c1B, c1A | c2B, c2A, -, -), Re = c0 + c1 x, Im = c2 sin (half_turn_tables).
struct RcBaArgs {
    const double* b;       // (3, S, C) sections -- read, or (graphic equaliser: gd.gain) designed here and written by the first bin block
    const double* a;
    int S, C, Nmid;
    const float* Wr;       // (Nmid, NIW) constant factor
    double g;              // anti-aliasing radius
    const cx<double>* Wd;  // float64 master twiddles
    int nfft, bin0, m_local;
    cx<float>* G;
    long g_pitch;
    cx<float>* H;
    long h_pitch;
    GeqDesign gd;
    unsigned pol;          // common.h: POL_RC_ST_NT
    long long* dbg;        // tuning, or null: per workgroup (after the design, after the tables, after the cascades, -) by s_memrealtime
};

// (a device function of the workgroup's coordinates (bx: block of 256 bin pairs, m: output row) and its LDS: the plain kernel of
// cascade2.hip and the launch that carries these workgroups beside the input's column pass, fusedfwd.hip, both run it)
template <int NIW>
__device__ __forceinline__ void rc_ba_body(const RcBaArgs& A, int bx, int m, char* smem) {
    const double* __restrict__ b = A.b;
    const double* __restrict__ a = A.a;
    const int S = A.S, C = A.C, Nmid = A.Nmid, nfft = A.nfft, bin0 = A.bin0, m_local = A.m_local;
    const float* __restrict__ Wr = A.Wr;
    const double g = A.g;
    const cx<double>* __restrict__ Wd = A.Wd;
    cx<float>* __restrict__ G = A.G;
    cx<float>* __restrict__ H = A.H;
    const long g_pitch = A.g_pitch, h_pitch = A.h_pitch;
    const GeqDesign gd = A.gd;
    const unsigned pol = A.pol;
    const int s_first = gd.gain ? 1 : 0, Seff = S - s_first;
    f4* tab = reinterpret_cast<f4*>(smem);                                   // [Nmid][basis 2][Seff][2]
    float* lw = reinterpret_cast<float*>(tab + (size_t)Nmid * 2 * Seff * 2);   // [Nmid][NIW]
    double* dt = reinterpret_cast<double*>(lw + ((Nmid * NIW + 3) & ~3));    // [Nmid][S][6] taps (b0 b1 b2 a0 a1 a2)
    // Every global operand of the workgroup is requested HERE, together: inside the launch pair (fusedfwd.hip) the memory
    // system is saturated by the column pass and a round trip costs 2-3 us -- the command gain, then the band constants, then
    // (behind two barriers) the twiddles were three of them in a row, a third of a workgroup's life.
    const int i0 = threadIdx.x;                      // (section i0 of the output row; beyond 256 of them: the loop below)
    const bool has_sec = i0 < Nmid * S;
    const int j0 = has_sec ? i0 / S : 0, sidx0 = has_sec ? i0 - j0 * S : 0, c0 = m * Nmid + j0;
    double raw0 = 0, ka0 = 0, kb0 = 0, tb0[3], ta0[3];
    if (gd.gain) {
        int ia, ib;
        geq_band_const_idx(sidx0, S, &ia, &ib);
        raw0 = geq_raw_gain(gd.gain, gd.in_kind, sidx0 * C + c0);
        ka0 = gd.k[ia];
        kb0 = gd.k[ib];
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            tb0[q] = b[(size_t)(q * S + sidx0) * C + c0];
            ta0[q] = a[(size_t)(q * S + sidx0) * C + c0];
        }
    }
    const float lw0 = Wr[threadIdx.x < Nmid * NIW ? threadIdx.x : 0];
    // a thread takes TWO ADJACENT BINS through the cascades (one set of table reads serves both): natural order elements
    // 2p, 2p + 1; row-major order (row 2r, column c) and the element one row below (bin + 1); the Nyquist element alone
    const int p = bx * 256 + threadIdx.x;
    int e[2];
    bool two, live = true;
    if (bin0 >= 0) {
        e[0] = 2 * p;
        live = e[0] < m_local;
        if (!live) e[0] = 0;
        two = live && e[0] + 1 < m_local;
        e[1] = two ? e[0] + 1 : e[0];
    } else {
        const int L2 = -bin0, L = nfft >> 1, L1 = L / L2, main = ((L1 + 1) >> 1) * L2;
        live = p <= main;
        if (p >= main) {
            e[0] = e[1] = L;
            two = false;
        } else {
            const int r = p / L2, c2 = p - r * L2;
            e[0] = 2 * r * L2 + c2;
            two = 2 * r + 1 < L1;
            e[1] = two ? e[0] + L2 : e[0];
        }
    }
    int kbin[2];
    cx<double> w1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        kbin[q] = bin_of(e[q], bin0, nfft);
        w1[q] = Wd[kbin[q] < nfft ? kbin[q] : kbin[q] - nfft];
    }
    if (A.dbg && threadIdx.x == 0) {      // (tuning: when this wavefront's operands have arrived)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        A.dbg[8 * ((size_t)m * 4096 + bx) + 3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (has_sec) {
        if (gd.gain) {
            geq_section_vals(geq_gain_of_raw(raw0, gd.in_kind), sidx0, S, ka0, kb0, tb0, ta0);
            if (bx == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    gd.b_out[(size_t)(q * S + sidx0) * C + c0] = tb0[q];
                    gd.a_out[(size_t)(q * S + sidx0) * C + c0] = ta0[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dt[(size_t)i0 * 6 + q] = tb0[q];
            dt[(size_t)i0 * 6 + 3 + q] = ta0[q];
        }
    }
    for (int i = 256 + threadIdx.x; i < Nmid * S; i += 256) {      // (more than 256 sections per output row: the rest, plainly)
        const int j = i / S, sidx = i - j * S;
        const int c = m * Nmid + j;
        double tb[3], ta[3];
        if (gd.gain) {
            geq_section_of(gd.gain, gd.in_kind, sidx * C + c, sidx, S, gd.k, tb, ta);
            if (bx == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    gd.b_out[(size_t)(q * S + sidx) * C + c] = tb[q];
                    gd.a_out[(size_t)(q * S + sidx) * C + c] = ta[q];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                tb[q] = b[(size_t)(q * S + sidx) * C + c];
                ta[q] = a[(size_t)(q * S + sidx) * C + c];
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dt[(size_t)i * 6 + q] = tb[q];
            dt[(size_t)i * 6 + 3 + q] = ta[q];
        }
    }
    for (int i = 256 + threadIdx.x; i < Nmid * NIW; i += 256) lw[i] = Wr[i];
    if (threadIdx.x < Nmid * NIW) lw[threadIdx.x] = lw0;
    __syncthreads();
    long long* dbg = A.dbg ? A.dbg + 8 * ((size_t)m * 4096 + bx) : nullptr;
    if (dbg && threadIdx.x == 0) dbg[0] = (long long)__builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < Nmid * Seff; i += 256) {
        const int j = i / Seff, se = i - j * Seff;
        const double* t = dt + (size_t)(j * S + s_first + se) * 6;
        // (graphic equaliser: band 0 is b = (g0, 0, 0), a = (1, 0, 0) -- its factor g0 multiplies band 1's numerator)
        const double sc = (s_first && se == 0) ? dt[(size_t)(j * S) * 6] : 1.0;
        const double g2 = g * g;
        const double SB = sc * (t[0] + g2 * t[2]), TB = sc * g * t[1], DB = sc * (t[0] - g2 * t[2]);
        const double SA = t[3] + g2 * t[5], TA = g * t[4], DA = t[3] - g2 * t[5];
        f4* lo = tab + ((size_t)(j * 2 + 0) * Seff + se) * 2;
        f4* hi = tab + ((size_t)(j * 2 + 1) * Seff + se) * 2;
        lo[0] = f4{(float)(SB + TB), (float)(SA + TA), (float)(-SB), (float)(-SA)};
        lo[1] = f4{(float)DB, (float)DA, 0.f, 0.f};
        hi[0] = f4{(float)(TB - SB), (float)(TA - SA), (float)SB, (float)SA};
        hi[1] = f4{(float)DB, (float)DA, 0.f, 0.f};
    }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[1] = (long long)__builtin_amdgcn_s_memrealtime();
    if (!live) return;
    bool low[2];
    float xr[2], xi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        low[q] = 4 * (long)kbin[q] < nfft;
        xr[q] = (float)(low[q] ? 1.0 - w1[q].x : 1.0 + w1[q].x);      // 1 -+ cos(omega), formed in double
        xi[q] = (float)(-w1[q].y);                                     // sin(omega)
    }
    const bool same = low[0] == low[1];
    f2 acc[2][NIW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < NIW; ++n) acc[q][n] = f2{0.f, 0.f};
    for (int j = 0; j < Nmid; ++j) {
        f2 Pr[2], Pi[2];      // (prod B, prod A): real and imaginary parts
        // (explicit fused operations: both walks below -- the shared-table one and the per-bin one of a thread whose two bins
        // straddle nfft / 4 -- must round alike whatever the compiler would contract)
        auto first = [&](int q, f4 e0, f4 e1) {
            Pr[q] = __builtin_elementwise_fma(f2{e0.z, e0.w}, f2{xr[q], xr[q]}, f2{e0.x, e0.y});
            Pi[q] = f2{e1.x, e1.y} * xi[q];
        };
        auto step = [&](int q, f4 e0, f4 e1) {
            const f2 R = __builtin_elementwise_fma(f2{e0.z, e0.w}, f2{xr[q], xr[q]}, f2{e0.x, e0.y});
            const f2 I = f2{e1.x, e1.y} * xi[q];
            f2 nr = Pi[q] * I;
            nr = __builtin_elementwise_fma(Pr[q], R, -nr);
            f2 ni = Pr[q] * I;
            ni = __builtin_elementwise_fma(Pi[q], R, ni);
            Pr[q] = nr;
            Pi[q] = ni;
        };
        if (same) {
            // (a section's table entry is requested one section ahead of its use: beside the column pass this role has one
            // wavefront per SIMD and nothing else to cover an LDS round trip with)
            const f4* tb = tab + (size_t)(j * 2 + (low[0] ? 0 : 1)) * Seff * 2;
            f4 n0 = tb[2];
            f2 n1 = *reinterpret_cast<const f2*>(tb + 3);
            first(0, tb[0], tb[1]);
            first(1, tb[0], tb[1]);
#pragma unroll 5
            for (int se = 1; se < Seff; ++se) {
                const f4 e0 = n0;
                const f4 e1 = f4{n1.x, n1.y, 0.f, 0.f};
                const int sn = se + 1 < Seff ? se + 1 : se;
                n0 = tb[2 * sn];
                n1 = *reinterpret_cast<const f2*>(tb + 2 * sn + 1);
                step(0, e0, e1);
                step(1, e0, e1);
            }
        } else {
            for (int q = 0; q < 2; ++q) {
                const f4* tb = tab + (size_t)(j * 2 + (low[q] ? 0 : 1)) * Seff * 2;
                if (q == 0) first(0, tb[0], tb[1]);
                else first(1, tb[0], tb[1]);
                for (int se = 1; se < Seff; ++se) {
                    if (q == 0) step(0, tb[2 * se], tb[2 * se + 1]);
                    else step(1, tb[2 * se], tb[2 * se + 1]);
                }
            }
        }
        const f4 w0 = *reinterpret_cast<const f4*>(lw + j * NIW);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float Bx = Pr[q].x, By = Pi[q].x, Ax = Pr[q].y, Ay = Pi[q].y;
            f2 hf;
            if (Ax != 0.f || Ay != 0.f) {
                const float inv = __builtin_amdgcn_rcpf(Ax * Ax + Ay * Ay);      // (1 ulp; an IEEE division is ten instructions per bin and cascade)
                hf = f2{(Bx * Ax + By * Ay) * inv, (By * Ax - Bx * Ay) * inv};
            } else {
                hf = f2{eps_of<float>(), 0.f};
            }
            if (q == 0 || two) {
                // G is read again by the backward pass only, a pipeline later: optionally past the caches (POL_RC_ST_NT)
                f2* gp = reinterpret_cast<f2*>(G + (size_t)(m * Nmid + j) * g_pitch + e[q]);
                if (pol & POL_RC_ST_NT) __builtin_nontemporal_store(hf, gp);
                else *gp = hf;
            }
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const float w = NIW >= 4 && n < 4 ? w0[n & 3] : lw[j * NIW + n];
                acc[q][n] = hf * w + acc[q][n];
            }
        }
    }
    if (dbg && threadIdx.x == 0) dbg[2] = (long long)__builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (q == 0 || two) {
#pragma unroll
            for (int n = 0; n < NIW; ++n) H[(size_t)(m * NIW + n) * h_pitch + e[q]] = cx<float>(acc[q][n].x, acc[q][n].y);
        }
}

}  // namespace fl
