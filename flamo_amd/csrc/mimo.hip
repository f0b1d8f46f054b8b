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

// HC: H does not depend on the bin (Gain / Matrix, hs_f = 0): its address is then uniform over the wavefront and the
// compiler fetches it with scalar loads -- MT*Ni fewer vector loads per thread in the small composition launches
template <typename T, int MT, int BT, int NU, bool HC = false>
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
    const cx<T>* Hf = HC ? H : H + (long)f * hs_f;
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
                if (HC && (conj_h & 2)) {     // real constant matrix (strides in real elements): no complex copy of it exists
                    h[u][mm] = (nv && m < No) ? cx<T>(reinterpret_cast<const T*>(H)[(long)m * hs_m + (long)n * hs_n], 0) : cx<T>(0, 0);
                } else {
                    h[u][mm] = (nv && m < No) ? Hf[(long)m * hs_m + (long)n * hs_n] : cx<T>(0, 0);
                    if (conj_h & 1) h[u][mm].y = -h[u][mm].y;
                }
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
    cx<T>* __restrict__ dH, long dh_pitch, T scale, int B, int M, int No, int Ni, int K, const T* __restrict__ dev_scale) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= M) return;
    if (dev_scale) scale *= *dev_scale;      // a factor that lives on the device (an objective's 2 g / N): no pass over dH for it
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

// ---------------------------------------------------------------- matrix-valued signals: MFMA
// When the signal carries >= 8 columns per bin and the channel count is >= 16 -- Recursion pushing the identity
// through a 32 x 32 loop (system.py:417-419), Shell.get_freq_response(identity=True), the matrix gradients of
// those -- the per-bin product is a genuine dense contraction: 8 N^3 flop against 24 N^2 bytes per bin,
// 10.7 flop/B at N = 32, and the lane-per-bin FMA stream above is bound by vector issue (measured 1.88 ms
// for 192001 bins of 32x32x32: 27 TFLOP/s, 1.7 TB/s) rather than by HBM (0.6 ms).  The matrix cores take the
// bin-planar layout as it is through the BLOCKED fp32 instruction v_mfma_f32_4x4x1_16b_f32: one instruction is
// 16 independent 4x4 outer-product updates, one block per bin --
//     lane = 4*bin + q :   srcA = A[4rb+q][t]   srcB = B[t][4cb+q]   acc[v] = D[4rb+v][4cb+q]
// so 16 adjacent bins of one plane are one 128-byte segment per q and nothing is transposed or staged.
// A wavefront owns 16 bins x (4 RB rows) x 8 columns: RB*2 accumulator tiles for the real and for the
// imaginary part (128 VGPRs at RB = 8); the four real products of a complex one are four MFMAs
// (Dr += Ar Br, Dr += (-Ai) Bi, Di += Ar Bi, Di += Ai Br).  The four wavefronts of a workgroup take
// adjacent column tiles of the same bins, so A is fetched from HBM once and from the L1/L2 after.
//   D[i, j, f] = scale * sum_t opA(A)[i, t, f] opB(B)[t, j, f],    t = (t1, t2),  j = (j1, j2)
// covers the forward product (t = input channel, j = (batch, column)) and the per-bin matrix gradient
// (t = (batch, column), j = input channel, B conjugated).
typedef float v4f __attribute__((ext_vector_type(4)));

struct MmaArgs {
    const cx<float>* A;
    long sa_f, sa_i, sa_t1, sa_t2;
    const cx<float>* B;
    long sb_j1, sb_j2, sb_t1, sb_t2;
    cx<float>* D;
    long sd_i, sd_j1, sd_j2;
    cx<float>* part;     // non-null: sum over bins instead of storing D -- one (NI x J) partial per bin-tile slot
    float scale;
    int conj_a, conj_b;
    int M, NI, J1, J2, T1, T2, nct, nrt;
    int vec_store;       // 16-byte stores through LDS (D and its strides 16-byte aligned)
};

// Operands are staged through LDS.  (A first version fed the MFMAs straight from global memory: a wavefront's 16
// bins are one 128-byte piece per plane, the planes of a 32 x 32 matrix are 1.5 MB apart, and HBM delivered 2.2 TB/s
// whatever the prefetch depth.)  A workgroup owns 64 adjacent bins and a 16 x 16 tile
// of the output: every wavefront loads whole 512-byte rows of a plane (as the lane-per-bin kernels do), the
// 16 + 16 operand planes of a contraction step sit in LDS ([plane][64 bins], double-buffered, one barrier per
// step), and wavefront w feeds its MFMAs for bins 16w..16w+15 from there: lane (bin, q) reads plane 4rb+q.
// Plane rows are padded to 80 elements so the four q of a read land in different bank halves.
// WB bin groups of 16 x WC column parts = the 4 wavefronts; a wavefront accumulates (4 RBW) rows x (4 CBW) columns.
//   <4,1,4,4>: 64 bins, 16 x 16 output tile;   <2,2,8,4>: 32 bins, 32 x 32 tile (each operand element is loaded
//   once per 32 output columns/rows instead of once per 16: the kernel is bound by bytes per CU, ~10 B/cycle)
template <int WB, int WC, int RBW, int CBW>
__global__ void __launch_bounds__(256) mimo_mfma_lds_kernel(MmaArgs a) {
    constexpr int BINS = 16 * WB, PST = BINS + 16, RT = 4 * RBW, CT = 4 * CBW * WC, PPP = 256 / BINS;   // planes per pass
    constexpr int NPA = RT / PPP, NPB = CT / PPP;
    static_assert(WB * WC == 4 && RT % PPP == 0 && CT % PPP == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) cx<float> smem[2 * (RT + CT) * PST];
    cx<float>(*sA)[RT][PST] = reinterpret_cast<cx<float>(*)[RT][PST]>(smem);
    cx<float>(*sB)[CT][PST] = reinterpret_cast<cx<float>(*)[CT][PST]>(smem + 2 * RT * PST);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane & 3, bl = lane >> 2;
    // XCD-aware order: the nrt*nct tiles of one bin tile run back to back on one XCD and share its L2.
    // Reduction mode: a workgroup walks bin tiles bt0, bt0 + slots, ... and keeps accumulating.
    const int inner = a.nrt * a.nct;
    const bool red = a.part != nullptr;
    int bt, rem, slots;
    if (red) {
        bt = blockIdx.x / inner;
        rem = blockIdx.x % inner;
        slots = gridDim.x / inner;
    } else {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        bt = (r / inner) * 8 + xcd;
        rem = r % inner;
        slots = 1 << 30;
    }
    const int bt0 = bt;
    if ((long)bt * BINS >= a.M) return;               // uniform over the workgroup, before any barrier
    const int rt = rem / a.nct, ct = rem % a.nct;
    const int ncols = a.J1 * a.J2;
    const int lb = tid % BINS, lp = tid / BINS;       // loader role: bin lb of planes lp, lp+PPP, ... of A and of B
    const int g = w % WB, h = w / WB;                 // MFMA role: bin group g, column part h
    const int bin = 16 * g + bl;
    const float sga = a.conj_a ? -1.f : 1.f, sgb = a.conj_b ? -1.f : 1.f;
    const int T = a.T1 * a.T2;
    v4f dr[RBW][CBW], di[RBW][CBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) dr[rb][cb] = di[rb][cb] = (v4f)(0.f);
  for (; (long)bt * BINS < a.M; bt += slots) {
    const int fl = min(bt * BINS + lb, a.M - 1);
    const float binm = (bt * BINS + lb < a.M) ? 1.f : 0.f;      // bins past the end add nothing to a bin sum
    const cx<float>* ga[NPA];
    const cx<float>* gb[NPB];
    float ma[NPA], mb[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int row = rt * RT + lp + PPP * i;
        ma[i] = row < a.NI ? binm : 0.f;
        ga[i] = a.A + (long)fl * a.sa_f + (long)(row < a.NI ? row : 0) * a.sa_i;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int col = ct * CT + lp + PPP * i;
        const bool cv = col < ncols;
        const int j1 = cv ? col / a.J2 : 0, j2 = cv ? col - j1 * a.J2 : 0;
        mb[i] = cv ? 1.f : 0.f;
        gb[i] = a.B + (long)j1 * a.sb_j1 + (long)j2 * a.sb_j2 + fl;
    }
    long aoff = 0, boff = 0;
    int t2 = 0;
    cx<float> ra[NPA], rb_[NPB];
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NPA; ++i) ra[i] = ga[i][aoff];
#pragma unroll
        for (int i = 0; i < NPB; ++i) rb_[i] = gb[i][boff];
        if (++t2 == a.T2) {
            t2 = 0;
            aoff += a.sa_t1 - (long)(a.T2 - 1) * a.sa_t2;
            boff += a.sb_t1 - (long)(a.T2 - 1) * a.sb_t2;
        } else {
            aoff += a.sa_t2;
            boff += a.sb_t2;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) sA[buf][lp + PPP * i][lb] = cx<float>(ra[i].x * ma[i], ra[i].y * (sga * ma[i]));
#pragma unroll
        for (int i = 0; i < NPB; ++i) sB[buf][lp + PPP * i][lb] = cx<float>(rb_[i].x * mb[i], rb_[i].y * (sgb * mb[i]));
    };
    fetch();
    stage(0);
    if (T > 1) fetch();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        cx<float> av[RBW], bv[CBW];
#pragma unroll
        for (int k = 0; k < RBW; ++k) av[k] = sA[buf][4 * k + q][bin];
#pragma unroll
        for (int k = 0; k < CBW; ++k) bv[k] = sB[buf][h * 4 * CBW + 4 * k + q][bin];
        if (t + 1 < T) stage(buf ^ 1);            // step t+1: fetched during step t-1, free to overwrite (read in t-1)
        if (t + 2 < T) fetch();                   // step t+2 in flight across the MFMAs below
        // two sweeps over the accumulator tiles: the second update of a tile is far behind the first
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                dr[rb][cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rb].x, bv[cb].x, dr[rb][cb], 0, 0, 0);
                di[rb][cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rb].x, bv[cb].y, di[rb][cb], 0, 0, 0);
            }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            const float nay = -av[rb].y;
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                dr[rb][cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(nay, bv[cb].y, dr[rb][cb], 0, 0, 0);
                di[rb][cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rb].y, bv[cb].x, di[rb][cb], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (red) continue;
    if (a.vec_store) {
        // The output tile leaves through LDS: the accumulator layout gives a store instruction 128-byte pieces of
        // four planes at 8 bytes per lane, and the 1.6 GB result was half of the kernel's time (store issue, not
        // bandwidth).  Per row block: 4 rows x CT columns = 4 CT planes of BINS bins are written to LDS, then
        // every thread stores 16 bytes (two bins) of a plane row -- whole rows of 256/512 bytes per plane.
        constexpr int OPS = BINS + 2, HB = BINS / 2, PPS = 256 / HB, NPASS = 4 * CT / PPS;
        static_assert(4 * CT * OPS <= 2 * (RT + CT) * PST, "output block fits in the operand buffer");
        cx<float>(*sO)[OPS] = reinterpret_cast<cx<float>(*)[OPS]>(smem);
        const int pl = tid / HB, bp = tid % HB;
        const int f0 = bt * BINS + 2 * bp;
        long dcol[NPASS];
        bool cvv[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int col = ct * CT + (pl + PPS * i) % CT;
            cvv[i] = col < ncols;
            const int j1 = cvv[i] ? col / a.J2 : 0, j2 = cvv[i] ? col - j1 * a.J2 : 0;
            dcol[i] = (long)j1 * a.sd_j1 + (long)j2 * a.sd_j2 + f0;
        }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    sO[v * CT + h * 4 * CBW + 4 * cb + q][bin] = cx<float>(a.scale * dr[rb][cb][v], a.scale * di[rb][cb][v]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int p = pl + PPS * i;
                const int row = rt * RT + 4 * rb + p / CT;
                if (cvv[i] && row < a.NI && f0 < a.M) {
                    const float4 val = *reinterpret_cast<const float4*>(&sO[p][2 * bp]);
                    cx<float>* d = a.D + dcol[i] + (long)row * a.sd_i;
                    if (f0 + 1 < a.M) *reinterpret_cast<float4*>(d) = val;
                    else *d = cx<float>(val.x, val.y);
                }
            }
            __syncthreads();
        }
        return;
    }
    const int f = bt * BINS + bin;
    if (f >= a.M) return;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
        const int col = ct * CT + h * 4 * CBW + 4 * cb + q;
        if (col >= ncols) continue;
        const int j1 = col / a.J2, j2 = col - j1 * a.J2;
        cx<float>* d = a.D + (long)j1 * a.sd_j1 + (long)j2 * a.sd_j2 + f;
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = rt * RT + 4 * rb + v;
                if (i < a.NI) d[(long)i * a.sd_i] = cx<float>(a.scale * dr[rb][cb][v], a.scale * di[rb][cb][v]);
            }
    }
    return;
  }
    // ---- reduction mode: every accumulator holds 16 per-block (bin mod 16) sums, one per lane group of 4; fold
    // them (fixed butterfly: deterministic), then the WB bin groups through LDS, and write this slot's partial
    cx<float>(*sred)[RT][CT + 1] = reinterpret_cast<cx<float>(*)[RT][CT + 1]>(smem);
    static_assert(WB * RT * (CT + 1) <= 2 * (RT + CT) * PST, "partials fit in the operand buffer");
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float x = dr[rb][cb][v], y = di[rb][cb][v];
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) {
                    x += __shfl_xor(x, off, 64);
                    y += __shfl_xor(y, off, 64);
                }
                if (bl == 0) sred[g][4 * rb + v][h * 4 * CBW + 4 * cb + q] = cx<float>(x, y);
            }
    __syncthreads();
    for (int e = tid; e < RT * CT; e += 256) {
        const int i = e / CT, j = e % CT;
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int gg = 0; gg < WB; ++gg) {
            x += sred[gg][i][j].x;
            y += sred[gg][i][j].y;
        }
        const int row = rt * RT + i, col = ct * CT + j;
        if (row < a.NI && col < ncols) a.part[((size_t)bt0 * a.NI + row) * ncols + col] = cx<float>(a.scale * x, a.scale * y);
    }
}

static int g_mfma_vec = 1;   // tuning: 0 = direct 8-byte stores from the accumulator layout
static int g_mfma_enabled = 1, g_mfma_tile16 = 0;   // tuning: off / always the 64-bin 16x16 tile

// rows >= 16, >= 8 output columns and a contraction >= 8: below that the product is HBM-bound and the
// 512-byte-per-plane accesses of the lane-per-bin kernels serve it better
static bool mfma_applies(int rows, int cols, int depth) { return g_mfma_enabled && rows >= 16 && cols >= 8 && depth >= 8; }

static int launch_mfma(MmaArgs a, hipStream_t st, int red_slots = 0, int* slots_used = nullptr) {
    const bool wide = !g_mfma_tile16 && a.NI > 16 && a.J1 * a.J2 > 16;
    const int rtile = wide ? 32 : 16, ctile = wide ? 32 : 16, bins = wide ? 32 : 64;
    a.nrt = cdiv_i(a.NI, rtile);
    a.nct = cdiv_i(a.J1 * a.J2, ctile);
    long nb = (long)cdiv_i(cdiv_i(a.M, bins), 8) * 8 * a.nrt * a.nct;
    if (a.part) nb = (long)min(red_slots, cdiv_i(a.M, bins)) * a.nrt * a.nct;   // red_slots bin-tile slots
    a.vec_store = !a.part && g_mfma_vec && ((uintptr_t)a.D % 16 == 0) && a.sd_i % 2 == 0 && a.sd_j1 % 2 == 0 && a.sd_j2 % 2 == 0;
    FL_REQUIRE(nb < (1ll << 31), "mimo: grid too large");
    if (wide) hipLaunchKernelGGL((mimo_mfma_lds_kernel<2, 2, 8, 4>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mimo_mfma_lds_kernel<4, 1, 4, 4>), dim3((unsigned)nb), dim3(256), 0, st, a);
    FL_CHECK_LAUNCH("mimo_mfma_lds");
    if (slots_used) *slots_used = (int)(nb / (a.nrt * a.nct));
    return FL_OK;
}

// ---------------------------------------------------------------- streaming form of the per-bin product (8 x 8, vector signals)
// Y[b,:,f] = H[:,:,f] X[b,:,f] with the SIGNAL streamed through an LDS ring by LDS-DMA (global_load_lds, 16 bytes per lane:
// one instruction moves two 512-byte plane rows of a 64-bin tile, no VGPR round trip) and the RESPONSE of the tile held in
// registers for the whole batch: one wavefront per workgroup owns 64 bins, fetches its 64 response values per lane once
// (H crosses the fabric exactly once, whatever the L2 does), then walks the batch in chunks of CB columns -- the chunk
// t+1 is in flight while chunk t is multiplied and stored.  No barriers (single wavefront), ~32 KB of LDS per workgroup so
// that the ~3 workgroups a CU gets are resident together.  Bin tiles past M read clamped addresses and store nothing.
template <int NCH, int CB>
__global__ void __launch_bounds__(64) mimo_stream_kernel(const cx<float>* __restrict__ H, long hs_m, long hs_n, int conj_h,
                                                         const cx<float>* __restrict__ X, long xs_b, long xs_n,
                                                         cx<float>* __restrict__ Y, long ys_b, long ys_m, int B, int M, long x_rows_end) {
    __shared__ __attribute__((aligned(16))) cx<float> xt[2][CB][NCH][64];
    const int lane = threadIdx.x;
    // XCD-aware order is not needed: nothing is shared between workgroups
    const int f0 = blockIdx.x * 64;
    const int f = f0 + lane;
    const bool live = f < M;
    const int fc = live ? f : M - 1;
    cx<float> h[NCH][NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m)
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            h[m][n] = H[(long)m * hs_m + (long)n * hs_n + fc];
            if (conj_h) h[m][n].y = -h[m][n].y;
        }
    // DMA of one chunk: instruction (c, np) moves plane rows n = 2 np and 2 np + 1 of column c: lanes 0..31 the first row,
    // 32..63 the second, 16 bytes (two bins) each; the LDS image [c][n][64 bins] is lane-linear as the instruction requires
    const int half = lane >> 5, piece = lane & 31;
    auto dma = [&](int chunk, int buf) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int col = chunk * CB + c;
            const cx<float>* colp = X + (long)(col < B ? col : B - 1) * xs_b;
#pragma unroll
            for (int np = 0; np < NCH / 2; ++np) {
                long off = (long)(2 * np + half) * xs_n + f0 + 2 * piece;
                if (off + 2 > x_rows_end) off = x_rows_end - 2;          // last tile of the last plane: stay inside the allocation
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(colp + off),
                                                 (__attribute__((address_space(3))) void*)(&xt[buf][c][2 * np][0]), 16, 0, 0);
            }
        }
    };
    // gridDim.y batch splits: workgroup (tile, s) walks columns [s*per, (s+1)*per)
    const int per = ((B + (int)gridDim.y - 1) / (int)gridDim.y + CB - 1) / CB * CB;
    const int c_begin = blockIdx.y * per, c_end = min(B, c_begin + per);
    if (c_begin >= c_end) return;
    const int chunk0 = c_begin / CB, nchunk = (c_end - c_begin + CB - 1) / CB;
    dma(chunk0, 0);
    for (int tt = 0; tt < nchunk; ++tt) {
        const int t = chunk0 + tt;
        const int buf = tt & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): chunk t has landed (and the previous chunk's stores have left)
        if (tt + 1 < nchunk) dma(t + 1, buf ^ 1);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int col = t * CB + c;
            if (col >= c_end) break;
            cx<float> x[NCH];
#pragma unroll
            for (int n = 0; n < NCH; ++n) x[n] = xt[buf][c][n][lane];
            cx<float>* yp = Y + (long)col * ys_b + f;
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                cx<float> acc(0.f, 0.f);
#pragma unroll
                for (int n = 0; n < NCH; ++n) fma_cx(acc, h[m][n], x[n]);
                if (live) yp[(long)m * ys_m] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------- host dispatch
static int g_mimo_variant = 0;   // tuning hook: mt*100 + bt*10 + nu (0 = default choice)

static int g_mimo_stream_split = 1, g_mimo_stream_cb = 4;
static int g_mimo_stream = 0;   // 8x8 vector-signal products through the LDS-DMA streaming kernel (tuning hook: gradw_cap -16)
static int g_mimo_hc = 1;
static int g_gradh_tile = 4;   // 4 = 4x4 tiles (default), 84 = 8x4, 8 = 8x8 where the matrix allows (tuning hook)   // tuning: 0 = constant matrices through the per-bin addressing (gradw_cap -4)

template <typename T, int MT, int BT, int NU>
static void launch_full_one(dim3 grid, int nct, int nmt, hipStream_t st, const cx<T>* H, long hs_f, long hs_m, long hs_n, int conj_h,
                            const cx<T>* X, long xs_b, long xs_n, long xs_k, cx<T>* Y, long ys_b, long ys_m, long ys_k,
                            int B, int M, int No, int Ni, int K) {
    if (NU == 1 && hs_f == 0 && (g_mimo_hc || (conj_h & 2)))
        hipLaunchKernelGGL((mimo_full_kernel<T, MT, BT, 1, true>), grid, dim3(256), 0, st, H, hs_f, hs_m, hs_n, conj_h, X, xs_b, xs_n,
                           xs_k, Y, ys_b, ys_m, ys_k, B, M, No, Ni, K, nct, nmt);
    else
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
    const bool h_real = (conj_h & 2) != 0;
    FL_REQUIRE(!h_real || (hs_f == 0 && g_mimo_variant == 0), "mimo: a real matrix must be frequency independent (hs_f = 0)");
    if constexpr (sizeof(T) == 4) {
        if (g_mimo_variant == 0 && !h_real && mfma_applies(No, ncols, Ni)) {
            MmaArgs a = {};
            a.A = (const cx<float>*)H; a.sa_f = hs_f; a.sa_i = hs_m; a.sa_t1 = 0; a.sa_t2 = hs_n; a.conj_a = conj_h;
            a.B = (const cx<float>*)X; a.sb_j1 = xs_b; a.sb_j2 = xs_k; a.sb_t1 = 0; a.sb_t2 = xs_n; a.conj_b = 0;
            a.D = (cx<float>*)Y; a.sd_i = ys_m; a.sd_j1 = ys_b; a.sd_j2 = ys_k; a.scale = 1.f;
            a.M = M; a.NI = No; a.J1 = B; a.J2 = K; a.T1 = 1; a.T2 = Ni;
            return launch_mfma(a, (hipStream_t)stream);
        }
    }
    if constexpr (sizeof(T) == 4) {
        // 8 x 8 per-bin responses on vector signals with 16-byte-aligned planes: the streaming kernel
        if (g_mimo_stream && g_mimo_variant == 0 && !h_real && hs_f == 1 && K == 1 && No == 8 && Ni == 8 && B >= 4 &&
            reinterpret_cast<uintptr_t>(X) % 16 == 0 && xs_n % 2 == 0 && xs_b % 2 == 0) {
            const long x_rows_end = (long)(Ni - 1) * xs_n + (xs_n < M + 2 ? xs_n : ((M + 1) & ~1L));   // elements addressable in a column
            const int bs = g_mimo_stream_split > 0 ? g_mimo_stream_split : 1;
            if (g_mimo_stream_cb == 2)
                hipLaunchKernelGGL((mimo_stream_kernel<8, 2>), dim3(cdiv_i(M, 64), bs), dim3(64), 0, (hipStream_t)stream,
                                   (const cx<float>*)H, hs_m, hs_n, conj_h, (const cx<float>*)X, xs_b, xs_n, (cx<float>*)Y, ys_b,
                                   ys_m, B, M, x_rows_end);
            else
                hipLaunchKernelGGL((mimo_stream_kernel<8, 4>), dim3(cdiv_i(M, 64), bs), dim3(64), 0, (hipStream_t)stream,
                                   (const cx<float>*)H, hs_m, hs_n, conj_h, (const cx<float>*)X, xs_b, xs_n, (cx<float>*)Y, ys_b,
                                   ys_m, B, M, x_rows_end);
            FL_CHECK_LAUNCH("mimo_stream");
            return FL_OK;
        }
    }
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
                      void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream,
                      const void* dev_scale_ = nullptr) {
    const T* dev_scale = (const T*)dev_scale_;
    FL_REQUIRE(G && X && dH, "mimo_gradh: null pointer");
    FL_REQUIRE(dh_pitch >= M, "mimo_gradh: dh_pitch must be >= M");
    FL_REQUIRE(B >= 0 && M >= 0 && No > 0 && Ni > 0 && K > 0, "mimo_gradh: bad sizes");
    if (M == 0) return FL_OK;
    hipStream_t st = (hipStream_t)stream;
    if constexpr (sizeof(T) == 4) {
        if (g_mimo_variant == 0 && !dev_scale && mfma_applies(No, Ni, B * K)) {      // (a device-side factor: the lane kernels)
            MmaArgs a = {};
            a.A = (const cx<float>*)G; a.sa_f = 1; a.sa_i = gs_m; a.sa_t1 = gs_b; a.sa_t2 = gs_k; a.conj_a = 0;
            a.B = (const cx<float>*)X; a.sb_j1 = 0; a.sb_j2 = xs_n; a.sb_t1 = xs_b; a.sb_t2 = xs_k; a.conj_b = 1;
            a.D = (cx<float>*)dH; a.sd_i = (long)Ni * dh_pitch; a.sd_j1 = 0; a.sd_j2 = dh_pitch; a.scale = (float)scale;
            a.M = M; a.NI = No; a.J1 = 1; a.J2 = Ni; a.T1 = B; a.T2 = K;
            return launch_mfma(a, st);
        }
    }
    // Larger register tiles read G and X once per bin instead of once per 4x4 tile (config 2: 326 -> 221 MB of fabric
    // traffic per launch) -- and measure SLOWER: 4x4 65 us, 8x4 68 us, 8x8 88 us (tools/dbg/archive/gradh_tile.py, cold caches):
    // the second read of a 4x4 tile comes from the L2, while 64 / 128 accumulator registers cost occupancy.  Kept
    // behind the tuning hook.
    if (sizeof(T) == 4 && No >= 8 && Ni >= 8 && g_gradh_tile != 4) {
        const int tn = g_gradh_tile == 84 ? 4 : 8;
        dim3 grid(cdiv_i(M, 256), cdiv_i(No, 8), cdiv_i(Ni, tn));
        FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradh: too many channels");
        if (tn == 8)
            hipLaunchKernelGGL((mimo_gradh_kernel<T, 8, 8>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                               (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K, dev_scale);
        else
            hipLaunchKernelGGL((mimo_gradh_kernel<T, 8, 4>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                               (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K, dev_scale);
    } else if (No >= 4 && Ni >= 4) {
        dim3 grid(cdiv_i(M, 256), cdiv_i(No, 4), cdiv_i(Ni, 4));
        FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradh: too many channels");
        hipLaunchKernelGGL((mimo_gradh_kernel<T, 4, 4>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                           (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K, dev_scale);
    } else {
        dim3 grid(cdiv_i(M, 256), No, Ni);
        FL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "mimo_gradh: too many channels");
        hipLaunchKernelGGL((mimo_gradh_kernel<T, 1, 1>), grid, dim3(256), 0, st, (const cx<T>*)G, gs_b, gs_m, gs_k,
                           (const cx<T>*)X, xs_b, xs_n, xs_k, (cx<T>*)dH, dh_pitch, (T)scale, B, M, No, Ni, K, dev_scale);
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
                                                              cx<T>* __restrict__ dW, int real_out) {
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
    // real_out: the matrix is real (its complex cast never existed): its gradient is the real part, stored as a real array
    if (lane == 0) {
        if (real_out) reinterpret_cast<T*>(dW)[e] = vr;
        else dW[e] = cx<T>(vr, vi);
    }
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
                      void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream, int real_out = 0) {
    FL_REQUIRE(G && X && part && dW, "mimo_gradw: null pointer");
    FL_REQUIRE(B >= 0 && M >= 0 && No > 0 && Ni > 0 && K > 0, "mimo_gradw: bad sizes");
    // 8x8 tiles where the matrix allows: every element of G and X is then read No/8 (Ni/8) times
    // instead of No/4 -- at N = 32 with matrix-valued signals that is 12 GB instead of 25 GB per launch
    if constexpr (sizeof(T) == 4) {
        if (g_mimo_variant == 0 && mfma_applies(No, Ni, B * K)) {
            // per-bin outer products on the matrix cores, summed over bins in the accumulators (mimo_mfma_lds_kernel)
            MmaArgs a = {};
            a.A = (const cx<float>*)G; a.sa_f = 1; a.sa_i = gs_m; a.sa_t1 = gs_b; a.sa_t2 = gs_k; a.conj_a = 0;
            a.B = (const cx<float>*)X; a.sb_j1 = 0; a.sb_j2 = xs_n; a.sb_t1 = xs_b; a.sb_t2 = xs_k; a.conj_b = 1;
            a.part = (cx<float>*)part; a.scale = 1.f;
            a.M = M; a.NI = No; a.J1 = 1; a.J2 = Ni; a.T1 = B; a.T2 = K;
            int slots = 0;
            int rc = launch_mfma(a, (hipStream_t)stream, gradw_blocks(M), &slots);
            if (rc) return rc;
            hipLaunchKernelGGL((mimo_gradw_final_kernel<T>), dim3(cdiv_i((long)No * Ni, 4)), dim3(256), 0, (hipStream_t)stream,
                               (const cx<T>*)part, slots, No * Ni, (cx<T>*)dW, real_out);
            FL_CHECK_LAUNCH("mimo_gradw_final");
            return FL_OK;
        }
    }
    // (double: 256 accumulator registers -- pays once a bin carries several columns, 3.5 -> 2.2 ms at N = 32 with
    // matrix-valued signals, 215 -> 160 us at N = 16, batch 8; a single column per bin keeps the 4x4 tile)
    const bool big = No >= 8 && Ni >= 8 && (sizeof(T) == 4 || (long)B * K >= 4);
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
                       (const cx<T>*)part, (int)grid.x, No * Ni, (cx<T>*)dW, real_out);
    FL_CHECK_LAUNCH("mimo_gradw_final");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {

int fl_mimo_gradw_blocks(int M) { return gradw_blocks(M); }
int fl_debug_set_mimo_variant(int variant, int gradw_cap) {
    g_mfma_enabled = variant != -1;        // -1: lane-per-bin kernels everywhere (default tiles)
    g_mfma_tile16 = variant == -14;        // -14: MFMA kernels with the 64-bin 16x16 tile everywhere
    if (variant < 0) variant = 0;
    g_mimo_variant = variant;
    g_mfma_vec = gradw_cap != -2;          // gradw_cap -2: direct stores in the MFMA kernels
    g_mimo_hc = gradw_cap != -4;
    g_mimo_stream = gradw_cap <= -1600 || gradw_cap == -16;      // -16, or -(1600 + 10*splits + cb)
    if (gradw_cap <= -1600) {
        g_mimo_stream_split = ((-gradw_cap - 1600) / 10);
        g_mimo_stream_cb = (-gradw_cap - 1600) % 10;
    }
    g_gradh_tile = gradw_cap == -88 ? 8 : (gradw_cap == -84 ? 84 : 4);
    if (gradw_cap < 0) gradw_cap = 0;
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

int fl_mimo_gradw_re_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                         void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream) {
    return gradw_impl<float>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, part, dW, B, M, No, Ni, K, stream, 1);
}
int fl_mimo_gradw_re_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                          void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream) {
    return gradw_impl<double>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, part, dW, B, M, No, Ni, K, stream, 1);
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
int fl_mimo_gradh_scaled_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                             void* dH, long dh_pitch, double scale, const void* dev_scale, int B, int M, int No, int Ni, int K, void* stream) {
    FL_REQUIRE(dev_scale, "mimo_gradh_scaled: null pointer");
    return gradh_impl<float>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, dH, dh_pitch, scale, B, M, No, Ni, K, stream, dev_scale);
}
int fl_mimo_gradh_scaled_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                              void* dH, long dh_pitch, double scale, const void* dev_scale, int B, int M, int No, int Ni, int K, void* stream) {
    FL_REQUIRE(dev_scale, "mimo_gradh_scaled: null pointer");
    return gradh_impl<double>(G, gs_b, gs_m, gs_k, X, xs_b, xs_n, xs_k, dH, dh_pitch, scale, B, M, No, Ni, K, stream, dev_scale);
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
