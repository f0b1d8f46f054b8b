// Shared device/host helpers for the flamo_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/flamo_hip.h"

namespace fl {

// ---------------------------------------------------------------- complex value type
template <typename T>
struct alignas(2 * sizeof(T)) cx {
    T x, y;
    __host__ __device__ cx() {}
    __host__ __device__ cx(T re, T im) : x(re), y(im) {}
};

template <typename T> __host__ __device__ inline cx<T> operator+(cx<T> a, cx<T> b) { return cx<T>(a.x + b.x, a.y + b.y); }
template <typename T> __host__ __device__ inline cx<T> operator-(cx<T> a, cx<T> b) { return cx<T>(a.x - b.x, a.y - b.y); }
template <typename T> __host__ __device__ inline cx<T> operator*(cx<T> a, cx<T> b) {
    return cx<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> __host__ __device__ inline cx<T> operator*(T s, cx<T> a) { return cx<T>(s * a.x, s * a.y); }
template <typename T> __host__ __device__ inline cx<T> mul_plain(cx<T> a, cx<T> b) { return cx<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __host__ __device__ inline cx<T> conj(cx<T> a) { return cx<T>(a.x, -a.y); }
// a * conj(b)
template <typename T> __host__ __device__ inline cx<T> mulc(cx<T> a, cx<T> b) {
    return cx<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// acc += a*b
template <typename T> __host__ __device__ inline void fma_cx(cx<T>& acc, cx<T> a, cx<T> b) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
// acc += a*conj(b)
template <typename T> __host__ __device__ inline void fma_cxc(cx<T>& acc, cx<T> a, cx<T> b) {
    acc.x += a.x * b.x + a.y * b.y;
    acc.y += a.y * b.x - a.x * b.y;
}
#ifdef FL_PACKED_COMPLEX
// ---- float: the same operators on 2-element vectors (opt-in: the transform kernels define FL_PACKED_COMPLEX), so that they become packed instructions (v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32, two floats per lane per issue slot) with the swizzles and sign flips of complex
// arithmetic as operand selectors; written with scalars the compiler packs about half of them and pays for the
// other half with register shuffles (v_mov was a fifth of the FFT kernels' instructions).  Not for the ALU-bound
// kernels (solves, cascades): a packed FMA has the throughput of two scalar ones on this part, and the sign flips cost
// extra instructions there (the N = 32 solve went 1.95 -> 3.1 ms with these operators).
typedef float f2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline f2 v2(cx<float> a) { return f2{a.x, a.y}; }
__host__ __device__ inline cx<float> c2(f2 v) { return cx<float>(v.x, v.y); }
// ---- packed complex arithmetic with the swaps and sign flips as VOP3P operand modifiers (device code).  hipcc folds a whole-
// vector negation into v_pk_add (a - b) and a broadcast half into op_sel, but NOT "swap the halves AND negate one": i*b, and
// the (-y, y) / (-ti, tr) operand of a complex product, came out as v_xor + v_mov per use -- a fifth of the FFT stages'
// instructions (32 + 19 of the 271 of a 16-point stage).  The operators below are the single instructions the hardware has:
//   op_sel[i] / op_sel_hi[i]: which half of source i feeds the low / high result lane;  neg_lo / neg_hi: negate source i there.
// Plain `asm` (not volatile): pure functions of their operands, free to be scheduled, merged and removed.
#define FL_PK_ASM 1
#if defined(__HIP_DEVICE_COMPILE__)
#define FL_PK_DEV 1
#else
#define FL_PK_DEV 0      // host pass: the same functions in plain C++ (never executed; device code is what runs)
#endif
__device__ __forceinline__ f2 pk_add_i(f2 a, f2 b) {      // a + i b = (a.x - b.y, a.y + b.x)
#if FL_PK_DEV
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return f2{a.x - b.y, a.y + b.x};
#endif
}
__device__ __forceinline__ f2 pk_sub_i(f2 a, f2 b) {      // a - i b = (a.x + b.y, a.y - b.x)
#if FL_PK_DEV
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return f2{a.x + b.y, a.y - b.x};
#endif
}
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 t) {       // a t: two instructions
#if FL_PK_DEV
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));                                 // (x tr, x ti)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "v"(t));   // (-y ti, y tr) +
    return r;
#else
    return f2{a.x * t.x - a.y * t.y, a.x * t.y + a.y * t.x};
#endif
}
__device__ __forceinline__ f2 pk_cmulc(f2 a, f2 t) {      // a conj(t)
#if FL_PK_DEV
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));                     // (x tr, -x ti)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "v"(t));             // (y ti, y tr) +
    return r;
#else
    return f2{a.x * t.x + a.y * t.y, a.y * t.x - a.x * t.y};
#endif
}
// the same product with a wavefront-uniform factor (a compile-time twiddle): it rides in an SGPR pair
__device__ __forceinline__ f2 pk_cmul_s(f2 a, f2 t) {
#if FL_PK_DEV
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(t));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "s"(t));
    return r;
#else
    return f2{a.x * t.x - a.y * t.y, a.x * t.y + a.y * t.x};
#endif
}
__device__ __forceinline__ f2 pk_rot_s(f2 a, f2 k) {      // (a.y k.x, a.x k.y): i a with k = (-1, 1), -i a with k = (1, -1)
#if FL_PK_DEV
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(k));
    return r;
#else
    return f2{a.y * k.x, a.x * k.y};
#endif
}
// a + s (i b) and a - s (i b), s a compile-time real: the +-i times a real-scaled difference of the radix-3 butterfly
__device__ __forceinline__ f2 pk_fma_i(float s, f2 b, f2 a) {      // a + i s b = (a.x - s b.y, a.y + s b.x)
#if FL_PK_DEV
    f2 r;
    const f2 sv = {s, s};       // (an SGPR pair)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "s"(sv), "v"(b), "v"(a));
    return r;
#else
    return f2{a.x - s * b.y, a.y + s * b.x};
#endif
}
__device__ __forceinline__ f2 pk_fms_i(float s, f2 b, f2 a) {      // a - i s b = (a.x + s b.y, a.y - s b.x)
#if FL_PK_DEV
    f2 r;
    const f2 sv = {s, s};
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(r) : "s"(sv), "v"(b), "v"(a));
    return r;
#else
    return f2{a.x + s * b.y, a.y - s * b.x};
#endif
}
__host__ __device__ inline cx<float> operator+(cx<float> a, cx<float> b) { return c2(v2(a) + v2(b)); }
__host__ __device__ inline cx<float> operator-(cx<float> a, cx<float> b) { return c2(v2(a) - v2(b)); }
__host__ __device__ inline cx<float> operator*(cx<float> a, cx<float> b) {
#if FL_PK_DEV
    return c2(pk_cmul(v2(a), v2(b)));
#endif
    f2 r = f2{a.x, a.x} * v2(b);
    return c2(__builtin_elementwise_fma(f2{-a.y, a.y}, f2{b.y, b.x}, r));
}
__host__ __device__ inline cx<float> operator*(float s, cx<float> a) { return c2(f2{s, s} * v2(a)); }
__host__ __device__ inline cx<float> mul_plain(cx<float> a, cx<float> b) {     // a b in compiler-visible form (see regfft.h, PK)
    f2 r = f2{a.x, a.x} * v2(b);
    return c2(__builtin_elementwise_fma(f2{-a.y, a.y}, f2{b.y, b.x}, r));
}
__host__ __device__ inline cx<float> mulc(cx<float> a, cx<float> b) {     // a * conj(b)
#if FL_PK_DEV
    return c2(pk_cmulc(v2(a), v2(b)));
#endif
    f2 r = f2{b.x, b.x} * v2(a);
    return c2(__builtin_elementwise_fma(f2{b.y, -b.y}, f2{a.y, a.x}, r));
}
// (the accumulating products acc += a*b keep the scalar form: four v_fma_f32 with no operand shuffles -- a packed
// FMA has the throughput of two scalar ones on this part, and its sign flip costs an extra instruction)
__host__ __device__ inline cx<float> axpy(float s, cx<float> a, cx<float> acc) {     // acc + s*a (real s)
    return c2(__builtin_elementwise_fma(f2{s, s}, v2(a), v2(acc)));
}
#endif
// acc + s*a (real s)
template <typename T> __host__ __device__ inline cx<T> axpy(T s, cx<T> a, cx<T> acc) { return cx<T>(acc.x + s * a.x, acc.y + s * a.y); }

template <typename T> __host__ __device__ inline cx<T> mul_i(cx<T> a) { return cx<T>(-a.y, a.x); }     // i*a
template <typename T> __host__ __device__ inline cx<T> mul_mi(cx<T> a) { return cx<T>(a.y, -a.x); }    // -i*a
template <typename T> __host__ __device__ inline cx<T> cdiv(cx<T> a, cx<T> b) {
    // Smith's algorithm (robust against overflow in |b|^2)
    if (fabs(b.x) >= fabs(b.y)) {
        T r = b.y / b.x, d = b.x + b.y * r;
        return cx<T>((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    } else {
        T r = b.x / b.y, d = b.x * r + b.y;
        return cx<T>((a.x * r + a.y) / d, (a.y * r - a.x) / d);
    }
}

// Wavefront reduce-scatter of N per-lane values: after the call the lane holds in a[0..cnt) the
// wavefront totals of the original entries off .. off+cnt-1.  Each halving step trades half of the
// lane's values with its partner (N/2 shuffles instead of N); when the count turns odd the rest is
// reduced by plain butterflies.  72 running sums cost 90 shuffles instead of 432 (the epilogue, not
// the bin loop, used to dominate the cascade backward).  Lanes whose index has a bit of dup_mask set
// hold duplicates.
template <typename V, int N, int MASK>
__device__ inline void wave_reduce_scatter(V (&a)[N], int lane, int& off, int& cnt, int& dup_mask) {
    if constexpr (MASK == 0) {
        cnt = N;
    } else if constexpr (N % 2 == 0) {
        const bool up = (lane & MASK) != 0;
        V k[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const V keep = up ? a[N / 2 + i] : a[i];
            const V send = up ? a[i] : a[N / 2 + i];
            k[i] = keep + __shfl_xor(send, MASK, 64);
        }
        if (up) off += N / 2;
        wave_reduce_scatter<V, N / 2, MASK / 2>(k, lane, off, cnt, dup_mask);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) a[i] = k[i];
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] += __shfl_xor(a[i], MASK, 64);
        dup_mask |= MASK;   // lanes differing in this bit now hold the same totals
        wave_reduce_scatter<V, N, MASK / 2>(a, lane, off, cnt, dup_mask);
    }
}

// 1 / b without branches: scale by max(|re|,|im|) (no overflow in |b|^2), two real divisions instead of
// the six (three per branch) of Smith's quotient -- the pivot reciprocal of the LU sits on its critical path
template <typename T> __host__ __device__ inline cx<T> crecip(cx<T> b) {
    const T s = fmax(fabs(b.x), fabs(b.y));
    const T is = (T)1 / s;
    const T x = b.x * is, y = b.y * is;
    const T d = is / (x * x + y * y);
    return cx<T>(x * d, -y * d);
}

// ---------------------------------------------------------------- error plumbing (host)
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define FL_CHECK_LAUNCH(what)                                  \
    do {                                                       \
        int _rc = fl::check_hip(hipGetLastError(), what);      \
        if (_rc) return _rc;                                   \
    } while (0)

#define FL_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) {                                         \
            fl::set_error(__VA_ARGS__);                        \
            return FL_ERR_BAD_ARG;                             \
        }                                                      \
    } while (0)

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- cache policy of the pipeline's data streams (host)
// Each stream of the fused Shell pipeline (spectral.hip, specwalk.hip) and of the cascade kernels is loaded / stored either with
// the default policy or non-temporally; which is right depends on who touched the data last and who reads it next (DESIGN 4.10:
// a plain read of data that left the Infinity Cache runs at half the rate of a non-temporal one behind a writing pass).  One bit
// per stream; `site` 1 selects the second set of column-pass bits (the gradient's transform: its input was written by the
// previous launch, the forward transform's was not).  fl_set_stream_policy (flamo_hip.h) sets both.
enum StreamPolicy : unsigned {
    POL_COLS_LD_NT = 1u << 0,    // spec_cols_fwd: loads of the time-domain input
    POL_COLS_ST_NT = 1u << 1,    // spec_cols_fwd: stores of the scratch rows
    POL_INV_LD_NT = 1u << 2,     // spec_cols_inv: loads of the scratch rows
    POL_INV_ST_NT = 1u << 3,     // spec_cols_inv: stores of the time-domain output
    POL_WALK_S_NT = 1u << 4,     // spec_mid_walk: LDS-DMA of the scratch rows
    POL_WALK_S2_PLAIN = 1u << 5, // spec_mid_walk: scratch rows out with the default policy (non-temporal otherwise)
    POL_WALK_XP_PLAIN = 1u << 6, // spec_mid_walk: kept spectrum out with the default policy (non-temporal otherwise)
    POL_GRADH_SG_NT = 1u << 7,   // spec_gradh_walk: LDS-DMA of the gradient's scratch rows
    POL_GRADH_XP_NT = 1u << 8,   // spec_gradh_walk: LDS-DMA of the kept spectrum
    POL_GRADH_DH_NT = 1u << 9,   // spec_gradh_walk: stores of dL/dH
    POL_LANES_G_NT = 1u << 10,   // sos_bwd_lanes: loads of the cascade response G (written a forward pass ago)
    POL_LANES_GH_NT = 1u << 11,  // sos_bwd_lanes: loads of dL/dH (written by the previous launch)
    POL_RC_ST_NT = 1u << 12,     // sos_response_rc_ba: stores of G (read again only by the backward pass)
    POL_INV_SG_NT = 1u << 13,    // spec_cols_inv<..., FUSE>: stores of the gradient's column pass (read by the next launch but one)
    POL_GRADH_FORWARD = 1u << 14, // spec_gradh_walk: the batch items of a slice first to last (default: last to first -- the inverse column
                                  // pass wrote the last items' rows last: 285.3 against 286.7 us per replayed step, eight interleaved rounds)
    POL_SITE1_SHIFT = 16,        // bits 16, 17: POL_COLS_LD_NT / POL_COLS_ST_NT of site 1
};
unsigned stream_policy();        // the current mask
int stream_site();               // 0 / 1

}  // namespace fl
