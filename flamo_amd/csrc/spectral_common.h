// Device helpers shared by the fused Shell pipeline's kernels (spectral.hip, specwalk.hip); gfx950 only.
#pragma once
#ifndef FL_PACKED_COMPLEX
#define FL_PACKED_COMPLEX 1
#endif
#include "common.h"
#include "regfft.h"
#include <type_traits>

// The pipeline's kernels exist in float32 and float64: spectral.hip is compiled twice (the Makefile's second rule adds
// -DFL_F64: real_t = double, exports *_f64 / *_c128), each build in a namespace of its own so that the two sets of kernels
// and argument structs do not collide.  specwalk.hip (float32 only) sees the float32 names.
#ifdef FL_F64
#define FL_SPEC_NS sp64
#else
#define FL_SPEC_NS sp32
#endif

namespace fl {
namespace FL_SPEC_NS {

#ifdef FL_F64
typedef double real_t;
typedef double2 real2;
__device__ __forceinline__ real2 make_real2(double a, double b) { return make_double2(a, b); }
#else
typedef float real_t;
typedef float2 real2;
__device__ __forceinline__ real2 make_real2(float a, float b) { return make_float2(a, b); }
#endif
typedef cx<real_t> cf;
constexpr unsigned ESZ = sizeof(cf), RSZ = sizeof(real_t);      // bytes per complex / real element (32-bit lane offsets)

// lane <-> lane^1 exchange (DPP quad_perm [1,0,3,2]), one dword at a time
__device__ __forceinline__ float swap1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ double swap1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

struct ColsArgs {
    const real_t* x;      // forward: real (Bn, t_len, G)
    real_t* y;            // inverse: real (Bn, t_len, G)
    cf* S;                // (Bn, L1, L2, G)
    const cf* W;          // W_n^j, j < n
    int n, L, L1, L2, G, cgs /* log2 CG */, CT, nct /* L2 / CT */, ngt /* G / CG */;
    int t_len, t_lim;
    real_t scale;
    double env_log2;
    const real_t* dev_scale;   // inverse, optional: a DEVICE scalar multiplied into `scale` (the objective's 2 g / N in the gradient
                          // of the input: ops.mean_square) -- a multiplication pass over the result otherwise
    double* sumsq;        // inverse, optional: sumsq[workgroup] = sum of the squares of the samples this workgroup stored (the
                          // objective's reduction rides in the pass that produces y: ops.mean_square never re-reads it)
    cf* Sg;               // inverse, optional: (Bn, L1, L2, G) -- the FORWARD column pass of the samples this launch stores (what
                          // fl_spec_cols_fwd of y leaves: the first pass of the gradient's transform when g_y is a multiple of y),
                          // formed from the tile in registers: y is not read back (spec_cols_inv<..., FUSE>)
    unsigned pol;         // cache policy of this launch's streams: bit 0 non-temporal loads, bit 1 non-temporal stores (common.h: StreamPolicy)
    long long* stamp;     // measurement, or null: the inverse pass's workgroup 0 leaves the device clock here when it starts
                          // (with spec_mid_walk's own stamps: that kernel's whole slot in a replayed step; fl_debug_set_walk_stamps)
};

// Global accesses as (workgroup-uniform base pointer) + (32-bit byte offset per lane): the address then costs one
// VGPR per access (scalar base + vector offset form) instead of a 64-bit pair -- with 16..50 accesses in flight per
// thread that is a wavefront per SIMD.  Every array addressed this way spans < 4 GB per batch item.
template <typename P>
__device__ __forceinline__ P& at(P* base, unsigned byte_off) {
    return *reinterpret_cast<P*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(base)) + byte_off);
}

// streaming data (read once / written once): non-temporal, so that it does not evict the response slices the batch
// items of a row pair share in the XCD's L2
typedef real_t v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ld_nt(const cf* base, unsigned byte_off) {
    const v2f q = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(base) + byte_off));
    return cf(q.x, q.y);
}
__device__ __forceinline__ void st_nt(cf* base, unsigned byte_off, cf v) {
    v2f q;
    q.x = v.x;
    q.y = v.y;
    __builtin_nontemporal_store(q, reinterpret_cast<v2f*>(reinterpret_cast<char*>(base) + byte_off));
}

// the same accesses with the policy as a template parameter (the kernels branch ONCE, workgroup-uniformly, around a whole group
// of accesses: see ColsArgs::pol)
template <bool NT>
__device__ __forceinline__ v2f ldv(const void* base, unsigned byte_off) {
    const v2f* p = reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(base) + byte_off);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void stv(void* base, unsigned byte_off, v2f q) {
    v2f* p = reinterpret_cast<v2f*>(reinterpret_cast<char*>(base) + byte_off);
    if constexpr (NT) __builtin_nontemporal_store(q, p);
    else *p = q;
}

#ifdef FL_F64
__device__ __forceinline__ double env_at(double env_log2, int t) { return exp2(env_log2 * (double)t); }
#else
__device__ __forceinline__ float env_at(double env_log2, int t) { return exp2f((float)(env_log2 * (double)t)); }
#endif

// bin pair (k, L-k) number p of primary row r: where the partner sits (slot, column); false when p owns no pair
__device__ __forceinline__ bool pair_of(int r, bool selfm, int p, int LEN, int& slotB, int& colB, bool& dc) {
    dc = false;
    if (r == 0) {
        if (2 * p > LEN) return false;
        dc = p == 0;
        slotB = 0;
        colB = (2 * p == LEN || p == 0) ? p : LEN - p;
    } else if (selfm) {
        if (2 * p >= LEN) return false;
        slotB = 0;
        colB = LEN - 1 - p;
    } else {
        slotB = 1;
        colB = LEN - 1 - p;
    }
    return true;
}


// ---------------------------------------------------------------- LDS-DMA, LDS-only barrier, explicit counter waits
// LDS-DMA of 16 bytes per lane: 64 lanes' pieces land at lds_byte_addr + 16*lane (wave-uniform base in M0), straight from
// the per-lane global address -- no VGPR round trip.  Written as inline assembly ON PURPOSE: issued through the builtin,
// the compiler orders every later LDS read of the kernel behind the transfer (it cannot tell the staging buffer from the
// row buffers: s_waitcnt vmcnt(0) in front of the next ds_read), which is exactly the overlap this kernel exists for.
// The kernel waits for its transfers itself (vmcnt(0) in front of the barrier that precedes their first read).
__device__ __forceinline__ void dma16(const void* g, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte_addr) : "memory");
}
// the same with a wavefront-uniform base in an SGPR pair and a 32-bit per-lane byte offset (no 64-bit address arithmetic per piece)
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// ... and with the non-temporal policy (streamed once by this CU, not wanted in L2 / the Infinity Cache afterwards)
template <bool NT>
__device__ __forceinline__ void dma16p(const void* g, unsigned lds_byte_addr) {
    if constexpr (NT) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds_byte_addr) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte_addr) : "memory");
}
template <bool NT>
__device__ __forceinline__ void dma16sp(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// workgroup barrier that orders LDS traffic only (a plain __syncthreads() also drains the vector-memory queue: the
// prefetch in flight and the previous unit's stores)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // s_waitcnt vmcnt(0)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }    // s_waitcnt lgkmcnt(0)

// s_waitcnt vmcnt(n) for a wavefront-uniform n (the count of this wavefront's NEWER transfers that may stay in flight); the
// counter's field is split in the encoding (bits 3:0 and 15:14).  Unknown counts wait for everything: always safe.
__device__ __forceinline__ void wait_vm_le(int n) {
    switch (n) {
        case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
        case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
        case 16: __builtin_amdgcn_s_waitcnt(0x4F70); break;
        case 20: __builtin_amdgcn_s_waitcnt(0x4F74); break;
        case 24: __builtin_amdgcn_s_waitcnt(0x4F78); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
}
// the same for a compile-time count (0 ... 63)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
}

}  // namespace FL_SPEC_NS

// specwalk.hip: where the launch behind spec_mid_walk stamps its start (null: measurement off)
long long* walk_successor_stamp();

// the fused plan of a transform length (spectral.hip, float32 build)
int spec_plan(int nfft, int& L1, int& L2);
int spec_plan_lean(int nfft);

}  // namespace fl
