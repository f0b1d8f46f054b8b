// What the part's memory system sustains on hand-written streaming kernels (gfx950 / MI355X): the ceiling the
// roofline fractions of the path's HBM-bound passes (spec_cols_*, spec_mid_walk, spec_gradh_walk) are read against.
//
// Not part of the path: measurement only (bench.py's "device" object, tools/dbg/hbm_probe.py).  torch's copy_ is a 50/50
// read/write kernel of torch's own; the passes of the path are read-mostly (spec_gradh_walk: 8 bytes read per byte
// written) and their scratch (98 MB) fits the 256 MiB Infinity Cache, so the ceiling has to be measured per access mix,
// per buffer size and per cache policy with kernels shaped like the path's own: persistent grid of k x 256 workgroups,
// 16 bytes per lane and access, eight independent accesses in flight per lane.
#include "common.h"

namespace fl {

typedef float pf4 __attribute__((ext_vector_type(4)));

constexpr int PROBE_THREADS = 256;
constexpr int PROBE_UNROLL = 8;
constexpr size_t PROBE_CHUNK = (size_t)PROBE_THREADS * PROBE_UNROLL;   // float4 elements per workgroup and trip (32 KB)

template <bool NT> __device__ __forceinline__ pf4 probe_ld(const pf4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void probe_st(pf4* p, pf4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// KIND 0: read only (one float per workgroup leaves the chip); 1: write only; 2: copy (1 read : 1 write);
// 3: read-mostly, 8 bytes read per byte written (the mix of spec_gradh_walk: two signal spectra in, one response out).
template <int KIND, bool NTL, bool NTS>
__global__ void __launch_bounds__(PROBE_THREADS) hbm_probe_kernel(const pf4* __restrict__ src, pf4* __restrict__ dst, size_t n4,
                                                                  int reverse, float* __restrict__ partial) {
    const size_t chunks = n4 / PROBE_CHUNK;
    pf4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const size_t cc = reverse ? chunks - 1 - c : c;
        const size_t base = cc * PROBE_CHUNK + threadIdx.x;
        pf4 v[PROBE_UNROLL];
        if constexpr (KIND != 1) {
#pragma unroll
            for (int u = 0; u < PROBE_UNROLL; ++u) v[u] = probe_ld<NTL>(src + base + (size_t)u * PROBE_THREADS);
        }
        if constexpr (KIND == 0) {
#pragma unroll
            for (int u = 0; u < PROBE_UNROLL; ++u) acc += v[u];
        } else if constexpr (KIND == 1) {
            const pf4 k = {1.f, 2.f, 3.f, (float)cc};
#pragma unroll
            for (int u = 0; u < PROBE_UNROLL; ++u) probe_st<NTS>(dst + base + (size_t)u * PROBE_THREADS, k);
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int u = 0; u < PROBE_UNROLL; ++u) probe_st<NTS>(dst + base + (size_t)u * PROBE_THREADS, v[u]);
        } else {
            pf4 s = v[0];
#pragma unroll
            for (int u = 1; u < PROBE_UNROLL; ++u) s += v[u];
            probe_st<NTS>(dst + cc * PROBE_THREADS + threadIdx.x, s);       // one 16-byte store per eight loads
        }
    }
    if constexpr (KIND == 0) {
        float s = acc.x + acc.y + acc.z + acc.w;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        __shared__ float red[PROBE_THREADS / 64];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}


// the same streams with 8 bytes per lane and access (the column passes' width: a lane owns one complex value / one channel pair
// of one sample): what the access width alone costs
typedef float pf2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(PROBE_THREADS) hbm_probe8_kernel(const pf2* __restrict__ src, pf2* __restrict__ dst, size_t n2,
                                                                   float* __restrict__ partial) {
    constexpr int U = 2 * PROBE_UNROLL;                       // the same bytes in flight per lane
    const size_t chunk = (size_t)PROBE_THREADS * U, chunks = n2 / chunk;
    pf2 acc = {0.f, 0.f};
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        pf2 v[U];
        if constexpr (KIND != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[base + (size_t)u * PROBE_THREADS];
        }
        if constexpr (KIND == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        } else if constexpr (KIND == 1) {
            const pf2 k = {1.f, (float)c};
#pragma unroll
            for (int u = 0; u < U; ++u) dst[base + (size_t)u * PROBE_THREADS] = k;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) dst[base + (size_t)u * PROBE_THREADS] = v[u];
        }
    }
    if constexpr (KIND == 0) {
        float s = acc.x + acc.y;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        __shared__ float red[PROBE_THREADS / 64];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

template <int KIND>
static int probe_launch(const void* src, void* dst, size_t n4, int wgs, int flags, float* partial, hipStream_t st) {
    const int rev = (flags >> 2) & 1;
    const pf4* s = (const pf4*)src;
    pf4* d = (pf4*)dst;
    switch (flags & 3) {
        case 0: hipLaunchKernelGGL((hbm_probe_kernel<KIND, false, false>), dim3(wgs), dim3(PROBE_THREADS), 0, st, s, d, n4, rev, partial); break;
        case 1: hipLaunchKernelGGL((hbm_probe_kernel<KIND, true, false>), dim3(wgs), dim3(PROBE_THREADS), 0, st, s, d, n4, rev, partial); break;
        case 2: hipLaunchKernelGGL((hbm_probe_kernel<KIND, false, true>), dim3(wgs), dim3(PROBE_THREADS), 0, st, s, d, n4, rev, partial); break;
        default: hipLaunchKernelGGL((hbm_probe_kernel<KIND, true, true>), dim3(wgs), dim3(PROBE_THREADS), 0, st, s, d, n4, rev, partial); break;
    }
    return FL_OK;
}

}  // namespace fl

extern "C" int fl_hbm_probe(int kind, const void* src, void* dst, size_t bytes, int workgroups, int flags, void* partials,
                            void* stream) {
    using namespace fl;
    FL_REQUIRE(kind >= 0 && kind <= 3, "fl_hbm_probe: kind 0 (read) / 1 (write) / 2 (copy) / 3 (8 reads per write)");
    FL_REQUIRE(bytes >= 16 * PROBE_CHUNK && bytes % (16 * PROBE_CHUNK) == 0, "fl_hbm_probe: bytes must be a multiple of 32 KiB");
    FL_REQUIRE(workgroups >= 1 && workgroups <= 65536, "fl_hbm_probe: 1..65536 workgroups");
    FL_REQUIRE((kind == 1 || src) && (kind == 0 || dst) && (kind != 0 || partials), "fl_hbm_probe: null buffer");
    const size_t n4 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    if (flags & 8) {      // 8 bytes per lane and access (kinds 0 / 1 / 2, default policy)
        FL_REQUIRE(kind <= 2, "fl_hbm_probe: the 8-byte form has kinds 0, 1, 2");
        const pf2* s2 = (const pf2*)src;
        pf2* d2 = (pf2*)dst;
        if (kind == 0) hipLaunchKernelGGL((hbm_probe8_kernel<0>), dim3(workgroups), dim3(PROBE_THREADS), 0, st, s2, d2, bytes / 8, (float*)partials);
        else if (kind == 1) hipLaunchKernelGGL((hbm_probe8_kernel<1>), dim3(workgroups), dim3(PROBE_THREADS), 0, st, s2, d2, bytes / 8, (float*)partials);
        else hipLaunchKernelGGL((hbm_probe8_kernel<2>), dim3(workgroups), dim3(PROBE_THREADS), 0, st, s2, d2, bytes / 8, (float*)partials);
        FL_CHECK_LAUNCH("fl_hbm_probe");
        return FL_OK;
    }
    switch (kind) {
        case 0: probe_launch<0>(src, dst, n4, workgroups, flags, (float*)partials, st); break;
        case 1: probe_launch<1>(src, dst, n4, workgroups, flags, (float*)partials, st); break;
        case 2: probe_launch<2>(src, dst, n4, workgroups, flags, (float*)partials, st); break;
        default: probe_launch<3>(src, dst, n4, workgroups, flags, (float*)partials, st); break;
    }
    FL_CHECK_LAUNCH("fl_hbm_probe");
    return FL_OK;
}
