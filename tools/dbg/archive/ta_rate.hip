// L2-resident load throughput per CU by access width: what the vector-memory path (TA/L1) delivers when the data comes
// from the XCD's L2 -- the regime of the response fetch in the fused row kernel.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <typename V>
__global__ void __launch_bounds__(256) rd(const V* __restrict__ src, size_t nelem_per_wg, int iters, float* out) {
    const V* base = src + (size_t)(blockIdx.x % 64) * nelem_per_wg;     // 64 distinct windows: L2-resident working set
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (size_t i = threadIdx.x; i < nelem_per_wg; i += 256 * 8) {
            V v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(i + 256 * u) % nelem_per_wg];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += ((const float*)&v[u])[0];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename V>
static void run(const char* name, void* buf, size_t bytes_per_wg, int nwg) {
    const size_t n = bytes_per_wg / sizeof(V);
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20;
    hipLaunchKernelGGL(rd<V>, dim3(nwg), dim3(256), 0, 0, (const V*)buf, n, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rd<V>, dim3(nwg), dim3(256), 0, 0, (const V*)buf, n, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * iters * bytes_per_wg;
    printf("%-28s %8.1f GB/s  = %6.1f B/clk/CU at 2.1 GHz\n", name, bytes / ms / 1e6, bytes / ms / 1e6 * 1e9 / 256 / 2.1e9 / 1e0 / 1e0 / 1.0 / 1.0 / 1.0 / 1.0 / 1.0 / 1.0 / 1.0 / 1e0 / 1e0 / 1.0);
}

int main() {
    void* buf;
    const size_t per_wg = 256 * 1024;             // 256 KB window, 64 windows = 16 MB (fits the 32 MB aggregate L2 / 4 MB per XCD partly)
    hipMalloc(&buf, per_wg * 64);
    hipMemset(buf, 0, per_wg * 64);
    for (int nwg : {1024, 4096}) {
        printf("workgroups %d\n", nwg);
        run<float>("dword  (4 B/lane)", buf, per_wg, nwg);
        run<float2>("dwordx2 (8 B/lane)", buf, per_wg, nwg);
        run<float4>("dwordx4 (16 B/lane)", buf, per_wg, nwg);
    }
    return 0;
}
