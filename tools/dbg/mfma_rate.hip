// Issue rate of the fp32 MFMA shapes on gfx950: N independent accumulator chains, ITER rounds.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/mfma_rate.hip -o tools/dbg/bin/mfma_rate && tools/dbg/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters) {
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float s = 0;
    if constexpr (KIND == 0) {          // 4x4x1 x16 blocks: 512 flop
        v4f acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (v4f)(0.f);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else if constexpr (KIND == 1) {   // 16x16x4: 2048 flop
        v4f acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (v4f)(0.f);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else if constexpr (KIND == 2) {   // 32x32x2: 4096 flop
        v16f acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v16f)(0.f);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    } else {                             // packed fp32 FMA: 256 flop per instruction
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f2)(0.f);
        const f2 av = {a, b}, bv = {b, a};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(av, bv, acc[i]);
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, double flop_per_instr, int per_round, int blocks = 2048) {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<KIND><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * per_round;
    const double simds = blocks >= 256 ? 1024 : blocks * 4;
    printf("%-12s blocks %5d %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", name, blocks, ms,
           instr * flop_per_instr / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (instr / simds));
    hipFree(out);
}

int main() {
    run<0>("4x4x1_16b", 512, 8);
    run<1>("16x16x4", 2048, 8);
    run<2>("32x32x2", 4096, 4);
    run<3>("v_pk_fma_f32", 256, 16);
    run<0>("4x4x1_16b", 512, 8, 256);      // one workgroup per CU: one wavefront per SIMD
    run<1>("16x16x4", 2048, 8, 256);
    run<3>("v_pk_fma_f32", 256, 16, 256);
    return 0;
}
