// Trivial kernels for tools/dbg/replay_min.py: does the graph-replay hazard of DESIGN 4.5 need anything of libflamo_hip?
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/dbg/tiny_kernels.hip -o tools/dbg/bin/libtiny.so
#include <hip/hip_runtime.h>
struct Big { float v[56]; const float* in; float* out; int n; };      // 240-byte by-value argument
__global__ void k_plain(const float* in, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.5f;
}
__global__ void k_big(Big a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] * a.v[i % 56];
}
__global__ void k_lds(const float* in, float* out, int n) {
    extern __shared__ float sm[];
    const int i = blockIdx.x * 256 + threadIdx.x;
    sm[threadIdx.x * 100] = i < n ? in[i] : 0.f;
    __syncthreads();
    if (i < n) out[i] = sm[threadIdx.x * 100] * 1.5f;
}
extern "C" {
int tiny_plain(const float* in, float* out, int n, void* st) {
    hipLaunchKernelGGL(k_plain, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, in, out, n);
    return (int)hipGetLastError();
}
int tiny_big(const float* in, float* out, int n, void* st) {
    Big a;
    for (int i = 0; i < 56; ++i) a.v[i] = 1.5f;
    a.in = in; a.out = out; a.n = n;
    hipLaunchKernelGGL(k_big, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, a);
    return (int)hipGetLastError();
}
int tiny_lds(const float* in, float* out, int n, void* st) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024); set = true; }
    hipLaunchKernelGGL(k_lds, dim3((n + 255) / 256), dim3(256), 102400 + 1024, (hipStream_t)st, in, out, n);
    return (int)hipGetLastError();
}
}
