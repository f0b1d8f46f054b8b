// Mean-square of a real signal block and its gradient (gfx950 / MI355X).
//
//   loss = (1/count) * sum y[r, c]^2          g_y[r, c] = (2/count) * g_loss * y[r, c]
//
// the scalar objective the training step puts on the output of the path (trainer.py:179-191
// reduce criterion(estimations, targets) to one scalar and call backward()).  Written as
// (y ** 2).mean() in torch this is five elementwise/reduction launches that read or write the
// whole (B, T, N) output eight times; here it is one streaming read forward and one read + one
// write backward, in whatever layout the irfft left the signal (rows with a pitch).
//
// Deterministic: per-block partial sums in double, combined in a fixed order by a second one-block
// launch.  (A single launch with a "last block done" ticket needs a device-scope fence per block,
// which on this multi-XCD part writes back / invalidates the XCD's L2 under the streaming reads:
// measured 57-85 us for the 98 MB pass instead of ~20.)
#include "common.h"

namespace fl {

template <typename T> struct Vec4 { T v[4]; };
template <> struct alignas(16) Vec4<float> { float v[4]; };
template <> struct alignas(32) Vec4<double> { double v[4]; };

constexpr int MS_MAX_BLOCKS = 4096;

template <typename T, int VEC>
__global__ void __launch_bounds__(256) mean_square_kernel(const T* __restrict__ y, long rows, long cols, long pitch,
                                                         double* __restrict__ partial) {
    double acc = 0.0;
    const long cv = cols / VEC;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {
        const T* row = y + r * pitch;
        const long stride = (long)gridDim.x * 256;
        long c = (long)blockIdx.x * 256 + threadIdx.x;
        if constexpr (VEC == 4) {
            // four independent 16-byte loads in flight per lane
            for (; c + 3 * stride < cv; c += 4 * stride) {
                Vec4<T> q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const Vec4<T>*>(row + (c + u * stride) * 4);
                T s[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    s[u] = q[u].v[0] * q[u].v[0] + q[u].v[1] * q[u].v[1] + q[u].v[2] * q[u].v[2] + q[u].v[3] * q[u].v[3];
                acc += (double)(s[0] + s[1]) + (double)(s[2] + s[3]);
            }
            for (; c < cv; c += stride) {
                const Vec4<T> q = *reinterpret_cast<const Vec4<T>*>(row + c * 4);
                acc += (double)(q.v[0] * q.v[0] + q.v[1] * q.v[1] + q.v[2] * q.v[2] + q.v[3] * q.v[3]);
            }
        } else {
            for (; c < cv; c += stride) {
                const T q = row[c];
                acc += (double)(q * q);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ void __launch_bounds__(256) mean_square_final_kernel(const double* __restrict__ partial, int nblocks,
                                                               double inv_count, T* __restrict__ loss) {
    // a few thousand partials behind one workgroup: the round trips, not the adds, are the time -- up to sixteen loads per lane
    // requested together and unconditionally (an index beyond the range reads partial[0] and adds zero: with the guard around the
    // load the 1792 partials behind the first 2048 of the metric's 3840 were seven round trips in a row); the order of the
    // additions is fixed
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = threadIdx.x;
    {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = i + u * 256;
            v[u] = partial[k < nblocks ? k : 0];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) q[u & 7] += (i + u * 256 < nblocks) ? v[u] : 0.0;
        i += 16 * 256;
    }
    for (; i + 7 * 256 < nblocks; i += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] += v[u];
    }
    for (int u = 0; i < nblocks; i += 256, ++u) q[u & 7] += partial[i];
    double s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (T)((red[0] + red[1] + red[2] + red[3]) * inv_count);
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) mean_square_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gloss,
                                                             double two_inv_count, T* __restrict__ gy, long rows,
                                                             long cols, long pitch) {
    const T k = (T)(two_inv_count * (double)gloss[0]);
    const long cv = cols / VEC;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {
        const T* row = y + r * pitch;
        T* out = gy + r * pitch;
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < cv; c += (long)gridDim.x * 256) {
            if constexpr (VEC == 4) {
                Vec4<T> q = *reinterpret_cast<const Vec4<T>*>(row + c * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) q.v[i] *= k;
                *reinterpret_cast<Vec4<T>*>(out + c * 4) = q;
            } else {
                out[c] = row[c] * k;
            }
        }
    }
}

static dim3 ms_grid(long rows, long cols, int vec) {
    const long cv = cols / vec;
    long gx = (cv + 255) / 256;
    if (rows == 1) {
        if (gx > 2048) gx = 2048;   // 8 blocks per CU, each lane keeps several 16-byte loads in flight
        return dim3((unsigned)(gx < 1 ? 1 : gx), 1);
    }
    if (gx > 16) gx = 16;
    long gy = rows;
    const long cap = MS_MAX_BLOCKS / (gx < 1 ? 1 : gx);
    if (gy > cap) gy = cap;
    return dim3((unsigned)(gx < 1 ? 1 : gx), (unsigned)gy);
}

template <typename T>
static int pick_vec(const void* p, const void* q, long cols, long pitch, long rows) {
    const uintptr_t a = (uintptr_t)p | (uintptr_t)q;
    const bool ok = (a % (4 * sizeof(T)) == 0) && (cols % 4 == 0) && (rows == 1 || pitch % 4 == 0);
    return ok ? 4 : 1;
}

template <typename T>
static int mean_square_impl(const void* y, long rows, long cols, long pitch, void* loss, void* scratch, void* stream) {
    FL_REQUIRE(y && loss && scratch, "mean_square: null pointer");
    FL_REQUIRE(rows > 0 && cols > 0 && pitch >= cols, "mean_square: bad sizes (rows, cols > 0, pitch >= cols)");
    if (pitch == cols) { cols *= rows; rows = 1; pitch = cols; }
    const int vec = pick_vec<T>(y, nullptr, cols, pitch, rows);
    const dim3 grid = ms_grid(rows, cols, vec);
    double* partial = reinterpret_cast<double*>(scratch);
    const double inv = 1.0 / ((double)rows * (double)cols);
    if (vec == 4)
        hipLaunchKernelGGL((mean_square_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, rows, cols,
                           pitch, partial);
    else
        hipLaunchKernelGGL((mean_square_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, rows, cols,
                           pitch, partial);
    FL_CHECK_LAUNCH("mean_square");
    hipLaunchKernelGGL((mean_square_final_kernel<T>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial,
                       (int)(grid.x * grid.y), inv, (T*)loss);
    FL_CHECK_LAUNCH("mean_square_final");
    return FL_OK;
}

template <typename T>
static int mean_square_bwd_impl(const void* y, const void* gloss, void* gy, long rows, long cols, long pitch, void* stream) {
    FL_REQUIRE(y && gloss && gy, "mean_square_bwd: null pointer");
    FL_REQUIRE(rows > 0 && cols > 0 && pitch >= cols, "mean_square_bwd: bad sizes");
    const double two_inv = 2.0 / ((double)rows * (double)cols);
    if (pitch == cols) { cols *= rows; rows = 1; pitch = cols; }
    const int vec = pick_vec<T>(y, gy, cols, pitch, rows);
    dim3 grid = ms_grid(rows, cols, vec);
    if (rows == 1) grid.x = (unsigned)((cols / vec + 1023) / 1024 < 1 ? 1 : (cols / vec + 1023) / 1024);  // 4 vectors per lane
    if (vec == 4)
        hipLaunchKernelGGL((mean_square_bwd_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y,
                           (const T*)gloss, two_inv, (T*)gy, rows, cols, pitch);
    else
        hipLaunchKernelGGL((mean_square_bwd_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y,
                           (const T*)gloss, two_inv, (T*)gy, rows, cols, pitch);
    FL_CHECK_LAUNCH("mean_square_bwd");
    return FL_OK;
}

// ---------------------------------------------------------------- mean squared error against a target
// loss = mean_r (sum_{c < ncols} y[r][c] - t[r])^2 over `rows` rows of `ncols` contiguous values: ncols = 1 is
// nn.MSELoss()(y, t) on equal shapes (examples/e7_biquad.py:82), ncols = N_out the reference's criterion
// flamo/optimize/loss.py:66-103 (`mse_loss.forward`: the prediction summed over its last axis, then nn.MSELoss against the
// squeezed target) -- one streaming pass each way (forward reads y and t; backward reads both and writes
// g_y[r][c] = (2 g / rows) (sum_c y[r][c] - t[r])) instead of torch's sum / sub / pow / mean and their four backward kernels.
template <typename T, int NC>
__global__ void __launch_bounds__(256) mse_kernel(const T* __restrict__ y, const T* __restrict__ t, long rows, int ncols,
                                                  double* __restrict__ partial) {
    double acc = 0.0;
    const long stride = (long)gridDim.x * 256;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += stride) {
        T s = (T)0;
        if constexpr (NC == 1) {
            s = y[r];
        } else if constexpr (NC == 4 || NC == 8 || NC == 16) {
#pragma unroll
            for (int q = 0; q < NC / 4; ++q) {
                const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(y + r * NC + 4 * q);
                s += (v.v[0] + v.v[1]) + (v.v[2] + v.v[3]);
            }
        } else {
            for (int c = 0; c < ncols; ++c) s += y[r * ncols + c];
        }
        const T d = s - t[r];
        acc += (double)(d * d);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T, int NC>
__global__ void __launch_bounds__(256) mse_bwd_kernel(const T* __restrict__ y, const T* __restrict__ t, const T* __restrict__ gloss,
                                                      double two_inv_rows, T* __restrict__ gy, long rows, int ncols) {
    const T k = (T)(two_inv_rows * (double)gloss[0]);
    const long stride = (long)gridDim.x * 256;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += stride) {
        T s = (T)0;
        if constexpr (NC == 1) {
            s = y[r];
        } else if constexpr (NC == 4 || NC == 8 || NC == 16) {
#pragma unroll
            for (int q = 0; q < NC / 4; ++q) {
                const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(y + r * NC + 4 * q);
                s += (v.v[0] + v.v[1]) + (v.v[2] + v.v[3]);
            }
        } else {
            for (int c = 0; c < ncols; ++c) s += y[r * ncols + c];
        }
        const T d = k * (s - t[r]);
        if constexpr (NC == 1) {
            gy[r] = d;
        } else if constexpr (NC == 4 || NC == 8 || NC == 16) {
            Vec4<T> o;
            o.v[0] = o.v[1] = o.v[2] = o.v[3] = d;
#pragma unroll
            for (int q = 0; q < NC / 4; ++q) *reinterpret_cast<Vec4<T>*>(gy + r * NC + 4 * q) = o;
        } else {
            for (int c = 0; c < ncols; ++c) gy[r * ncols + c] = d;
        }
    }
}

static int mse_blocks(long rows) {
    long g = (rows + 255) / 256;
    if (g > 2048) g = 2048;
    return (int)(g < 1 ? 1 : g);
}

template <typename T>
static int mse_impl(const void* y, const void* t, long rows, int ncols, void* loss, void* scratch, void* stream) {
    FL_REQUIRE(y && t && loss && scratch, "mse: null pointer");
    FL_REQUIRE(rows > 0 && ncols > 0 && ncols <= 4096, "mse: bad sizes");
    const int nblk = mse_blocks(rows);
    double* partial = reinterpret_cast<double*>(scratch);
    const bool al = (reinterpret_cast<uintptr_t>(y) % (4 * sizeof(T))) == 0;
#define FL_MSE(NC_)                                                                                                           \
    hipLaunchKernelGGL((mse_kernel<T, NC_>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)t, rows, ncols, partial)
    if (ncols == 1) FL_MSE(1);
    else if (ncols == 4 && al) FL_MSE(4);
    else if (ncols == 8 && al) FL_MSE(8);
    else if (ncols == 16 && al) FL_MSE(16);
    else FL_MSE(0);
#undef FL_MSE
    FL_CHECK_LAUNCH("mse");
    hipLaunchKernelGGL((mean_square_final_kernel<T>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial, nblk,
                       1.0 / (double)rows, (T*)loss);
    FL_CHECK_LAUNCH("mse_final");
    return FL_OK;
}

template <typename T>
static int mse_bwd_impl(const void* y, const void* t, const void* gloss, void* gy, long rows, int ncols, void* stream) {
    FL_REQUIRE(y && t && gloss && gy, "mse_bwd: null pointer");
    FL_REQUIRE(rows > 0 && ncols > 0 && ncols <= 4096, "mse_bwd: bad sizes");
    long g = (rows + 1023) / 1024;      // four rows per lane
    if (g < 1) g = 1;
    if (g > 65535) g = 65535;
    const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gy)) % (4 * sizeof(T))) == 0;
#define FL_MSEB(NC_)                                                                                                          \
    hipLaunchKernelGGL((mse_bwd_kernel<T, NC_>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)t,  \
                       (const T*)gloss, 2.0 / (double)rows, (T*)gy, rows, ncols)
    if (ncols == 1) FL_MSEB(1);
    else if (ncols == 4 && al) FL_MSEB(4);
    else if (ncols == 8 && al) FL_MSEB(8);
    else if (ncols == 16 && al) FL_MSEB(16);
    else FL_MSEB(0);
#undef FL_MSEB
    FL_CHECK_LAUNCH("mse_bwd");
    return FL_OK;
}

// ---------------------------------------------------------------- gradient buckets of a replayed step
// The parameter gradients of a captured training step, packed into ONE of two flat buffers -- alternately, chosen ON THE
// DEVICE from a counter the kernel itself advances, so that the packing is a node of the captured graph like any other
// (no host-side copy between two replays: on this runtime an eager launch between two graph launches costs ~10 us of
// idle device on either side) and the data-parallel all-reduce of step k can still be in flight on bucket k & 1 while
// replay k + 1 fills the other one (DistributedDataParallel's gradient-as-bucket-view, double-buffered).
//   table: count entries of three 64-bit words (source address, byte offset in the bucket, bytes; all multiples of 4)
//   state: [0] packs done so far (its parity chooses the bucket), [1] arrivals of the current launch
// One workgroup per entry; the last one to arrive -- every workgroup has read the parity before it arrives -- advances the
// counter and clears the arrivals for the next launch.
__global__ void __launch_bounds__(256) pack_toggle_kernel(const unsigned long long* __restrict__ table, int count, char* flat0,
                                                          char* flat1, int* state) {
    __shared__ int par;
    if (threadIdx.x == 0) par = __atomic_load_n(&state[0], __ATOMIC_RELAXED) & 1;
    __syncthreads();
    const unsigned long long* e = table + 3 * (size_t)blockIdx.x;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(e[0]);
    uint32_t* dst = reinterpret_cast<uint32_t*>((par ? flat1 : flat0) + e[1]);
    const size_t n = e[2] / 4;
    for (size_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&state[1], 1) == count - 1) {
            __atomic_store_n(&state[1], 0, __ATOMIC_RELAXED);
            __threadfence();
            atomicAdd(&state[0], 1);
        }
    }
}

// ---------------------------------------------------------------- sparsity criterion of a mixing matrix
// loss = mean_c (sum_ij |A_c[i][j]| - N sqrt(N)) / (N (1 - sqrt(N)))  over C matrices of (N, N) -- flamo/optimize/loss.py:12-63
// (`sparsity_loss`: the second criterion of the colorless-FDN training, examples/e8_colorless_fdn.py:138) in ONE launch each way
// instead of torch's abs / sum / sub / div / neg kernels and their backward: the matrix is 16 x 16, every one of those launches
// is pure latency inside a captured training step.  One workgroup; the sums are taken in a fixed order.
template <typename T>
__global__ void __launch_bounds__(256) sparsity_kernel(const T* __restrict__ A, int C, int N, T* __restrict__ loss) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double sn = sqrt((double)N), k = 1.0 / ((double)N * (1.0 - sn));
    double total = 0.0;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int i = threadIdx.x; i < N * N; i += 256) s += fabs((double)A[(size_t)c * N * N + i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        total += ((red[0] + red[1]) + (red[2] + red[3]) - (double)N * sn) * k;
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (T)(total / (double)C);
}

// g_A = gloss sign(A) / (C N (1 - sqrt(N)))   (sign(0) = 0, as torch.sign)
template <typename T>
__global__ void __launch_bounds__(256) sparsity_bwd_kernel(const T* __restrict__ A, const T* __restrict__ gloss, int C, int N,
                                                           T* __restrict__ gA) {
    const double sn = sqrt((double)N);
    const T k = (T)((double)gloss[0] / ((double)C * (double)N * (1.0 - sn)));
    const size_t n = (size_t)C * N * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T a = A[i];
        gA[i] = a > (T)0 ? k : (a < (T)0 ? -k : (a == (T)0 ? (T)0 : a * k));      // (a NaN entry stays NaN)
    }
}

template <typename T>
static int sparsity_impl(const void* A, int C, int N, void* loss, void* stream) {
    FL_REQUIRE(A && loss, "sparsity: null pointer");
    FL_REQUIRE(C >= 1 && N >= 2, "sparsity: at least one matrix of at least 2 x 2");
    hipLaunchKernelGGL((sparsity_kernel<T>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const T*)A, C, N, (T*)loss);
    FL_CHECK_LAUNCH("sparsity");
    return FL_OK;
}
template <typename T>
static int sparsity_bwd_impl(const void* A, const void* gloss, int C, int N, void* gA, void* stream) {
    FL_REQUIRE(A && gloss && gA, "sparsity_bwd: null pointer");
    FL_REQUIRE(C >= 1 && N >= 2, "sparsity_bwd: at least one matrix of at least 2 x 2");
    const size_t n = (size_t)C * N * N;
    size_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL((sparsity_bwd_kernel<T>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const T*)A, (const T*)gloss, C, N,
                       (T*)gA);
    FL_CHECK_LAUNCH("sparsity_bwd");
    return FL_OK;
}

// ---------------------------------------------------------------- magnitude of a spectrum
// out = |z| and its backward g_z = g z / |z| (0 where z = 0: torch's sgn) over rows of `cols` contiguous complex values `pitch`
// apart -- the output layer of the reference's magnitude-domain examples, dsp.Transform(lambda x: torch.abs(x))
// (examples/e7_biquad.py:76, e8_colorless_fdn.py:102), one launch each way where torch runs abs, then sgn and a complex multiply.
template <typename T>
__global__ void __launch_bounds__(256) cabs_kernel(const cx<T>* __restrict__ z, T* __restrict__ out, long rows, long cols, long pitch,
                                                   long opitch) {
    for (long r = blockIdx.y; r < rows; r += gridDim.y)
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long)gridDim.x * 256) {
            const cx<T> v = z[r * pitch + c];
            out[r * opitch + c] = (T)hypot(v.x, v.y);
        }
}
template <typename T>
__global__ void __launch_bounds__(256) cabs_bwd_kernel(const cx<T>* __restrict__ z, const T* __restrict__ g, cx<T>* __restrict__ gz,
                                                       long rows, long cols, long pitch, long gpitch) {
    for (long r = blockIdx.y; r < rows; r += gridDim.y)
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long)gridDim.x * 256) {
            const cx<T> v = z[r * pitch + c];
            const T m = (T)hypot(v.x, v.y), gg = g[r * gpitch + c];
            gz[r * pitch + c] = m > (T)0 ? cx<T>(gg * (v.x / m), gg * (v.y / m)) : cx<T>((T)0 * gg, (T)0 * gg);
        }
}
static dim3 cabs_grid(long rows, long cols) {
    long gx = (cols + 1023) / 1024;      // four values per lane
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    long gy = rows;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}
template <typename T>
static int cabs_impl(const void* z, void* out, long rows, long cols, long pitch, long opitch, void* stream) {
    FL_REQUIRE(z && out, "cabs: null pointer");
    FL_REQUIRE(rows > 0 && cols > 0 && pitch >= cols && opitch >= cols, "cabs: bad sizes");
    hipLaunchKernelGGL((cabs_kernel<T>), cabs_grid(rows, cols), dim3(256), 0, (hipStream_t)stream, (const cx<T>*)z, (T*)out, rows, cols,
                       pitch, opitch);
    FL_CHECK_LAUNCH("cabs");
    return FL_OK;
}
template <typename T>
static int cabs_bwd_impl(const void* z, const void* g, void* gz, long rows, long cols, long pitch, long gpitch, void* stream) {
    FL_REQUIRE(z && g && gz, "cabs_bwd: null pointer");
    FL_REQUIRE(rows > 0 && cols > 0 && pitch >= cols && gpitch >= cols, "cabs_bwd: bad sizes");
    hipLaunchKernelGGL((cabs_bwd_kernel<T>), cabs_grid(rows, cols), dim3(256), 0, (hipStream_t)stream, (const cx<T>*)z, (const T*)g,
                       (cx<T>*)gz, rows, cols, pitch, gpitch);
    FL_CHECK_LAUNCH("cabs_bwd");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {
size_t fl_mean_square_scratch_bytes(void) { return MS_MAX_BLOCKS * sizeof(double); }
int fl_mean_square_f32(const void* y, long rows, long cols, long pitch, void* loss, void* scratch, void* stream) {
    return mean_square_impl<float>(y, rows, cols, pitch, loss, scratch, stream);
}
int fl_mean_square_f64(const void* y, long rows, long cols, long pitch, void* loss, void* scratch, void* stream) {
    return mean_square_impl<double>(y, rows, cols, pitch, loss, scratch, stream);
}
int fl_mean_square_final_f32(const void* parts, int n_parts, double inv_count, void* loss, void* stream) {
    FL_REQUIRE(parts && loss && n_parts > 0, "mean_square_final: bad arguments");
    hipLaunchKernelGGL((mean_square_final_kernel<float>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)parts, n_parts, inv_count, (float*)loss);
    FL_CHECK_LAUNCH("mean_square_final");
    return FL_OK;
}
int fl_mean_square_final_f64(const void* parts, int n_parts, double inv_count, void* loss, void* stream) {
    FL_REQUIRE(parts && loss && n_parts > 0, "mean_square_final: bad arguments");
    hipLaunchKernelGGL((mean_square_final_kernel<double>), dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)parts, n_parts, inv_count, (double*)loss);
    FL_CHECK_LAUNCH("mean_square_final");
    return FL_OK;
}
int fl_mse_f32(const void* y, const void* t, long rows, int ncols, void* loss, void* scratch, void* stream) {
    return mse_impl<float>(y, t, rows, ncols, loss, scratch, stream);
}
int fl_mse_f64(const void* y, const void* t, long rows, int ncols, void* loss, void* scratch, void* stream) {
    return mse_impl<double>(y, t, rows, ncols, loss, scratch, stream);
}
int fl_mse_bwd_f32(const void* y, const void* t, const void* gloss, void* gy, long rows, int ncols, void* stream) {
    return mse_bwd_impl<float>(y, t, gloss, gy, rows, ncols, stream);
}
int fl_mse_bwd_f64(const void* y, const void* t, const void* gloss, void* gy, long rows, int ncols, void* stream) {
    return mse_bwd_impl<double>(y, t, gloss, gy, rows, ncols, stream);
}
int fl_pack_toggle(const void* table, int count, void* flat0, void* flat1, void* state, void* stream) {
    FL_REQUIRE(table && flat0 && flat1 && state && count > 0 && count <= 65535, "pack_toggle: bad arguments");
    hipLaunchKernelGGL(pack_toggle_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)table, count,
                       (char*)flat0, (char*)flat1, (int*)state);
    FL_CHECK_LAUNCH("pack_toggle");
    return FL_OK;
}
int fl_mean_square_bwd_f32(const void* y, const void* gloss, void* gy, long rows, long cols, long pitch, void* stream) {
    return mean_square_bwd_impl<float>(y, gloss, gy, rows, cols, pitch, stream);
}
int fl_mean_square_bwd_f64(const void* y, const void* gloss, void* gy, long rows, long cols, long pitch, void* stream) {
    return mean_square_bwd_impl<double>(y, gloss, gy, rows, cols, pitch, stream);
}
int fl_sparsity_f32(const void* A, int C, int N, void* loss, void* stream) { return sparsity_impl<float>(A, C, N, loss, stream); }
int fl_sparsity_f64(const void* A, int C, int N, void* loss, void* stream) { return sparsity_impl<double>(A, C, N, loss, stream); }
int fl_sparsity_bwd_f32(const void* A, const void* gloss, int C, int N, void* gA, void* stream) {
    return sparsity_bwd_impl<float>(A, gloss, C, N, gA, stream);
}
int fl_sparsity_bwd_f64(const void* A, const void* gloss, int C, int N, void* gA, void* stream) {
    return sparsity_bwd_impl<double>(A, gloss, C, N, gA, stream);
}
int fl_cabs_c64(const void* z, void* out, long rows, long cols, long pitch, long opitch, void* stream) {
    return cabs_impl<float>(z, out, rows, cols, pitch, opitch, stream);
}
int fl_cabs_c128(const void* z, void* out, long rows, long cols, long pitch, long opitch, void* stream) {
    return cabs_impl<double>(z, out, rows, cols, pitch, opitch, stream);
}
int fl_cabs_bwd_c64(const void* z, const void* g, void* gz, long rows, long cols, long pitch, long gpitch, void* stream) {
    return cabs_bwd_impl<float>(z, g, gz, rows, cols, pitch, gpitch, stream);
}
int fl_cabs_bwd_c128(const void* z, const void* g, void* gz, long rows, long cols, long pitch, long gpitch, void* stream) {
    return cabs_bwd_impl<double>(z, g, gz, rows, cols, pitch, gpitch, stream);
}
}
