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

}  // namespace fl
