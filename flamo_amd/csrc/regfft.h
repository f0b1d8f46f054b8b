// In-register butterflies and fully unrolled mixed-radix FFTs shared by the transform kernels (gfx950).
#pragma once
#include "common.h"

namespace fl {

// ---------------------------------------------------------------- radix butterflies (in registers)
// PK: the packed-asm forms of common.h (float only).  A kernel can opt out (spec_cols_inv does: with the asm forms its
// schedule changes and it runs 40 % slower -- measured, both for the butterflies alone and for the products alone).
template <typename T, int R, bool INV, bool PK = true>
struct Bfly;

template <typename T, bool INV, bool PK>
struct Bfly<T, 2, INV, PK> {
    static __device__ inline void run(cx<T>* v, const cx<T>*, int) {
        cx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <typename T, bool INV, bool PK>
struct Bfly<T, 4, INV, PK> {
    static __device__ inline void run(cx<T>* v, const cx<T>*, int) {
        cx<T> t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], d = v[1] - v[3];
#ifdef FL_PK_ASM
        if constexpr (PK && sizeof(T) == 4) {       // t1 +- i d as one packed add each (common.h)
            v[0] = t0 + t2;
            v[2] = t0 - t2;
            v[1] = c2(INV ? pk_add_i(v2(t1), v2(d)) : pk_sub_i(v2(t1), v2(d)));
            v[3] = c2(INV ? pk_sub_i(v2(t1), v2(d)) : pk_add_i(v2(t1), v2(d)));
            return;
        }
#endif
        cx<T> t3 = INV ? mul_i(d) : mul_mi(d);
        v[0] = t0 + t2;
        v[1] = t1 + t3;
        v[2] = t0 - t2;
        v[3] = t1 - t3;
    }
};

template <typename T, bool INV, bool PK>
struct Bfly<T, 3, INV, PK> {
    static __device__ inline void run(cx<T>* v, const cx<T>*, int) {
        const T h = (T)0.86602540378443864676;  // sin(2pi/3)
        const cx<T> t = v[1] + v[2];
        const cx<T> u = axpy((T)-0.5, t, v[0]);
        const cx<T> d = v[1] - v[2];
#ifdef FL_PK_ASM
        if constexpr (PK && sizeof(T) == 4) {       // u +- i h d as one packed multiply-add each
            v[0] = v[0] + t;
            v[1] = c2(INV ? pk_fma_i((float)h, v2(d), v2(u)) : pk_fms_i((float)h, v2(d), v2(u)));
            v[2] = c2(INV ? pk_fms_i((float)h, v2(d), v2(u)) : pk_fma_i((float)h, v2(d), v2(u)));
            return;
        }
#endif
        const cx<T> w = h * (INV ? mul_i(d) : mul_mi(d));
        v[0] = v[0] + t;
        v[1] = u + w;
        v[2] = u - w;
    }
};

template <typename T, bool INV, bool PK>
struct Bfly<T, 5, INV, PK> {
    static __device__ inline void run(cx<T>* v, const cx<T>*, int) {
        const T k1c = (T)0.30901699437494742410;   // cos(2pi/5)
        const T k2c = (T)-0.80901699437494742410;  // cos(4pi/5)
        const T s1 = (T)0.95105651629515357212;   // sin(2pi/5)
        const T s2 = (T)0.58778525229247312917;   // sin(4pi/5)
        const cx<T> a1 = v[1] + v[4], a2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
        const cx<T> m1 = axpy(k2c, a2, axpy(k1c, a1, v[0]));
        const cx<T> m2 = axpy(k1c, a2, axpy(k2c, a1, v[0]));
        const cx<T> e1 = axpy(s2, d2, s1 * d1);
        const cx<T> e2 = axpy(-s1, d2, s2 * d1);
#ifdef FL_PK_ASM
        if constexpr (PK && sizeof(T) == 4) {       // m +- i e as one packed add each
            v[0] = v[0] + a1 + a2;
            v[1] = c2(INV ? pk_add_i(v2(m1), v2(e1)) : pk_sub_i(v2(m1), v2(e1)));
            v[4] = c2(INV ? pk_sub_i(v2(m1), v2(e1)) : pk_add_i(v2(m1), v2(e1)));
            v[2] = c2(INV ? pk_add_i(v2(m2), v2(e2)) : pk_sub_i(v2(m2), v2(e2)));
            v[3] = c2(INV ? pk_sub_i(v2(m2), v2(e2)) : pk_add_i(v2(m2), v2(e2)));
            return;
        }
#endif
        const cx<T> j1 = INV ? mul_i(e1) : mul_mi(e1);
        const cx<T> j2 = INV ? mul_i(e2) : mul_mi(e2);
        v[0] = v[0] + a1 + a2;
        v[1] = m1 + j1;
        v[4] = m1 - j1;
        v[2] = m2 + j2;
        v[3] = m2 - j2;
    }
};

// generic odd prime radix: direct O(R^2) DFT with w_R^j = tw[j * step] (forward table)
template <typename T, int R, bool INV, bool PK>
struct Bfly {
    static __device__ inline void run(cx<T>* v, const cx<T>* tw, int step) {
        cx<T> o[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            cx<T> acc = v[0];
#pragma unroll
            for (int j = 1; j < R; ++j) {
                cx<T> w = tw[((j * k) % R) * step];
                if (INV) w = conj(w);
                fma_cx(acc, v[j], w);
            }
            o[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = o[k];
    }
};

constexpr double kPi = 3.141592653589793238462643383279502884;

constexpr double c_sin_small(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int k = 1; k < 14; ++k) {
        term *= -x2 / ((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double c_cos_small(double x) {
    double x2 = x * x, term = 1, sum = 1;
    for (int k = 1; k < 14; ++k) {
        term *= -x2 / ((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
// cos / sin of 2*pi*m/R with exact octant reduction
constexpr double c_cos2pi(int m, int R) {
    m %= R;
    if (m < 0) m += R;
    // fold to [0, R/2]: cos(2pi m/R) = cos(2pi (R-m)/R)
    if (2 * m > R) m = R - m;
    // now angle in [0, pi]; cos(pi - x) = -cos x
    bool neg = false;
    if (4 * m > R) { m = R - 2 * m; neg = true; /* angle' = pi - angle = pi*(R-2m)/R -> use half-angle form below */
        // angle' = pi * m' / R with m' = R - 2m_old ; handle by separate formula
        double x = kPi * (double)m / (double)R;           // in [0, pi/2)
        double v = (x <= kPi / 4) ? c_cos_small(x) : c_sin_small(kPi / 2 - x);
        return -v;
    }
    (void)neg;
    double x = 2 * kPi * (double)m / (double)R;           // in [0, pi/2]
    return (x <= kPi / 4) ? c_cos_small(x) : c_sin_small(kPi / 2 - x);
}
constexpr double c_sin2pi(int m, int R) {
    m %= R;
    if (m < 0) m += R;
    bool neg = false;
    if (2 * m > R) { m = R - m; neg = true; }              // sin(2pi - x) = -sin x
    double v = 0;
    if (4 * m > R) {                                       // angle in (pi/2, pi]: sin(pi - x)
        double x = kPi * (double)(R - 2 * m) / (double)R;  // pi - angle, in [0, pi/2)
        v = (x <= kPi / 4) ? c_sin_small(x) : c_cos_small(kPi / 2 - x);
    } else {
        double x = 2 * kPi * (double)m / (double)R;
        v = (x <= kPi / 4) ? c_sin_small(x) : c_cos_small(kPi / 2 - x);
    }
    return neg ? -v : v;
}

template <int R>
struct TwTab {
    double re[R], im[R];  // W_R^m = exp(-2 pi i m / R)
    constexpr TwTab() : re{}, im{} {
        for (int m = 0; m < R; ++m) {
            re[m] = c_cos2pi(m, R);
            im[m] = -c_sin2pi(m, R);
        }
    }
};

constexpr bool is_base_radix(int R) { return R == 2 || R == 3 || R == 4 || R == 5 || R == 7 || R == 11 || R == 13; }
constexpr int first_factor(int R) {
    if (R % 4 == 0) return 4;
    if (R % 2 == 0) return 2;
    if (R % 3 == 0) return 3;
    if (R % 5 == 0) return 5;
    if (R % 7 == 0) return 7;
    if (R % 11 == 0) return 11;
    return 13;
}

// Natural-order in-register FFT of size R: v[k] <- sum_t v[t] W_R^(+-tk)
template <typename T, int R, bool INV, bool PK = true>
struct RegFFT {
    static __device__ __forceinline__ void run(cx<T> (&v)[R]) {
        if constexpr (R == 1) {
            return;
        } else if constexpr (R == 7 || R == 11 || R == 13) {
            constexpr TwTab<R> tw = TwTab<R>();
            cx<T> o[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                cx<T> acc = v[0];
#pragma unroll
                for (int j = 1; j < R; ++j) {
                    const int m = (j * k) % R;
                    const cx<T> w((T)tw.re[m], INV ? (T)(-tw.im[m]) : (T)tw.im[m]);
                    fma_cx(acc, v[j], w);
                }
                o[k] = acc;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = o[k];
        } else if constexpr (is_base_radix(R)) {
            Bfly<T, R, INV, PK>::run(v, nullptr, 0);
        } else {
            constexpr int R1 = first_factor(R), R2 = R / R1;
            constexpr TwTab<R> tw = TwTab<R>();
            cx<T> w[R];
#pragma unroll
            for (int t2 = 0; t2 < R2; ++t2) {
                cx<T> sub[R1];
#pragma unroll
                for (int t1 = 0; t1 < R1; ++t1) sub[t1] = v[t1 * R2 + t2];
                RegFFT<T, R1, INV, PK>::run(sub);
#pragma unroll
                for (int k1 = 0; k1 < R1; ++k1) {
                    const int m = (t2 * k1) % R;
                    if (m == 0) {
                        w[k1 * R2 + t2] = sub[k1];
                    } else {
#ifdef FL_PK_ASM
                        if constexpr (PK && sizeof(T) == 4) {
                            // a compile-time twiddle: a quarter turn is one packed multiply by (+-1, -+1) with the halves swapped,
                            // everything else the two-instruction product with the factor in an SGPR pair
                            if ((4 * m) % R == 0) {
                                const int q = ((4 * m) / R) % 4;          // W^m = (-i)^q forward, (+i)^q inverse
                                if (q == 2) w[k1 * R2 + t2] = c2(-v2(sub[k1]));
                                else w[k1 * R2 + t2] = c2(pk_rot_s(v2(sub[k1]), ((q == 1) != INV) ? f2{1.f, -1.f} : f2{-1.f, 1.f}));
                            } else {
                                w[k1 * R2 + t2] = c2(pk_cmul_s(v2(sub[k1]), f2{(float)tw.re[m], INV ? (float)(-tw.im[m]) : (float)tw.im[m]}));
                            }
                            continue;
                        }
#endif
                        const cx<T> tf((T)tw.re[m], INV ? (T)(-tw.im[m]) : (T)tw.im[m]);
                        w[k1 * R2 + t2] = mul_plain(sub[k1], tf);
                    }
                }
            }
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                cx<T> sub[R2];
#pragma unroll
                for (int t2 = 0; t2 < R2; ++t2) sub[t2] = w[k1 * R2 + t2];
                RegFFT<T, R2, INV, PK>::run(sub);
#pragma unroll
                for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = sub[k2];
            }
        }
    }
};

}  // namespace fl
