// Per-bin eigenvalues (and right eigenvectors) of small complex matrices (gfx950 / MI355X).
//
//   A_f v = lambda v,   A_f = A[:, :, f]  (N x N, general complex),  one matrix per frequency bin
//
// replacing torch.linalg.eigvals in flamo.functional.get_eigenvalues (flamo/functional.py:24-39), the
// first thing the active-acoustics training loop does with the output of the path
// (examples/e8_active_acoustics.py:586-603: the loss is taken on |eig| of the (B, F, n_M, n_M) loop
// matrix at a subset of the bins).  SURVEY 8-f4: the next bin-parallel kernel after the solve.
//
// One wavefront per matrix, the matrix (H), the accumulated unitary (Q) and the eigenvector scratch
// in LDS.  Classical dense path, every loop with wavefront-uniform control flow:
//   1. Householder reduction to upper Hessenberg form, Q accumulated;
//   2. explicitly shifted QR sweeps (Wilkinson shift, exceptional shifts at iterations 10 and 20,
//      deflation on |h(k,k-1)| <= eps (|h(k-1,k-1)| + |h(k,k)|)) with Givens rotations applied to the
//      whole rows / columns, so that H ends as the Schur form T = Q^H A Q;
//   3. eigenvalues = diag T; eigenvectors by back substitution in T, one vector per lane, V = Q Y.
// A rotation's row update is parallel over columns (lane = column), its column update over rows
// (lane = row); the chain of rotations itself is sequential -- this kernel is latency bound by
// design, its parallelism is across bins (thousands of matrices in flight).
#include "common.h"

namespace fl {

template <typename T> __device__ inline T cabs1(cx<T> a) { return fabs(a.x) + fabs(a.y); }
template <typename T> __device__ inline T cabs2(cx<T> a) { return a.x * a.x + a.y * a.y; }
template <typename T> __device__ inline T cabsv(cx<T> a) { return sqrt(a.x * a.x + a.y * a.y); }
template <typename T> __device__ inline cx<T> csqrt_(cx<T> a) {
    const T r = cabsv(a);
    if (r == (T)0) return cx<T>(0, 0);
    T re = sqrt((T)0.5 * (r + fabs(a.x)));
    T im = a.y / ((T)2 * re);
    if (a.x < 0) {   // principal branch
        const T t = re;
        re = fabs(im);
        im = (a.y < 0) ? -t : t;
    }
    return cx<T>(re, im);
}
template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; static constexpr float tiny = 1e-30f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; static constexpr double tiny = 1e-290; };

template <typename T>
__global__ void __launch_bounds__(64) eig_kernel(const cx<T>* __restrict__ A, long a_pitch, int N, int M,
                                                cx<T>* __restrict__ lam, long l_pitch, cx<T>* __restrict__ V,
                                                long v_pitch, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = N + 1;
    cx<T>* H = reinterpret_cast<cx<T>*>(smem);   // [N][NP]
    cx<T>* Q = H + N * NP;                        // [N][NP]
    cx<T>* Y = Q + N * NP;                        // [N][NP] eigenvectors of T (only when V != nullptr)
    cx<T>* vv = Y + (V ? N * NP : 0);             // [N] Householder vector / rotation sines
    T* rc = reinterpret_cast<T*>(vv + N);         // [N] rotation cosines
    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= M) return;
    const T eps = Eps<T>::v;

    for (int idx = lane; idx < N * N; idx += 64) {
        const int i = idx / N, j = idx - i * N;
        H[i * NP + j] = A[(size_t)idx * a_pitch + f];
        Q[i * NP + j] = cx<T>(i == j ? (T)1 : (T)0, 0);
    }
    __syncthreads();

    // ---- 1. Hessenberg reduction: H <- P_k H P_k, Q <- Q P_k, P_k = I - beta v v^H
    for (int k = 0; k + 2 < N; ++k) {
        const bool in = lane > k && lane < N;
        const cx<T> xi = in ? H[lane * NP + k] : cx<T>(0, 0);
        const T nrm2 = wave_sum(cabs2(xi));
        const cx<T> x0 = H[(k + 1) * NP + k];
        const T tail2 = nrm2 - cabs2(x0);
        if (!(tail2 > Eps<T>::tiny)) continue;           // column already in Hessenberg form (uniform)
        const T nrm = sqrt(nrm2), a0 = cabsv(x0);
        const cx<T> ph = (a0 > 0) ? cx<T>(x0.x / a0, x0.y / a0) : cx<T>(1, 0);
        const cx<T> alpha(-ph.x * nrm, -ph.y * nrm);     // x -> alpha e1
        const cx<T> v0 = x0 - alpha;
        const T beta = (T)2 / (tail2 + cabs2(v0));
        __syncthreads();
        if (in) vv[lane] = (lane == k + 1) ? v0 : xi;
        __syncthreads();
        // left: column j (lane j): w = sum_i conj(v_i) H[i][j];  H[i][j] -= beta v_i w
        if (lane < N) {
            cx<T> w(0, 0);
            for (int i = k + 1; i < N; ++i) fma_cx(w, conj(vv[i]), H[i * NP + lane]);
            w = cx<T>(beta * w.x, beta * w.y);
            for (int i = k + 1; i < N; ++i) H[i * NP + lane] = H[i * NP + lane] - vv[i] * w;
        }
        __syncthreads();
        // right: row i (lane i): w = sum_j H[i][j] v_j;  H[i][j] -= beta w conj(v_j);  the same on Q
        if (lane < N) {
            cx<T> w(0, 0), wq(0, 0);
            for (int j = k + 1; j < N; ++j) {
                fma_cx(w, H[lane * NP + j], vv[j]);
                fma_cx(wq, Q[lane * NP + j], vv[j]);
            }
            w = cx<T>(beta * w.x, beta * w.y);
            wq = cx<T>(beta * wq.x, beta * wq.y);
            for (int j = k + 1; j < N; ++j) {
                const cx<T> vc = conj(vv[j]);
                H[lane * NP + j] = H[lane * NP + j] - w * vc;
                Q[lane * NP + j] = Q[lane * NP + j] - wq * vc;
            }
        }
        __syncthreads();
        if (in && lane > k + 1) H[lane * NP + k] = cx<T>(0, 0);   // annihilated entries: exactly zero
        if (lane == k + 1) H[lane * NP + k] = alpha;
        __syncthreads();
    }

    // ---- 2. shifted QR on the Hessenberg matrix -> Schur form
    int hi = N - 1, iter = 0, total = 0, fail = 0;
    while (hi >= 0) {
        int lo = hi;
        while (lo > 0) {   // every lane scans the same entries: uniform
            const T s = cabs1(H[(lo - 1) * NP + lo - 1]) + cabs1(H[lo * NP + lo]);
            if (cabs1(H[lo * NP + lo - 1]) <= eps * s) break;
            --lo;
        }
        if (lo > 0) {
            __syncthreads();
            if (lane == 0) H[lo * NP + lo - 1] = cx<T>(0, 0);
            __syncthreads();
        }
        if (lo == hi) {
            --hi;
            iter = 0;
            continue;
        }
        if (++total > 60 * N) {
            fail = hi + 1;
            break;
        }
        ++iter;
        // Wilkinson shift: the eigenvalue of the trailing 2x2 closer to its last diagonal entry
        const cx<T> a = H[(hi - 1) * NP + hi - 1], b = H[(hi - 1) * NP + hi], c = H[hi * NP + hi - 1], d = H[hi * NP + hi];
        cx<T> mu;
        if (iter == 10 || iter == 20) {
            const T e = fabs(H[hi * NP + hi - 1].x) + ((hi >= 2) ? fabs(H[(hi - 1) * NP + hi - 2].x) : (T)0);
            mu = cx<T>(d.x + e, d.y);
        } else {
            const cx<T> hd = (T)0.5 * (a - d);
            const cx<T> disc = csqrt_(hd * hd + b * c);
            const cx<T> m1 = (T)0.5 * (a + d) + disc, m2 = (T)0.5 * (a + d) - disc;
            mu = (cabs2(m1 - d) <= cabs2(m2 - d)) ? m1 : m2;
        }
        __syncthreads();
        if (lane >= lo && lane <= hi) H[lane * NP + lane] = H[lane * NP + lane] - mu;
        __syncthreads();
        // R = G_{hi-1} ... G_lo (H - mu I): rotation k acts on rows k, k+1, columns k .. N-1 (lane = column)
        for (int k = lo; k < hi; ++k) {
            const cx<T> x = H[k * NP + k], y = H[(k + 1) * NP + k];
            const T ax = cabsv(x), r = sqrt(cabs2(x) + cabs2(y));
            T cs;
            cx<T> sn;
            if (!(r > 0)) {
                cs = 1;
                sn = cx<T>(0, 0);
            } else if (!(ax > 0)) {
                cs = 0;
                const T ay = cabsv(y);
                sn = cx<T>(y.x / ay, -y.y / ay);                 // conj(y)/|y|
            } else {
                cs = ax / r;
                const cx<T> px(x.x / ax, x.y / ax);              // x/|x|
                sn = px * cx<T>(y.x / r, -y.y / r);              // (x/|x|) conj(y) / r
            }
            __syncthreads();
            if (lane == 0) {
                rc[k] = cs;
                vv[k] = sn;
            }
            if (lane >= k && lane < N) {
                const cx<T> t1 = H[k * NP + lane], t2 = H[(k + 1) * NP + lane];
                H[k * NP + lane] = cs * t1 + sn * t2;
                H[(k + 1) * NP + lane] = cs * t2 - conj(sn) * t1;
            }
            __syncthreads();
        }
        // H <- R G_lo^H ... G_{hi-1}^H + mu I, Q <- Q G^H: rotation k acts on columns k, k+1 (lane = row);
        // a lane only touches its own row, so the chain needs no barrier
        if (lane < N) {
            for (int k = lo; k < hi; ++k) {
                const T cs = rc[k];
                const cx<T> sn = vv[k];
                if (lane <= k + 1) {
                    const cx<T> t1 = H[lane * NP + k], t2 = H[lane * NP + k + 1];
                    H[lane * NP + k] = cs * t1 + conj(sn) * t2;
                    H[lane * NP + k + 1] = cs * t2 - sn * t1;
                }
                const cx<T> q1 = Q[lane * NP + k], q2 = Q[lane * NP + k + 1];
                Q[lane * NP + k] = cs * q1 + conj(sn) * q2;
                Q[lane * NP + k + 1] = cs * q2 - sn * q1;
            }
        }
        __syncthreads();
        if (lane >= lo && lane <= hi) H[lane * NP + lane] = H[lane * NP + lane] + mu;
        __syncthreads();
    }
    if (lane == 0 && info) info[f] = fail;
    if (lane < N) lam[(size_t)lane * l_pitch + f] = H[lane * NP + lane];
    if (!V) return;

    // ---- 3. eigenvectors of T by back substitution (lane i: vector i), then V = Q Y
    T tnorm = 0;
    if (lane < N)
        for (int j = lane; j < N; ++j) tnorm += cabs1(H[lane * NP + j]);
    tnorm = wave_sum(tnorm);
    const T smin = fmax(eps * tnorm, Eps<T>::tiny);
    if (lane < N) {
        const int i = lane;
        const cx<T> li = H[i * NP + i];
        for (int j = N - 1; j > i; --j) Y[j * NP + i] = cx<T>(0, 0);
        Y[i * NP + i] = cx<T>(1, 0);
        for (int j = i - 1; j >= 0; --j) {
            cx<T> sacc(0, 0);
            for (int k = j + 1; k <= i; ++k) fma_cx(sacc, H[j * NP + k], Y[k * NP + i]);
            cx<T> den = H[j * NP + j] - li;
            if (cabs1(den) < smin) den = cx<T>(smin, 0);       // repeated eigenvalue: perturb like LAPACK's trevc
            Y[j * NP + i] = cdiv(cx<T>(-sacc.x, -sacc.y), den);
        }
    }
    __syncthreads();
    if (lane < N) {
        const int j = lane;                                     // column j of V = Q Y(:, j), Y upper triangular
        T n2 = 0;
        for (int i = 0; i < N; ++i) {
            cx<T> sacc(0, 0);
            for (int k = 0; k <= j; ++k) fma_cx(sacc, Q[i * NP + k], Y[k * NP + j]);
            H[i * NP + j] = sacc;                               // T is no longer needed: reuse H for V
            n2 += cabs2(sacc);
        }
        const T inv = (n2 > 0) ? (T)1 / sqrt(n2) : (T)1;       // unit 2-norm columns
        for (int i = 0; i < N; ++i) {
            const cx<T> v = H[i * NP + j];
            V[((size_t)i * N + j) * v_pitch + f] = cx<T>(v.x * inv, v.y * inv);
        }
    }
}

template <typename T>
static int eig_impl(const void* A, long a_pitch, int N, int M, void* lam, long l_pitch, void* V, long v_pitch, void* info,
                    void* stream) {
    FL_REQUIRE(A && lam, "eig: null pointer");
    FL_REQUIRE(N >= 1 && N <= 64 && M >= 0 && a_pitch >= M && l_pitch >= M && (!V || v_pitch >= M), "eig: bad sizes (1 <= N <= 64, pitches >= M)");
    if (M == 0) return FL_OK;
    const size_t lds = ((size_t)(V ? 3 : 2) * N * (N + 1) + N) * sizeof(cx<T>) + (size_t)N * sizeof(T);
    if (lds > 160 * 1024) {
        set_error("eig: N=%d does not fit in LDS at this precision", N);
        return FL_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&eig_kernel<T>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "eig LDS size");
        if (rc) return rc;
    }
    hipLaunchKernelGGL((eig_kernel<T>), dim3(M), dim3(64), lds, (hipStream_t)stream, (const cx<T>*)A, a_pitch, N, M,
                       (cx<T>*)lam, l_pitch, (cx<T>*)V, v_pitch, (int*)info);
    FL_CHECK_LAUNCH("eig");
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {
int fl_eig_c64(const void* A, long a_pitch, int N, int M, void* lam, long l_pitch, void* V, long v_pitch, void* info,
               void* stream) {
    return eig_impl<float>(A, a_pitch, N, M, lam, l_pitch, V, v_pitch, info, stream);
}
int fl_eig_c128(const void* A, long a_pitch, int N, int M, void* lam, long l_pitch, void* V, long v_pitch, void* info,
                void* stream) {
    return eig_impl<double>(A, a_pitch, N, M, lam, l_pitch, V, v_pitch, info, stream);
}
}
