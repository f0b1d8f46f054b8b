// The input's column pass and the cascade response in ONE grid (see fusedfwd.h).  gfx950 only.
//
// Workgroup i of T = n_cols + n_rc takes the response role when floor((i + 1) n_rc / T) > floor(i n_rc / T) -- the response's
// workgroups are spread evenly through the grid, so that every CU holds both kinds for the whole launch: the response's packed
// multiply-adds issue while the column pass's wavefronts wait for HBM.  Both bodies are the device functions the plain kernels
// run (spec_cols_body.h, rc_ba_body.h): same arithmetic, same results bit for bit.
#include "spectral_common.h"
#include "spec_cols_body.h"
#include "fusedfwd.h"

namespace fl {
using namespace sp32;

// n_mix: the response's workgroups are spread evenly over the FIRST n_mix workgroups of the grid (n_rc <= n_mix <= n_cols +
// n_rc); the rest are column-pass workgroups.  n_mix = n_cols + n_rc is the even spread; a smaller value puts more of the
// response's wavefronts on every SIMD while they last (one response workgroup per CU is one wavefront per SIMD: its dependent
// packed chains then issue at a fraction of the rate).
template <int A, int B, int VT, int RG, bool PLAIN, int NIW>
__global__ void __launch_bounds__(256, 5) cols_fwd_rc_kernel(ColsArgs a, RcBaArgs r, int n_mix, int n_rc, int rc_gx, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int i = blockIdx.x;
    int role = 0;
    if (dbg && threadIdx.x == 0) dbg[4 * (size_t)i] = (long long)__builtin_amdgcn_s_memrealtime();
    if (i >= n_mix) {
        spec_cols_fwd_body<A, B, VT, RG, PLAIN>(a, i - n_rc, smem);
    } else {
        const int before = (int)(((long)i * n_rc) / n_mix), after = (int)(((long)(i + 1) * n_rc) / n_mix);
        if (after > before) {
            role = 1;
            rc_ba_body<NIW>(r, before % rc_gx, before / rc_gx, smem);
        } else {
            spec_cols_fwd_body<A, B, VT, RG, PLAIN>(a, i - before, smem);
        }
    }
    if (dbg && threadIdx.x == 0) {      // (tuning, fl_debug_set_pair_stamps: start, end, role, HW_ID | XCC_ID << 32 of every workgroup)
        dbg[4 * (size_t)i + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        dbg[4 * (size_t)i + 2] = role;
        dbg[4 * (size_t)i + 3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
}

static thread_local bool t_pair_mode = false, t_have = false;
static thread_local PendingRc t_pending;
static long long* g_pair_dbg = nullptr;      // fl_debug_set_pair_stamps
static long g_pair_launches = 0;      // grids issued with both roles (tests check that the path under test is this one)
// response workgroups per column-pass workgroup in the mixed part of the grid, in percent of the even spread's ratio (100)
static int g_pair_density = [] { const char* e = getenv("FLAMO_PAIR_DENSITY"); return e ? atoi(e) : 140; }();
static int g_pair_enabled = [] { const char* e = getenv("FLAMO_LAUNCH_PAIR"); return e ? atoi(e) : 1; }();

long long* pair_dbg() { return g_pair_dbg; }
bool pair_mode() { return t_pair_mode && g_pair_enabled; }
void pending_rc_put(const PendingRc& p) {
    t_pending = p;
    t_have = true;
}
bool pending_rc_take(PendingRc& out) {
    if (!t_have) return false;
    out = t_pending;
    t_have = false;
    return true;
}

int fused_cols_rc_launch(const ColsArgs& a, unsigned n_cols, const PendingRc& rc, hipStream_t st) {
    // the metric's shapes: 200-point columns (nfft = 96000), the 16-wide tile with one load group, 4 or 8 channels on the factor
    const int vt = a.CT << a.cgs;
    if (a.L1 != 200 || vt != 16 || !(rc.niw == 8 || rc.niw == 4)) return FL_ERR_UNSUPPORTED;
    constexpr int A = 8, B = 25, LEN = A * B, LENP = LEN | 1;
    size_t lds = ((size_t)16 * LENP + LEN + (size_t)16 * B) * sizeof(cf);
    if (rc.lds > lds) lds = rc.lds;
    const long n_rc = (long)rc.gx * rc.gy, total = (long)n_cols + n_rc;
    if (total >= (1l << 31)) return FL_ERR_UNSUPPORTED;
    const bool plain = a.env_log2 == 0.0 && a.t_lim >= a.n;
    long n_mix = n_rc + (long)n_cols * 100 / (g_pair_density > 0 ? g_pair_density : 100);
    if (n_mix > total) n_mix = total;
    if (n_mix < n_rc) n_mix = n_rc;
    RcBaArgs rargs = rc.args;
    rargs.dbg = g_pair_dbg ? g_pair_dbg + 4 * 8192 : nullptr;      // (phase stamps of the response role behind the per-workgroup records)
#define FL_PAIR(PLAIN_, NIW_)                                                                                          \
    hipLaunchKernelGGL((cols_fwd_rc_kernel<A, B, 16, 1, PLAIN_, NIW_>), dim3((unsigned)total), dim3(256), lds, st, a, rargs, \
                       (int)n_mix, (int)n_rc, rc.gx, g_pair_dbg)
    if (plain) {
        if (rc.niw == 8) FL_PAIR(true, 8);
        else FL_PAIR(true, 4);
    } else {
        if (rc.niw == 8) FL_PAIR(false, 8);
        else FL_PAIR(false, 4);
    }
#undef FL_PAIR
    FL_CHECK_LAUNCH("cols_fwd_rc");
    ++g_pair_launches;
    return FL_OK;
}

}  // namespace fl

using namespace fl;

extern "C" {

int fl_launch_pair_begin(void) {
    t_pair_mode = true;
    t_have = false;
    return FL_OK;
}

int fl_launch_pair_pending(void) { return t_have ? 1 : 0; }

long fl_debug_launch_pair_count(void) { return g_pair_launches; }

int fl_debug_set_pair_stamps(void* buf) {
    g_pair_dbg = (long long*)buf;
    return FL_OK;
}

int fl_launch_pair_flush(void* stream) {
    t_pair_mode = false;
    PendingRc p;
    if (pending_rc_take(p)) return rc_ba_launch_now(p, (hipStream_t)stream);
    return FL_OK;
}

}  // extern "C"
