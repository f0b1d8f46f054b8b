// One launch for two independent kernels of the forward pass (gfx950 / MI355X): the cascade response (sos_response_rc_ba:
// bound by packed-VALU issue, depends on the PARAMETERS only) beside the input's column pass (spec_cols_fwd: bound by HBM,
// depends on the INPUT only).  Captured graph branches do not run concurrently on this runtime and the step is the serial sum
// of its kernels, so the overlap is made inside one grid: a workgroup takes either role (fusedfwd.hip).
//
// Host protocol ("launch pair"): between fl_launch_pair_begin() and fl_launch_pair_flush() on a thread, rc_ba_launch
// (cascade2.hip) RECORDS its launch instead of issuing it; the next forward column pass (spectral.hip: cols_launch) issues both
// as one grid when it has the shape for it, otherwise the recorded launch goes out first, alone.  flush issues whatever is
// still recorded.  The caller guarantees that nothing reads the response between the two (flamo_amd/ops.py: paired_launch).
#pragma once
#include "rc_ba_body.h"

namespace fl {
namespace sp32 { struct ColsArgs; }

struct PendingRc {
    RcBaArgs args;
    int niw;               // input channels of the constant factor (template parameter of the kernel)
    int gx, gy;            // grid of the plain launch: blocks of 256 bin pairs x output rows
    size_t lds;
};

long long* pair_dbg();                              // fl_debug_set_pair_stamps' buffer, or null
bool pair_mode();                                   // this thread is between begin and flush
void pending_rc_put(const PendingRc& p);
bool pending_rc_take(PendingRc& out);               // true (and the slot is cleared) when a launch was recorded
int rc_ba_launch_now(const PendingRc& p, hipStream_t st);      // cascade2.hip: the plain launch
// fusedfwd.hip: both in one grid; FL_ERR_UNSUPPORTED (nothing launched, no error text) when the column pass's shape is not taken
int fused_cols_rc_launch(const sp32::ColsArgs& a, unsigned n_cols_blocks, const PendingRc& rc, hipStream_t st);

}  // namespace fl
