"""ctypes binding of libflamo_hip.so (C ABI declared in include/flamo_hip.h).

The shared library is built in-tree (``flamo_amd/libflamo_hip.so``) by ``build()`` /
``make -C flamo_amd/csrc``.  There is NO fallback: if the library is missing or a tensor is
not on a ROCm device the ops raise -- the product path never routes through torch.fft,
torch.einsum, torch.linalg or any CPU code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLAMO_HIP_LIB") or os.path.join(_HERE, "libflamo_hip.so")   # override: A/B builds
CSRC = os.path.join(_HERE, "csrc")

_lock = threading.Lock()
_lib = None

_vp, _i, _l, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/flamo_hip.h one to one
_SIGNATURES = {
    "fl_version": (_i, []),
    "fl_last_error": (C.c_char_p, []),
    "fl_twiddle_fill_f32": (_i, [_vp, _i, _vp]),
    "fl_twiddle_fill_f64": (_i, [_vp, _i, _vp]),
    "fl_fft_plan": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "fl_fft_scratch_elems": (_sz, [_i, _i, _i]),
    "fl_debug_set_fft_max_single": (_i, [_i]),
    "fl_debug_set_fft_fast": (_i, [_i]),
    "fl_rfft_f32": (_i, [_vp, _l, _i, _vp, _l, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_rfft_f64": (_i, [_vp, _l, _i, _vp, _l, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_rfft_ci_f32": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_rfft_ci_f64": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_irfft_f32": (_i, [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_irfft_f64": (_i, [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _i, _d, _d, _i, _vp]),
    "fl_transpose": (_i, [_vp, _vp, _i, _i, _i, _l, _i, _vp]),
    "fl_spec_plan": (_i, [_i, C.POINTER(_i), C.POINTER(_i)]),
    "fl_spec_supports": (_i, [_i, _i, _i]),
    "fl_spec_supports_f64": (_i, [_i, _i, _i]),
    "fl_spec_aux_elems": (_sz, [_i]),
    "fl_spec_aux_fill_f32": (_i, [_vp, _i, _vp]),
    "fl_spec_aux_fill_f64": (_i, [_vp, _i, _vp]),
    "fl_debug_set_spec": (_i, [_i, _i]),
    "fl_debug_set_spec_times": (_i, [_vp]),
    "fl_spec_cols_fwd_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _d, _vp]),
    "fl_spec_cols_fwd_f64": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _d, _vp]),
    "fl_spec_gradh_loop_supports_f32": (_i, [_i, _i, _i]),
    "fl_spec_gradh_loop_supports_f64": (_i, [_i, _i, _i]),
    "fl_spec_gradh_loop_f32": (_i, [_vp, _vp, _l, _l, _vp, _l, _l, _vp, _i, _i, _i, _i, _d, _i, _d, _vp, _vp]),
    "fl_spec_gradh_loop_f64": (_i, [_vp, _vp, _l, _l, _vp, _l, _l, _vp, _i, _i, _i, _i, _d, _i, _d, _vp, _vp]),
    "fl_spec_mid_f32": (_i, [_vp, _vp, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "fl_spec_mid_f64": (_i, [_vp, _vp, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "fl_spec_walk_supports": (_i, [_i, _i, _i]),
    "fl_spec_walk_spectrum_elems": (_sz, [_i, _i, _i]),
    "fl_spec_walk_workgroups": (_i, [_i, _i]),
    "fl_spec_walk_partition": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "fl_spec_mid_walk_f32": (_i, [_vp, _vp, _vp, _vp, _l, _l, _i, _vp, _i, _i, _i, _i, _d, _i, _i, _vp, _vp]),
    "fl_spec_gradh_slices": (_i, [_i, _i]),
    "fl_spec_gradh_walk_f32": (_i, [_vp, _vp, _vp, _l, _l, _l, _i, _vp, _i, _i, _i, _i, _d, _i, _vp]),
    "fl_sum_parts_c64": (_i, [_vp, _l, _i, _vp, _l, _vp]),
    "fl_debug_set_walk": (_i, [_i, _i, _i, _vp]),
    "fl_debug_set_walk_stamps": (_i, [_vp]),
    "fl_wall_clock_khz": (_i, []),
    "fl_spec_cols_inv_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _d, _vp]),
    "fl_spec_cols_inv_sumsq_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp]),
    "fl_spec_cols_inv_sumsq_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp]),
    "fl_spec_cols_blocks_f32": (_i, [_i, _i, _i]),
    "fl_spec_cols_blocks_f64": (_i, [_i, _i, _i]),
    "fl_spec_cols_inv_inplace_ok_f32": (_i, [_i, _i]),
    "fl_spec_cols_inv_inplace_ok_f64": (_i, [_i, _i]),
    "fl_spec_cols_inv_grad_supported_f32": (_i, [_i, _i]),
    "fl_spec_cols_inv_grad_supported_f64": (_i, [_i, _i]),
    "fl_spec_cols_inv_sumsq_grad_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _d, _vp, _vp]),
    "fl_spec_cols_inv_sumsq_grad_f64": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _d, _vp, _vp]),
    "fl_spec_cols_inv_scaled_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _vp, _d, _vp]),
    "fl_spec_cols_inv_scaled_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _vp, _d, _vp]),
    "fl_spec_gradh_walk_scaled_f32": (_i, [_vp, _vp, _vp, _l, _l, _l, _i, _vp, _i, _i, _i, _i, _d, _i, _vp, _vp]),
    "fl_mean_square_final_f32": (_i, [_vp, _i, _d, _vp, _vp]),
    "fl_mean_square_final_f64": (_i, [_vp, _i, _d, _vp, _vp]),
    "fl_pack_toggle": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "fl_spec_cols_inv_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _d, _d, _vp]),
    "fl_permute_bins_c64": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp]),
    "fl_permute_bins_c128": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp]),
    "fl_mimo_c64": (_i, [_vp, _l, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_c128": (_i, [_vp, _l, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_diag_c64": (_i, [_vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_mimo_diag_c128": (_i, [_vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_c64": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _d, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_c128": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _d, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_scaled_c64": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _d, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_scaled_c128": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _d, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_diag_c64": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradh_diag_c128": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _l, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradw_blocks": (_i, [_i]),
    "fl_debug_set_mimo_variant": (_i, [_i, _i]),
    "fl_mimo_gradw_c64": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradw_c128": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradw_re_c64": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_mimo_gradw_re_c128": (_i, [_vp, _l, _l, _l, _vp, _l, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fl_delay_response_c64": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _l, _vp]),
    "fl_delay_response_c128": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _l, _vp]),
    "fl_sos_response_c64": (_i, [_vp, _vp, _i, _i, _d, _vp, _i, _i, _i, _vp, _l, _vp]),
    "fl_sos_response_f32eval_c64": (_i, [_vp, _vp, _i, _i, _d, _vp, _i, _i, _i, _vp, _l, _vp]),
    "fl_sos_response_c128": (_i, [_vp, _vp, _i, _i, _d, _vp, _i, _i, _i, _vp, _l, _vp]),
    "fl_sos_bwd_blocks": (_i, [_i, _i, _i, _i]),
    "fl_debug_set_sos_chunk": (_i, [_i]),
    "fl_debug_set_rc_fast": (_i, [_i]),
    "fl_sos_response_bwd_c64": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _d, _vp, _i, _i, _i, _vp, _vp]),
    "fl_sos_response_bwd_c128": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _d, _vp, _i, _i, _i, _vp, _vp]),
    "fl_sos_response_apply_max_ni": (_i, [_i]),
    "fl_sos_response_apply_c64": (_i, [_vp, _vp, _i, _i, _i, _vp, _l, _l, _i, _d, _vp, _i, _i, _i, _vp, _l, _vp, _l, _l, _vp]),
    "fl_sos_response_bwd_outer_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _i, _i, _i, _vp, _l, _vp, _vp, _i, _d, _vp, _i, _i, _i, _vp, _vp]),
    "fl_sos_response_rc_c64": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _l, _vp, _l, _i, _vp]),
    "fl_geq_response_c64": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _d, _vp, _i, _i, _i, _vp, _l, _i, _vp]),
    "fl_geq_response_rc_c64": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _l, _vp, _l, _i, _vp]),
    "fl_sos_response_bwd_rc_c64": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "fl_geq_sections": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fl_geq_sections_bwd": (_i, [_vp, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp]),
    "fl_geq_sections_bwd_w": (_i, [_vp, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "fl_geq_sections_bwd_w64": (_i, [_vp, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "fl_sos_response_rc_c128": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _l, _vp, _l, _vp]),
    "fl_geq_response_rc_c128": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _l, _vp, _l, _vp]),
    "fl_sos_response_bwd_rc_c128": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "fl_geq_bwd_lanes_blocks": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "fl_geq_bwd_lanes_wrows": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "fl_geq_response_bwd_lanes_c64": (_i, [_i, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fl_geq_sections_bwd_lanes": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "fl_geq_bwd_lanes_blocks_f64": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "fl_geq_bwd_lanes_wrows_f64": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "fl_geq_response_bwd_lanes_c128": (_i, [_i, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _d, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "fl_geq_sections_bwd_lanes_f64": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "fl_debug_set_cascade_lanes": (_i, [_i, _i, _i]),
    "fl_debug_set_cascade_stamps": (_i, [_vp, _i]),
    "fl_solve_max_n": (_i, [_i]),
    "fl_solve_ws_max_n": (_i, []),
    "fl_solve_ws_bytes": (_l, [_i, _i, _i]),
    "fl_solve_ws_c64": (_i, [_vp, _l, _i, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp]),
    "fl_solve_ws_c128": (_i, [_vp, _l, _i, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp]),
    "fl_solve_c64": (_i, [_vp, _l, _i, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_c128": (_i, [_vp, _l, _i, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_scaled_c64": (_i, [_vp, _l, _vp, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_scaled_c128": (_i, [_vp, _l, _vp, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_kept_lu_elems": (_sz, [_i, _i, _i]),
    "fl_solve_kept_piv_elems": (_sz, [_i, _i, _i]),
    "fl_solve_scaled_keep_c64": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp]),
    "fl_solve_scaled_keep_c128": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp]),
    "fl_solve_kept_adjoint_c64": (_i, [_vp, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_kept_adjoint_c128": (_i, [_vp, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_fdn_keep_tile": (_i, [_i, _i]),
    "fl_solve_fdn_wadj_supported": (_i, [_i]),
    "fl_solve_dud2_grads_w_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _l, _vp, _l,
                                       _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_solve_dud2_grads_w_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _l, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _l, _vp, _l,
                                       _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_solve_fdn_wadj_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _l, _vp]),
    "fl_solve_fdn_wadj_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _l, _vp]),
    "fl_solve_fdn_keep_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _vp, _vp]),
    "fl_solve_fdn_keep_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp, _vp, _vp]),
    "fl_solve_kept_adjoint_rank1_c64": (_i, [_vp, _vp, _i, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    "fl_solve_kept_adjoint_rank1_c128": (_i, [_vp, _vp, _i, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    "fl_solve_dud_c64": (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_dud_c128": (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_dud_grads_blocks": (_i, [_i, _i]),
    "fl_solve_fdn_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _i, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    "fl_solve_fdn_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _i, _vp, _i, _vp, _l, _vp, _i, _vp, _l, _vp, _l, _l, _l, _i, _i, _i, _vp]),
    "fl_solve_dud2_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_dud2_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp]),
    "fl_solve_dud2_grads_c64": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_solve_dud2_grads_c128": (_i, [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_solve_dud_grads_c64": (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp]),
    "fl_solve_dud_grads_c128": (_i, [_vp, _l, _l, _vp, _vp, _l, _l, _vp, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp]),
    "fl_debug_set_solve_variant": (_i, [_i]),
    "fl_mean_square_scratch_bytes": (_sz, []),
    "fl_mean_square_f32": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp]),
    "fl_mean_square_f64": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp]),
    "fl_mean_square_bwd_f32": (_i, [_vp, _vp, _vp, _l, _l, _l, _vp]),
    "fl_mean_square_bwd_f64": (_i, [_vp, _vp, _vp, _l, _l, _l, _vp]),
    "fl_cabs_c64": (_i, [_vp, _vp, _l, _l, _l, _l, _vp]),
    "fl_cabs_c128": (_i, [_vp, _vp, _l, _l, _l, _l, _vp]),
    "fl_cabs_bwd_c64": (_i, [_vp, _vp, _vp, _l, _l, _l, _l, _vp]),
    "fl_cabs_bwd_c128": (_i, [_vp, _vp, _vp, _l, _l, _l, _l, _vp]),
    "fl_sparsity_f32": (_i, [_vp, _i, _i, _vp, _vp]),
    "fl_sparsity_f64": (_i, [_vp, _i, _i, _vp, _vp]),
    "fl_sparsity_bwd_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fl_sparsity_bwd_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fl_mse_f32": (_i, [_vp, _vp, _l, _i, _vp, _vp, _vp]),
    "fl_mse_f64": (_i, [_vp, _vp, _l, _i, _vp, _vp, _vp]),
    "fl_mse_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _vp]),
    "fl_mse_bwd_f64": (_i, [_vp, _vp, _vp, _vp, _l, _i, _vp]),
    "fl_matrix_exp_stash_elems": (_sz, [_i]),
    "fl_debug_set_expm_mfma": (_i, [_i]),
    "fl_matrix_exp_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_cplx_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_cplx_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_cplx_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_cplx_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_both_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "fl_matrix_exp_both_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_both_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "fl_matrix_exp_bwd_both_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "fl_eig_c64": (_i, [_vp, _l, _i, _i, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_eig_c128": (_i, [_vp, _l, _i, _i, _vp, _l, _vp, _l, _vp, _vp]),
    "fl_launch_pair_begin": (_i, []),
    "fl_launch_pair_pending": (_i, []),
    "fl_debug_launch_pair_count": (_l, []),
    "fl_debug_set_pair_stamps": (_i, [_vp]),
    "fl_launch_pair_flush": (_i, [_vp]),
    "fl_set_stream_policy": (_i, [C.c_uint, _i]),
    "fl_hbm_probe": (_i, [_i, _vp, _vp, _sz, _i, _i, _vp, _vp]),
}

EXPORTS = tuple(_SIGNATURES)


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into flamo_amd/libflamo_hip.so (hipcc cross-compiles
    without a GPU).  Returns the library path."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libflamo_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-8000:])
    return LIB_PATH


# A launch pair is open on this thread (ops.paired_launch): a recorded response launch must go out before ANY other library
# call than the column pass that carries it -- every call site fetches the handle through lib(), which is where that is enforced.
_pair = threading.local()


def lib(pair_ok: bool = False) -> C.CDLL:
    """The loaded library.  Raises (never falls back) when it has not been built.  ``pair_ok``: the caller is the launch that
    carries a recorded one (see ops.paired_launch); every other call flushes it first."""
    global _lib
    if not pair_ok and getattr(_pair, "stream_of", None) is not None and _lib is not None and _lib.fl_launch_pair_pending():
        rc = _lib.fl_launch_pair_flush(_pair.stream_of())
        _lib.fl_launch_pair_begin()
        if rc != 0:
            raise RuntimeError(f"libflamo_hip launch pair flush failed (code {rc}): " + _lib.fl_last_error().decode("utf-8", "replace"))
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} not found: the HIP extension is required (run "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C flamo_amd/csrc`)."
                    )
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(handle, name)  # AttributeError if the symbol is missing
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().fl_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libflamo_hip {what} failed (code {rc}): {msg}")
