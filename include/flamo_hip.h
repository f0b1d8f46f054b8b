/*
 * flamo_hip.h -- C ABI of libflamo_hip.so, the MI355X (gfx950) kernels behind the
 * flamo.processor.dsp / flamo.processor.system operator API.
 *
 * The reference (gdalsanto/flamo v0.2.13) has no FFI layer of its own: its hot path is three
 * torch primitives called from Python.  Every entry point below names the reference call
 * site(s) it replaces (paths relative to the reference checkout).  The library is bound with
 * ctypes from flamo_amd/_lib.py (see INTEGRATION.md for the stub a maintainer would add to
 * the reference).
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (PyTorch's caching allocator),
 *    including scratch.  The library allocates nothing and never synchronises the device;
 *    every launch goes to `stream` (a hipStream_t passed as void*).
 *  - Return value: FL_OK (0) or a negative FL_ERR_* code; fl_last_error() returns a
 *    thread-local message for the last failure on the calling thread.  No C++ exception
 *    crosses the ABI.  Functions are re-entrant.
 *  - Layout ("bin-planar"): a frequency-domain tensor with logical shape (B, M, N, K) is
 *    stored with the bin axis contiguous: address = b*s_b + n*s_n + k*s_k + f.  The distance
 *    between consecutive bin rows (the "pitch", >= the number of bins) is a parameter wherever
 *    a kernel writes or reads whole rows, so callers can pad rows to an aligned length
 *    (flamo_amd.ops pads to a multiple of 32 elements = 256 bytes in c64).  A time-domain
 *    tensor (B, T, N) is stored signal-planar: address = sig*stride + t, sig = b*N + n.
 *    Complex numbers are interleaved (re, im) pairs of the real type (f32 -> "c64",
 *    f64 -> "c128").  Strides are in ELEMENTS of the array's own element type.
 *  - Suffixes: _f32/_c64 single precision, _f64/_c128 double precision.
 */
#ifndef FLAMO_HIP_H
#define FLAMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FL_OK 0
#define FL_ERR_BAD_ARG (-1)
#define FL_ERR_UNSUPPORTED (-2)
#define FL_ERR_HIP (-3)

#define FL_ABI_VERSION 1

int fl_version(void);
const char* fl_last_error(void);

/* ------------------------------------------------------------------ transforms
 * Replace torch.fft.rfft / torch.fft.irfft along dim=1 in dsp.FFT (flamo/processor/dsp.py:88),
 * dsp.iFFT (dsp.py:114), dsp.FFTAntiAlias (dsp.py:161-162), dsp.iFFTAntiAlias (dsp.py:204-205)
 * and Shell.get_time_response / get_freq_response (flamo/processor/system.py:1050-1052,
 * 1119-1128), and the autograd backward of both (rfft' is an irfft with halved interior
 * bins, irfft' is an rfft with doubled interior bins).
 */

/* W[j] = exp(-2*pi*i*j/nfft), j in [0, nfft): master twiddle table shared by the FFT passes,
 * the real-FFT split step, the integer-delay response and the SOS response. */
int fl_twiddle_fill_f32(void* W, int nfft, void* stream);
int fl_twiddle_fill_f64(void* W, int nfft, void* stream);

/* Plan query: nfft must be even with nfft/2 = 2^a 3^b 5^c 7^d 11^e 13^f.  L1*L2 = nfft/2;
 * L1 == 1 means a single in-LDS pass.  Returns FL_ERR_UNSUPPORTED otherwise. */
int fl_fft_plan(int nfft, int is_f64, int* L1, int* L2);
/* complex elements of scratch needed for nsig signals (0 for single-pass plans) */
size_t fl_fft_scratch_elems(int nfft, int is_f64, int nsig);
/* test hook: largest half-length handled in one pass (0 restores the default) */
int fl_debug_set_fft_max_single(int max_half_len);
/* test hook: 0 forces the generic Stockham kernels where the two-register-stage fast kernels
 * would be chosen (default 1); 2 = fast kernels, inverse column pass without mirror-column pairing;
 * 16 / 32 = column-tile width forced; + 1000 * rows per workgroup of the row pass */
int fl_debug_set_fft_fast(int enabled);

/* X[sig, k] = scale * w_k * sum_t x[sig, t] * e(t) * exp(-2 pi i k t / nfft),  k in [0, nfft/2]
 *   x: real, signal `sig` starts at x + sig*x_sig_stride and has t_in valid samples (zero
 *      padded / truncated to nfft as torch.fft.rfft(n=nfft) does);
 *   e(t) = 2^(env_log2 * t) (anti-alias envelope gamma^-t; env_log2 = 0 disables it);
 *   w_k = 1, or (interior_x2 != 0) 2 for 0 < k < nfft/2 (used by irfft's backward);
 *   X: complex, signal `sig` at X + sig*X_sig_stride (X_sig_stride >= nfft/2+1). */
int fl_rfft_f32(const void* x, long x_sig_stride, int t_in, void* X, long X_sig_stride, void* scratch, const void* W,
                int nsig, int nfft, double scale, double env_log2, int interior_x2, void* stream);
int fl_rfft_f64(const void* x, long x_sig_stride, int t_in, void* X, long X_sig_stride, void* scratch, const void* W,
                int nsig, int nfft, double scale, double env_log2, int interior_x2, void* stream);

/* Same transform reading the reference's channel-innermost time tensor directly: x is (B, t_in, n_chan)
 * contiguous, signal sig = b*n_chan + n has sample t at x[(b*t_in + t)*n_chan + n]; nsig = B*n_chan.
 * The layout conversion is fused into the first FFT pass (no separate transpose pass). */
int fl_rfft_ci_f32(const void* x, int n_chan, int t_in, void* X, long X_sig_stride, void* scratch, const void* W,
                   int nsig, int nfft, double scale, double env_log2, int interior_x2, void* stream);
int fl_rfft_ci_f64(const void* x, int n_chan, int t_in, void* X, long X_sig_stride, void* scratch, const void* W,
                   int nsig, int nfft, double scale, double env_log2, int interior_x2, void* stream);

/* y[sig, t] = scale * e(t) * sum_k w_k' Re(v_k X[sig,k] exp(+2 pi i k t / nfft)),  t in [0, t_out)
 *   C2R semantics of torch.fft.irfft: w_k' = 1 for k in {0, nfft/2} (imaginary part ignored),
 *   2 otherwise; v_k = 1, or (interior_half != 0) 1/2 for interior bins (rfft's backward);
 *   t_out <= nfft samples are written per signal at y + sig*y_sig_stride. */
int fl_irfft_f32(const void* X, long X_sig_stride, void* y, long y_sig_stride, int t_out, void* scratch, const void* W,
                 int nsig, int nfft, double scale, double env_log2, int interior_half, void* stream);
int fl_irfft_f64(const void* X, long X_sig_stride, void* y, long y_sig_stride, int t_out, void* scratch, const void* W,
                 int nsig, int nfft, double scale, double env_log2, int interior_half, void* stream);

/* dst[b*cols*dst_pitch + c*dst_pitch + r] = src[b][r][c]  (batched 2-D transpose through LDS;
 * elem_bytes in {4, 8, 16}; dst_pitch >= rows lets the caller pad the output rows).
 * Converts the reference's channel-innermost (B, T, N) / (B, M, N) tensors to the planar layout
 * and back. */
int fl_transpose(const void* src, void* dst, int nbatch, int rows, int cols, long dst_pitch, int elem_bytes, void* stream);

/* ------------------------------------------------------------------ fused Shell pipeline (float32)
 * Replaces the whole of Shell.forward (flamo/processor/system.py:839-855) when the input layer is
 * dsp.FFT / FFTAntiAlias (dsp.py:88, 161-162), the core is a chain of per-bin products (Series.forward,
 * system.py:299-300, over dsp.py:466, 552, 922-924, 1021) and the output layer is dsp.iFFT / iFFTAntiAlias
 * (dsp.py:114, 204-205):   y = irfft( H[f] . rfft(x) )   in three launches, the (B, M, N) spectrum never
 * making a round trip through HBM between the transforms and the product.  Time-domain tensors are taken
 * and produced channel-innermost (B, T, G) as the reference's users hand them in (no layout conversion).
 * nfft/2 = L1*L2 (fl_spec_plan).  Between the launches the data sits in a scratch array S (B, L1, L2, G)
 * of complex values; spectra and responses that cross the boundary (the spectrum kept for the backward
 * pass, H and its gradient) are bin-planar in ROW-MAJOR BIN ORDER: bin k = k1 + L1*k2 (k1 < L1, k2 < L2)
 * is element i = k1*L2 + k2 of its plane, the Nyquist bin nfft/2 is element nfft/2 (fl_permute_bins_c64
 * converts).  The backward pass runs the same three kernels (irfft' = weighted rfft, rfft' = weighted irfft).
 */
/* FL_OK and (L1, L2) when the fused pipeline supports this transform length */
int fl_spec_plan(int nfft, int* L1, int* L2);
/* The three kernels below take W = the master twiddle table (nfft entries, fl_twiddle_fill_f32) FOLLOWED by
 * fl_spec_aux_elems(nfft) entries of contiguous copies (W_L1^j, W_L2^j, W_n^(L1 j)) written by fl_spec_aux_fill_f32
 * (launch it behind the master fill on the same stream). */
size_t fl_spec_aux_elems(int nfft);
int fl_spec_aux_fill_f32(void* W, int nfft, void* stream);
/* 1 when (nfft, input channels, output channels) has a fused kernel */
int fl_spec_supports(int nfft, int n_in, int n_out);
/* tuning hook: virtual columns per workgroup of the column passes (16 | 32; 0 = per-shape choice), load group (1 | 2 | 4; 0 = auto) */
int fl_debug_set_spec(int vt, int rg);
/* tuning hook: device buffer of 8 int64 per workgroup of fl_spec_mid_f32 for cycle stamps at its phase boundaries (null: off) */
int fl_debug_set_spec_times(void* buf);
/* K1: S[b][k1][c][g] = W_L^(c k1) sum_t1 z[c + L2 t1] W_L1^(t1 k1),  z[j] = e(2j) x[b][2j][g] + i e(2j+1) x[b][2j+1][g];
 * x: real (Bn, t_len, G) contiguous, 8-byte aligned, G even (a power of two <= 32 or a multiple of 32); samples at
 * t >= min(t_len, nfft) count as zero. */
int fl_spec_cols_fwd_f32(const void* x, int Bn, int t_len, int G, void* S, const void* W, int nfft, double env_log2,
                         void* stream);
/* mid: rows k1 = r, L1 - r of all channels per workgroup.  Forward row FFTs and split step give the spectrum
 *   X[k] = spec_scale * w_k * rfft(.)[k]  (w_k = 2 on interior bins if spec_interior2), stored to Xs if non-null
 *   (Xs[b*xs_b + n*xs_n + i], row-major bin order);  if S2 is non-null:  Y[k] = op(H[k]) X[k]  (H null: Y = X,
 *   NI == NO; H[m*hs_m + n*hs_n + i]; conj_h: conj(H) -- pass swapped strides for H^H), interior bins halved if
 *   pre_half, Hermitian pre-step, inverse row FFTs and twiddle into S2 (Bn, L1, L2, NO). */
int fl_spec_mid_f32(const void* S, void* S2, void* Xs, long xs_b, long xs_n, const void* H, long hs_m, long hs_n, int conj_h,
                    const void* W, int nfft, int Bn, int NI, int NO, double spec_scale, int spec_interior2, int pre_half,
                    void* stream);
/* mid, batch-walking form (csrc/specwalk.hip): the same operator as fl_spec_mid_f32 with H present and S2 != S.
 * One workgroup per CU owns a contiguous range of (row pair, batch item) units and keeps the row pair's response slice in
 * registers, so H (the "fmn" operand of einsum("fmn,bfn...->bfm...") at dsp.py:922-924) crosses the L2 -> L1 path once per
 * row pair and workgroup instead of once per batch item; the next unit's scratch rows arrive by LDS-DMA under the current
 * unit's phases, and the inverse-side FFT stages of a unit run beside the forward-side stages of the next on the other
 * half of the wavefronts.  Xp (or null): the spectrum for the backward pass, PAIR-MAJOR and private to these two kernels:
 *   Xp[((r*Bn + b)*2 + e)*NI*L2 + n*L2 + p] = X[b][n][bin], bin = k (e = 0) or nfft/2 - k (e = 1) of pair p of row pair r
 * (fl_spec_walk_spectrum_elems complex values).  fl_spec_walk_supports: 1 when (nfft, channels) has this kernel. */
int fl_spec_walk_supports(int nfft, int n_in, int n_out);
size_t fl_spec_walk_spectrum_elems(int nfft, int Bn, int NI);
/* Work partition of the forward kernel: fl_spec_walk_workgroups = its grid size; fl_spec_walk_partition fills the HOST array
 * bounds[n_wg + 1] with contiguous unit ranges of about equal cost (a unit = (row pair, batch item), r * Bn + b; entering a row
 * pair costs a fill, self-mirrored row pairs are half units).  The caller keeps a DEVICE copy and passes it as `bounds`
 * (null: equal unit counts). */
int fl_spec_walk_workgroups(int nfft, int Bn);
int fl_spec_walk_partition(int nfft, int Bn, int n_wg, int* bounds);
int fl_spec_mid_walk_f32(const void* S, void* S2, void* Xp, const void* H, long hs_m, long hs_n, int conj_h, const void* W, int nfft, int Bn,
                         int NI, int NO, double spec_scale, int spec_interior2, int pre_half, const void* bounds, void* stream);
/* Backward of the product inside the row kernel -- the autograd of dsp.py:922-924 w.r.t. the response:
 *   dH[m][n][i] = sum_b gY[b,m,i] conj(X[b,n,i]),   gY = scale_g * w_k * rfft-spectrum of the rows in Sg (K1 of the output
 *   gradient, (Bn, L1, L2, NO); w_k = 2 on interior bins if interior2_g), X = the pair-major spectrum fl_spec_mid_walk_f32
 *   kept.  gY never goes through HBM.  Workgroup (row pair, batch slice) sums its slice in registers and writes partial
 *   plane set s < n_slices: dH_parts[s*ds_s + m*ds_m + n*ds_n + i], row-major bin order (every element of every set is
 *   written; fl_sum_parts_c64 or the consumer adds the sets -- fixed order, no atomics).
 *   fl_spec_gradh_slices: the slice count that fills the device (1 <= . <= Bn). */
int fl_spec_gradh_slices(int nfft, int Bn);
int fl_spec_gradh_walk_f32(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g, void* stream);
/* the same with dH multiplied by the device scalar out_scale[0] (float) on its way out: the factor 2 g / N of an objective
 * mean(y^2) whose gradient g_y = (2 g / N) y is never materialised -- K1 runs on y itself (ops.mean_square) */
int fl_spec_gradh_walk_scaled_f32(const void* Sg, const void* Xp, void* dH_parts, long ds_s, long ds_m, long ds_n, int n_slices,
                                  const void* W, int nfft, int Bn, int NI, int NO, double scale_g, int interior2_g,
                                  const void* out_scale, void* stream);
/* out[j] = sum_{s < n_parts} parts[s*part_stride + j], j < n complex values (16-byte aligned, n and part_stride even) */
int fl_sum_parts_c64(const void* parts, long part_stride, int n_parts, void* out, long n, void* stream);
/* tuning hook: mode 0 switches the walking kernels off (fl_spec_walk_supports -> 0); wgs / slices override the forward
 * kernel's workgroup count and the backward kernel's slice count (0 = per device); times: 8 int64 per workgroup or null */
int fl_debug_set_walk(int mode, int wgs, int slices, void* times);
/* Measurement: with a device buffer of 2 x 1024 + 1 int64 set, every launch of fl_spec_mid_walk_f32 issued (or captured)
 * afterwards stamps the constant-rate device clock (s_memrealtime, fl_wall_clock_khz) when each workgroup w starts and ends
 * (buf[2 w], buf[2 w + 1]), and the inverse column pass behind it (fl_spec_cols_inv_*) leaves its own start in buf[2048]:
 * buf[2048] - min(start) is the kernel's whole slot inside a replayed graph (its launch, its run, the drain of its stores), where
 * events cannot be recorded; max(end) - min(start) its active time.  NULL: off. */
int fl_debug_set_walk_stamps(void* buf);
int fl_wall_clock_khz(void);
/* K3: y[b][t][g] = scale * e(t) * (unnormalised inverse transform of S2), t < t_out <= t_len; y: real (Bn, t_len, G) */
int fl_spec_cols_inv_f32(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                         double env_log2, void* stream);
/* K3 that also leaves sumsq_parts[w] = sum of the squares of the samples workgroup w stored (double; fl_spec_cols_blocks
 * entries, every one written): the reduction of an objective mean(y^2) -- trainer.py:177-190 with a squared-error criterion --
 * rides in the pass that produces y; fl_mean_square_final_* combines the partials in a fixed order */
int fl_spec_cols_inv_sumsq_f32(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                               double env_log2, void* sumsq_parts, void* stream);
int fl_spec_cols_blocks_f32(int nfft, int Bn, int G);
/* K3 (plain shape: every sample stored, no envelope) that also leaves Sg = what fl_spec_cols_fwd_f32 of the y it stores would
 * leave -- the first pass of the GRADIENT's transform when the objective's g_y is a multiple of y (trainer.py:177-190 with a
 * squared-error criterion: loss.backward() starts with rfft(g_y), dsp.py:114 under autograd) -- formed from the tile in
 * registers: the backward pass does not re-read y.  fl_spec_cols_inv_grad_supported_*: 1 where the fused form exists
 * (every column plan whose first radix times the tile width is at most 256 -- all but the 441- and 800-point columns and 16
 * channels on 16-point radices -- both precisions), else the caller runs fl_spec_cols_fwd on y. */
int fl_spec_cols_inv_grad_supported_f32(int nfft, int G);
/* 1: fl_spec_cols_inv_* may be given y = S2 (the real (Bn, nfft, G) output over the scratch it is transformed from; t_len = t_out
 * = nfft): a workgroup reads its whole tile before its first store and the two tiles are the same bytes when one tile carries all
 * G channels */
int fl_spec_cols_inv_inplace_ok_f32(int nfft, int G);
int fl_spec_cols_inv_sumsq_grad_f32(const void* S2, void* y, void* Sg, int Bn, int G, const void* W, int nfft, double scale,
                                    void* sumsq_parts, void* stream);
/* K3 with a DEVICE scalar (float; double in the _f64 form) multiplied into `scale`: the gradient of the input under an
 * objective whose factor lives on the device (trainer.py:177-190: loss.backward() hands 2 g / N down as a tensor) -- a
 * multiplication pass over the (Bn, t_len, G) result otherwise */
int fl_spec_cols_inv_scaled_f32(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                                const void* dev_scale, double env_log2, void* stream);
/* per plane: dst[i] = src[k(i)] (inverse = 0, natural -> row-major bin order) or dst[k(i)] = src[i] (inverse = 1) */
int fl_permute_bins_c64(const void* src, long src_pitch, void* dst, long dst_pitch, int nplanes, int nfft, int inverse,
                        void* stream);
/* Backward of the pipeline, gradient of the response, in ONE launch (round 5) where the batch-walking kernel
 * fl_spec_gradh_walk_f32 does not apply -- float64, fewer than four batch items:
 *   dH[m][n][i] = out_scale * (*dev_scale) * sum_b gY[b][m][i] conj(Xs[b][n][i]),
 * gY = the spectrum of the output's gradient whose column pass is Sg (scale_g, interior bins doubled: the irfft backward of
 * dsp.py:110-115 under the einsum backward of dsp.py:922-924), Xs the spectrum fl_spec_mid_* kept.  Replaces
 * fl_spec_mid_*(S2 = null, Xs out) + fl_mimo_gradh_*: the gradient's spectrum is never written.  A workgroup owns (row pair,
 * output-channel group) and walks the batch; the sum over the batch is in a fixed order (no atomics).  240- / 256-bin rows,
 * equal channel counts of 2 / 4 / 8 (fl_spec_gradh_loop_supports_*). */
int fl_spec_gradh_loop_supports_f32(int nfft, int NI, int NO);
int fl_spec_gradh_loop_supports_f64(int nfft, int NI, int NO);
int fl_spec_gradh_loop_f32(const void* Sg, const void* Xs, long xs_b, long xs_n, void* dH, long ds_m, long ds_n, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2, double out_scale,
                           const void* dev_scale, void* stream);
int fl_spec_gradh_loop_f64(const void* Sg, const void* Xs, long xs_b, long xs_n, void* dH, long ds_m, long ds_n, const void* W,
                           int nfft, int Bn, int NI, int NO, double scale_g, int interior2, double out_scale,
                           const void* dev_scale, void* stream);

/* The same pipeline in float64 / complex128 (spectral.hip compiled a second time with real_t = double): same arguments, same
 * layouts, W from fl_twiddle_fill_f64 + fl_spec_aux_fill_f64.  One workgroup per (row pair, batch item) at every plan length,
 * equal channel counts only (fl_spec_supports_f64 also accounts for the doubled LDS of the row and column tiles). */
int fl_spec_supports_f64(int nfft, int n_in, int n_out);
int fl_spec_aux_fill_f64(void* W, int nfft, void* stream);
int fl_spec_cols_fwd_f64(const void* x, int Bn, int t_len, int G, void* S, const void* W, int nfft, double env_log2,
                         void* stream);
int fl_spec_mid_f64(const void* S, void* S2, void* Xs, long xs_b, long xs_n, const void* H, long hs_m, long hs_n, int conj_h,
                    const void* W, int nfft, int Bn, int NI, int NO, double spec_scale, int spec_interior2, int pre_half,
                    void* stream);
int fl_spec_cols_inv_f64(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                         double env_log2, void* stream);
int fl_spec_cols_inv_sumsq_f64(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                               double env_log2, void* sumsq_parts, void* stream);
int fl_spec_cols_blocks_f64(int nfft, int Bn, int G);
int fl_spec_cols_inv_scaled_f64(const void* S2, void* y, int Bn, int t_len, int t_out, int G, const void* W, int nfft, double scale,
                                const void* dev_scale, double env_log2, void* stream);
int fl_spec_cols_inv_grad_supported_f64(int nfft, int G);
int fl_spec_cols_inv_inplace_ok_f64(int nfft, int G);
int fl_spec_cols_inv_sumsq_grad_f64(const void* S2, void* y, void* Sg, int Bn, int G, const void* W, int nfft, double scale,
                                    void* sumsq_parts, void* stream);
int fl_permute_bins_c128(const void* src, long src_pitch, void* dst, long dst_pitch, int nplanes, int nfft, int inverse,
                         void* stream);

/* ------------------------------------------------------------------ per-bin complex MIMO product
 * Replace torch.einsum("fmn,bfn...->bfm...") (dsp.py:922-924, 3406-3408 and every Filter
 * subclass), einsum("mn,bfn...->bfm...") (Gain/Matrix, dsp.py:466-468) with hs_f = 0,
 * einsum("fn,bfn...->bfn...") (parallel*, dsp.py:1021-1023, 3504-3506) and
 * einsum("n,bfn...->bfn...") (parallelGain, dsp.py:552-554) with hs_f = 0, plus their
 * autograd backward.
 *
 * Y[b,m,k,f] = sum_n op(H[f,m,n]) X[b,n,k,f];  op = conj if conj_h & 1.
 * H element address: f*hs_f + m*hs_m + n*hs_n (hs_f = 0: frequency-independent matrix).
 * conj_h & 2 (fl_mimo_* only, hs_f = 0): H is a REAL matrix of the signal's precision, strides in real elements -- the
 * Gain / Matrix parameter as it is, without the real -> complex cast of dsp.py:466-468.
 * X/Y addresses: b*s_b + ch*s_n + k*s_k + f. */
int fl_mimo_c64(const void* H, long hs_f, long hs_m, long hs_n, int conj_h,
                const void* X, long xs_b, long xs_n, long xs_k,
                void* Y, long ys_b, long ys_m, long ys_k,
                int B, int M, int No, int Ni, int K, void* stream);
int fl_mimo_c128(const void* H, long hs_f, long hs_m, long hs_n, int conj_h,
                 const void* X, long xs_b, long xs_n, long xs_k,
                 void* Y, long ys_b, long ys_m, long ys_k,
                 int B, int M, int No, int Ni, int K, void* stream);

/* Y[b,n,k,f] = op(h[f,n]) X[b,n,k,f];  h address f*hs_f + n*hs_n */
int fl_mimo_diag_c64(const void* h, long hs_f, long hs_n, int conj_h,
                     const void* X, long xs_b, long xs_n, long xs_k,
                     void* Y, long ys_b, long ys_n, long ys_k,
                     int B, int M, int N, int K, void* stream);
int fl_mimo_diag_c128(const void* h, long hs_f, long hs_n, int conj_h,
                      const void* X, long xs_b, long xs_n, long xs_k,
                      void* Y, long ys_b, long ys_n, long ys_k,
                      int B, int M, int N, int K, void* stream);

/* dH[m,n,f] = scale * sum_{b,k} G[b,m,k,f] * conj(X[b,n,k,f])   (planar dH: (m*Ni+n)*dh_pitch + f).
 * The gradient of the product w.r.t. H (torch complex convention), and -scale = the
 * Recursion's dA = -dR out^H. */
int fl_mimo_gradh_c64(const void* G, long gs_b, long gs_m, long gs_k,
                      const void* X, long xs_b, long xs_n, long xs_k,
                      void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream);
int fl_mimo_gradh_c128(const void* G, long gs_b, long gs_m, long gs_k,
                       const void* X, long xs_b, long xs_n, long xs_k,
                       void* dH, long dh_pitch, double scale, int B, int M, int No, int Ni, int K, void* stream);
/* ... with a DEVICE scalar (float / double) multiplied into `scale` by the kernel: the factor an objective hands down as a
 * tensor (trainer.py:177-190: loss.backward()) -- a multiplication pass over dH otherwise */
int fl_mimo_gradh_scaled_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                             void* dH, long dh_pitch, double scale, const void* dev_scale, int B, int M, int No, int Ni, int K, void* stream);
int fl_mimo_gradh_scaled_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                              void* dH, long dh_pitch, double scale, const void* dev_scale, int B, int M, int No, int Ni, int K, void* stream);
/* dh[n,f] = sum_{b,k} G[b,n,k,f] * conj(X[b,n,k,f])   (planar dh: n*dh_pitch + f) */
int fl_mimo_gradh_diag_c64(const void* G, long gs_b, long gs_n, long gs_k,
                           const void* X, long xs_b, long xs_n, long xs_k,
                           void* dh, long dh_pitch, int B, int M, int N, int K, void* stream);
int fl_mimo_gradh_diag_c128(const void* G, long gs_b, long gs_n, long gs_k,
                            const void* X, long xs_b, long xs_n, long xs_k,
                            void* dh, long dh_pitch, int B, int M, int N, int K, void* stream);

/* dW[m,n] = sum_{b,k,f} G[b,m,k,f] * conj(X[b,n,k,f]): gradient of a frequency-independent matrix
 * (Gain/Matrix, dsp.py:466-468; the FDN mixing matrix) with the reduction over bins done in the
 * kernel.  part: complex scratch (nblk, No, Ni), nblk = fl_mimo_gradw_blocks(M), per-block partial tiles;
 * dW: complex (No, Ni) = their sum in block order (a second tiny launch: deterministic, no atomics). */
int fl_mimo_gradw_blocks(int M);
/* tuning hook: variant = mt*100 + bt*10 + nu (register tile MT x BT, nu input channels loaded per
 * trip; 0 = default; -1 = lane-per-bin kernels also where the MFMA kernels apply, -14 = MFMA kernels with
 * the 64-bin 16x16 tile only), gradw_cap = partial blocks of fl_mimo_gradw (0 = default; -2 = MFMA
 * kernels store straight from the accumulator layout instead of through LDS) */
int fl_debug_set_mimo_variant(int variant, int gradw_cap);
int fl_mimo_gradw_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                      void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream);
int fl_mimo_gradw_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                       void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream);
/* The same reduction for a REAL frequency-independent matrix (a Gain whose complex cast is never formed, see conj_h bit 1
 * of fl_mimo_*): dW is a real (No, Ni) array holding the real part of the sum. */
int fl_mimo_gradw_re_c64(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                         void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream);
int fl_mimo_gradw_re_c128(const void* G, long gs_b, long gs_m, long gs_k, const void* X, long xs_b, long xs_n, long xs_k,
                          void* part, void* dW, int B, int M, int No, int Ni, int K, void* stream);

/* ------------------------------------------------------------------ frequency responses
 * Integer delay lines, Delay/parallelDelay.get_freq_response with isint=True
 * (dsp.py:3356-3365, 3512-3521):  H[c, f] = amp[c] * exp(-2 pi i ((bin0+f) * m[c] mod nfft) / nfft)
 * -- the phase index is reduced in 64-bit integer arithmetic and looked up in W (bit-exact
 * indexing; the reference evaluates exp(-j*omega*m) in floating point).  amp[c] = gamma^m[c]
 * is supplied by the caller (real, same precision as H). H planar: c*h_pitch + f.
 * Bin range (here and in the cascade entry points below): element f of a row is bin bin0 + f, f < m_local
 * (bin0 >= 0: the contiguous range a rank owns under bin sharding); bin0 < 0 asks for the whole spectrum
 * (m_local = nfft/2+1) in the ROW-MAJOR bin order of the fused Shell pipeline with row length L2 = -bin0:
 * element k1*L2 + k2 is bin k1 + (nfft/2/L2)*k2, element nfft/2 the Nyquist bin (see fl_spec_plan). */
int fl_delay_response_c64(const int32_t* m, const void* amp, int C, const void* W, int nfft,
                          int bin0, int m_local, void* H, long h_pitch, void* stream);
int fl_delay_response_c128(const int32_t* m, const void* amp, int C, const void* W, int nfft,
                           int bin0, int m_local, void* H, long h_pitch, void* stream);

/* Second-order-section cascades, the tail shared by Biquad/SVF/GEQ/PEQ.get_poly_coeff
 * (dsp.py:1520-1526, 2587-2593):  per channel c and bin k,
 *   B_s = b[0,s,c] + b[1,s,c] g w + b[2,s,c] g^2 w^2,  A_s likewise,  w = exp(-2 pi i k/nfft),
 *   H[c,k] = prod_s B_s / prod_s A_s, or eps where |prod A| == 0.
 * b, a: DOUBLE (3, S, C) contiguous and Wd the FLOAT64 twiddle table (fl_twiddle_fill_f64)
 * whatever the precision of H (_c64 / _c128).  Evaluated directly (no (M,S,C) tensor is ever built), in
 * double precision whatever the storage type: the shelving sections cancel to ~1e-5 of their
 * terms at low frequency, which float32 evaluation (the reference's float32 mode) cannot hold. */
int fl_sos_response_c64(const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                        int nfft, int bin0, int m_local, void* H, long h_pitch, void* stream);
int fl_sos_response_c128(const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                         int nfft, int bin0, int m_local, void* H, long h_pitch, void* stream);
/* fl_sos_response_c64 with the cascade evaluated in float (section polynomials about w = +1 / -1 with coefficient sums
 * formed in double, two sections per packed instruction): response within 3e-7 of the double evaluation's, about twice
 * as fast.  For forward-only use: the mixed-precision backward reuses the saved forward response, and parameter maps
 * with cancellation (parametric equalisers) amplify the extra 2e-7 to 1e-5 .. 1e-4 in the gradient. */
int fl_sos_response_f32eval_c64(const void* b, const void* a, int S, int C, double gamma, const void* Wd,
                                int nfft, int bin0, int m_local, void* H, long h_pitch, void* stream);
/* Graphic equaliser from its command gains in ONE launch (fl_geq_sections + fl_sos_response[_f32eval]_c64): the float
 * kernel designs the sections in its prologue; b, a (double (3, nb, C) each) are outputs for the backward pass. */
int fl_geq_response_c64(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int C, double gamma, const void* Wd,
                        int nfft, int bin0, int m_local, void* H, long h_pitch, int float_eval, void* stream);
/* Backward: partial sums over bins of dL/db, dL/da.  part: double (nblk, 2, 3, S, C) where
 * nblk = fl_sos_bwd_blocks(m_local, C, S, mixed) (mixed = 1 for the _c64 route with H, the constant-factor and the
 * outer-product forms; the grid is sized to one round of resident workgroups); every entry is written (no zero-fill
 * needed), the caller sums over nblk.  H / h_pitch: the forward output.  _c64 with H != NULL takes the mixed-precision
 * route (section values in double, quotients and running sums in float in the basis
 * {1, d, d^2}, d = 1 - g w, converted back in double); H == NULL, and _c128 always, evaluates
 * everything in double. */
int fl_sos_bwd_blocks(int m_local, int C, int S, int mixed);
/* tuning hook: sections whose sums one thread keeps in registers (0 = default); + 100 * blocks per channel */
int fl_debug_set_sos_chunk(int sections_per_thread);
/* test hook: 0 = the float evaluations (fl_sos_response_f32eval_c64, fl_sos_response_rc_c64 with float_eval) fall back to double */
int fl_debug_set_rc_fast(int on);
int fl_sos_response_bwd_c64(const void* gH, long g_pitch, const void* H, long h_pitch, const void* b, const void* a, int S,
                            int C, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* part, void* stream);
int fl_sos_response_bwd_c128(const void* gH, long g_pitch, const void* H, long h_pitch, const void* b, const void* a, int S,
                             int C, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* part, void* stream);
/* A full (No, Ni) cascade applied to a signal with BX = 1 or 2 columns in one launch (float evaluation as
 * fl_sos_response_f32eval_c64): G[m*Ni + j, f] (planes of pitch g_pitch, kept for the backward pass) and
 * Y[b][m][f] = sum_j G[m][j][f] X[b][j][f]  (X planes b*xs_b + j*xs_n + f, Y planes b*ys_b + m*ys_m + f) -- dsp.py:922-924 over
 * the cascade tail dsp.py:1520-1526 without the product's own pass over the response.  Ni <= fl_sos_response_apply_max_ni(S). */
int fl_sos_response_apply_max_ni(int S);
int fl_sos_response_apply_c64(const void* b, const void* a, int S, int No, int Ni, const void* X, long xs_b, long xs_n, int BX,
                              double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* Y, long ys_b,
                              long ys_m, void* stream);
/* fl_sos_response_bwd_c64 (mixed-precision route) when the response was applied to a signal with few columns,
 * Y[b,:,f] = H[f] X[b,:,f] (dsp.py:922-924): dL/dH[m][n][f] = sum_b gY[b][m][f] conj(X[b][n][f]) is formed inside the
 * kernel from the two signals (planes b*s_b + channel*s_n + f) instead of being read from an (M, No, Ni) tensor.
 * H: the saved forward response, planes c = m*Ni + n of pitch h_pitch.  part as above. */
int fl_sos_response_bwd_outer_c64(const void* gY, long gy_sb, long gy_sn, const void* X, long x_sb, long x_sn, int B, int No, int Ni,
                                  const void* H, long h_pitch, const void* b, const void* a, int S, double gamma, const void* Wd,
                                  int nfft, int bin0, int m_local, void* part, void* stream);
/* Cascade response times a real constant matrix on the right -- Series(Matrix, <cascade-type filter>), system.py:299-300
 * over dsp.py:466-468 and dsp.py:922-924 (the reference applies the two modules one after the other):
 *   G[m*Nmid + j, f] as fl_sos_response_c64 (planes of pitch g_pitch; kept for the backward pass),
 *   H[m*Ni + n, f] = sum_j G[m][j] Wr[j][n]   (planes of pitch h_pitch), Wr float (Nmid, Ni) row-major, Ni in {2,4,8,16}.
 * float_eval != 0: the cascade in float as in fl_sos_response_f32eval_c64 (graphic-equaliser sections, whose parameter
 * map is benign: gradient within 2e-7 of the double evaluation's; or forward-only use). */
int fl_sos_response_rc_c64(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma,
                           const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch,
                           int float_eval, void* stream);
/* The same operator for a graphic equaliser given by its command gains (eq.py:57-111; gain / in_kind / consts as in
 * fl_geq_sections): the float kernel designs the sections in its prologue and writes them to b, a (double (3, nb, No*Nmid)
 * each, outputs: the backward pass reads them) -- one launch for fl_geq_sections + fl_sos_response_rc_c64. */
int fl_geq_response_rc_c64(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int No, int Nmid, int Ni,
                           const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch,
                           void* H, long h_pitch, int float_eval, void* stream);
/* The same backward pass when the cascade's response G (No x Nmid per bin, channel pair c = m*Nmid + j) was multiplied on
 * the right by a real constant matrix W (Nmid x Ni) -- Series(Matrix, <cascade-type filter>), system.py:299-300 over
 * dsp.py:466-468 and dsp.py:922-924:  H[m][n] = sum_j G[m][j] W[j][n].  gHfull: dL/dH, planes (m*Ni + n) of pitch g_pitch;
 * G: the saved forward response of the cascade (planes c, pitch h_pitch).  part as above (for G's coefficients);
 * partW: float (fl_sos_bwd_blocks(m_local, C, S, 1), No*Nmid, Ni), per-block partials of Re(conj(G[m][j]) dL/dH[m][n]) -- summed
 * over blocks and over m they are dL/dW[j][n].  Replaces two response-sized composition-backward passes.  Ni in {2,4,8,16}. */
int fl_sos_response_bwd_rc_c64(const void* gHfull, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                               int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                               int bin0, int m_local, void* part, void* partW, void* stream);
/* The three operators above in complex128 with a float64 constant factor (Wr, partW double; G, H, gHfull complex128) --
 * the same Series(Matrix, <cascade-type filter>) of system.py:299-300 when the model is built with dtype=torch.float64, the
 * default of the reference's example scripts (examples/e7_biquad.py:237).  Everything is evaluated in double (the kernels
 * of fl_sos_response_c128 / fl_sos_response_bwd_c128 with the composition folded in); the backward takes cascades of at most
 * 12 sections (one register-resident chunk, so that the Ni gradient planes are read once). */
int fl_sos_response_rc_c128(const void* b, const void* a, int S, int No, int Nmid, int Ni, const void* Wr, double gamma,
                            const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch, void* H, long h_pitch,
                            void* stream);
int fl_geq_response_rc_c128(const void* gain, int in_kind, int nb, const void* consts, void* b, void* a, int No, int Nmid, int Ni,
                            const void* Wr, double gamma, const void* Wd, int nfft, int bin0, int m_local, void* G, long g_pitch,
                            void* H, long h_pitch, void* stream);
int fl_sos_response_bwd_rc_c128(const void* gHfull, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                int bin0, int m_local, void* part, void* partW, void* stream);

/* Graphic-equaliser design: command gains -> the float32-rounded second-order sections of
 * GEQ / parallelGEQ for all C channel pairs at once (replaces the Python double loop over
 * flamo/auxiliary/eq.py:57-111 `geq` in dsp.py:2573-2585, 2661-2672), and its backward.
 *   gain: (nb, C), in_kind 0: command gains in dB, double;
 *                  in_kind 1 / 2: the module's RAW parameters x (double / float) under its default
 *                  map 20 log10|x| (dsp.py:2526) -- the linear gain is then |x| and the map, its
 *                  backward and the dtype casts fold into these two launches;
 *                  in_kind 3 / 4: RAW parameters x (double / float) under the map 20 log10(sigmoid(x)) that the
 *                  reference's FDN examples give their attenuation filters (e8_fdn.py:97): linear gain sigmoid(x);
 *   b, a: double (3, nb, C) holding float32-representable values;
 *   consts: double [t_lo, t_hi, t2_lo, t2_hi, st_lo, st_hi, pk_t[nb-3], pk_c[nb-3]] -- tan/cos of
 *   the float32 band frequencies as the host evaluates them (flamo/functional.py:555-675).
 * Backward: gb, ga: double (nblk, 3, nb, C) partial gradients, blk_stride elements between blocks
 * (fl_sos_response_bwd's `part` is consumed directly: gb = part, ga = part + 3*nb*C,
 * blk_stride = 6*nb*C); ggain: (nb, C) in gain's type. */
int fl_geq_sections(const void* gain, int in_kind, int nb, int C, const void* consts, void* b, void* a, void* stream);
int fl_geq_sections_bwd(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                        int C, const void* consts, void* ggain, void* stream);
/* fl_geq_sections_bwd with a second small reduction in the same launch: gW[e] = sum_r partW[r*wn + e], r < wrows,
 * e < wn (float) -- the partials fl_sos_response_bwd_rc_c64 leaves for the constant factor's gradient (wrows =
 * blocks * No, wn = Nmid * Ni). */
int fl_geq_sections_bwd_w(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                          int C, const void* consts, void* ggain, const void* partW, int wrows, int wn, void* gW, void* stream);
/* ... with double partW / gW (the partials of fl_sos_response_bwd_rc_c128) */
int fl_geq_sections_bwd_w64(const void* gain, int in_kind, const void* gb, const void* ga, long blk_stride, int nblk, int nb,
                            int C, const void* consts, void* ggain, const void* partW, int wrows, int wn, void* gW, void* stream);

/* Second generation of the graphic equaliser's cascade backward (csrc/cascade2.hip): one lane per (channel pair, section)
 * walking the bins, instead of one lane per bin walking the sections -- the backward of the cascade tail dsp.py:1520-1526
 * for GEQ / parallelGEQ sections (dsp.py:2563-2593, eq.py:57-111) under the einsum dsp.py:922-924, with the composition
 * backward of Series(Matrix, GEQ) (system.py:299-300, dsp.py:466-468) folded in as fl_sos_response_bwd_rc_c64 does.
 *   mode 0: gH = dL/dG, planes c (No * Nmid channel pairs; Ni, Wr, partW unused);
 *   mode 1: gH = dL/dH of H = G W, planes (m * Ni + n); Wr float (Nmid, Ni); partW float (Nmid * Ni,
 *           fl_geq_bwd_lanes_wrows(...)) out (per-workgroup partials of sum_m Re(conj(G[m][j]) dL/dH[m][n]), a v_mfma_f32_16x16x4_f32
 *           contraction over the bins).
 * G: the saved response (planes c); b, a: the designed taps, double (3, S, No * Nmid).
 * Outputs: psum float (S * No * Nmid, nbx, 4) -- per section, pair and block the sums of Re(q / B~), Re(q / A~),
 * sin Im(q / B~), sin Im(q / A~) (q = conj(dL/dG) G, B~ / A~ the section polynomials turned by half a sample) -- and
 * pq float (No * Nmid, nbx), the sums of Re(q); nbx = fl_geq_bwd_lanes_blocks(...) (0: shape not taken, use the
 * first-generation entry points).  fl_geq_sections_bwd_lanes reduces them, recovers the third sum from
 * sum Re(q) = (S + T) G0 - S G1 - D G2, forms the tap gradients and runs the design's backward (as fl_geq_sections_bwd_w). */
int fl_geq_bwd_lanes_blocks(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw, int mode);
int fl_geq_bwd_lanes_wrows(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw);
int fl_geq_response_bwd_lanes_c64(int mode, const void* gH, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                  int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                  int bin0, int m_local, void* psum, void* pq, void* partW, void* stream);
int fl_geq_sections_bwd_lanes(const void* gain, int in_kind, const void* psum, const void* pq, int nbx, const void* b, const void* a,
                              double gamma, int nb, int C, const void* consts, void* ggain, const void* partW, int wrows, int wn,
                              void* gW, void* stream);
/* The same pair in double precision (round 5; float64 modules, the reference examples' default dtype): gH, G complex128, Wr,
 * psum, pq, partW, gW double; half the lanes per workgroup.  Two v_fma_f64 per packed float instruction cost the SAME issue
 * time, so the float64 cascade backward runs at the float32 one's speed (the lane-per-bin double kernel takes 3x as long).
 * 2 / 4 / 8 constant-factor columns (16 would spill: fl_geq_bwd_lanes_blocks_f64 answers 0 and the first generation serves). */
int fl_geq_bwd_lanes_blocks_f64(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw, int mode);
int fl_geq_bwd_lanes_wrows_f64(int m_local, int C, int S, int nfft, int bin0, int ppr, int niw);
int fl_geq_response_bwd_lanes_c128(int mode, const void* gH, long g_pitch, const void* G, long h_pitch, const void* b, const void* a,
                                   int S, int No, int Nmid, int Ni, const void* Wr, double gamma, const void* Wd, int nfft,
                                   int bin0, int m_local, void* psum, void* pq, void* partW, void* stream);
int fl_geq_sections_bwd_lanes_f64(const void* gain, int in_kind, const void* psum, const void* pq, int nbx, const void* b, const void* a,
                                  double gamma, int nb, int C, const void* consts, void* ggain, const void* partW, int wrows, int wn,
                                  void* gW, void* stream);
/* test / tuning hook: on = 0 routes every cascade through the first-generation kernels; blocks_per_cu, tile_bins >= 0 set
 * the grid's sizing (negative: unchanged).  Returns the previous `on`. */
int fl_debug_set_cascade_lanes(int on, int blocks_per_cu, int tile_bins);
/* tuning hook: device buffer of 6 int64 per (workgroup, wavefront) receiving the cycles spent per phase (null: off); skip bit 0 /
 * bit 1 leave out the lane-per-bin / lane-per-section phase (wrong results: timing only) */
int fl_debug_set_cascade_stamps(void* device_buffer, int skip);

/* ------------------------------------------------------------------ closed loop
 * Replace torch.linalg.solve(A, B) in system.Recursion.forward (system.py:420-425).
 * Per bin f:  A_f = (one_minus ? I - P[:,:,f] : P[:,:,f]);  if adjoint, A_f := A_f^H;
 *             OUT[b,:,k,f] = A_f^{-1} R[b,:,k,f]
 * LU with partial pivoting, factored ONCE per bin and applied to all B*K right-hand sides
 * (the reference factors the same matrix B times).  P planar: (i*N + j)*p_pitch + f.  N <= 64 (c64) / 32 (c128) in
 * registers (one matrix row per lane); above that one workgroup per bin with the matrix in LDS, up to fl_solve_max_n
 * (138 / 97): a correctness path for sizes torch.linalg.solve accepts and delay networks rarely use. */
int fl_solve_max_n(int f64);
int fl_solve_c64(const void* P, long p_pitch, int one_minus, int adjoint,
                 const void* R, long rs_b, long rs_n, long rs_k,
                 void* OUT, long os_b, long os_n, long os_k,
                 int B, int M, int N, int K, void* stream);
int fl_solve_c128(const void* P, long p_pitch, int one_minus, int adjoint,
                  const void* R, long rs_b, long rs_n, long rs_k,
                  void* OUT, long os_b, long os_n, long os_k,
                  int B, int M, int N, int K, void* stream);

/* The same solve for loops beyond fl_solve_max_n, to fl_solve_ws_max_n() channels (torch.linalg.solve, system.py:425, has no
 * bound): the matrix of a bin lives in a caller-owned workspace in global memory, fl_solve_ws_bytes(N, M, f64) bytes (0 when
 * the size needs none and fl_solve_* serves it; at most 1 GB: one N x (N + 1) slot per resident workgroup, which walks its bins).
 * Any N the plain entry points take is routed as they route it (the workspace is then ignored and may be null). */
int fl_solve_ws_max_n(void);
long fl_solve_ws_bytes(int N, int M, int f64);
int fl_solve_ws_c64(const void* P, long p_pitch, int one_minus, int adjoint,
                    const void* R, long rs_b, long rs_n, long rs_k,
                    void* OUT, long os_b, long os_n, long os_k,
                    int B, int M, int N, int K, void* ws, long ws_bytes, void* stream);
int fl_solve_ws_c128(const void* P, long p_pitch, int one_minus, int adjoint,
                     const void* R, long rs_b, long rs_n, long rs_k,
                     void* OUT, long os_b, long os_n, long os_k,
                     int B, int M, int N, int K, void* ws, long ws_bytes, void* stream);

/* fl_solve_* (one_minus) with a constant row scale of the materialised matrix: A_f = I - diag(l) P[f], l: N complex values
 * l_sn apart.  For loops whose feedforward path ends in per-channel gains (Series(Delay((N,N)), parallelGain(N)) around a
 * mixing matrix, the active-acoustics structure): P = D[f] U is formed once and the gains never make a pass over the
 * (M, N, N) tensor, forward or backward. */
int fl_solve_scaled_c64(const void* P, long p_pitch, const void* l, long l_sn, int adjoint, const void* R, long rs_b, long rs_n,
                        long rs_k, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);
int fl_solve_scaled_c128(const void* P, long p_pitch, const void* l, long l_sn, int adjoint, const void* R, long rs_b, long rs_n,
                         long rs_k, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);

/* The same forward system with its LU factors KEPT, and the adjoint system solved from them -- the two solves of
 * Recursion.forward and its autograd backward (flamo/processor/system.py:420-425: torch.linalg.solve, whose backward solves A^H
 * with a second factorisation).  LU: fl_solve_kept_lu_elems(N, M, f64) complex values, piv: fl_solve_kept_piv_elems int32 -- the
 * pivoted factors (L below the diagonal, U on and above it) and the pivot rows, tiled by the bins of a workgroup; opaque to the
 * caller, valid for the same (N, M, precision).  At N = 32 the elimination is ~90 % of a solve; the adjoint from kept factors
 * is one pass over 8 N^2 bytes per bin.  N <= 64 (c64) / 32 (c128). */
size_t fl_solve_kept_lu_elems(int N, int M, int f64);
size_t fl_solve_kept_piv_elems(int N, int M, int f64);
int fl_solve_scaled_keep_c64(const void* P, long p_pitch, const void* l, long l_sn, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                             long os_b, long os_n, long os_k, int B, int M, int N, int K, void* LU, void* piv, void* stream);
int fl_solve_scaled_keep_c128(const void* P, long p_pitch, const void* l, long l_sn, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                              long os_b, long os_n, long os_k, int B, int M, int N, int K, void* LU, void* piv, void* stream);
int fl_solve_kept_adjoint_c64(const void* LU, const void* piv, const void* R, long rs_b, long rs_n, long rs_k,
                              void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);
int fl_solve_kept_adjoint_c128(const void* LU, const void* piv, const void* R, long rs_b, long rs_n, long rs_k,
                               void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);
/* The same for the feedback delay network between its gains (fl_solve_fdn_*, forward system; reverb.py:117-199 / e8_fdn.py:60-100
 * through system.py:420-425): 8 < N <= 16.  fl_solve_fdn_keep_tile: bins per tile of the kept arrays for this size (0: this size
 * keeps nothing -- call fl_solve_fdn_* both ways); LU: ceil(M / tile) * tile * N^2 complex values, piv: ceil(M / tile) * tile * N
 * int32.  fl_solve_kept_adjoint_rank1_*: OUT = A^-H (conj(rv) . rs), the backward's adjoint system with the output-gain row rv
 * (N values, real or complex) times the output's gradient rs (B, M) as its right-hand side. */
int fl_solve_fdn_keep_tile(int N, int f64);
/* The forward FDN solve with the adjoint system's solution for the output-gain row beside it: wadj[n*wadj_sn + f] =
 * (A[f]^-H cw^H)[n], from the factors of the same launch.  A network with one output channel has the backward right-hand side
 * cw^H gy[b][f] -- the same vector times a scalar -- so A^-H (cw^H gy) = wadj . gy and Recursion's backward
 * (system.py:420-425 under autograd, between the two gains of reverb.py:117-199 / e8_fdn.py:60-100) needs no solve.
 * 4 < N <= 16 (fl_solve_fdn_wadj_supported); everything else as fl_solve_fdn_* with adjoint = 0. */
int fl_solve_fdn_wadj_supported(int N);
/* fl_solve_dud2_grads_c64 with the adjoint solution given in that form: gR[b][n][f] = W[n*w_sn + f] gy[b*gy_sb + f], formed where
 * the kernel consumes it (one column per batch item). */
int fl_solve_dud2_grads_w_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                              long r_sn, long r_sf, const void* W, long w_sn, const void* gy, long gy_sb, const void* OUT, long s_b,
                              long s_n, long s_k, int B, int M, int N, void* gl, long gl_sn, void* gr, long gr_sn, void* partU,
                              void* gU, void* gR0, const void* sx, long sx_b, const void* sy, long sy_b, void* g_side_real,
                              void* stream);
int fl_solve_dud2_grads_w_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                              long r_sn, long r_sf, const void* W, long w_sn, const void* gy, long gy_sb, const void* OUT, long s_b,
                              long s_n, long s_k, int B, int M, int N, void* gl, long gl_sn, void* gr, long gr_sn, void* partU,
                              void* gU, void* gR0, const void* sx, long sx_b, const void* sy, long sy_b, void* g_side_real,
                              void* stream);
int fl_solve_fdn_wadj_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                          long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                          void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* wadj, long wadj_sn,
                          void* stream);
int fl_solve_fdn_wadj_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                          long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                          void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* wadj, long wadj_sn,
                          void* stream);
int fl_solve_fdn_keep_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                          long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                          void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* LU, void* piv,
                          void* stream);
int fl_solve_fdn_keep_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                           long r_sn, long r_sf, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw, int cw_real,
                           void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* LU, void* piv,
                           void* stream);
int fl_solve_kept_adjoint_rank1_c64(const void* LU, const void* piv, int tile_bins, const void* rv, int rv_real, const void* rs, long rs_sb,
                                    void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream);
int fl_solve_kept_adjoint_rank1_c128(const void* LU, const void* piv, int tile_bins, const void* rv, int rv_real, const void* rs, long rs_sb,
                                     void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream);

/* Same solve with the loop matrix given in the factored form every feedback delay network has
 * (reverb.py:117-199, e8_fdn.py:60-100: delays and attenuation filters are diagonal, only the
 * mixing matrix is full):  A_f = I - diag(l[:,f]) U diag(r[:,f]),  U frequency independent (N x N,
 * row major), l / r per-bin (element (n,f) at n*sn + f*sf), constant (sf = 0) or absent (NULL = 1).
 * A is built in registers from 2N values per bin; the N^2-per-bin matrix of system.py:420-424 is
 * never formed.  adjoint != 0 solves with A_f^H (backward). */
int fl_solve_dud_c64(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, int adjoint,
                     const void* R, long rs_b, long rs_n, long rs_k, void* OUT, long os_b, long os_n, long os_k,
                     int B, int M, int N, int K, void* stream);
int fl_solve_dud_c128(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, int adjoint,
                      const void* R, long rs_b, long rs_n, long rs_k, void* OUT, long os_b, long os_n, long os_k,
                      int B, int M, int N, int K, void* stream);
/* Backward of the factored solve in one pass (replaces the five launches of the layered form in autograd over
 * system.py:420-424: two diagonal products, an outer-product reduction, a matrix product, a per-bin reduction).
 * gR = A^-H g (fl_solve_dud with adjoint) and OUT = A^-1 R share the strides (s_b, s_n, s_k), bins contiguous.
 *   gU[i][j]  = sum_{f,b,k} conj(l_i) gR_i conj(r_j out_j)        (N x N; needs partU: fl_solve_dud_grads_blocks(M,N) x N x N)
 *   gl[i][f]  = sum_{b,k} gR_i conj((U (r . out))_i)               (per-bin l only; rows gl_sn apart)
 *   gr[j][f]  = sum_{b,k} (U^H (conj(l) . gR))_j conj(out_j)       (per-bin r only)
 * Any of gl, gr, (partU, gU) may be NULL.  Deterministic (fixed-order reductions). */
int fl_solve_dud_grads_blocks(int M, int N);
int fl_solve_dud_grads_c64(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, const void* gR,
                           const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N, int K, void* gl, long gl_sn, void* gr,
                           long gr_sn, void* partU, void* gU, void* stream);
int fl_solve_dud_grads_c128(const void* l, long l_sn, long l_sf, const void* U, const void* r, long r_sn, long r_sf, const void* gR,
                            const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N, int K, void* gl, long gl_sn, void* gr,
                            long gr_sn, void* partU, void* gU, void* stream);
/* The factored solve and its one-pass backward with TWO left factors, l = l (.) l2 (l may be NULL), and optionally the
 * right-hand side scaled by l2 as well (rhs_l2, forward system): the loop of a feedback delay network has
 * P = diag(delays (.) attenuation) U and R = delays (.) (input gains x) (system.py:417-424 over reverb.py:117-199) -- the
 * delay factor is applied by the kernels where they load l and R instead of two diagonal-product launches forward and two
 * backward.  l2 carries no gradient.  In the backward: gl is the gradient of the FIRST factor (sum gR conj(U(r.out)) times
 * conj(l2)), and gR0 (NULL = not wanted; same layout as gR) = conj(l2) (.) gR, the gradient of the unscaled R. */
int fl_solve_dud2_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, int rhs_l2, const void* U,
                      const void* r, long r_sn, long r_sf, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                      long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);
int fl_solve_dud2_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, int rhs_l2, const void* U,
                       const void* r, long r_sn, long r_sf, int adjoint, const void* R, long rs_b, long rs_n, long rs_k, void* OUT,
                       long os_b, long os_n, long os_k, int B, int M, int N, int K, void* stream);
/* Side reductions (sx, sy non-NULL; K = 1): when the loop sits between an input-gain column b (R0 = b x) and an output-gain row
 * c (y = c . OUT) -- Series(Gain(N,1), Recursion, Gain(1,N)), reverb.py:117-199 -- pass x as sx and gy as sy (element (b, f)
 * at b*s_b + f): partU is then (blocks, N*N + 2N) and gU (N*N + 2N) = [gU | g_b | g_c] with
 *   g_b[i] = sum conj(l2_i) gR_i conj(x),   g_c[i] = sum gy conj(out_i)
 * -- the two gains' gradients without their own bin-reduction launches.  g_side_real (may be NULL): for REAL gain vectors,
 * a real (2N) array that receives (Re g_b, Re g_c) instead of the tail of gU. */
int fl_solve_dud2_grads_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                            long r_sn, long r_sf, const void* gR, const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N,
                            int K, void* gl, long gl_sn, void* gr, long gr_sn, void* partU, void* gU, void* gR0, const void* sx,
                            long sx_b, const void* sy, long sy_b, void* g_side_real, void* stream);
int fl_solve_dud2_grads_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                             long r_sn, long r_sf, const void* gR, const void* OUT, long s_b, long s_n, long s_k, int B, int M, int N,
                             int K, void* gl, long gl_sn, void* gr, long gr_sn, void* partU, void* gU, void* gR0, const void* sx,
                             long sx_b, const void* sy, long sy_b, void* g_side_real, void* stream);
/* The FDN loop between an input-gain column and an output-gain row (Series(Gain(N,1), Recursion, Gain(1,N)),
 * reverb.py:117-199): fl_solve_dud2 with the right-hand side built in the kernel, R_i = rv_i rs[b][f] (rs: the one-channel
 * spectrum, element (b, f) at b*rs_sb + f; rv: N gains, real values of the signal's precision if rv_real else complex;
 * conjugated for the adjoint system, scaled by l2 for the forward one), and -- forward system, cz non-NULL -- the contracted
 * output z[b][f] = sum_i cw_i OUT_i written beside OUT.  In-place kernels only: N <= 32 (c64) / 16 (c128). */
int fl_solve_fdn_c64(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                     long r_sn, long r_sf, int adjoint, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw,
                     int cw_real, void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream);
int fl_solve_fdn_c128(const void* l, long l_sn, long l_sf, const void* l2, long l2_sn, long l2_sf, const void* U, const void* r,
                      long r_sn, long r_sf, int adjoint, const void* rv, int rv_real, const void* rs, long rs_sb, const void* cw,
                      int cw_real, void* cz, long cz_sb, void* OUT, long os_b, long os_n, long os_k, int B, int M, int N, void* stream);
/* tuning/test hook: 0 (default) = N <= 16 factor with rows exchanged in place (compile-time DPP broadcasts,
 * threshold pivoting; two rows per lane for the factored loop at N in (4, 16]); 1 = the shuffle kernel with implicit
 * partial pivoting for every N; 4 = the in-place kernels with one row per lane */
int fl_debug_set_solve_variant(int variant);

/* ------------------------------------------------------------------------------------------------
 * Scalar objective on the output of the path: the training step reduces the model output to one
 * scalar and back-propagates it (flamo/optimize/trainer.py:179-191).  For the mean-square
 * objective  loss = mean(y^2)  (torch: (y ** 2).mean(), five launches and eight passes over y)
 *   fl_mean_square_*      loss[0] = (1/(rows*cols)) sum_{r,c} y[r*pitch + c]^2        (one read of y)
 *   fl_mean_square_bwd_*  gy[r*pitch + c] = (2/(rows*cols)) * gloss[0] * y[r*pitch + c]
 * y is real, rows of `cols` samples `pitch` apart (any layout the irfft leaves behind; padding
 * is neither read nor written).  scratch: fl_mean_square_scratch_bytes() bytes private to one
 * stream (per-block partials in double, combined in a fixed order by a second one-block launch:
 * the value is deterministic).  loss/gloss are device scalars of y's type. */
size_t fl_mean_square_scratch_bytes(void);
int fl_mean_square_f32(const void* y, long rows, long cols, long pitch, void* loss, void* scratch, void* stream);
int fl_mean_square_f64(const void* y, long rows, long cols, long pitch, void* loss, void* scratch, void* stream);
/* loss[0] = inv_count * sum of n_parts partial sums (double), added in a fixed order by one workgroup */
int fl_mean_square_final_f32(const void* parts, int n_parts, double inv_count, void* loss, void* stream);
int fl_mean_square_final_f64(const void* parts, int n_parts, double inv_count, void* loss, void* stream);
/* Gradient buckets of a replayed training step (the data-parallel gradient sum of trainer.py:172-191's loop over several
 * GPUs; the reference has no counterpart -- it trains on one device): copies `count` tensors into ONE of two flat buffers,
 * alternating from launch to launch by a counter kept on the device, so that the call can be a node of a captured HIP graph
 * and the all-reduce of step k may still run on its bucket while replay k + 1 fills the other.
 *   table: device array of count x 3 uint64 {source address, byte offset inside the bucket, bytes} (multiples of 4),
 *   state: device int32[2], zero before the first launch: [0] launches done (bucket of launch i is i & 1), [1] internal. */
int fl_pack_toggle(const void* table, int count, void* flat0, void* flat1, void* state, void* stream);
int fl_mean_square_bwd_f32(const void* y, const void* gloss, void* gy, long rows, long cols, long pitch, void* stream);
int fl_mean_square_bwd_f64(const void* y, const void* gloss, void* gy, long rows, long cols, long pitch, void* stream);
/* Magnitude of a spectrum, the output layer of the reference's magnitude-domain examples -- dsp.Transform(lambda x: torch.abs(x)),
 * examples/e7_biquad.py:76, e8_colorless_fdn.py:102 (flamo/processor/dsp.py:27-66 wraps the callable):
 *   out[r][c] = |z[r][c]|,   g_z[r][c] = g[r][c] z[r][c] / |z[r][c]|  (0 where z = 0, as torch's sgn),
 * rows of `cols` contiguous values, `pitch` (complex) / `opitch`, `gpitch` (real) apart: contiguous tensors (rows = 1) and the
 * bin-planar views of this library alike.  One launch each way (torch: abs; sgn and a complex multiply). */
int fl_cabs_c64(const void* z, void* out, long rows, long cols, long pitch, long opitch, void* stream);
int fl_cabs_c128(const void* z, void* out, long rows, long cols, long pitch, long opitch, void* stream);
int fl_cabs_bwd_c64(const void* z, const void* g, void* gz, long rows, long cols, long pitch, long gpitch, void* stream);
int fl_cabs_bwd_c128(const void* z, const void* g, void* gz, long rows, long cols, long pitch, long gpitch, void* stream);
/* Sparsity criterion of a mixing matrix, the second criterion of the colorless-FDN training (flamo/optimize/loss.py:12-63
 * `sparsity_loss`, examples/e8_colorless_fdn.py:138):
 *   loss = mean_c (sum_ij |A_c[i][j]| - N sqrt(N)) / (N (1 - sqrt(N))),   A: (C, N, N) contiguous (C = 1: the plain matrix),
 *   g_A = gloss sign(A) / (C N (1 - sqrt(N)))  (gloss a device scalar; sign(0) = 0 as torch.sign).
 * One launch each way instead of torch's abs / sum / sub / div / neg launches and their backward. */
int fl_sparsity_f32(const void* A, int C, int N, void* loss, void* stream);
int fl_sparsity_f64(const void* A, int C, int N, void* loss, void* stream);
int fl_sparsity_bwd_f32(const void* A, const void* gloss, int C, int N, void* gA, void* stream);
int fl_sparsity_bwd_f64(const void* A, const void* gloss, int C, int N, void* gA, void* stream);
/* Mean squared error against a target, the criterion the reference's training loops use (flamo/optimize/trainer.py:179-189
 * calling flamo/optimize/loss.py:66-103 `mse_loss`, or nn.MSELoss directly as examples/e7_biquad.py:82-87):
 *   loss = (1 / rows) sum_r (sum_{c < ncols} y[r][c] - t[r])^2,   y: (rows, ncols) contiguous, t: (rows).
 * ncols = 1: nn.MSELoss()(y, t) on equal shapes; ncols = N_out: loss.py:101-102 (the prediction summed over its last axis).
 * One streaming pass each way: g_y[r][c] = (2 gloss / rows) (sum_c y[r][c] - t[r]).  scratch as fl_mean_square. */
int fl_mse_f32(const void* y, const void* t, long rows, int ncols, void* loss, void* scratch, void* stream);
int fl_mse_f64(const void* y, const void* t, long rows, int ncols, void* loss, void* scratch, void* stream);
int fl_mse_bwd_f32(const void* y, const void* t, const void* gloss, void* gy, long rows, int ncols, void* stream);
int fl_mse_bwd_f64(const void* y, const void* t, const void* gloss, void* gy, long rows, int ncols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Orthogonal parameter map of dsp.Matrix: E = exp(A), A = X (skew = 0) or
 * A = triu(X,1) - triu(X,1)^T (skew != 0) -- torch.matrix_exp(skew_matrix(x)),
 * flamo/processor/dsp.py:649 with flamo/functional.py:42-56.  One workgroup, float64 arithmetic,
 * fixed scaling-and-squaring schedule (2^-10 scaling, order-10 series: |A| up to ~100 agrees with
 * torch.matrix_exp to 1e-13), no host synchronisation, so a training step stays capturable in a
 * HIP graph.  X, E, gE, gX: (N, N) row major in the parameter type (_f32 / _f64), N <= 64;
 * stash: fl_matrix_exp_stash_elems(N) doubles written by the forward and read by the backward. */
size_t fl_matrix_exp_stash_elems(int N);
/* test hook: 0 = the LDS kernels also at N = 16 (default 1: one wavefront on v_mfma_f64_16x16x4_f64, csrc/expm.hip); returns
 * the previous setting, a negative argument only queries */
int fl_debug_set_expm_mfma(int on);
int fl_matrix_exp_f32(const void* X, int N, int skew, void* E, void* stash, void* stream);
int fl_matrix_exp_f64(const void* X, int N, int skew, void* E, void* stash, void* stream);
int fl_matrix_exp_bwd_f32(const void* gE, int N, int skew, const void* stash, void* gX, void* stream);
int fl_matrix_exp_bwd_f64(const void* gE, int N, int skew, const void* stash, void* gX, void* stream);
/* The same pair with the result stored as the complex matrix (re, 0) that the per-bin kernels take -- E: 2 N^2 values --
 * and the backward reading the real part of a complex gradient gE (2 N^2 values): the real <-> complex passes between
 * the parameter map and the solve / product kernels (dsp.py:466-468's cast and its backward) never run. */
int fl_matrix_exp_cplx_f32(const void* X, int N, int skew, void* E, void* stash, void* stream);
int fl_matrix_exp_cplx_f64(const void* X, int N, int skew, void* E, void* stash, void* stream);
int fl_matrix_exp_bwd_cplx_f32(const void* gE, int N, int skew, const void* stash, void* gX, void* stream);
int fl_matrix_exp_bwd_cplx_f64(const void* gE, int N, int skew, const void* stash, void* gX, void* stream);
/* Both forms from one launch (E real N^2, Ec complex 2 N^2; either may be NULL), and the backward of both (gE real, gEc
 * complex, either may be NULL: dL/dX from gE + Re gEc): a step in which the model takes the complex matrix and a criterion
 * the real one (sparsity of the mixing matrix, optimize/loss.py:36-63) evaluates the map once each way. */
int fl_matrix_exp_both_f32(const void* X, int N, int skew, void* E, void* Ec, void* stash, void* stream);
int fl_matrix_exp_both_f64(const void* X, int N, int skew, void* E, void* Ec, void* stash, void* stream);
int fl_matrix_exp_bwd_both_f32(const void* gE, const void* gEc, int N, int skew, const void* stash, void* gX, void* stream);
int fl_matrix_exp_bwd_both_f64(const void* gE, const void* gEc, int N, int skew, const void* stash, void* gX, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-bin eigenvalues of small general complex matrices: torch.linalg.eigvals in
 * flamo.functional.get_eigenvalues (flamo/functional.py:24-39), applied by the active-acoustics loss to
 * the (B, F, N, N) loop matrices the path produces (examples/e8_active_acoustics.py:586-603).
 *   A: planar (N, N, a_pitch), element (i, j, f) at (i*N + j)*a_pitch + f, f < M;  N <= 64
 *   lam: (N, l_pitch) eigenvalues of every bin (order: position on the diagonal of the Schur form --
 *        NOT LAPACK's order; compare as sets)
 *   V:   NULL, or planar (N, N, v_pitch) right eigenvectors as columns, unit 2-norm (needed by the
 *        backward: g_A = V^-H diag(g_lambda) V^H, one fl_solve per bin)
 *   info: NULL, or int[M]: 0, or k > 0 if the QR iteration gave up with k eigenvalues unconverged.
 * One wavefront per matrix (Householder Hessenberg reduction, explicitly shifted QR with Wilkinson
 * shifts to Schur form, back substitution for the vectors), matrices in LDS. */
int fl_eig_c64(const void* A, long a_pitch, int N, int M, void* lam, long l_pitch, void* V, long v_pitch, void* info,
               void* stream);
int fl_eig_c128(const void* A, long a_pitch, int N, int M, void* lam, long l_pitch, void* V, long v_pitch, void* info,
                void* stream);

/* Launch pairs: two independent kernels of the forward pass in ONE grid -- the cascade-times-matrix response
 * (fl_geq_response_rc_c64 / fl_sos_response_rc_c64: parameters only; the reference's get_freq_response of flamo/processor/dsp.py:
 * 2563-2593 behind dsp.Matrix's map, dsp.py:649) beside the column pass of the input's transform (fl_spec_cols_fwd_f32: the first
 * half of torch.fft.rfft, dsp.py:88).  Between begin and flush on the calling thread the response launch is RECORDED; the next
 * fl_spec_cols_fwd_f32 issues both (shapes it does not take: the recorded launch first, alone); flush issues a launch that is
 * still recorded and ends the mode.  The caller guarantees that nothing reads the response's outputs in between.
 * fl_launch_pair_pending: 1 while a launch is recorded.  FLAMO_LAUNCH_PAIR=0 turns recording off. */
int fl_launch_pair_begin(void);
int fl_launch_pair_pending(void);
long fl_debug_launch_pair_count(void);      /* grids issued with both roles so far (process-wide) */
/* tuning: every workgroup w of the next grids with both roles leaves (start, end by the device's constant-rate clock, role
 * 0 = column pass / 1 = response, HW_ID | XCC_ID << 32) in buf[4 w .. 4 w + 3] (int64; w < 8192), and response block (bx, m) its
 * phase stamps (after the design, the tables, the cascades, when its operands arrived; the plain launch: start, end) in
 * buf[32768 + 8 (4096 m + bx) ..] -- 32768 + 8 * 8 * 4096 entries for eight output rows.  NULL: off. */
int fl_debug_set_pair_stamps(void* buf);
int fl_launch_pair_flush(void* stream);

/* Cache policy of the pipeline's data streams (process-wide mask; csrc/common.h: enum StreamPolicy names the bits -- one per
 * stream of the fused Shell pipeline: default or non-temporal loads / stores) and the launch SITE of the calling thread's next
 * column passes (0: the forward transform of the input, 1: the gradient's transform, whose input the previous launch wrote).
 * mask 0xFFFFFFFF / site < 0 leave the respective setting; returns the mask in force.  A tuning knob of the measurement
 * (tools/dbg/policy_sweep.py), no reference call site; the library's default is the measured best. */
int fl_set_stream_policy(unsigned mask, int site);

/* ------------------------------------------------------------------------------------------------
 * Measurement only (no reference call site: the reference has no device code).  What the memory system sustains on a
 * hand-written persistent streaming kernel -- the ceiling bench.py's "device" object states beside the 8 TB/s
 * specification and the roofline fractions of the HBM-bound passes (torch.fft.rfft / einsum / irfft of
 * flamo/processor/dsp.py:88, 114, 922-924) are read against.
 *   kind 0: read `bytes` from src (one float per workgroup written to partials[workgroups]);  1: write `bytes` to dst;
 *   2: copy src -> dst;  3: read `bytes`, write bytes/8 (the 8:1 mix of the response-gradient pass).
 *   bytes: multiple of 32 KiB;  workgroups: persistent grid (256 threads each, 16 bytes per lane and access, eight in flight);
 *   flags: bit 0 non-temporal loads, bit 1 non-temporal stores, bit 2 walk the buffer from its end (consumer-first order),
 *          bit 3 eight bytes per lane and access instead of sixteen (kinds 0 - 2, default policy). */
int fl_hbm_probe(int kind, const void* src, void* dst, size_t bytes, int workgroups, int flags, void* partials, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLAMO_HIP_H */
