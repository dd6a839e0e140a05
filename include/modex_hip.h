/*
 * modex_hip.h -- C ABI of libmodex_hip.so: the MI355X (gfx950) kernels behind the
 * mod_extraction hot path.
 *
 * The reference (christhetree/mod_extraction) is pure Python and has no FFI of its own; the
 * boundary a maintainer binds is this flat C ABI, called from the Python classes that mirror the
 * reference's modules (see INTEGRATION.md for the ctypes stub).  Conventions:
 *
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch tensor .data_ptr());
 *     nothing is allocated, freed or retained by the library; no global state
 *   - tensors are dense, row-major, float32 unless stated otherwise
 *   - `stream` is a hipStream_t (0 / NULL = default stream); calls are asynchronous and
 *     re-entrant per stream; the library never synchronises
 *   - return value: 0 = launched, MX_ERR_ARG (-1) bad argument, MX_ERR_UNSUPPORTED (-2) size not
 *     supported (e.g. delay line > LDS), MX_ERR_LAUNCH (-3) HIP launch error.  Never throws.
 *
 * Each entry point cites the reference code it replaces (file:line in /root/reference).
 */
#ifndef MODEX_HIP_H
#define MODEX_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MX_OK 0
#define MX_ERR_ARG (-1)
#define MX_ERR_UNSUPPORTED (-2)
#define MX_ERR_LAUNCH (-3)

/* ABI version, bumped on any signature change. */
int mx_abi_version(void);

/* ---- K1: LFO synthesis -- mod_extraction/modulations.py:16-57 (make_mod_signal) -------------
 * One row per LFO.  freq, phase, exp: (B,) float32; shape: (B,) int32 in
 * {0 cos, 1 rect_cos, 2 inv_rect_cos, 3 tri, 4 saw, 5 rsaw, 6 sqr} (NULL = cos); exp NULL = 1;
 * start: (B,) int32 sample offset into a longer signal (NULL = 0; datasets.py:442-449 crops the
 * phaser ground truth that way).  The signal has n_src points at rate sr; if n_out != n_src it
 * is resampled with util.py:15-29 (linear, align_corners=True).  out: (B, n_out). */
int mx_lfo_synth(const float *freq, const float *phase, const int32_t *shape, const float *exp,
                 const int32_t *start, int64_t B, int64_t n_src, int64_t n_out, float sr,
                 float *out, void *stream);

/* ---- util.py:15-29 (linear_interpolate_last_dim, align_corners=True): (rows,n_in)->(rows,n_out) */
int mx_interp_linear(const float *x, int64_t rows, int64_t n_in, int64_t n_out, float *y,
                     void *stream);

/* ---- K2: flanger / chorus -- mod_extraction/fx.py:72-119 (MonoFlangerChorusModule.apply_effect)
 * x (B,N) mono clips; mod (B,n_mod), n_mod == N or shorter (resampled in-kernel).
 * Per-clip float32 constants, each (B,):
 *   lfo_scale     = max_lfo_delay_samples * width            (fx.py:99)
 *   min_delay     = min_delay_width * max_min_delay_samples  (fx.py:98)
 *   feedback, depth, mix, one_minus_mix                      (fx.py:114-117)
 * max_delay (B,) int32: delay-line length M per clip (fx.py:42), max_delay_max = max over the
 * batch (<= 40000).  rows/n_rows: optional subset of clip indices to process (NULL = all B).
 * y (B,N) out, clipped to [-1,1].  Optional (NULL to skip): mod_up (B,N) resampled LFO,
 * dbg_prev (B,N) int64 and dbg_frac (B,N) = prev_idx_all / delay_read_fraction_all of
 * fx.py:101-102 for index-parity tests. */
int mx_flanger_fwd(const float *x, const float *mod, int64_t n_mod, const float *lfo_scale,
                   const float *min_delay, const float *feedback, const float *depth,
                   const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                   int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B, int64_t N,
                   float *y, float *mod_up, int64_t *dbg_prev, float *dbg_frac, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MODEX_HIP_H */
