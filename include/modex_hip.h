/*
 * modex_hip.h -- C ABI of libmodex_hip.so: the MI355X (gfx950) kernels behind the
 * mod_extraction hot path.
 *
 * The reference (christhetree/mod_extraction) is pure Python and has no FFI of its own; the
 * boundary a maintainer binds is this flat C ABI, called from the Python classes that mirror the
 * reference's modules (see INTEGRATION.md for the ctypes stub).  Conventions:
 *
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch tensor .data_ptr());
 *     nothing is allocated, freed or retained by the library; no caller-observable global state
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
/* The library keeps no caller-observable mutable state: every entry point is a function of its arguments and its stream.
 * (Internally: per-device latches of the dynamic-LDS function attribute, and kernel-variant knobs read once from MODEX_*
 * environment variables; neither changes a result.) */

/* ---- K1: LFO synthesis -- mod_extraction/modulations.py:16-57 (make_mod_signal) -------------
 * One row per LFO.  freq, phase, exp: (B,) float32; shape: (B,) int32 in
 * {0 cos, 1 rect_cos, 2 inv_rect_cos, 3 tri, 4 saw, 5 rsaw, 6 sqr} (NULL = cos); exp NULL = 1;
 * start: (B,) int32 sample offset into a longer signal (NULL = 0; datasets.py:442-449 crops the
 * phaser ground truth that way).  The signal has n_src points at rate sr; if n_out != n_src it
 * is resampled with util.py:15-29 (linear, align_corners=True).  out: (B, n_out). */
int mx_lfo_synth(const float *freq, const float *phase, const int32_t *shape, const float *exp,
                 const int32_t *start, int64_t B, int64_t n_src, int64_t n_out, float sr,
                 float *out, void *stream);

/* Synthetic dry audio of the benchmark / test batches (no reference counterpart: the reference trains on recorded
 * guitar): uniform noise in [lo, hi), Philox-4x32-10 keyed by (seed; sample, clip, counter).  Row of clip b at out + b * stride;
 * rows: optional list of n_rows clip indices (NULL = 0 .. n_rows - 1); length of a row = (lens ? lens[clip] : 0) + len_add,
 * at most max_len. */
int mx_uniform_rows(float *out, int64_t stride, const int32_t *rows, int64_t n_rows, const int32_t *lens, int64_t len_add,
                    int64_t max_len, uint64_t seed, uint32_t counter, float lo, float hi, void *stream);

/* ---- util.py:15-29 (linear_interpolate_last_dim, align_corners=True): (rows,n_in)->(rows,n_out) */
int mx_interp_linear(const float *x, int64_t rows, int64_t n_in, int64_t n_out, float *y,
                     void *stream);

/* Transpose of mx_interp_linear restricted to a window of the output axis: dy (rows, j_len) = d loss / d y[:, j0 : j0 + j_len]
 * (zero elsewhere) -> dx (rows, n_in).  Used by an UNFROZEN LFO model inside the TBPTT step (mod_extraction/lightning.py:
 * 361-366: each step back-propagates through its own chunk of the resampled LFO).  Deterministic gather, no atomics. */
int mx_interp_linear_bwd(const float *dy, int64_t dy_stride, int64_t rows, int64_t n_in, int64_t n_out, int64_t j0,
                         int64_t j_len, float *dx, void *stream);

/* ---- K2: flanger / chorus -- mod_extraction/fx.py:72-119 (MonoFlangerChorusModule.apply_effect)
 * x: mono clips, row b at x + b*x_stride (N samples each; x_stride = N for a dense (B,N) tensor);
 * mod (B,n_mod), n_mod == N or shorter (resampled in-kernel).
 * Per-clip float32 constants, each (B,):
 *   lfo_scale     = max_lfo_delay_samples * width            (fx.py:99)
 *   min_delay     = min_delay_width * max_min_delay_samples  (fx.py:98)
 *   feedback, depth, mix, one_minus_mix                      (fx.py:114-117)
 * max_delay (B,) int32: delay-line length M per clip (fx.py:42), max_delay_max = max over the
 * batch.  The delay line lives in LDS: max_delay_max (+ n_mod when the LFO is resampled in-kernel) <= 34784 floats
 * (FL_MAX_M = the CU's 160 KB minus the 24.7 KB record ring of eight 64-sample rows per chunk; it was 37872 with four
 * rows per chunk) = 789 ms at 44.1 kHz -- the shipped configs need 11 ms (flanger) / 40 ms (chorus); longer lines
 * return MX_ERR_UNSUPPORTED.   rows/n_rows: optional subset of clip indices to process (NULL = all B).
 * y: row b at y + b*y_stride, N samples, clipped to [-1,1].  Optional (NULL to skip): mod_up (B,N) resampled LFO,
 * dbg_prev (B,N) int64 and dbg_frac (B,N) = prev_idx_all / delay_read_fraction_all of
 * fx.py:101-102 for index-parity tests. */
int mx_flanger_fwd(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                   const float *min_delay, const float *feedback, const float *depth,
                   const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                   int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B, int64_t N,
                   float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac, void *stream);
/* Measurement twin of mx_flanger_fwd (bench.py, SURVEY.md section 8d "measured serial floor"): the same launch with NO
 * global-memory traffic inside the sample loop (constant inputs, only the last chunk stored), i.e. the kernel's
 * dependent chain alone.  Outputs are meaningless; never called by the product.  No reference counterpart. */
int mx_flanger_fwd_probe(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                   const float *min_delay, const float *feedback, const float *depth,
                   const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                   int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B, int64_t N,
                   float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac, void *stream);

/* Measurement aid (bench.py): `steps` dependent LDS round trips of the flanger lock-step's shape (two ds_read_b32 of the
 * slot the previous step wrote, the five fp32 operations of fx.py:113-115, one ds_write_b32) on one wavefront, nothing
 * else.  Its time per step x the lock-steps of a clip is a floor of mx_flanger_fwd that does not come from that kernel.
 * out: 1 float.  No reference counterpart. */
int mx_lds_roundtrip_probe(int64_t steps, float *out, void *stream);

/* ---- K3: phaser -- call site mod_extraction/datasets.py:455-482 (pedalboard==0.7.3 Phaser = JUCE
 * dsp::Phaser<float>: 6 first-order TPT all-pass stages + feedback, sine LFO at sr/4 on a log
 * frequency axis, linear dry/wet mix), then clip to [-1,1] (datasets.py:472).  Third-party
 * algorithm restated from its published source: parity unpinned.
 * x: source audio, row b at x + b*x_stride holding lead[b] + N samples; rate, depth, centre,
 * feedback, mix (B,) fp32; lead (B,) int32 = samples rendered before the output window (the
 * reference renders n + sr/rate samples and crops at a random offset, datasets.py:428-449), NULL = 0;
 * rows/n_rows: optional subset of clips.  y: row b at y + b*y_stride = processed[lead:lead+N];
 * dry_out (optional, same stride): the matching crop of x.  exact_order != 0: every sample in JUCE's operation order on
 * one wavefront per clip (the bit reference); 0: the same recurrence as a linear scan over time (512 chunks per clip run
 * in parallel, their affine maps chained; results within 1e-6).  workspace (optional): floats, row i of the processed
 * clips at workspace + i * workspace_stride, >= ceil((lead + N) / 4) each (the cut-off of every 4-sample group, kept
 * between the two passes of the scan; without it the cut-offs are evaluated twice). */
int mx_phaser_fwd(const float *x, int64_t x_stride, const float *rate, const float *depth,
                  const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                  const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                  float *y, int64_t y_stride, float *dry_out, float *workspace, int64_t workspace_stride, void *stream);
/* Measurement twin of mx_phaser_fwd (bench.py, SURVEY.md section 8d "measured serial floor"): the same launch with NO
 * global-memory traffic inside the sample loop (constant inputs, only the last chunk stored), i.e. the kernel's
 * dependent chain alone.  Outputs are meaningless; never called by the product.  No reference counterpart. */
int mx_phaser_fwd_probe(const float *x, int64_t x_stride, const float *rate, const float *depth,
                  const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                  const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                  float *y, int64_t y_stride, float *dry_out, float *workspace, int64_t workspace_stride, void *stream);

/* ---- K4: log-mel front end -- mod_extraction/models.py:170-181,199-208
 * (torchaudio MelSpectrogram: n_fft in {512, 1024, 2048} -- every shipped config: 1024 --, hann, centre/reflect, power 2,
 * mel filter bank `fb`; other n_fft: MX_ERR_UNSUPPORTED)
 * x (planes, N), planes = B*in_ch; window (n_fft,); twiddle (n_fft, 2) = exp(-2 pi i m / n_fft);
 * fb (n_fft/2+1, n_mels) row-major; band_lo/band_hi (n_mels,) int32 non-zero row range per mel band.
 * out (planes, n_mels, out_pitch) = log(clip(mel, eps)); out_pitch >= n_frames (352 for 345);
 * columns >= n_frames are zero.  SpecAugment: mel rows [f0,f1) and frames [t0,t1) are set to 0
 * before clip/log (0,0 = no mask). */
int mx_logmel_fwd(const float *x, int64_t planes, int64_t N, const float *window, const float *twiddle,
                  const float *fb, const int32_t *band_lo, const int32_t *band_hi, int64_t n_fft,
                  int64_t hop, int64_t n_mels, int64_t n_frames, int64_t out_pitch, float eps, int32_t f0,
                  int32_t f1, int32_t t0, int32_t t1, float *out, void *stream);

/* ---- K5/K6: one Spectral2DCNN block -- mod_extraction/models.py:183-195
 * Activation planes are (B, C, H, 352) fp32 (345 valid columns).  Weight packing:
 * flip=0: (Cout,Cin,5,13) -> [ci][kh][kw][co] (forward); flip=1 -> [co][4-kh][12-kw][ci] (dgrad). */
int mx_conv_pack_weights(const float *W, int64_t Cout, int64_t Cin, int32_t flip, float *wt, void *stream);

/* mean / rstd (eps inside the sqrt, biased variance) of f(x) per (b,c) plane over H x Wv;
 * f = PReLU(slope[c]) if slope != NULL else identity.  stats (B, C, 2). */
int mx_plane_stats(const float *x, const float *slope, int64_t B, int64_t C, int64_t H, int64_t Wv,
                   float eps, float *stats, void *stream);
/* The same statistics from the per-row partial sums the f16x3 forward kernels below leave when given slope_out /
 * stats_part: part (B, H, C, 2) = {sum, sum of squares} of PReLU(out) - PReLU(bias_c) over the Wv valid columns of one
 * pooled row (the shift keeps the fp32 row sums free of a common offset: no cancellation for near-constant planes);
 * fp64 across the rows.  bias (C,) / slope (C,): the SAME bias and slope_out the forward kernel was given.
 * The LayerNorm of models.py:186 of the NEXT block without re-reading the plane. */
int mx_plane_stats_finish(const float *part, const float *bias, const float *slope, int64_t B, int64_t C, int64_t H,
                          int64_t Wv, float eps, float *stats, void *stream);

/* LayerNorm -> Conv2d(5x13, dilation (1,dilation), same) -> +bias -> MaxPool(2,1), fused.
 * in (B,Cin,H,352): log-mel (first_layer=1) or the previous block's pooled pre-activations (their
 * PReLU, slope (Cin,), is applied on the fly).  out (B,64,H/2,352) pooled PRE-activations,
 * out_amax (B,64,H/2,352) uint8 = which of the two pooled rows won (first max on ties). */
int mx_conv_block_fwd(const float *in, const float *stats, const float *slope, const float *wt,
                      const float *bias, int64_t B, int64_t Cin, int64_t H, int64_t Wv, int32_t dilation,
                      int32_t first_layer, float *out, uint8_t *out_amax, void *stream);

/* data gradient of the block's convolution: G (B,64,H/2,352) = dL/d(pooled pre-activation),
 * routed through amax; wt_flipped = weights packed with flip=1; dxhat (B,64,H,352). */
int mx_conv_block_dgrad(const float *G, const uint8_t *amax, const float *wt_flipped, int64_t B, int64_t H,
                        int64_t Wv, int32_t dilation, float *dxhat, void *stream);

/* ---- K6 on the fp16 matrix cores with fp32-equivalent accuracy ("f16x3": every fp32 operand is split into
 * an fp16 pair hi + lo, products are hi*hi + hi*lo + lo*hi accumulated in fp32; 3 MFMAs at 16x the fp32 MFMA
 * rate).  Same reference semantics as mx_conv_block_fwd / mx_conv_block_dgrad (models.py:183-195) for the
 * 64->64 channel blocks.  Operands are prepared once per layer as channels-last fp16 pairs:
 *   w_hi, w_lo : 4*5*13*64*16 halfs each ([ci/16][kh][kw][co][16], weights * 256; flip = 1 for the data gradient)
 *   x_hi, x_lo : (B, H, 4, 352, 16) halfs (channel-block major): forward = split of (prelu(x) - mean) * rstd;
 *                dgrad = split of the max-pool routed gradient * S_dz, S_dz = 2^k chosen from max|G|
 *                (scale (2,) device floats receives {S_dz, 1/S_dz}; amax_ws = one uint32 workspace, or with
 *                amax_ready != 0 the bits of max|G| left there by mx_ln_prelu_bwd).  dz_hi = dz_lo = NULL: only the
 *                scale pair is produced (the sparse kernels below take the POOLED operand instead). */
int mx_conv_pack_weights_f16(const float *W, int32_t flip, void *w_hi, void *w_lo, void *stream);
int mx_conv_prep_fwd_f16(const float *x, const float *stats, const float *slope, int64_t B, int64_t H,
                         int64_t Wv, void *x_hi, void *x_lo, void *stream);
int mx_conv_prep_dgrad_f16(const float *G, const uint8_t *amax, int64_t B, int64_t H, int64_t Wv,
                           uint32_t *amax_ws, int32_t amax_ready, float *scale, void *dz_hi, void *dz_lo,
                           void *stream);
/* slope_out (64,) / stats_part (B, H/2, 64, 2), both optional: the PReLU slope that follows this block (models.py:194) and
 * the partial LayerNorm statistics of the next block's input for mx_plane_stats_finish. */
int mx_conv_block_fwd_f16(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo,
                          const float *bias, int64_t B, int64_t H, int64_t Wv, int32_t dilation, float *out,
                          uint8_t *out_amax, const float *slope_out, float *stats_part, void *stream);
int mx_conv_block_dgrad_f16(const void *dz_hi, const void *dz_lo, const void *w_hi, const void *w_lo,
                            const float *scale, int64_t B, int64_t H, int64_t Wv, int32_t dilation,
                            float *dxhat, void *stream);
/* First block (2 input channels) on the same matrix-core kernel: (kernel row, input channel) pairs become the 16
 * "channels" of the operand (k = kh*2 + ci, 10 used), so that one 16-deep MFMA k-step covers a tap column and the
 * K loop is a single stage.  Same reference semantics (models.py:183-195, first block: LayerNorm -> Conv2d(2,64,
 * (5,13)) -> +bias -> MaxPool(2,1)).  w_hi, w_lo: 13*2*64*8 halfs each; xk_hi, xk_lo: (B, H, 352, 16) halfs. */
int mx_conv_pack_weights_kvec_f16(const float *W, void *w_hi, void *w_lo, void *stream);
int mx_conv_prep_fwd_kvec_f16(const float *x, const float *stats, int64_t B, int64_t H, int64_t Wv, void *xk_hi,
                              void *xk_lo, void *stream);
int mx_conv_block1_fwd_f16(const void *xk_hi, const void *xk_lo, const void *w_hi, const void *w_lo, const float *bias,
                           int64_t B, int64_t H, int64_t Wv, float *out, uint8_t *out_amax, const float *slope_out,
                           float *stats_part, void *stream);
/* weight gradient of the first block from the same k-vector operand (torch Conv2d backward w.r.t. weight,
 * models.py:187).  G, amax (B,64,H/2,352): gradient w.r.t. the pooled output and the pooling argmax; amax_bits: bit
 * pattern of max|G| (mx_ln_prelu_bwd's gmax_bits); scale (2,) receives {S, 1/S}; part: workspace of
 * ceil(B*H/rows_per_slab)*13*64*16 floats; dW (64,2,5,13). */
int mx_conv_block1_wgrad_f16(const float *G, const uint8_t *amax, const uint32_t *amax_bits, const void *xk_hi,
                             const void *xk_lo, int64_t B, int64_t H, int64_t Wv, int64_t rows_per_slab, float *scale,
                             float *part, float *dW, void *stream);
/* The same with the gradient handed over as f16x3 PAIRS: Gp (B,64,H/2,352) uint32 = (hi | lo << 16) of G * scale[0] per element
 * (fp16 bit patterns), as mx_ln_prelu_bwd_pair leaves them; scale {S, 1/S} from mx_ln_bwd_finish.  Same reference lines. */
int mx_conv_block1_wgrad_pair_f16(const void *Gp, const uint8_t *amax, const float *scale, const void *xk_hi, const void *xk_lo,
                                  int64_t B, int64_t H, int64_t Wv, int64_t rows_per_slab, float *part, float *dW, void *stream);

/* weight gradient from the same prepared operands (dz pair of mx_conv_prep_dgrad_f16, x pair of
 * mx_conv_prep_fwd_f16); part = workspace of ceil(B*H/rows_per_slab)*65*64*64 floats; dW (64,64,5,13). */
int mx_conv_block_wgrad_f16(const void *dz_hi, const void *dz_lo, const void *x_hi, const void *x_lo,
                            const float *scale, int64_t B, int64_t H, int32_t dilation, int64_t rows_per_slab,
                            float *part, float *dW, void *stream);

/* weight gradient: x (B,Cin,H,352) = block input before PReLU/LayerNorm (Cin = 64, or 2 for the
 * first block with slope = NULL); part = workspace of ceil(B*H/rows_per_slab)*65*64*Cin floats;
 * dW (64,Cin,5,13) torch layout, overwritten. */
int mx_conv_block_wgrad(const float *G, const uint8_t *amax, const float *x, const float *stats,
                        const float *slope, int64_t B, int64_t Cin, int64_t H, int64_t Wv, int32_t dilation,
                        int64_t rows_per_slab, float *part, float *dW, void *stream);

/* The same weight gradient on the SPARSE fp16 matrix instruction (v_smfmac_f32_32x32x32_f16): with K ordered as
 * (position, row of the pooling pair) the routed gradient is 2:4 structured sparse -- its compressed form is the pooled
 * gradient G itself, its index bits are the pooling argmax -- so one instruction covers both rows of a pooling pair
 * (half the matrix instructions of mx_conv_block_wgrad_f16; identical results up to fp32 summation order).
 * g_hi, g_lo (B,H/2,4,352,16) halfs = the channels-last split of G * S and gidx (B,64,H/2,22,2) uint16 index words, both
 * from mx_conv_prep_gpool_cl_f16 below (the pair is the operand of the sparse data gradient as well; its [position]
 * [channel] rows become the A fragments through transposing LDS reads).  rows_per_slab counts POOLED rows;
 * Wv <= 351 (MX_ERR_UNSUPPORTED otherwise: the zero pad column of the x operand is the kernel's halo source);
 * part: ceil(B*(H/2)/rows_per_slab)*65*64*64 floats. */
int mx_conv_block_wgrad_sp_f16(const void *g_hi, const void *g_lo, const void *gidx, const void *x_hi, const void *x_lo,
                               const float *scale, int64_t B, int64_t H, int64_t Wv, int32_t dilation,
                               int64_t rows_per_slab, float *part, float *dW, void *stream);

/* The data gradient on the sparse matrix instruction as well.  The sparse operand must be A, so tiles are computed
 * transposed ([position][ci]) with K = (output channel, row of the pooling pair): compressed A = the pooled gradient
 * in channels-last form, B = the weights of the two kernel rows the pair meets, packed in fragment order and read
 * from global memory; 12 K stages instead of 20.  Same results as mx_conv_block_dgrad_f16 up to summation order.
 *   mx_conv_pack_weights_sp_f16: W (64,64,5,13) -> w_hi, w_lo: 4*3*2*13*2*64*16 halfs each
 *   mx_conv_prep_gpool_cl_f16: G, amax (B,64,H/2,352), scale {S, 1/S} -> g_hi, g_lo (B,H/2,4,352,16) halfs,
 *                              g_idx (B,H/2,4,352) uint32 index words of the data gradient; gidx (or NULL):
 *                              (B,64,H/2,22,2) uint16 index words of mx_conv_block_wgrad_sp_f16 from the same pass
 *                              (mx_conv_prep_dgrad_f16 with dz_hi = dz_lo = NULL then only computes the scale pair)
 *   mx_conv_block_dgrad_sp_f16: dxhat (B,64,H,352); Wv <= 351 (zero pad column = halo source).  x_hi, x_lo, ln_part
 *                              (all or NULL): the block's forward operand pair (B,H,4,352,16) = the normalised input
 *                              xhat, and ln_part (B,64,H,2,2) floats <- {sum dxhat, sum dxhat * xhat} per (plane, row,
 *                              position half): the plane statistics mx_ln_prelu_bwd needs, taken while the values are
 *                              in registers instead of by a sweep over dxhat and p.  gx_bits (2 uint32, zeroed by the
 *                              caller; optional, needs ln_part): receives the bit patterns of max|dxhat| and max|xhat|
 *                              of the launch (atomic max) for mx_ln_bwd_finish */
int mx_conv_pack_weights_sp_f16(const float *W, void *w_hi, void *w_lo, void *stream);
int mx_conv_prep_gpool_cl_f16(const float *G, const uint8_t *amax, const float *scale, int64_t B, int64_t H, int64_t Wv,
                              void *g_hi, void *g_lo, void *g_idx, void *gidx, void *stream);
int mx_conv_block_dgrad_sp_f16(const void *g_hi, const void *g_lo, const void *g_idx, const void *w_hi, const void *w_lo,
                               const float *scale, int64_t B, int64_t H, int64_t Wv, int32_t dilation, float *dxhat,
                               const void *x_hi, const void *x_lo, float *ln_part, uint32_t *gx_bits, void *stream);
/* LayerNorm / PReLU backward (torch autograd of models.py:186,194) written STRAIGHT INTO the pooled operand of the block below,
 * for blocks whose two gradients both take the pooled channels-last pair: dL/dp never exists in fp32 (8 B per element less
 * than mx_ln_prelu_bwd + mx_conv_prep_gpool_cl_f16).  The f16x3 scale is needed before the pass, so it comes from an upper
 * bound on max|G| instead of the maximum itself:
 *   mx_ln_bwd_finish: ln_part (B,C,H,2,2), stats (B,C,2), slope (C,), gx_bits (2,) as left by mx_conv_block_dgrad_sp_f16 ->
 *                     m12 (B,C,2) = plane means {dxhat, dxhat * xhat}; scale (2,) = {S, 1/S}, S the power of two that puts
 *                     max_p rstd_p max(1,|slope|) (max|dxhat| + |m1_p| + max|xhat| |m2_p|) into [512, 1024); bound_ws: 1 uint32
 *   mx_ln_prelu_bwd_gpool_f16: p, dxhat, amax (B,64,Hp,352) -> g_hi, g_lo (B,Hp,4,352,16), g_idx (B,Hp,4,352), gidx
 *                     (B,64,Hp,22,2; optional) exactly as mx_conv_prep_gpool_cl_f16 would from G; part: workspace of
 *                     B*64*Hp*6*2 floats; dslope_part, gsum_part (B*64,): the per-plane sums mx_ln_prelu_bwd returns */
int mx_ln_bwd_finish(const float *ln_part, const float *stats, const float *slope, const uint32_t *gx_bits, int64_t B,
                     int64_t C, int64_t H, int64_t Wv, float *m12, uint32_t *bound_ws, float *scale, void *stream);
int mx_ln_prelu_bwd_gpool_f16(const float *p, const float *dxhat, const uint8_t *amax, const float *stats,
                              const float *slope, const float *m12, const float *scale, int64_t B, int64_t Hp, int64_t Wv,
                              void *g_hi, void *g_lo, void *g_idx, void *gidx, float *part, float *dslope_part,
                              float *gsum_part, void *stream);

/* LayerNorm backward fused with the backward of the PReLU in front of it.  p (B,C,H,352): input of
 * that PReLU; dxhat_inout: in = grad w.r.t. the normalised tensor, out = G = dL/dp (in place);
 * dslope_part (B*C,) per-plane partial of dL/dslope.  Optional by-products of the same pass (NULL = skip):
 * gsum_part (B*C,) = per-plane sum of G (the bias gradient partials mx_plane_sum would produce) and
 * gmax_bits: atomicMax of the bit pattern of |G| into one zero-initialised uint32 (the amax_ready input of
 * mx_conv_prep_dgrad_f16).  ln_part (NULL = compute them here): (B,C,H,2,2) partial sums {sum dxhat, sum dxhat *
 * xhat} left by mx_conv_block_dgrad_sp_f16; with them the kernel reads each tensor once instead of twice. */
int mx_ln_prelu_bwd(const float *p, float *dxhat_inout, const float *stats, const float *slope, int64_t B,
                    int64_t C, int64_t H, int64_t Wv, float *dslope_part, float *gsum_part, uint32_t *gmax_bits,
                    const float *ln_part, void *stream);
/* mx_ln_prelu_bwd that leaves G as f16x3 pairs in place: element -> (hi | lo << 16) of G * scale[0] (the operand of
 * mx_conv_block1_wgrad_pair_f16); scale = {S, 1/S} from mx_ln_bwd_finish, ln_part required.  Same reference lines. */
int mx_ln_prelu_bwd_pair(const float *p, float *dxhat_inout, const float *stats, const float *slope, int64_t B, int64_t C,
                         int64_t H, int64_t Wv, float *dslope_part, float *gsum_part, const float *ln_part, const float *scale,
                         void *stream);

/* out[c] (+)= sum_r part[r*C + c]  (fp64 accumulate; deterministic) */
int mx_reduce_rows(const float *part, int64_t R, int64_t C, int32_t accumulate, float *out, void *stream);

/* out[plane] = sum over the H x Wv valid region of each (H,352) plane (bias gradients). */
int mx_plane_sum(const float *x, int64_t planes, int64_t H, int64_t Wv, float *out, void *stream);

/* ---- K7: head -- mod_extraction/models.py:209-215: PReLU -> mean over bins -> Conv1d(C->L,k=1) ->
 * sigmoid.  p6 (B,C,Hl,352); latent (B,C,Wv) and out (B,L,Wv) dense.  L <= 4. */
int mx_head_fwd(const float *p6, const float *slope, const float *wout, const float *bout, int64_t B,
                int64_t C, int64_t Hl, int64_t Wv, int64_t L, float *latent, float *out, void *stream);
/* gmax_bits (optional, zeroed by the caller): receives the bit pattern of max|G6| (atomic max) -- the f16x3 gradient scale of
 * the last block without a sweep over G6 (mx_conv_prep_dgrad_f16 with amax_ready = 1). */
int mx_head_bwd(const float *p6, const float *slope, const float *wout, const float *latent,
                const float *out, const float *d_out, const float *d_latent, int64_t B, int64_t C,
                int64_t Hl, int64_t Wv, int64_t L, float *G6, float *dwout_part, float *dbout_part,
                float *dslope_part, uint32_t *gmax_bits, void *stream);

/* ---- K8: LFO loss -- mod_extraction/lightning.py:33-62 + losses.py:70-102 (l1, fdl1, sdl1, mse,
 * 'mean' reductions).  y_hat, y (B,n), n <= 2048; part (B,4) workspace; losses (5,) = l1, fdl1,
 * sdl1, mse, weighted total (weights <= 0 are logged but not added); grad (B,n) or NULL. */
int mx_lfo_loss(const float *y_hat, const float *y, int64_t B, int64_t n, float w_l1, float w_fdl1,
                float w_sdl1, float w_mse, float *part, float *losses, float *grad, void *stream);

/* ---- K9: LFO post-processing -- mod_extraction/modulations.py:219-363; all bit-exact fp32.
 * x (rows, n) dense. */
/* smoothen (modulations.py:359-363): out (rows, n-k+1) = moving average over k frames. */
int mx_smoothen(const float *x, int64_t rows, int64_t n, int64_t k, float *out, void *stream);
/* find_corners (modulations.py:219-238): top/bot (rows, n) float maps. */
int mx_find_corners(const float *x, int64_t rows, int64_t n, float *top, float *bot, void *stream);
/* stretch_corners after smoothing (modulations.py:260-307): out (rows, n). */
int mx_stretch_corners(const float *x, int64_t rows, int64_t n, int64_t max_n_corners, float *out,
                       void *stream);
/* Gradient of mx_stretch_corners w.r.t. its input (mod_extraction/modulations.py:260-291 is written with differentiable torch
 * ops; lightning.py:294-296 with an unfrozen LFO model back-propagates through it): x = that call's input, dout = gradient
 * w.r.t. its output -> dx.  Corner positions / targets are discrete and carry no gradient. */
int mx_stretch_corners_bwd(const float *x, const float *dout, int64_t rows, int64_t n, int64_t max_n_corners, float *dx,
                           void *stream);
/* check_mod_sig (modulations.py:311-343) per row: valid (rows,) int32 0/1;
 * min_gap = int(min_fraction_between_corners * n). */
int mx_check_mod_sig(const float *x, int64_t rows, int64_t n, int32_t min_top, int32_t max_top,
                     int32_t min_bot, int32_t max_bot, int32_t min_gap, int32_t *valid, void *stream);

/* ---- K10: LSTM-64 effect model -- mod_extraction/models.py:311-339 (nn.LSTM(2,64) -> Linear(64,1) ->
 * + x -> tanh, input order (lfo, audio)) and its truncated BPTT, lightning.py:355-384.
 * x, lfo, y: B rows of T samples with row strides (chunks are views into (B,1,n) tensors).
 * Parameters in torch layout: w_ih (256,2), w_hh (256,64), b_ih (256), b_hh (256), fc_w (64), fc_b (1).
 * h_in, c_in (B,64): state entering the chunk (read only); h_out, c_out (B,64): state leaving it (may alias the
 * inputs when no backward follows: HiddenStateModel.update_hidden, models.py:298-301).  stash (B,T,384) = per-step
 * (i,f,g,o,c,h) for the backward, or NULL. */
int mx_lstm_fwd(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w, const float *fc_b,
                const float *h_in, const float *c_in, float *h_out, float *c_out, float *y, int64_t y_stride,
                float *stash, int64_t B, int64_t T, void *stream);
/* Measurement twin of mx_lstm_fwd (bench.py, SURVEY.md section 8d "measured serial floor"): the same launch with NO
 * global-memory traffic inside the sample loop (constant inputs, only the last chunk stored), i.e. the kernel's
 * dependent chain alone.  Outputs are meaningless; never called by the product.  No reference counterpart. */
int mx_lstm_fwd_probe(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w, const float *fc_b,
                const float *h_in, const float *c_in, float *h_out, float *c_out, float *y, int64_t y_stride,
                float *stash, int64_t B, int64_t T, void *stream);
/* BPTT of one chunk with nn.L1Loss fused: loss = loss_scale * sum |y - wet| (loss_scale = w/(B*T)).
 * h_init, c_init (B,64): state at the chunk start (detached, lightning.py:383).  part (B,17473):
 * per-clip gradient rows in state-dict order [weight_ih | weight_hh | bias_ih | bias_hh | fc.weight |
 * fc.bias]; sum them with mx_reduce_rows. */
int mx_lstm_bwd_l1(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                   int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                   const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                   float loss_scale, float *part, int64_t B, int64_t T, void *stream);
/* The same BPTT for ANY loss -- mod_extraction/lightning.py:380-382 back-propagates whatever calc_and_log_losses returns
 * (losses.py:142-160): dy (B rows, stride dy_stride) = d loss / d y of every output sample of the chunk, produced by
 * mx_effect_loss_grad (L1 / MSE / ESR / DC), mx_mrstft_loss (dx) or their sum.  Other arguments as mx_lstm_bwd_l1. */
int mx_lstm_bwd(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                int64_t y_stride, const float *dy, int64_t dy_stride, const float *stash, const float *w_hh,
                const float *fc_w, const float *h_init, const float *c_init, float *part, int64_t B, int64_t T,
                void *stream);
/* Measurement twin of mx_lstm_bwd_l1 (bench.py, SURVEY.md section 8d "measured serial floor"): the same launch with NO
 * global-memory traffic inside the sample loop (constant inputs, only the last chunk stored), i.e. the kernel's
 * dependent chain alone.  Outputs are meaningless; never called by the product.  No reference counterpart. */
int mx_lstm_bwd_l1_probe(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                   int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                   const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                   float loss_scale, float *part, int64_t B, int64_t T, void *stream);

/* mx_lstm_bwd_l1 / mx_lstm_bwd that ALSO leaves the gate gradients dgate (B, T, 256) (row order of weight_ih_l0): wet != NULL
 * selects the fused nn.L1Loss (loss_scale), else dy = d loss / d y.  mx_lstm_dlfo: dlfo[b][t] = sum_r weight_ih[r][0] dgate[b][t][r]
 * = d loss / d lfo of every sample -- mod_extraction/lightning.py:258,344-366 with freeze_lfo_model: false (the LFO model is
 * re-run inside every TBPTT step and trained through the effect model). */
int mx_lstm_bwd_dgate(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                      int64_t y_stride, const float *wet, int64_t wet_stride, const float *dy, int64_t dy_stride,
                      const float *stash, const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                      float loss_scale, float *part, float *dgate, int64_t B, int64_t T, void *stream);
int mx_lstm_dlfo(const float *dgate, const float *w_ih, int64_t B, int64_t T, float *dlfo, int64_t dlfo_stride, void *stream);

/* Measurement aid (bench.py, `frac_of_independent_floor` of the phaser scan): `steps` samples of the bare 6-stage all-pass
 * cascade + feedback (38 flops) for eight independent state vectors per lane -- the shape of the scan's phase A -- on ONE
 * 512-lane workgroup: no loads, stores, cut-off evaluation, chunk maps or chaining.  time / (steps x 8) x ceil((lead + N) / 512)
 * x 9 runs is a floor of mx_phaser_fwd that does not come from that kernel.  out: >= 512 floats.  No reference counterpart. */
int mx_phaser_cascade_probe(int64_t steps, float *out, void *stream);

/* Measurement aid (bench.py, `frac_of_independent_floor` of the LSTM kernels): `steps` dependent recurrent steps of the bare
 * shape of models.py:333 (kind 0: LDS broadcast of h -> 16 packed FMAs -> cross-lane adds -> v_exp / v_rcp gate -> exchange ->
 * cell update -> tanh -> LDS write -> s_barrier) or of its BPTT (kind 1: gate gradients from LDS -> 16 packed FMAs -> all-reduce
 * over 16 row groups -> dh, dc, dg -> LDS write -> s_barrier) on ONE 512-lane workgroup, or (kind 2) the forward step in the
 * 256-lane decomposition that mx_lstm_fwd launches since round 6 (32 packed FMAs per lane, one cross-lane add): no global memory,
 * no input term, no stash, no output layer, no weight gradients.  Time / steps x T is a floor of a T-step launch that does not come from the
 * product kernels.  out: 1 float (keeps the chain live).  No reference counterpart. */
int mx_lstm_step_probe(int32_t kind, int64_t steps, float *out, void *stream);

/* ---- TCN extractors -- mod_extraction/tcn.py:106-302 (TCNBlock / TCN) under models.py:72-125,218-289
 * (SpectralTCN / SpectralDSTCN).  Activations: (B, C, 352) fp32 planes with T <= 352 valid columns.
 * mx_sgemm_f32: general fp32 GEMM on the matrix cores (exact fp32), element strides for every operand:
 *   C[m c_rs + n c_cs] (+)= sum_k A[m a_rs + k a_cs] B[k b_rs + n b_cs], n_batch batches a_bs / b_bs apart;
 *   batches_per_group consecutive batches are summed inside the kernel, group g writes C + g c_bs.
 * mx_tcn_im2col: col[(ci ksz + k)][b To + t'] = xhat[b][ci][t' stride + (k - ksz/2) dilation] (0 outside [0,T));
 *   xhat = (x - stats[2b]) stats[2b+1] (LayerNorm([C,T]) statistics from mx_plane_stats(x, NULL, B, 1, C, T)) or x.
 * mx_tcn_col2im: the transposed gather (gradient w.r.t. xhat).
 * mx_tcn_act_fwd: z += bias (kept: PReLU input); y = PReLU(z) + res.  mx_tcn_act_bwd: dz = dy PReLU'(zb);
 *   part (B*C, 2) = row sums of dz and of dy zb [zb <= 0] (bias / slope gradients after mx_reduce_rows).
 * mx_tcn_ln_bwd: dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) + add, per clip. */
int mx_sgemm_f32(const float *a, int64_t a_rs, int64_t a_cs, int64_t a_bs, const float *b, int64_t b_rs, int64_t b_cs,
                 int64_t b_bs, float *c, int64_t c_rs, int64_t c_cs, int64_t c_bs, int64_t M, int64_t N, int64_t K,
                 int64_t n_batch, int64_t batches_per_group, int32_t accumulate, void *stream);
int mx_tcn_im2col(const float *x, const float *stats, int64_t B, int64_t C, int64_t T, int64_t To, int64_t ksz,
                  int64_t dilation, int64_t stride, float *col, void *stream);
int mx_tcn_col2im(const float *dcol, int64_t B, int64_t C, int64_t T, int64_t To, int64_t ksz, int64_t dilation,
                  int64_t stride, float *dx, void *stream);
int mx_tcn_act_fwd(float *z, const float *bias, const float *slope, const float *res, int64_t B, int64_t C, int64_t T,
                   float *y, void *stream);
int mx_tcn_act_bwd(const float *dy, const float *zb, const float *slope, int64_t B, int64_t C, int64_t T, float *dz,
                   float *part, void *stream);
int mx_tcn_ln_bwd(const float *x, const float *dxhat, const float *stats, const float *add, int64_t B, int64_t C,
                  int64_t T, float *dx, void *stream);

/* ---- Spectral2DCNN outside the 5x13 / 64-channel / pool (2,1) family -- mod_extraction/models.py:127-215 with any kernel
 * size, channel list, dilations, MaxPool2d((p,1)), use_ln, in_ch, frame count and latent_dim (the class's own defaults
 * are such a configuration).  Dense NCHW fp32 tensors, exact fp32 arithmetic; the three convolution products are
 * mx_sgemm_f32 on the matrices these gathers build.
 * mx_im2col2d: col[(ci kh + i) kw + j][(b H + h) W + w] = x[b][ci][h + i dh - pt][w + j dw - pl] (0 outside the image) for
 *   the nb clips at x; "same" padding (models.py:187): pt = dh (kh - 1) / 2, pl = dw (kw - 1) / 2 (aten: the remainder goes
 *   after).  mx_col2im2d: the transposed gather, dx[b][ci][y][x] = sum over taps of dcol[...] (gradient w.r.t. x).
 * mx_rowln_fwd / _bwd: nn.LayerNorm([bins, frames], elementwise_affine=False) (models.py:186) of `rows` contiguous rows of n
 *   elements: y = (x - mean) rstd, stats (rows, 2) = {mean, rstd}; dx = rstd (dy - mean(dy) - y mean(dy y)).
 * mx_pool_prelu_fwd / _bwd: Conv2d bias + MaxPool2d((p,1)) + PReLU(C) (models.py:187-189) of the products z (planes, H, W),
 *   planes = B C: v (planes, H/p, W) = window maximum of z + bias[c] (first maximum wins; rows beyond (H/p) p are dropped), out = v > 0 ? v : slope[c] v,
 *   amax = row offset of the maximum; backward: dz (planes, H, W) routed (zero elsewhere), part (planes, 2) = {sum of dz
 *   (bias gradient), sum of g v where v <= 0 (slope gradient)} -> mx_reduce_rows over the clips.
 * mx_binmean_head_fwd / _bwd: latent (B, C, W) = mean over bins of x (B, C, H, W), out (B, L, W) = sigmoid(Conv1d(C, L, 1))
 *   (models.py:209-215); backward from d_out (B, L, W) and / or d_latent (B, C, W) (either may be NULL): ds (B, L, W) = the
 *   gradient at the Conv1d output (its weight / bias gradients are an mx_sgemm_f32 / mx_row_sums of it), dx (B, C, H, W).
 * mx_row_sums: out[r] = sum of the n contiguous elements of row r (fp64 accumulate). */
int mx_im2col2d(const float *x, int64_t nb, int64_t Cin, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t dh, int64_t dw,
                int64_t pt, int64_t pl, float *col, void *stream);
int mx_col2im2d(const float *dcol, int64_t nb, int64_t Cin, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t dh, int64_t dw,
                int64_t pt, int64_t pl, float *dx, void *stream);
int mx_rowln_fwd(const float *x, int64_t rows, int64_t n, float eps, float *y, float *stats, void *stream);
int mx_rowln_bwd(const float *dy, const float *y, const float *stats, int64_t rows, int64_t n, float *dx, void *stream);
int mx_row_sums(const float *x, int64_t rows, int64_t n, float *out, void *stream);
int mx_pool_prelu_fwd(const float *z, const float *bias, int64_t planes, int64_t C, int64_t H, int64_t W, int64_t p,
                      const float *slope, float *v, float *out, uint8_t *amax, void *stream);
int mx_pool_prelu_bwd(const float *g, const float *v, const uint8_t *amax, int64_t planes, int64_t C, int64_t H, int64_t W,
                      int64_t p, const float *slope, float *dz, float *part, void *stream);
int mx_binmean_head_fwd(const float *x, int64_t B, int64_t C, int64_t H, int64_t W, const float *wout, const float *bout,
                        int64_t L, float *latent, float *out, void *stream);
int mx_binmean_head_bwd(const float *d_out, const float *d_latent, const float *out, const float *wout, int64_t B, int64_t C,
                        int64_t H, int64_t W, int64_t L, float *ds, float *dx, void *stream);

/* ---- LSTMEffectModel of any size -- mod_extraction/models.py:311-339 with in_ch / out_ch / n_hidden / latent_dim other than
 * the shipped 1 / 1 / 64 / 1 (a param_model, lightning.py:344-347, widens latent_dim).  Only the recurrences are kernels of
 * their own; the input projection, the output layer and every gradient product are mx_sgemm_f32 calls over all steps.
 * mx_lstmg_fwd: zin (B, T, 4 Hn) = W_ih u_t (gate order i, f, g, o); per step gates = act(zin + bias_ih + bias_hh + W_hh h),
 *   c = f c + i g, h = o tanh(c); stash (B, T, 6, Hn) = (i, f, g, o, c, h) of every step; (h1, c1) = the final state.
 * mx_lstmg_bwd: dhfc (B, T, Hn) = d loss / d h_t through the output layer; dgate (B, T, 4 Hn) = d loss / d gate
 *   pre-activations (BPTT inside the chunk; the incoming state is a constant, lightning.py:353,383).
 * mx_lstmg_out_fwd: y (B, Co, T) = tanh(fc (B, T, out_ch) + bias + x (B, in_ch, T)), Co = max(out_ch, in_ch), torch's
 *   broadcast (models.py:337-338).  mx_lstmg_out_bwd: dpre (B, T, out_ch) = dy (1 - y^2) summed over the broadcast channels. */
int mx_lstmg_fwd(const float *zin, const float *bias_ih, const float *bias_hh, const float *w_hh, const float *h0, const float *c0,
                 int64_t B, int64_t T, int64_t Hn, float *stash, float *h1, float *c1, void *stream);
int mx_lstmg_bwd(const float *stash, const float *dhfc, const float *w_hh, const float *c0, int64_t B, int64_t T, int64_t Hn,
                 float *dgate, void *stream);
int mx_lstmg_out_fwd(const float *fc, const float *bias, const float *x, int64_t B, int64_t T, int64_t out_ch, int64_t in_ch,
                     float *y, void *stream);
int mx_lstmg_out_bwd(const float *dy, const float *y, int64_t B, int64_t T, int64_t out_ch, int64_t Co, float *dpre, void *stream);

/* ---- TCN variants outside SpectralTCN / SpectralDSTCN -- mod_extraction/tcn.py:14-103,130-195: explicit padding with the causal /
 * centre crop of the residual branch, the cached streaming convolution, FiLM with or without its BatchNorm1d.  Dense
 * (B, C, T) fp32 tensors of any length; convolutions = mx_im2col2d (one bin row) + mx_sgemm_f32, LayerNorm = mx_rowln_*.
 * mx_chan_stats: stats (C, 2) = {mean, biased variance} of every channel over (clips, frames) (BatchNorm1d training mode).
 * mx_chan_norm_fwd: xhat = (z - norm[c][0]) norm[c][1], norm (C, 2) = {mean, rstd}; _bwd: dz = rstd (g - mean(g) - xhat
 *   mean(g xhat)) with the means over (clips, frames) when train != 0 (batch statistics), else dz = rstd g.
 * mx_film_fwd: a = xhat gb[b][c] + gb[b][C + c], gb (B, 2 C) = the adaptor's output (tcn.py:95-102); _bwd: dxhat = da gain,
 *   dgb = {sum over frames of da xhat, of da}.
 * mx_prelu_res_fwd: y = PReLU(a; slope[c]) (slope NULL: identity) + res (NULL: none) (tcn.py:185-191); _bwd: da = dy PReLU'(a),
 *   part (B C,) = sum over frames of dy a where a <= 0 (slope gradient after mx_reduce_rows over the clips). */
int mx_chan_stats(const float *z, int64_t B, int64_t C, int64_t T, float *stats, void *stream);
int mx_chan_norm_fwd(const float *z, const float *norm, int64_t B, int64_t C, int64_t T, float *xhat, void *stream);
int mx_chan_norm_bwd(const float *g, const float *xhat, const float *norm, int64_t B, int64_t C, int64_t T, int32_t train, float *dz,
                     void *stream);
int mx_film_fwd(const float *xhat, const float *gb, int64_t B, int64_t C, int64_t T, float *a, void *stream);
int mx_film_bwd(const float *da, const float *xhat, const float *gb, int64_t B, int64_t C, int64_t T, float *dxhat, float *dgb,
                void *stream);
int mx_prelu_res_fwd(const float *a, const float *slope, const float *res, int64_t B, int64_t C, int64_t T, float *y, void *stream);
int mx_prelu_res_bwd(const float *dy, const float *a, const float *slope, int64_t B, int64_t C, int64_t T, float *da, float *part,
                     void *stream);

/* ---- effect-model losses -- mod_extraction/losses.py:14-67 (ESR, DC) and nn.L1Loss:
 * part (B,4) = per-clip sums of |y - y_hat|, (y - y_hat)^2, y^2, (y - y_hat). */
int mx_effect_loss_sums(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride, int64_t B,
                        int64_t T, float *part, void *stream);
/* d (w_l1 L1 + w_mse MSE + w_esr ESR + w_dc DC) / d y_hat ('mean' reductions; nn.L1Loss, nn.MSELoss, losses.py:33-38,
 * 61-66) into dy (B rows, stride dy_stride); accumulate != 0 adds onto what dy holds (e.g. the MR-STFT gradient).
 * The backward half of losses.py:142-160 for the effect model (lightning.py:380-382). */
int mx_effect_loss_grad(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride, int64_t B,
                        int64_t T, float w_l1, float w_mse, float w_esr, float w_dc, float eps, int32_t accumulate,
                        float *dy, int64_t dy_stride, void *stream);

/* ---- K11: multi-resolution STFT loss -- mod_extraction/losses.py:155-156 (auraloss==0.4.0
 * MultiResolutionSTFTLoss(reduction="mean"), third-party: restated from its published defaults,
 * parity unpinned): mean over resolutions of [ w_sc * ||Y-X||_F/||Y||_F + w_log * mean|log X - log Y| ],
 * mag = sqrt(clamp(re^2+im^2, eps)).  y_hat, y: B rows of T samples (row strides).
 * fft_sizes, hops: HOST int32 arrays of n_res entries (fft sizes in {512,1024,2048}) -- the only host
 * pointers of this ABI.  windows (n_res, 2048) device: row r = n_fft_r-long analysis window (hann of
 * win_length centred in the frame).  twiddle (2048,2) device = exp(-2 pi i m/2048).
 * terms (2*n_res+1) device: [sc_0, logmag_0, ..., total].  dx: d total / d y_hat rows (stride dx_stride)
 * or NULL.  Workspaces (device): part >= 3*B*max_r ceil(frames_r/8) doubles, coef n_res floats (the scalar alpha_r =
 * w_sc / (n_res ||Y-X||_F ||Y||_F) of each resolution), scratch (needed only when dx != NULL) >= sum_r 2*B*(frames_r*hop_r +
 * runs_r*max(n_fft_r - hop_r, 0)) floats with frames_r = 1 + T/hop_r, runs_r = ceil(frames_r/F_r), F_r = max(32,
 * ceil(n_fft_r/hop_r)) rounded up to even: the two linear components g1_r, g2_r of the time-domain gradient
 * (d total / d y_hat = sum_r alpha_r g1_r + g2_r) as per-run overlap-add sums plus one tail per run; nothing per bin is stored. */
int mx_mrstft_loss(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride, int64_t B,
                   int64_t T, int32_t n_res, const int32_t *fft_sizes, const int32_t *hops,
                   const float *windows, const float *twiddle, float w_sc, float w_log, float eps,
                   double *part, float *coef, float *scratch, float *terms, float *dx, int64_t dx_stride,
                   void *stream);

/* ---- K12: AdamW -- torch.optim.AdamW (configs/opt/adam_w.yml), flat fp32 buffers of n elements;
 * step = 1-based step index; grad_scale multiplies the gradient first (1/world after a sum
 * all-reduce). */
int mx_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, void *stream);

/* mx_reduce_rows + mx_adamw_step in ONE launch for small parameter sets whose gradient arrives as one row per clip (the
 * LSTM-64's 17 473 parameters, 83 optimizer steps per TBPTT batch: lightning.py:355-384 / configs/opt/adam_w.yml):
 * part (R, n) -> grad (n,) = column sums (fp64, the order of mx_reduce_rows), then the AdamW update of mx_adamw_step on them.
 * Bit-identical to the two calls. */
int mx_reduce_rows_adamw_step(const float *part, int64_t R, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                              int64_t n, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MODEX_HIP_H */
