// flanger.hip -- K2: mono flanger / chorus (reference: mod_extraction/fx.py:72-119).
//
// One wavefront per clip; the circular delay line (M <= 40k floats) lives in LDS.
// The reference executes 88 200 dependent python iterations per batch.  Here a wave walks the
// clip in chunks of 64*V samples.  For every sample the fp32 index bookkeeping of fx.py:95-103
// (write slot, fractional read position, prev/next slot) is evaluated with exactly the
// reference's rounding sequence (no FMA contraction; build uses -ffp-contract=off and explicit
// __f*_rn).  From the *integer* slots the wave derives, per chunk, the shortest distance G back
// to a sample whose write is read inside the chunk; any run of <= G consecutive samples has no
// internal read-after-write dependency, so it is processed by G lanes in lock step (all reads,
// then all writes -- exactly the reference's read-before-write order, fx.py:111-115).
// Chorus (min delay >= 485 samples) always runs 256 samples per step; a flanger near zero delay
// degrades gracefully down to the reference's one-sample-at-a-time order (G = 1).
//
// Results are bit-identical to the reference for identical mod_sig input.
// Algorithmic HBM traffic: 12 B/sample (x, mod in; y out), 8 B/sample with the 882-point LFO
// resampled in-kernel (util.py:15-29).
#include "common.h"

#define FL_V 4                 // samples per lane per chunk
#define FL_CHUNK (64 * FL_V)
#define FL_MAX_M 40000         // 160 KB LDS = 40960 floats

struct FlSample {
    float x, frac;
    int w, prev, next;
    int dep;                   // distance (in samples) to the most recent slot write it reads
};

__global__ __launch_bounds__(64) void flanger_kernel(
    const float *__restrict__ x, long long x_stride, const float *__restrict__ mod, int n_mod, float mod_scale,
    const float *__restrict__ lfo_scale, const float *__restrict__ min_delay,
    const float *__restrict__ feedback, const float *__restrict__ depth,
    const float *__restrict__ mix, const float *__restrict__ one_minus_mix,
    const int *__restrict__ max_delay, const int *__restrict__ rows, int N, int lfo_off,
    float *__restrict__ y, long long y_stride, float *__restrict__ mod_up, long long *__restrict__ dbg_prev,
    float *__restrict__ dbg_frac, int probe)
{
    extern __shared__ float buf[];           // [M delay line | n_mod LFO row (when resampled in-kernel)]
    const int lane = threadIdx.x;
    const int b = rows ? rows[blockIdx.x] : (int)blockIdx.x;
    const int M = max_delay[b];
    const float Mf = (float)M;
    const float ls = lfo_scale[b], md = min_delay[b], fb = feedback[b], dp = depth[b];
    const float mx = mix[b], omm = one_minus_mix[b];
    const float *xb = x + (size_t)b * x_stride;
    const float *mb = mod + (size_t)b * n_mod;
    float *yb = y + (size_t)b * y_stride;

    for (int i = lane; i < M; i += 64) buf[i] = 0.0f;  // fx.py:92 (LDS ops of one wave are in order)

    const bool resample = (n_mod != N);
    float *lfo = buf + lfo_off;                        // the short LFO row lives in LDS: no gather latency per chunk
    if (resample)
        for (int i = lane; i < n_mod; i += 64) lfo[i] = mb[i];
    int w_chunk = 0;                                   // c0 % M, carried instead of a per-sample integer modulo
    float xr[FL_V], mr[FL_V];
    // software prefetch of the first chunk
#pragma unroll
    for (int j = 0; j < FL_V; ++j) {
        int n = j * 64 + lane;
        xr[j] = n < N && !probe ? xb[n] : 0.25f;
        if (!resample) mr[j] = n < N && !probe ? mb[n] : 0.5f;
    }

    for (int c0 = 0; c0 < N; c0 += FL_CHUNK) {
        FlSample s[FL_V];
        int g = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < FL_V; ++j) {
            const int n = c0 + j * 64 + lane;
            const bool valid = n < N;
            float m;
            if (resample) {
                InterpTap t = interp_tap(mod_scale, valid ? n : 0, n_mod);
                m = interp_combine(t, lfo[t.i0], lfo[t.i1]);
                if (mod_up && valid) mod_up[(size_t)b * N + n] = m;
            } else {
                m = mr[j];
            }
            int w = w_chunk + j * 64 + lane;                   // fx.py:95: n % M without a division
            while (w >= M) w -= M;
            const float d = __fadd_rn(__fmul_rn(ls, m), md);   // fx.py:99
            const float r1 = __fadd_rn(__fsub_rn((float)w, d), Mf);  // fx.py:100
            float r;
            if (r1 >= 0.0f && r1 < Mf) r = r1;                 // fmod is the identity here
            else if (r1 >= Mf && r1 < __fadd_rn(Mf, Mf)) r = __fsub_rn(r1, Mf);  // exact (Sterbenz)
            else r = torch_remainderf(r1, Mf);                 // out-of-contract mod_sig: generic path
            const float fl = floorf(r);
            int prev = (int)fl;                                // fx.py:102
            if (prev < 0) prev = 0;                            // NaN / garbage guard (never hit in contract)
            if (prev >= M) prev = M - 1;
            const int next = prev + 1 == M ? 0 : prev + 1;     // fx.py:103
            s[j].x = xr[j];
            s[j].frac = __fsub_rn(r, fl);                      // fx.py:101
            s[j].w = w;
            s[j].prev = prev;
            s[j].next = next;
            int dp_ = w - prev; if (dp_ <= 0) dp_ += M;        // slot w itself is "M samples ago"
            int dn_ = w - next; if (dn_ <= 0) dn_ += M;
            s[j].dep = valid ? min(dp_, dn_) : 0x7fffffff;
            g = min(g, s[j].dep);
            if (dbg_prev && valid) dbg_prev[(size_t)b * N + n] = prev;
            if (dbg_frac && valid) dbg_frac[(size_t)b * N + n] = s[j].frac;
        }
        g = wave_min_i32(g);

        // prefetch the next chunk while this one is in flight
        float xn[FL_V], mn[FL_V];
#pragma unroll
        for (int j = 0; j < FL_V; ++j) {
            int n = c0 + FL_CHUNK + j * 64 + lane;
            xn[j] = n < N && !probe ? xb[n] : 0.25f;
            if (!resample) mn[j] = n < N && !probe ? mb[n] : 0.5f;
        }

        float o[FL_V];
        if (g >= FL_CHUNK) {
            // whole chunk is dependency-free: 4 rows of 64 lanes, all reads before all writes
            float pv[FL_V], nv[FL_V];
#pragma unroll
            for (int j = 0; j < FL_V; ++j) { pv[j] = buf[s[j].prev]; nv[j] = buf[s[j].next]; }
#pragma unroll
            for (int j = 0; j < FL_V; ++j) {
                float it = __fadd_rn(__fmul_rn(s[j].frac, nv[j]), __fmul_rn(__fsub_rn(1.0f, s[j].frac), pv[j]));
                buf[s[j].w] = __fadd_rn(s[j].x, __fmul_rn(fb, it));   // fx.py:114
                o[j] = __fadd_rn(s[j].x, __fmul_rn(dp, it));          // fx.py:115
            }
        } else {
            const int gr = g < 64 ? g : 64;
#pragma unroll
            for (int j = 0; j < FL_V; ++j) {
                o[j] = 0.0f;
                for (int g0 = 0; g0 < 64; g0 += gr) {
                    if (lane >= g0 && lane < g0 + gr) {
                        float pv = buf[s[j].prev], nv = buf[s[j].next];     // fx.py:111-112
                        float it = __fadd_rn(__fmul_rn(s[j].frac, nv), __fmul_rn(__fsub_rn(1.0f, s[j].frac), pv));
                        buf[s[j].w] = __fadd_rn(s[j].x, __fmul_rn(fb, it));
                        o[j] = __fadd_rn(s[j].x, __fmul_rn(dp, it));
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
#pragma unroll
        for (int j = 0; j < FL_V; ++j) {
            const int n = c0 + j * 64 + lane;
            if (n < N && (!probe || n + FL_CHUNK >= N)) {     // probe: only the last chunk is stored (keeps the chain live)
                float v = __fadd_rn(__fmul_rn(omm, s[j].x), __fmul_rn(mx, o[j]));   // fx.py:117
                yb[n] = fminf(fmaxf(v, -1.0f), 1.0f);                              // fx.py:118
            }
            xr[j] = xn[j];
            mr[j] = mn[j];
        }
        w_chunk += FL_CHUNK;
        while (w_chunk >= M) w_chunk -= M;
    }
}

// C ABI ---------------------------------------------------------------------------------------
// x: row b at x + b*x_stride (N samples); y likewise with y_stride; mod (B,n_mod) with n_mod == N or any shorter length (resampled in-kernel,
// align_corners=True); per-clip fp32 constants lfo_scale = max_lfo_delay_samples*width,
// min_delay = min_delay_width*max_min_delay_samples, feedback, depth, mix, one_minus_mix;
// max_delay (B,) int32 = delay-line length M per clip (flanger and chorus clips may be mixed in
// one batch); rows: optional list of n_rows clip indices to process (others untouched).
// Optional outputs: mod_up (B,N) resampled LFO; dbg_prev (B,N) int64 / dbg_frac (B,N) for the
// index-parity tests.
static int flanger_fwd_launch(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream, int probe)
{
    if (!x || !mod || !lfo_scale || !min_delay || !feedback || !depth || !mix || !one_minus_mix ||
        !max_delay || !y || B <= 0 || N <= 0 || n_mod <= 0)
        return MX_ERR_ARG;
    if (max_delay_max < 2 || x_stride < N || y_stride < N) return MX_ERR_ARG;
    if (max_delay_max > FL_MAX_M || N >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const int64_t items = rows ? n_rows : B;
    if (items <= 0) return MX_OK;
    static bool attr_set[64] = {};                       // per device: one process may drive several GPUs
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)flanger_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            FL_MAX_M * sizeof(float));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    // LDS: delay line (max over the batch) + the LFO row when it is resampled in-kernel (n_mod < N)
    const int lfo_off = max_delay_max;
    const size_t lds_floats = (size_t)max_delay_max + (n_mod != N ? (size_t)n_mod : 0);
    if (lds_floats > FL_MAX_M) return MX_ERR_UNSUPPORTED;
    const size_t lds = lds_floats * sizeof(float);
    hipLaunchKernelGGL(flanger_kernel, dim3((unsigned)items), dim3(64), lds, (hipStream_t)stream, x, (long long)x_stride, mod,
                       (int)n_mod, interp_scale_host(n_mod, N), lfo_scale, min_delay, feedback, depth,
                       mix, one_minus_mix, max_delay, rows, (int)N, lfo_off, y, (long long)y_stride, mod_up, (long long *)dbg_prev,
                       dbg_frac, probe);
    return mx_launch_status();
}

MX_EXPORT int mx_flanger_fwd(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream)
{
    return flanger_fwd_launch(x, x_stride, mod, n_mod, lfo_scale, min_delay, feedback, depth, mix, one_minus_mix, max_delay, max_delay_max, rows, n_rows, B, N, y, y_stride, mod_up, dbg_prev, dbg_frac, stream, 0);
}

// Measurement twin (bench.py's serial floor): the SAME launch with no global-memory traffic inside the sample loop -- inputs are constants, only the last chunk is stored.  Results are meaningless; nothing in the product calls it.
MX_EXPORT int mx_flanger_fwd_probe(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream)
{
    return flanger_fwd_launch(x, x_stride, mod, n_mod, lfo_scale, min_delay, feedback, depth, mix, one_minus_mix, max_delay, max_delay_max, rows, n_rows, B, N, y, y_stride, mod_up, dbg_prev, dbg_frac, stream, 1);
}
