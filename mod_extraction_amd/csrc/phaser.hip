// phaser.hip -- K3: 6-stage all-pass phaser with feedback (reference call site:
// mod_extraction/datasets.py:455-482 -> pedalboard==0.7.3 Phaser == JUCE dsp::Phaser<float>;
// third-party source absent from the reference tree: PARITY UNPINNED, checked against
// oracle/csrc/oracle_ref.c:orc_phaser, which restates the published JUCE algorithm).
//
// Two kernels.  phaser_kernel: one wavefront per clip, every sample in JUCE's operation order, strictly serial -- the bit
// reference (exact_order != 0).  phaser_scan_kernel (default): the same recurrence as a linear scan over time, one
// workgroup per clip, one chunk of the clip per lane (see its header below).
// sin / pow / log10 are evaluated in fp64 and rounded once, which reproduces the host libm's (correctly rounded) float
// results; the LFO phase accumulator advances in fp32 exactly as JUCE's does.
// `lead` samples are processed (filter warm-up, LFO phase) before the N output samples: the
// reference renders n + sr/rate samples and crops at a random offset (datasets.py:428-449).
// Algorithmic HBM traffic: 8 B/sample (+4 B/sample when the cropped dry clip is also written).
#include "common.h"

#define PH_BLOCK 256
#ifndef PH_WPB
#define PH_WPB 4          // clips (wavefronts) per workgroup: packs the long-running serial chains onto few CUs so
                          // that the 1-workgroup-per-CU matrix kernels of the train step keep the other CUs
#endif

__global__ __launch_bounds__(64 * PH_WPB) void phaser_kernel(const float *__restrict__ x, long long x_stride,
                                                    const float *__restrict__ rate,
                                                    const float *__restrict__ depth,
                                                    const float *__restrict__ centre,
                                                    const float *__restrict__ feedback,
                                                    const float *__restrict__ mix,
                                                    const int *__restrict__ lead_arr,
                                                    const int *__restrict__ rows, int n_items, int N, float sr_f,
                                                    double sr, float *__restrict__ y, long long y_stride,
                                                    float *__restrict__ dry_out)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * PH_WPB + (threadIdx.x >> 6);
    if (item >= n_items) return;                        // whole wave exits; waves never synchronise with each other
    const int b = rows ? rows[item] : item;
    const int lead = lead_arr ? lead_arr[b] : 0;
    const int total = lead + N;
    const float *xb = x + (size_t)b * x_stride;
    float *yb = y + (size_t)b * y_stride;
    float *db = dry_out ? dry_out + (size_t)b * y_stride : nullptr;

    const float two_pi = 6.283185307179586476925286766559f;
    const float pi_f = 3.14159265358979323846f;
    const float fmax_hz = (float)fmin(20000.0, 0.49 * sr);
    const float log_min = (float)log10(20.0), log_max = (float)log10((double)fmax_hz);
    const float inc = __fmul_rn(__fdiv_rn(two_pi, (float)(sr / 4.0)), rate[b]);
    const float norm_centre = __fdiv_rn(__fsub_rn((float)log10((double)centre[b]), log_min), __fsub_rn(log_max, log_min));
    const float osc_vol = __fmul_rn(depth[b], 0.5f);
    const float fb = feedback[b];
    const float wet_g = mix[b], dry_g = __fsub_rn(1.0f, mix[b]);
    (void)sr_f;

    float phase = 0.0f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, last = 0.f;

    for (int n0 = 0; n0 < total; n0 += PH_BLOCK) {
        // (1) coalesced load of 256 input samples: lane l holds samples n0 + j*64 + l, j = 0..3
        float xr[PH_BLOCK / 64], yr[PH_BLOCK / 64];
#pragma unroll
        for (int j = 0; j < PH_BLOCK / 64; ++j) {
            const int n = n0 + j * 64 + lane;
            xr[j] = n < total ? xb[n] : 0.0f;
            yr[j] = 0.0f;
        }
        // (2) sequential fp32 phase accumulation; lane k keeps the phase of update k
        float my_phase = 0.0f;
        for (int k = 0; k < PH_BLOCK / 4; ++k) {
            if (lane == k) my_phase = phase;
            phase = __fadd_rn(phase, inc);
            while (phase >= two_pi) phase = __fsub_rn(phase, two_pi);
        }
        // (3) one cut-off update per lane: lane k holds G of samples 4k .. 4k+3 of this block
        float Greg;
        {
            float osc = (float)sin((double)__fsub_rn(my_phase, pi_f));
            float lfo = __fadd_rn(__fmul_rn(osc, osc_vol), norm_centre);
            lfo = lfo < 0.0f ? 0.0f : (lfo > 1.0f ? 1.0f : lfo);
            float fc = (float)pow(10.0, (double)__fadd_rn(__fmul_rn(lfo, __fsub_rn(log_max, log_min)), log_min));
            float g = (float)tan(3.14159265358979323846 * (double)fc / sr);
            Greg = __fdiv_rn(g, __fadd_rn(1.0f, g));
        }
        // (4) the dependent chain, wave-uniform; inputs and coefficients are broadcast from lane
        //     registers (v_readlane) so that no LDS round trip sits between two samples
        const int cnt = min(PH_BLOCK, total - n0);
#pragma unroll
        for (int j = 0; j < PH_BLOCK / 64; ++j) {
            const int lim = min(64, cnt - j * 64);
            for (int li = 0; li < lim; ++li) {
                const float in = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr[j]), li));
                float out = __fsub_rn(in, last);
                const float G = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Greg), (j * 64 + li) >> 2));
                float v, yk;
#define PH_STAGE(S)                                   \
    v = __fmul_rn(G, __fsub_rn(out, S));             \
    yk = __fadd_rn(v, S);                            \
    S = __fadd_rn(v, yk);                            \
    out = __fsub_rn(__fmul_rn(2.0f, yk), out);
                PH_STAGE(s0) PH_STAGE(s1) PH_STAGE(s2) PH_STAGE(s3) PH_STAGE(s4) PH_STAGE(s5)
#undef PH_STAGE
                last = __fmul_rn(out, fb);
                float m = __fadd_rn(__fmul_rn(out, wet_g), __fmul_rn(in, dry_g));
                m = m < -1.0f ? -1.0f : (m > 1.0f ? 1.0f : m);
                yr[j] = lane == li ? m : yr[j];
            }
        }
        // (5) coalesced store of the samples that fall inside the output window
#pragma unroll
        for (int j = 0; j < PH_BLOCK / 64; ++j) {
            const int n = n0 + j * 64 + lane;
            if (n >= lead && n < total) {
                yb[n - lead] = yr[j];
                if (db) db[n - lead] = xr[j];
            }
        }
    }
}

// ---- the default kernel: the recurrence as a LINEAR SCAN over time -------------------------------------------------
// For a given cut-off sequence the phaser is a linear time-varying system in its 7-vector state z = (s0..s5, lastOut):
//     z(n+1) = A(G_n) z(n) + b(G_n) x(n).
// The cut-offs do not depend on the state (LFO only), so the clip is cut into PS_P = 512 chunks, one per LANE, and
//   A  every lane runs its chunk EIGHT times in registers -- from the seven unit states with silent input and from the
//      zero state with the real input -- which gives the chunk's affine map  z_end = M z_start + v  (the scalar all-pass
//      cascade costs 38 flops per sample and run; a dense 8 x 8 step would cost 128), and on the way evaluates its
//      cut-off updates (one fp64 sin / pow / tan per 4 samples: each update of the clip is still evaluated exactly once)
//      and parks them in a workspace;
//   B  seven lanes of one wave chain the 512 maps (a 7 x 7 mat-vec per chunk, ~25 us): the state at every chunk start;
//   C  every lane re-runs its chunk from its true start state in JUCE's operation order and writes the output window.
// No sample waits for its predecessor outside a chunk of ~345: 85 clips x (2 s + lead) take ~0.4 ms where the
// state-space step on a producer / consumer wave pair (round 2, one dependent 8 x 8 step per sample) took 7.8 ms.
// Rounding: phase C is the reference's own arithmetic; only the chunk-start states carry the re-association error of the
// maps (~1e-7 of the state, it decays like any state perturbation of the stable all-pass loop).
// The LFO phase at a chunk start must equal what 44 100 sequential fp32 additions give (JUCE accumulates in fp32; a
// closed form in fp64 is 1e-5 rad off after a few thousand steps): ps_phase_after() below reproduces them exactly in
// O(binades) steps.
#define PS_WAVES 8
#define PS_P (64 * PS_WAVES)       // chunks (lanes) per clip
#define PS_MV 57                   // floats per chunk map: M column-major (49) + v (7) + pad
#define PS_LDS_FLOATS (PS_P * PS_MV + (PS_P + 1) * 8)

// phase after g cut-off updates:  p <- fl(p + inc);  while (p >= 2 pi) p <- fl(p - 2 pi)   (oracle_ref.c:orc_phaser)
// Inside one binade [2^e, 2^(e+1)) every representable p is a multiple of ulp = 2^(e-23), so fl(p + inc) = p + d with ONE
// constant d = rn(inc / ulp) ulp -- except that a tie (inc / ulp = I + 1/2 exactly) rounds to even and can make the FIRST
// step from an odd multiple differ from all later ones.  Hence: three real steps; if the last two are equal and all
// three values share a binade, jump  p += j d  (exact in fp32: stays inside the binade, below 2 pi, one step of margin),
// then continue with real steps across the binade edge / the wrap.  ~25 rounds per LFO period; checked against the
// sequential loop for 600 (rate, count) pairs including constructed ties (tools/probe/check_phase_jump.py).
__device__ __forceinline__ float ps_step(float p, float inc, float two_pi)
{
    p = __fadd_rn(p, inc);
    while (p >= two_pi) p = __fsub_rn(p, two_pi);
    return p;
}
__device__ float ps_phase_after(int g, float inc, float two_pi)
{
    float p = 0.0f;
    int rem = g;
    while (rem > 0) {
        const float p1 = ps_step(p, inc, two_pi);
        p = p1;
        if (--rem == 0) break;
        const float p2 = ps_step(p1, inc, two_pi);
        p = p2;
        if (--rem == 0) break;
        const float p3 = ps_step(p2, inc, two_pi);
        p = p3;
        if (--rem == 0) break;
        const float d1 = __fsub_rn(p2, p1), d2 = __fsub_rn(p3, p2);
        const int e1 = (__float_as_int(p1) >> 23) & 0xff, e2 = (__float_as_int(p2) >> 23) & 0xff, e3 = (__float_as_int(p3) >> 23) & 0xff;
        if (d1 == d2 && d2 > 0.0f && e1 == e2 && e2 == e3 && e3 > 0 && e3 < 0xfe) {
            const double edge = (double)__int_as_float((e3 + 1) << 23);          // 2^(e+1)
            const double lim = edge < (double)two_pi ? edge : (double)two_pi;
            long long j = (long long)floor((lim - (double)p3) / (double)d2) - 1;
            if (j > rem) j = rem;
            if (j > 0) {
                p = (float)((double)p3 + (double)j * (double)d2);
                rem -= (int)j;
            }
        }
    }
    return p;
}

__global__ __launch_bounds__(PS_P) void phaser_scan_kernel(const float *__restrict__ x, long long x_stride,
                                                           const float *__restrict__ rate,
                                                           const float *__restrict__ depth,
                                                           const float *__restrict__ centre,
                                                           const float *__restrict__ feedback,
                                                           const float *__restrict__ mix,
                                                           const int *__restrict__ lead_arr,
                                                           const int *__restrict__ rows, int n_items, int N, double sr,
                                                           float *__restrict__ y, long long y_stride,
                                                           float *__restrict__ dry_out, float *__restrict__ gws,
                                                           long long gws_stride, int probe)
{
    extern __shared__ __attribute__((aligned(16))) float ps_lds[];
    float *mv = ps_lds, *zs = ps_lds + PS_P * PS_MV;
    const int p = threadIdx.x, lane = p & 63;
    const int item = blockIdx.x;
    const int b = rows ? rows[item] : item;
    const int lead = lead_arr ? lead_arr[b] : 0;
    const int total = lead + N;
    const float *xb = x + (size_t)b * x_stride;
    float *yb = y + (size_t)b * y_stride;
    float *db = dry_out ? dry_out + (size_t)b * y_stride : nullptr;
    float *gw = gws + (size_t)item * gws_stride;

    const float two_pi = 6.283185307179586476925286766559f;
    const float pi_f = 3.14159265358979323846f;
    const float fmax_hz = (float)fmin(20000.0, 0.49 * sr);
    const float log_min = (float)log10(20.0), log_max = (float)log10((double)fmax_hz);
    const float inc = __fmul_rn(__fdiv_rn(two_pi, (float)(sr / 4.0)), rate[b]);
    const float norm_centre = __fdiv_rn(__fsub_rn((float)log10((double)centre[b]), log_min), __fsub_rn(log_max, log_min));
    const float osc_vol = __fmul_rn(depth[b], 0.5f);
    const float fb = feedback[b];
    const float wet_g = mix[b], dry_g = __fsub_rn(1.0f, mix[b]);

    const int n_groups = (total + 3) >> 2;                       // cut-off updates = groups of 4 samples from sample 0
    const int gpc = (n_groups + PS_P - 1) / PS_P;                // groups per chunk
    const int g0 = p * gpc, g1 = min(g0 + gpc, n_groups);        // this lane's groups [g0, g1) (empty beyond the clip)

    const bool use_ws = gws != nullptr && (long long)n_groups <= gws_stride;
    // one cut-off update (oracle_ref.c:orc_phaser; sin / pow / tan in fp64 and rounded once = the host libm's float results)
    auto cutoff = [&](float ph) {
        const float osc = (float)sin((double)__fsub_rn(ph, pi_f));
        float lfo = __fadd_rn(__fmul_rn(osc, osc_vol), norm_centre);
        lfo = lfo < 0.0f ? 0.0f : (lfo > 1.0f ? 1.0f : lfo);
        const float fc = (float)pow(10.0, (double)__fadd_rn(__fmul_rn(lfo, __fsub_rn(log_max, log_min)), log_min));
        const float gg = (float)tan(3.14159265358979323846 * (double)fc / sr);
        return __fdiv_rn(gg, __fadd_rn(1.0f, gg));
    };

#define PS_STAGE(S)                                   \
    v = __fmul_rn(G, __fsub_rn(out, S));             \
    yk = __fadd_rn(v, S);                            \
    S = __fadd_rn(v, yk);                            \
    out = __fsub_rn(__fmul_rn(2.0f, yk), out);

    // ---- A: the chunk's affine map, the cut-offs on the way
    float S[8][7];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 7; ++c) S[r][c] = r == c ? 1.0f : 0.0f;
    float phase = ps_phase_after(min(g0, n_groups), inc, two_pi);
    for (int g = g0; g < g1; ++g) {
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * g + j;
            xv[j] = probe ? 0.25f : (n < total ? xb[n] : 0.0f);
        }
        const float G = cutoff(phase);
        phase = ps_step(phase, inc, two_pi);
        if (use_ws) gw[g] = G;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float in = r == 7 ? xv[j] : 0.0f;
                float out = __fsub_rn(in, S[r][6]), v, yk;
                PS_STAGE(S[r][0]) PS_STAGE(S[r][1]) PS_STAGE(S[r][2]) PS_STAGE(S[r][3]) PS_STAGE(S[r][4]) PS_STAGE(S[r][5])
                S[r][6] = __fmul_rn(out, fb);
            }
        }
    }
    {
        float *m = mv + p * PS_MV;
#pragma unroll
        for (int c = 0; c < 7; ++c)
#pragma unroll
            for (int r = 0; r < 7; ++r) m[c * 7 + r] = S[c][r];      // column c = image of unit state c
#pragma unroll
        for (int r = 0; r < 7; ++r) m[49 + r] = S[7][r];
    }
    __syncthreads();

    // ---- B: chain the maps (wave 0; lane r < 7 owns row r, the new state is broadcast with v_readlane)
    if (p < 64) {
        const int r = lane < 7 ? lane : 0;
        float z[7], mine = 0.0f;
#pragma unroll
        for (int c = 0; c < 7; ++c) z[c] = 0.0f;
        for (int q = 0; q < PS_P; ++q) {
            if (lane < 7) zs[q * 8 + lane] = mine;
            const float *m = mv + q * PS_MV;
            float acc = m[49 + r];
#pragma unroll
            for (int c = 0; c < 7; ++c) acc = __builtin_fmaf(m[c * 7 + r], z[c], acc);
            mine = acc;
#pragma unroll
            for (int c = 0; c < 7; ++c) z[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), c));
        }
    }
    __syncthreads();

    // ---- C: the chunk from its true start state, JUCE's operation order, output window
    {
        float s0 = zs[p * 8 + 0], s1 = zs[p * 8 + 1], s2 = zs[p * 8 + 2], s3 = zs[p * 8 + 3], s4 = zs[p * 8 + 4],
              s5 = zs[p * 8 + 5], last = zs[p * 8 + 6];
        float phase_c = use_ws ? 0.0f : ps_phase_after(min(g0, n_groups), inc, two_pi);
        for (int g = g0; g < g1; ++g) {
            float G;
            if (use_ws) G = gw[g];
            else {                                               // workspace too small for this clip: evaluate the cut-off again
                G = cutoff(phase_c);
                phase_c = ps_step(phase_c, inc, two_pi);
            }
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 4 * g + j;
                xv[j] = probe ? 0.25f : (n < total ? xb[n] : 0.0f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 4 * g + j;
                const float in = xv[j];
                float out = __fsub_rn(in, last), v, yk;
                PS_STAGE(s0) PS_STAGE(s1) PS_STAGE(s2) PS_STAGE(s3) PS_STAGE(s4) PS_STAGE(s5)
                last = __fmul_rn(out, fb);
                float m = __fadd_rn(__fmul_rn(out, wet_g), __fmul_rn(in, dry_g));
                m = m < -1.0f ? -1.0f : (m > 1.0f ? 1.0f : m);
                if (n >= lead && n < total && (!probe || g + 1 == g1)) {
                    yb[n - lead] = m;
                    if (db) db[n - lead] = in;
                }
            }
        }
    }
#undef PS_STAGE
}

// x: source audio, row b at x + b*x_stride, at least lead[b] + N samples; rate, depth, centre,
// feedback, mix: (B,) fp32; lead: (B,) int32 warm-up samples (NULL = 0); rows/n_rows: optional
// subset of clip indices.  y: row b at y + b*y_stride, N samples = processed[lead : lead+N];
// dry_out (optional, same stride as y): the matching crop of the source.
// exact_order != 0: the whole clip sample by sample on one wavefront in JUCE's operation order (the bit reference);
// 0: the time-parallel linear scan above.  workspace: floats, row `item` at workspace + item * workspace_stride, at least
// ceil((lead + N) / 4) floats per processed clip (the cut-off of every 4-sample group, kept between the two passes of
// the scan); optional: NULL or too short a row makes the kernel evaluate the cut-offs twice instead.
static int phaser_fwd_launch(const float *x, int64_t x_stride, const float *rate, const float *depth,
                            const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                            const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                            float *y, int64_t y_stride, float *dry_out, float *workspace, int64_t workspace_stride,
                            void *stream, int probe)
{
    if (!x || !rate || !depth || !centre || !feedback || !mix || !y || B <= 0 || N <= 0 || sr <= 0.0) return MX_ERR_ARG;
    if (workspace && workspace_stride <= 0) return MX_ERR_ARG;
    if (N >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const int64_t items = rows ? n_rows : B;
    if (items <= 0) return MX_OK;
    if (exact_order) {
        const dim3 grid((unsigned)((items + PH_WPB - 1) / PH_WPB)), block(64 * PH_WPB);
        hipLaunchKernelGGL(phaser_kernel, grid, block, 0, (hipStream_t)stream, x, (long long)x_stride, rate, depth, centre,
                           feedback, mix, lead, rows, (int)items, (int)N, (float)sr, sr, y, (long long)y_stride, dry_out);
    } else {
        const size_t lds = PS_LDS_FLOATS * sizeof(float);
        static bool attr_set[64] = {};                       // per device: one process may drive several GPUs
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            (void)hipFuncSetAttribute((const void *)phaser_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(phaser_scan_kernel, dim3((unsigned)items), dim3(PS_P), lds, (hipStream_t)stream, x, (long long)x_stride,
                           rate, depth, centre, feedback, mix, lead, rows, (int)items, (int)N, sr, y, (long long)y_stride,
                           dry_out, workspace, (long long)workspace_stride, probe);
    }
    return mx_launch_status();
}

MX_EXPORT int mx_phaser_fwd(const float *x, int64_t x_stride, const float *rate, const float *depth,
                            const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                            const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                            float *y, int64_t y_stride, float *dry_out, float *workspace, int64_t workspace_stride,
                            void *stream)
{
    return phaser_fwd_launch(x, x_stride, rate, depth, centre, feedback, mix, lead, rows, n_rows, B, N, sr, exact_order, y, y_stride, dry_out, workspace, workspace_stride, stream, 0);
}

// Measurement twin (bench.py's serial floor): the SAME launch with no global-memory traffic inside the sample loop -- inputs are constants, only the last chunk is stored.  Results are meaningless; nothing in the product calls it.
MX_EXPORT int mx_phaser_fwd_probe(const float *x, int64_t x_stride, const float *rate, const float *depth,
                            const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                            const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                            float *y, int64_t y_stride, float *dry_out, float *workspace, int64_t workspace_stride,
                            void *stream)
{
    return phaser_fwd_launch(x, x_stride, rate, depth, centre, feedback, mix, lead, rows, n_rows, B, N, sr, exact_order, y, y_stride, dry_out, workspace, workspace_stride, stream, 1);
}

// ---- independent floor of the scan (bench.py) -----------------------------------------------------------
// `mx_phaser_fwd_probe` re-runs the product kernel without global traffic: that floor inherits its schedule.  This
// microbenchmark is nothing but the arithmetic NO schedule of the scan can avoid: the 6-stage all-pass cascade of one sample
// (JUCE's FirstOrderTPTFilter step x 6 + feedback, 38 flops) for EIGHT independent state vectors per lane -- the shape of
// phase A (seven unit states + the driven zero state) -- on one 512-lane workgroup, with a constant cut-off and input: no loads,
// no stores, no fp64 sin / pow / tan, no chunk maps, no chaining.  time / (steps x 8) = the cost of one (sample, run) at the
// kernel's occupancy; a clip needs ceil((lead + N) / 512) samples per lane x 9 runs (8 of phase A + phase C).
__global__ __launch_bounds__(PS_P) void phaser_cascade_probe_kernel(int steps, float G, float fb, float *__restrict__ out)
{
    float s[8][6], last[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        last[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) s[r][k] = (r == k) ? 1.0f : 0.001f * (float)(threadIdx.x & 7);
    }
    float in = 0.25f + 1e-3f * (float)(threadIdx.x & 15);
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float o = __fsub_rn(r == 7 ? in : 0.0f, last[r]);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float v = __fmul_rn(G, __fsub_rn(o, s[r][k]));
                const float yk = __fadd_rn(v, s[r][k]);
                s[r][k] = __fadd_rn(v, yk);
                o = __fsub_rn(__fmul_rn(2.0f, yk), o);
            }
            last[r] = __fmul_rn(o, fb);
        }
        in = -in;
    }
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += last[r] + s[r][5];
    if (acc == 123.456f) out[threadIdx.x] = acc;             // keeps the chains live; never true for these constants
    if (threadIdx.x == 0) out[0] = acc;
}

// `steps` samples x 8 independent cascade runs per lane on one 512-lane workgroup (see above); out: >= 512 floats
MX_EXPORT int mx_phaser_cascade_probe(int64_t steps, float *out, void *stream)
{
    if (!out || steps <= 0 || steps >= (1ll << 30)) return MX_ERR_ARG;
    hipLaunchKernelGGL(phaser_cascade_probe_kernel, dim3(1), dim3(PS_P), 0, (hipStream_t)stream, (int)steps, 0.37f, 0.45f, out);
    return mx_launch_status();
}
