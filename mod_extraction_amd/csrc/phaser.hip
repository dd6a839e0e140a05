// phaser.hip -- K3: 6-stage all-pass phaser with feedback (reference call site:
// mod_extraction/datasets.py:455-482 -> pedalboard==0.7.3 Phaser == JUCE dsp::Phaser<float>;
// third-party source absent from the reference tree: PARITY UNPINNED, checked against
// oracle/csrc/oracle_ref.c:orc_phaser, which restates the published JUCE algorithm).
//
// One wavefront per clip (16 clips per workgroup).  The recurrence (in - lastOut -> 6 first-order TPT all-pass stages ->
// out; lastOut = out * feedback) is strictly serial per sample, so the wave splits the work by
// kind: per block of 256 samples the 64 lanes evaluate the 64 LFO / cut-off updates in parallel
// (sin, pow, fp64 tan -> G = g / (1 + g), one per lane), samples are loaded/stored 4 per lane
// coalesced (lane registers, broadcast with v_readlane), and only the 256-sample dependent chain
// runs wave-uniformly.
// sin / pow / log10 are evaluated in fp64 and rounded once, which reproduces the host libm's (correctly
// rounded) float results; the LFO phase accumulator is advanced sequentially in fp32 (as JUCE does) to stay bit-faithful.
// `lead` samples are processed (filter warm-up, LFO phase) before the N output samples: the
// reference renders n + sr/rate samples and crops at a random offset (datasets.py:428-449).
// Algorithmic HBM traffic: 8 B/sample (+4 B/sample when the cropped dry clip is also written).
#include "common.h"

#define PH_BLOCK 256
#ifndef PH_WPB
#define PH_WPB 4          // clips (wavefronts) per workgroup: packs the long-running serial chains onto few CUs so
                          // that the 1-workgroup-per-CU matrix kernels of the train step keep the other CUs
#endif

template <bool FAST>
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
        // FAST: the same all-pass stage written as out = (2G-1) x + (2-2G) s, s' = 2G x + (1-2G) s: one FMA
        // per stage on the sample-to-sample critical path instead of five dependent operations (4-5x
        // shorter chain).  Algebraically identical to the JUCE order, rounding differs (<= 1e-6 on audio).
        const float c1r = 2.0f * Greg - 1.0f, c2r = 2.0f - 2.0f * Greg, c3r = 2.0f * Greg, c4r = 1.0f - 2.0f * Greg;
        float fc1 = 0.f, fc2 = 0.f, fc3 = 0.f, fc4 = 0.f;
#pragma unroll
        for (int j = 0; j < PH_BLOCK / 64; ++j) {
            const int lim = min(64, cnt - j * 64);
            for (int li = 0; li < lim; ++li) {
                const float in = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr[j]), li));
                float out = __fsub_rn(in, last);
                if (FAST) {
                    // coefficients change every 4 samples (one cut-off update): re-broadcast only then
                    if ((li & 3) == 0) {
                        const int src = (j * 64 + li) >> 2;
                        fc1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c1r), src));
                        fc2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c2r), src));
                        fc3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c3r), src));
                        fc4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c4r), src));
                    }
#define PH_FSTAGE(S)                              \
    {                                             \
        const float xin = out;                    \
        out = fmaf(fc1, xin, fc2 * S);            \
        S = fmaf(fc3, xin, fc4 * S);              \
    }
                    PH_FSTAGE(s0) PH_FSTAGE(s1) PH_FSTAGE(s2) PH_FSTAGE(s3) PH_FSTAGE(s4) PH_FSTAGE(s5)
#undef PH_FSTAGE
                    last = __fmul_rn(out, fb);
                    float m = __fadd_rn(__fmul_rn(out, wet_g), __fmul_rn(in, dry_g));
                    m = m < -1.0f ? -1.0f : (m > 1.0f ? 1.0f : m);
                    yr[j] = lane == li ? m : yr[j];
                    continue;
                }
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

// x: source audio, row b at x + b*x_stride, at least lead[b] + N samples; rate, depth, centre,
// feedback, mix: (B,) fp32; lead: (B,) int32 warm-up samples (NULL = 0); rows/n_rows: optional
// subset of clip indices.  y: row b at y + b*y_stride, N samples = processed[lead : lead+N];
// dry_out (optional, same stride as y): the matching crop of the source.
// exact_order != 0: evaluate every all-pass stage in JUCE's operation order (v = G (x - s); y = v + s;
// s = v + y; out = 2 y - x); 0: the algebraically identical FMA form with a 4-5x shorter dependency chain.
MX_EXPORT int mx_phaser_fwd(const float *x, int64_t x_stride, const float *rate, const float *depth,
                            const float *centre, const float *feedback, const float *mix, const int32_t *lead,
                            const int32_t *rows, int64_t n_rows, int64_t B, int64_t N, double sr, int32_t exact_order,
                            float *y, int64_t y_stride, float *dry_out, void *stream)
{
    if (!x || !rate || !depth || !centre || !feedback || !mix || !y || B <= 0 || N <= 0 || sr <= 0.0) return MX_ERR_ARG;
    if (N >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const int64_t items = rows ? n_rows : B;
    if (items <= 0) return MX_OK;
    if (exact_order)
        hipLaunchKernelGGL((phaser_kernel<false>), dim3((unsigned)((items + PH_WPB - 1) / PH_WPB)), dim3(64 * PH_WPB), 0,
                           (hipStream_t)stream, x, (long long)x_stride, rate, depth, centre, feedback, mix, lead, rows,
                           (int)items, (int)N, (float)sr, sr, y, (long long)y_stride, dry_out);
    else
        hipLaunchKernelGGL((phaser_kernel<true>), dim3((unsigned)((items + PH_WPB - 1) / PH_WPB)), dim3(64 * PH_WPB), 0,
                           (hipStream_t)stream, x, (long long)x_stride, rate, depth, centre, feedback, mix, lead, rows,
                           (int)items, (int)N, (float)sr, sr, y, (long long)y_stride, dry_out);
    return mx_launch_status();
}
