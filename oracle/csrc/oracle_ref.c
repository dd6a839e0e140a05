/*
 * oracle_ref.c -- CPU restatement of the reference's sample-recurrent effects.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in mod_extraction_amd/ may link, load or
 * call this file; it is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * Compile with -O2 -ffp-contract=off (see oracle/Makefile): every expression
 * below must round exactly where the reference's separate torch ops round.
 *
 *   orc_flanger      follows mod_extraction/fx.py:72-119 (apply_effect)
 *   orc_phaser       restates pedalboard==0.7.3 Phaser == JUCE dsp::Phaser<float>
 *                    (third-party, source absent from /root/reference:
 *                    PARITY UNPINNED -- see oracle/README.md)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* torch.remainder for float32 (aten BinaryOpsKernel.cpp remainder_kernel):
 * mod = fmod(a, b); if (mod != 0 && ((b < 0) != (mod < 0))) mod += b;      */
static inline float torch_remainderf(float a, float b)
{
    float mod = fmodf(a, b);
    if ((mod != 0.0f) && ((b < 0.0f) != (mod < 0.0f))) mod += b;
    return mod;
}

/*
 * fx.py:72-119.  One mono channel per clip (n_ch == 1 on every call site).
 *   x, mod       : (B, N) float32
 *   lfo_scale[b] : max_lfo_delay_samples * width            (fx.py:99, first product)
 *   min_delay[b] : min_delay_width * max_min_delay_samples  (fx.py:98)
 *   feedback, depth, mix, one_minus_mix : (B,) float32
 *   M            : max_delay_samples (fx.py:42)
 *   y            : (B, N) out;  idx_prev / frac (optional, may be NULL): the
 *                  bookkeeping arrays of fx.py:100-102 for index-parity tests.
 */
void orc_flanger(const float *x, const float *mod, const float *lfo_scale,
                 const float *min_delay, const float *feedback, const float *depth,
                 const float *mix, const float *one_minus_mix, int64_t B, int64_t N,
                 int64_t M, float *y, int64_t *idx_prev, float *frac_out)
{
    float *buf = (float *)malloc(sizeof(float) * (size_t)M);
    const float Mf = (float)M;
    for (int64_t b = 0; b < B; ++b) {
        memset(buf, 0, sizeof(float) * (size_t)M);               /* fx.py:92 */
        const float *xb = x + b * N, *mb = mod + b * N;
        float *yb = y + b * N;
        for (int64_t n = 0; n < N; ++n) {
            int64_t w = n % M;                                   /* fx.py:95 */
            float t = lfo_scale[b] * mb[n];                      /* fx.py:99 */
            float d = t + min_delay[b];
            float r0 = (float)w - d;                             /* fx.py:100 */
            float r1 = r0 + Mf;
            float r = torch_remainderf(r1, Mf);
            float fl = floorf(r);
            float frac = r - fl;                                 /* fx.py:101 */
            int64_t prev = (int64_t)fl;                          /* fx.py:102 */
            int64_t next = (prev + 1) % M;                       /* fx.py:103 */
            if (idx_prev) idx_prev[b * N + n] = prev;
            if (frac_out) frac_out[b * N + n] = frac;
            float pv = buf[prev], nv = buf[next];                /* fx.py:111-112 */
            float a = frac * nv;                                 /* fx.py:113 */
            float c = (1.0f - frac) * pv;
            float interp = a + c;
            float fbv = feedback[b] * interp;                    /* fx.py:114 */
            buf[w] = xb[n] + fbv;
            float dv = depth[b] * interp;                        /* fx.py:115 */
            float o = xb[n] + dv;
            float dry = one_minus_mix[b] * xb[n];                /* fx.py:117 */
            float wet = mix[b] * o;
            float s = dry + wet;
            yb[n] = s < -1.0f ? -1.0f : (s > 1.0f ? 1.0f : s);   /* fx.py:118 */
        }
    }
    free(buf);
}

/*
 * JUCE dsp::Phaser<float> as wrapped by pedalboard.Phaser (call site
 * datasets.py:455-482).  Restated from the published JUCE 6/7 sources:
 *   - 6 FirstOrderTPTFilter all-pass stages, G = g/(1+g), g = (float)tan(pi*fc/sr) in double
 *   - sine LFO evaluated at sr/4 (maxUpdateCounter = 4): osc = sin(phase - pi),
 *     phase advanced by 2*pi*rate/(sr/4) and wrapped at 2*pi, first value at phase 0
 *   - lfo = clamp(osc*depth*0.5 + mapFromLog10(centre, 20, min(20000, 0.49 sr)), 0, 1)
 *     fc  = mapToLog10(lfo, 20, min(20000, 0.49 sr))
 *   - per sample: in - lastOut -> 6 stages -> out; lastOut = out*feedback
 *   - linear dry/wet mix: y = wet*mix + dry*(1-mix); then clip to [-1,1] (datasets.py:472)
 *   Parameter smoothers are snapped by prepare()/reset() so constants apply from sample 0.
 *   x, y: (B, N);  rate, depth, centre, feedback, mix: (B,)
 *   lfo_out (optional): (B, ceil(N/4)) normalised lfo in [0,1] for diagnostics.
 */
void orc_phaser(const float *x, const float *rate, const float *depth,
                const float *centre, const float *feedback, const float *mix,
                int64_t B, int64_t N, double sr, float *y, float *lfo_out)
{
    const float two_pi = 6.283185307179586476925286766559f;
    const float pi_f = 3.14159265358979323846f;
    const float fmax = (float)fmin(20000.0, 0.49 * sr);
    const float log_min = log10f(20.0f), log_max = log10f(fmax);
    const int64_t nd = (N + 3) / 4;
    for (int64_t b = 0; b < B; ++b) {
        const float *xb = x + b * N;
        float *yb = y + b * N;
        float s[6] = {0, 0, 0, 0, 0, 0};
        float last = 0.0f, G = 0.0f;
        float phase = 0.0f;
        const float inc = (two_pi / (float)(sr / 4.0)) * rate[b];
        const float norm_centre = (log10f(centre[b]) - log_min) / (log_max - log_min);
        const float osc_vol = depth[b] * 0.5f;
        const float wet_g = mix[b], dry_g = 1.0f - mix[b];
        for (int64_t n = 0; n < N; ++n) {
            if ((n & 3) == 0) {
                float osc = sinf(phase - pi_f);
                phase += inc;
                while (phase >= two_pi) phase -= two_pi;
                float lfo = osc * osc_vol + norm_centre;
                lfo = lfo < 0.0f ? 0.0f : (lfo > 1.0f ? 1.0f : lfo);
                if (lfo_out) lfo_out[b * nd + (n >> 2)] = lfo;
                float fc = powf(10.0f, lfo * (log_max - log_min) + log_min);
                float g = (float)tan(3.14159265358979323846 * (double)fc / sr);
                G = g / (1.0f + g);
            }
            float in = xb[n];
            float out = in - last;
            for (int k = 0; k < 6; ++k) {
                float v = G * (out - s[k]);
                float yk = v + s[k];
                s[k] = v + yk;
                out = 2.0f * yk - out;
            }
            last = out * feedback[b];
            float m = out * wet_g + in * dry_g;
            yb[n] = m < -1.0f ? -1.0f : (m > 1.0f ? 1.0f : m);
        }
    }
}
