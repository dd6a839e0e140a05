// melspec.hip -- K4: STFT -> |.|^2 -> mel filter bank -> SpecAugment mask -> clip -> log.
// Reference: mod_extraction/models.py:170-181,199-208 (torchaudio.transforms.MelSpectrogram with
// n_fft 1024, hop 256, periodic hann, centre + reflect pad, power 2, HTK mels, no norm).
//
// One 256-thread workgroup produces MEL_FR consecutive frames of one (clip, channel) plane.
// Per frame: windowed, reflect-padded load -> 1024-point complex radix-4 Stockham FFT in LDS
// (5 passes, twiddles from a host-built fp64->fp32 table) -> power of bins 0..512 -> each thread
// accumulates one mel band over its non-zero filter range [lo, hi).  The MEL_FR results per band
// are buffered in LDS and written as 64-byte runs along the frame axis.
// Output planes use a padded row pitch (`out_pitch` floats, 352 for 345 frames) so that the conv
// kernels can load 16-byte aligned vectors; columns >= n_frames are written as 0.
//
// HBM-bound by design: 2*N*4 B read + 2*n_mels*frames*4 B written per clip (1.41 MB at N=88200);
// the FFT is ~17.7 MFLOP per clip.
#include "common.h"

#define MEL_NFFT 1024
#define MEL_FR 16

struct cfloat { float re, im; };
__device__ __forceinline__ cfloat cmul(cfloat a, cfloat b)
{
    cfloat r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
__device__ __forceinline__ cfloat cadd(cfloat a, cfloat b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cfloat csub(cfloat a, cfloat b) { return {a.re - b.re, a.im - b.im}; }

__global__ __launch_bounds__(256) void melspec_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ window,
    const float2 *__restrict__ twiddle,     // exp(-2 pi i m / 1024), m in [0,1024)
    const float *__restrict__ fb,           // (513, n_mels) row-major (torchaudio mel_scale.fb)
    const int *__restrict__ band_lo, const int *__restrict__ band_hi, int n_mels, int hop,
    int n_frames, int out_pitch, float eps, int f0, int f1, int t0, int t1, float *__restrict__ out)
{
    __shared__ cfloat bufA[MEL_NFFT];
    __shared__ cfloat bufB[MEL_NFFT];
    __shared__ float power[MEL_NFFT / 2 + 1];
    extern __shared__ float melbuf[];       // n_mels * (MEL_FR + 1)
    const int tid = threadIdx.x;
    const int plane = blockIdx.y;           // clip * in_ch + channel
    const int tile0 = blockIdx.x * MEL_FR;
    const float *xp = x + (size_t)plane * N;

    for (int fl = 0; fl < MEL_FR; ++fl) {
        const int t = tile0 + fl;
        if (t < n_frames) {                                   // block-uniform
            // ---- windowed load, centre=True reflect padding (torch.stft) ----
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tid + q * 256;
                int s = t * hop + n - MEL_NFFT / 2;
                if (s < 0) s = -s;
                if (s >= N) s = 2 * (N - 1) - s;
                s = s < 0 ? 0 : s;                            // clips shorter than n_fft/2 are rejected on the host
                bufA[n].re = xp[s] * window[n];
                bufA[n].im = 0.0f;
            }
            __syncthreads();
            // ---- 5 radix-4 Stockham passes: A->B->A->B->A->B ----
            cfloat *src = bufA, *dst = bufB;
#pragma unroll
            for (int pass = 0; pass < 5; ++pass) {
                const int Ns = 1 << (2 * pass);
                const int j = tid;
                const int k = j & (Ns - 1);
                const int tw_step = k * (MEL_NFFT / (Ns * 4));
                cfloat v0 = src[j], v1 = src[j + 256], v2 = src[j + 512], v3 = src[j + 768];
                if (pass > 0) {
                    float2 w1 = twiddle[tw_step], w2 = twiddle[2 * tw_step], w3 = twiddle[3 * tw_step];
                    v1 = cmul(v1, {w1.x, w1.y});
                    v2 = cmul(v2, {w2.x, w2.y});
                    v3 = cmul(v3, {w3.x, w3.y});
                }
                cfloat a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), d = csub(v1, v3);
                cfloat a3 = {d.im, -d.re};                    // -i * (v1 - v3)
                const int j0 = ((j - k) << 2) + k;
                dst[j0] = cadd(a0, a2);
                dst[j0 + Ns] = cadd(a1, a3);
                dst[j0 + 2 * Ns] = csub(a0, a2);
                dst[j0 + 3 * Ns] = csub(a1, a3);
                __syncthreads();
                cfloat *tmp = src; src = dst; dst = tmp;
            }
            // result is in `src` (= bufB after 5 swaps)
            for (int kk = tid; kk <= MEL_NFFT / 2; kk += 256) {
                cfloat z = src[kk];
                float mag = sqrtf(z.re * z.re + z.im * z.im);  // torchaudio: spec.abs().pow(2)
                power[kk] = mag * mag;
            }
            __syncthreads();
            for (int m = tid; m < n_mels; m += 256) {
                float acc = 0.0f;
                const int lo = band_lo[m], hi = band_hi[m];
                for (int kk = lo; kk < hi; ++kk) acc = fmaf(fb[(size_t)kk * n_mels + m], power[kk], acc);
                melbuf[m * (MEL_FR + 1) + fl] = acc;
            }
            __syncthreads();
        }
    }
    // ---- mask, clip, log, store (runs of MEL_FR frames per band) ----
    float *op = out + (size_t)plane * n_mels * out_pitch;
    for (int idx = tid; idx < n_mels * MEL_FR; idx += 256) {
        const int m = idx / MEL_FR, fl = idx % MEL_FR;
        const int t = tile0 + fl;
        if (t < out_pitch) {
            float v = 0.0f;
            if (t < n_frames) {
                float p = melbuf[m * (MEL_FR + 1) + fl];
                if ((m >= f0 && m < f1) || (t >= t0 && t < t1)) p = 0.0f;   // SpecAugment fill value 0
                v = logf(fmaxf(p, eps));
            }
            op[(size_t)m * out_pitch + t] = v;
        }
    }
}

// x: (planes, N) fp32 with planes = B*in_ch; window (1024,), twiddle (1024,) float2, fb (513,n_mels),
// band_lo/band_hi (n_mels,) int32 = non-zero row range of each fb column;
// out: (planes, n_mels, out_pitch) = log(clip(mel, eps)); mask ranges [f0,f1) x [t0,t1) (0,0 = none).
MX_EXPORT int mx_logmel_fwd(const float *x, int64_t planes, int64_t N, const float *window,
                            const float *twiddle, const float *fb, const int32_t *band_lo,
                            const int32_t *band_hi, int64_t n_fft, int64_t hop, int64_t n_mels,
                            int64_t n_frames, int64_t out_pitch, float eps, int32_t f0, int32_t f1,
                            int32_t t0, int32_t t1, float *out, void *stream)
{
    if (!x || !window || !twiddle || !fb || !band_lo || !band_hi || !out || planes <= 0) return MX_ERR_ARG;
    if (n_fft != MEL_NFFT || N <= MEL_NFFT / 2 || N >= (1ll << 30) || planes > 65535 || n_mels > 2048 ||
        out_pitch < n_frames)
        return MX_ERR_UNSUPPORTED;
    const int tiles = (int)((out_pitch + MEL_FR - 1) / MEL_FR);
    const size_t lds = (size_t)n_mels * (MEL_FR + 1) * sizeof(float);
    hipLaunchKernelGGL(melspec_kernel, dim3(tiles, (unsigned)planes), dim3(256), lds, (hipStream_t)stream,
                       x, (int)N, window, (const float2 *)twiddle, fb, band_lo, band_hi, (int)n_mels, (int)hop,
                       (int)n_frames, (int)out_pitch, eps, f0, f1, t0, t1, out);
    return mx_launch_status();
}
