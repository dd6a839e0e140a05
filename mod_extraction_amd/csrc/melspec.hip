// melspec.hip -- K4: STFT -> |.|^2 -> mel filter bank -> SpecAugment mask -> clip -> log.
// Reference: mod_extraction/models.py:170-181,199-208 (torchaudio.transforms.MelSpectrogram with
// n_fft 1024, hop 256, periodic hann, centre + reflect pad, power 2, HTK mels, no norm).
//
// One 256-thread workgroup produces MEL_FR consecutive frames of one (clip, channel) plane, ONE FRAME PER WAVEFRONT at a
// time: windowed, reflect-padded load -> 1024-point complex radix-4 Stockham FFT in wave-private LDS (5 passes, 4
// butterflies per lane and pass, twiddles from a host-built fp64->fp32 table) -> power of bins 0..512 -> each lane
// accumulates its mel bands over their non-zero filter range [lo, hi).  A wave executes in lockstep and the LDS serves
// its instructions in order, so a pass reads all its inputs into registers and writes the outputs back IN PLACE with no
// workgroup barrier at all (the first version spread a frame over the 256 threads: nine __syncthreads per frame, 144
// per workgroup, and the kernel was bound by them; same butterflies in the same order -> same bits).  The MEL_FR
// results per band are buffered in LDS and written as 64-byte runs along the frame axis.
// Output planes use a padded row pitch (`out_pitch` floats, 352 for 345 frames) so that the conv
// kernels can load 16-byte aligned vectors; columns >= n_frames are written as 0.
//
// HBM-bound by design: 2*N*4 B read + 2*n_mels*frames*4 B written per clip (1.41 MB at N=88200);
// the FFT is ~17.7 MFLOP per clip.
#include "wave_fft.h"

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
    int n_frames, int out_pitch, float eps, int f0, int f1, int t0, int t1, int coef_cap, float *__restrict__ out)
{
    __shared__ cfloat fftbuf[4][MEL_NFFT + 64];   // exchange buffer of a wave (padded layouts, see below)
    __shared__ float powbuf[4][MEL_NFFT / 2 + 1];
    __shared__ float2 tw_s[MEL_NFFT];        // the twiddle table: 12 global loads per lane and pass otherwise
    extern __shared__ float melbuf[];       // n_mels * (MEL_FR + 1) results, then the filter bank's non-zero coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int plane = blockIdx.y;           // clip * in_ch + channel
    const int tile0 = blockIdx.x * MEL_FR;
    const float *xp = x + (size_t)plane * N;
    cfloat *buf = fftbuf[wave];
    float *power = powbuf[wave];
    // The non-zero filter coefficients of every band, packed band after band in LDS once per workgroup: the band loop
    // below read them from global memory with one dependent L2 round trip per bin and frame.  boff[m] = start of band m
    // (exclusive prefix sum of the band widths, serial: n_mels is a few hundred); coef_cap floats are available.
    float *coef = melbuf + n_mels * (MEL_FR + 1);
    int *boff = reinterpret_cast<int *>(coef + coef_cap);
    for (int i = tid; i < MEL_NFFT; i += 256) tw_s[i] = twiddle[i];
    if (tid == 0) {
        int acc = 0;
        for (int m = 0; m < n_mels; ++m) { boff[m] = acc; acc += band_hi[m] - band_lo[m]; }
        boff[n_mels] = acc;
    }
    __syncthreads();
    const bool packed = boff[n_mels] <= coef_cap;              // block-uniform; otherwise the bands read fb directly
    if (packed) {
        for (int m = tid; m < n_mels; m += 256) {
            const int lo = band_lo[m], hi = band_hi[m], o = boff[m];
            for (int kk = lo; kk < hi; ++kk) coef[o + kk - lo] = fb[(size_t)kk * n_mels + m];
        }
    }
    __syncthreads();

    for (int fl = wave; fl < MEL_FR; fl += 4) {
        const int t = tile0 + fl;
        if (t >= n_frames) continue;                           // wave-uniform
        // ---- windowed load, centre=True reflect padding (torch.stft), straight into registers: lane a computes the
        //      butterflies j = a + 64 b (b = 0..3) of pass 0, whose inputs are samples j + 256 c ----
        cfloat R[4][4];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int n = lane + 64 * bq + 256 * cq;
                int sidx = t * hop + n - MEL_NFFT / 2;
                if (sidx < 0) sidx = -sidx;
                if (sidx >= N) sidx = 2 * (N - 1) - sidx;
                sidx = sidx < 0 ? 0 : sidx;                    // clips shorter than n_fft/2 are rejected on the host
                R[bq][cq].re = xp[sidx] * window[n];
                R[bq][cq].im = 0.0f;
            }
        // One radix-4 Stockham butterfly of pass `pass` (Ns = 4^pass): inputs v[c] = src[j + 256 c], outputs o[r] =
        // dst[j0 + r Ns].  The arithmetic (and so every bit of the result) does not depend on which lane runs it.
        auto bfly = [&](int pass, int j, const cfloat (&vin)[4], cfloat (&o)[4]) {
            const int Ns = 1 << (2 * pass);
            const int k = j & (Ns - 1);
            const int tw_step = k * (MEL_NFFT / (Ns * 4));
            cfloat v0 = vin[0], v1 = vin[1], v2 = vin[2], v3 = vin[3];
            if (pass > 0) {
                float2 w1 = tw_s[tw_step], w2 = tw_s[2 * tw_step], w3 = tw_s[3 * tw_step];
                v1 = cmul(v1, {w1.x, w1.y});
                v2 = cmul(v2, {w2.x, w2.y});
                v3 = cmul(v3, {w3.x, w3.y});
            }
            cfloat a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), d = csub(v1, v3);
            cfloat a3 = {d.im, -d.re};                        // -i * (v1 - v3)
            o[0] = cadd(a0, a2);
            o[1] = cadd(a1, a3);
            o[2] = csub(a0, a2);
            o[3] = csub(a1, a3);
        };
        auto out_pos = [](int pass, int j, int r) {
            const int Ns = 1 << (2 * pass), k = j & (Ns - 1);
            return ((j - k) << 2) + k + r * Ns;
        };
        // The five passes run in THREE register stages.  A lane that computes butterflies a + 64 b of an even pass p
        // holds exactly the inputs of four butterflies of pass p + 1 (positions j' + 256 c with j' = out_pos(p, a, r)),
        // so passes (0,1) and (2,3) need no exchange between them; two LDS exchanges remain (was: five LDS round trips
        // of the whole frame, the kernel's bottleneck), both with padded layouts that spread a half-wave over all banks.
        cfloat O[4][4], P[4][4];
        // -- passes 0 + 1
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) bfly(0, lane + 64 * bq, R[bq], O[bq]);      // O[b][r] = position 4 a + 256 b + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cfloat vin[4] = {O[0][r], O[1][r], O[2][r], O[3][r]};
            bfly(1, 4 * lane + r, vin, P[r]);                                          // positions 16 a + r + 4 r'
        }
        // -- exchange 1: position q lives at q + (q >> 4)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) {
                const int q = out_pos(1, 4 * lane + r, r2);
                buf[q + (q >> 4)] = P[r][r2];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int bq = 0; bq < 4; ++bq)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int q = lane + 64 * bq + 256 * cq;
                R[bq][cq] = buf[q + (q >> 4)];
            }
        __builtin_amdgcn_wave_barrier();
        // -- passes 2 + 3
        const int k2 = lane & 15;
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) bfly(2, lane + 64 * bq, R[bq], O[bq]);      // O[b][r] = position 4 (a - k2) + k2 + 16 r + 256 b
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cfloat vin[4] = {O[0][r], O[1][r], O[2][r], O[3][r]};
            bfly(3, 4 * (lane - k2) + k2 + 16 * r, vin, P[r]);
        }
        // -- exchange 2: position q lives at q + 16 (q >> 8)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) {
                const int q = out_pos(3, 4 * (lane - k2) + k2 + 16 * r, r2);
                buf[q + 16 * (q >> 8)] = P[r][r2];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int bq = 0; bq < 4; ++bq)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int q = lane + 64 * bq + 256 * cq;
                R[bq][cq] = buf[q + 16 * (q >> 8)];
            }
        __builtin_amdgcn_wave_barrier();
        // -- pass 4: butterfly j = a + 64 b delivers bins j + 256 r; the power of bins 0..512 goes to LDS for the bands
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            const int j = lane + 64 * bq;
            bfly(4, j, R[bq], O[bq]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int bin = j + 256 * r;
                if (bin <= MEL_NFFT / 2) {
                    const cfloat z = O[bq][r];
                    const float mag = sqrtf(z.re * z.re + z.im * z.im);  // torchaudio: spec.abs().pow(2)
                    power[bin] = mag * mag;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < n_mels; m += 64) {
            float acc = 0.0f;
            const int lo = band_lo[m], hi = band_hi[m];
            if (packed) {
                const float *cm = coef + boff[m] - lo;
                for (int kk = lo; kk < hi; ++kk) acc = fmaf(cm[kk], power[kk], acc);
            } else {
                for (int kk = lo; kk < hi; ++kk) acc = fmaf(fb[(size_t)kk * n_mels + m], power[kk], acc);
            }
            melbuf[m * (MEL_FR + 1) + fl] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
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

// ---- n_fft = 512 / 2048: the same pipeline on the shared wavefront FFT (wave_fft.h) ---------------------------------
// One frame per wavefront (two for 512) as a complex transform of (x, 0): the same passes, exchanges and twiddle table as
// the MR-STFT loss's forward transform; power of bins 0..NF/2 -> LDS -> mel bands -> mask, clip, log exactly as above.
// (The reference's own configs all use n_fft = 1024 -- the kernel above; this one closes the constructor's other sizes.)
template <int NF>
__global__ __launch_bounds__(WF<NF>::WAVES * 64) void melspec_wf_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ window, const float2 *__restrict__ twiddle,
    const float *__restrict__ fb, const int *__restrict__ band_lo, const int *__restrict__ band_hi, int n_mels, int hop,
    int n_frames, int out_pitch, float eps, int f0, int f1, int t0, int t1, int coef_cap, float *__restrict__ out)
{
    constexpr int L = WF<NF>::L, E = WF<NF>::E, NB = WF<NF>::NB, WAVES = WF<NF>::WAVES, FW = WF<NF>::FW, NT = WAVES * 64;
    __shared__ cf fftbuf[WAVES * FW][WF<NF>::LEN];
    __shared__ float powbuf[WAVES * FW][NF / 2 + 1];
    __shared__ cf tw_s[NF];
    extern __shared__ float melbuf[];       // n_mels * (MEL_FR + 1) results, then the filter bank's non-zero coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane / L, a = lane % L;
    const int plane = blockIdx.y;
    const int tile0 = blockIdx.x * MEL_FR;
    const float *xp = x + (size_t)plane * N;
    cf *buf = fftbuf[wave * FW + g];
    float *power = powbuf[wave * FW + g];
    float *coef = melbuf + n_mels * (MEL_FR + 1);
    int *boff = reinterpret_cast<int *>(coef + coef_cap);
    for (int i = tid; i < NF; i += NT) {
        const float2 w = twiddle[i];
        tw_s[i] = {w.x, w.y};
    }
    if (tid == 0) {
        int acc = 0;
        for (int m = 0; m < n_mels; ++m) { boff[m] = acc; acc += band_hi[m] - band_lo[m]; }
        boff[n_mels] = acc;
    }
    __syncthreads();
    const bool packed = boff[n_mels] <= coef_cap;
    if (packed) {
        for (int m = tid; m < n_mels; m += NT) {
            const int lo = band_lo[m], hi = band_hi[m], o = boff[m];
            for (int kk = lo; kk < hi; ++kk) coef[o + kk - lo] = fb[(size_t)kk * n_mels + m];
        }
    }
    __syncthreads();
    FftLane<NF> fl_;
    fft_lane_setup<NF, 1>(fl_, buf, tw_s, twiddle, a);

    for (int fb0 = wave * FW; fb0 < MEL_FR; fb0 += WAVES * FW) {
        const int fl = fb0 + g;                                 // this stream's frame slot
        if (tile0 + fb0 >= n_frames) break;                     // wave-uniform
        const int t = tile0 + fl;
        const bool live = t < n_frames && fl < MEL_FR;
        const int tt = live ? t : n_frames - 1;
        cf R[NB][4], Z[E];
#pragma unroll
        for (int bq = 0; bq < NB; ++bq)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int n = a + L * bq + (NF / 4) * cq;
                int sidx = tt * hop + n - NF / 2;
                if (sidx < 0) sidx = -sidx;
                if (sidx >= N) sidx = 2 * (N - 1) - sidx;
                sidx = sidx < 0 ? 0 : sidx;                    // clips shorter than n_fft/2 are rejected on the host
                R[bq][cq] = {xp[sidx] * window[n], 0.0f};
            }
        wave_fft<NF, false>(R, Z, fl_);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int bin = pos_final<NF>(i, a);
            if (bin <= NF / 2) {
                const float mag = sqrtf(Z[i].x * Z[i].x + Z[i].y * Z[i].y);  // torchaudio: spec.abs().pow(2)
                power[bin] = mag * mag;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            for (int m = a; m < n_mels; m += L) {
                float acc = 0.0f;
                const int lo = band_lo[m], hi = band_hi[m];
                if (packed) {
                    const float *cm = coef + boff[m] - lo;
                    for (int kk = lo; kk < hi; ++kk) acc = fmaf(cm[kk], power[kk], acc);
                } else {
                    for (int kk = lo; kk < hi; ++kk) acc = fmaf(fb[(size_t)kk * n_mels + m], power[kk], acc);
                }
                melbuf[m * (MEL_FR + 1) + fl] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float *op = out + (size_t)plane * n_mels * out_pitch;
    for (int idx = tid; idx < n_mels * MEL_FR; idx += NT) {
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

// x: (planes, N) fp32 with planes = B*in_ch; window (n_fft,), twiddle (n_fft,) float2 = exp(-2 pi i m / n_fft), fb (n_fft/2+1,n_mels),
// n_fft in {512, 1024, 2048};
// band_lo/band_hi (n_mels,) int32 = non-zero row range of each fb column;
// out: (planes, n_mels, out_pitch) = log(clip(mel, eps)); mask ranges [f0,f1) x [t0,t1) (0,0 = none).
MX_EXPORT int mx_logmel_fwd(const float *x, int64_t planes, int64_t N, const float *window,
                            const float *twiddle, const float *fb, const int32_t *band_lo,
                            const int32_t *band_hi, int64_t n_fft, int64_t hop, int64_t n_mels,
                            int64_t n_frames, int64_t out_pitch, float eps, int32_t f0, int32_t f1,
                            int32_t t0, int32_t t1, float *out, void *stream)
{
    if (!x || !window || !twiddle || !fb || !band_lo || !band_hi || !out || planes <= 0) return MX_ERR_ARG;
    if ((n_fft != 512 && n_fft != 1024 && n_fft != 2048) || N <= n_fft / 2 || N >= (1ll << 30) || planes > 65535 || n_mels > 2048 ||
        out_pitch < n_frames)
        return MX_ERR_UNSUPPORTED;
    const int tiles = (int)((out_pitch + MEL_FR - 1) / MEL_FR);
    const int coef_cap = 2 * ((int)n_fft / 2 + 1) + 2 * (int)n_mels;    // triangular filters: every bin lies in <= 2 bands
    const size_t lds = ((size_t)n_mels * (MEL_FR + 1) + coef_cap + n_mels + 1) * sizeof(float);
    // static (FFT exchange buffers, twiddles) + dynamic (mel tile, packed filter coefficients) LDS must fit the CU's 160 KB:
    // a large n_mels is MX_ERR_UNSUPPORTED here, not a failed launch; above 64 KB the dynamic part needs the function attribute
    const void *fn = n_fft == MEL_NFFT ? (const void *)melspec_kernel
                     : n_fft == 512    ? (const void *)melspec_wf_kernel<512>
                                       : (const void *)melspec_wf_kernel<2048>;
    static size_t static_lds[3] = {0, 0, 0};                      // a constant of the code object, cached
    static MxLdsLatch latch[3] = {};
    const int ki = n_fft == MEL_NFFT ? 0 : n_fft == 512 ? 1 : 2;
    if (!static_lds[ki]) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return MX_ERR_LAUNCH;
        static_lds[ki] = fa.sharedSizeBytes ? fa.sharedSizeBytes : 1;
    }
    const size_t lds_cap = 160 * 1024;
    if (static_lds[ki] + lds > lds_cap) return MX_ERR_UNSUPPORTED;
    if (static_lds[ki] + lds > 64 * 1024 && mx_set_dyn_lds(latch[ki], fn, lds_cap - static_lds[ki]) != MX_OK) return MX_ERR_LAUNCH;
    if (n_fft != MEL_NFFT) {
        if (n_fft == 512)
            hipLaunchKernelGGL((melspec_wf_kernel<512>), dim3(tiles, (unsigned)planes), dim3(WF<512>::WAVES * 64), lds, (hipStream_t)stream,
                               x, (int)N, window, (const float2 *)twiddle, fb, band_lo, band_hi, (int)n_mels, (int)hop, (int)n_frames,
                               (int)out_pitch, eps, f0, f1, t0, t1, coef_cap, out);
        else
            hipLaunchKernelGGL((melspec_wf_kernel<2048>), dim3(tiles, (unsigned)planes), dim3(WF<2048>::WAVES * 64), lds, (hipStream_t)stream,
                               x, (int)N, window, (const float2 *)twiddle, fb, band_lo, band_hi, (int)n_mels, (int)hop, (int)n_frames,
                               (int)out_pitch, eps, f0, f1, t0, t1, coef_cap, out);
        return mx_launch_status();
    }
    hipLaunchKernelGGL(melspec_kernel, dim3(tiles, (unsigned)planes), dim3(256), lds, (hipStream_t)stream,
                       x, (int)N, window, (const float2 *)twiddle, fb, band_lo, band_hi, (int)n_mels, (int)hop,
                       (int)n_frames, (int)out_pitch, eps, f0, f1, t0, t1, coef_cap, out);
    return mx_launch_status();
}
