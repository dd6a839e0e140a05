// mrstft.hip -- K11: multi-resolution STFT loss, forward value and gradient w.r.t. the prediction
// (reference: mod_extraction/losses.py:155-156 -> auraloss==0.4.0 MultiResolutionSTFTLoss(reduction=
// "mean"); third-party, source absent from the reference tree: PARITY UNPINNED, checked against
// oracle/losses.py:MultiResolutionSTFTLoss, which restates the published auraloss defaults:
// (n_fft, hop, win) = (1024,120,600), (2048,240,1200), (512,50,240), periodic hann centred in the FFT
// frame, centre=True reflect padding, mag = sqrt(clamp(re^2 + im^2, eps)),
// loss = mean over resolutions of [ ||Y - X||_F / ||Y||_F  +  mean |log X - log Y| ]).
//
// Three passes per resolution (12 B/sample algorithmic: x, y in, grad out):
//   A  stats : per frame, ONE complex FFT of x + i*y (one frame per wavefront, register-staged radix-4 Stockham, see
//              below) -> both spectra by Hermitian separation -> per-workgroup partial sums of
//              (Ym - Xm)^2, Ym^2, |log Xm - log Ym|
//   B  grad  : same FFT, per-bin dL/dX from the global norms evaluated by the lane that feeds the bin into the
//              inverse FFT, window, store the frame's time-domain gradient to scratch (frames x n_fft)
//   C  fold  : overlap-add as a GATHER (each sample sums the frames that cover it, incl. the reflect-
//              padded positions) -> deterministic, no atomics
// Measured per 256 clips x 4 s (all three resolutions, value + gradient): 19.2 ms (frame spread over 256 threads, an LDS
// round trip and a __syncthreads per pass, twiddles from global memory) -> 14.9 (twiddles staged in LDS) -> 11.4 ms
// (this file).  The FFT passes are VALU-issue bound: ~1 800 vector instructions per 1024-point frame.
#include "common.h"

#define MR_MAXN 2048
#define MR_FPG 16        // frames per workgroup (passes A and B); the partial-sum workspace is sized for >= 8

struct cf { float re, im; };
// Complex multiply, every product rounded on its own (the file is compiled with -ffp-contract=off).  Measured with the two
// twiddle FMAs written by hand (2 mul + 2 fma instead of 4 mul + 2 add): the three resolutions take 9.95 ms instead of
// 10.95 ms per 256 x 4 s -- and x == y no longer gives loss == 0 and gradient == 0 exactly as the reference does
// (tests/test_gpu_mrstft.py::test_mrstft_identical_signals): the loss packs the two real signals as x + i y into ONE
// transform, and the spectra separate exactly for x == y only while the arithmetic is symmetric under the index mirror
// k -> N - k, which maps a butterfly's twiddle w to -i conj(w) and thereby SWAPS the two products of a complex multiply;
// an FMA rounds one of them and not the other, whichever way it is written.
__device__ __forceinline__ cf cmulf(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf caddf(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csubf(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }

// ---- the FFT: one frame per wavefront (two per wavefront for N = 512), passes fused in registers -----------------
// A frame of N points is held by L lanes, E = N / L complex values per lane.  Radix-4 Stockham passes p = 0.. with
// Ns = 4^p (butterfly j reads src[j + (N/4) c], writes dst[4 (j - k) + k + r Ns], k = j mod Ns) plus one radix-2
// pass for 512 / 2048.  A lane that runs butterflies a + L b of an EVEN pass holds exactly the inputs of E / 4
// butterflies of the next pass, so passes (0,1) and (2,3) run back to back in registers; 2048's radix-2 pass fuses
// with pass 4 the same way.  Three register stages, two exchanges through a wave-private padded LDS buffer, no
// workgroup barrier (first version: the frame spread over 256 threads, one LDS round trip and one __syncthreads per
// pass, twiddles from global memory: 19.2 ms per 256 x 4 s; twiddles in LDS: 14.9 ms).  The index algebra was
// checked against numpy.fft for the three sizes, both directions, before it was written down here.
template <int N> struct WF {
    static constexpr int L = (N == 512) ? 32 : 64;     // lanes per frame
    static constexpr int E = N / L;                    // complex values per lane: 16, 16, 32
    static constexpr int NB = E / 4;                   // radix-4 butterflies per lane and pass: 4, 4, 8
    static constexpr int NBQ = NB / 4;                 // 1, 1, 2
    static constexpr int LEN = N + N / 16;             // padded exchange buffer (complex values)
    static constexpr int WAVES = (N == 2048) ? 2 : 4;  // wavefronts per workgroup (static LDS <= 64 KB)
    static constexpr int FW = 64 / L;                  // frames a wavefront works on at a time
};

template <int N, bool INV, int P>
__device__ __forceinline__ void bfly4(int j, const cf &i0, const cf &i1, const cf &i2, const cf &i3, cf (&o)[4],
                                      const float2 *tw_s)
{
    constexpr int Ns = 1 << (2 * P);
    cf v0 = i0, v1 = i1, v2 = i2, v3 = i3;
    if (P > 0) {
        const int step = (j & (Ns - 1)) * (N / (Ns * 4));
        float2 w1 = tw_s[step], w2 = tw_s[2 * step], w3 = tw_s[3 * step];
        if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
        v1 = cmulf(v1, {w1.x, w1.y});
        v2 = cmulf(v2, {w2.x, w2.y});
        v3 = cmulf(v3, {w3.x, w3.y});
    }
    const cf a0 = caddf(v0, v2), a1 = csubf(v0, v2), a2 = caddf(v1, v3), d = csubf(v1, v3);
    const cf a3 = INV ? cf{-d.im, d.re} : cf{d.im, -d.re};              // (+/-) i (v1 - v3)
    o[0] = caddf(a0, a2);
    o[1] = caddf(a1, a3);
    o[2] = csubf(a0, a2);
    o[3] = csubf(a1, a3);
}
template <int P> __device__ __forceinline__ int out_pos4(int j, int r)
{
    constexpr int Ns = 1 << (2 * P);
    const int k = j & (Ns - 1);
    return ((j - k) << 2) + k + r * Ns;
}

// passes P and P + 1 on the registers of one lane: R[b][c] = value at position a + L b + (N/4) c on entry, the
// pass-(P+1) outputs go to the exchange buffer through PAD (position -> padded slot), and R is re-read in the same
// layout from the buffer
template <int N, bool INV, int P, typename PAD>
__device__ __forceinline__ void fused_pair(cf (&R)[WF<N>::NB][4], cf *buf, const float2 *tw_s, int a, PAD pad)
{
    constexpr int L = WF<N>::L, NB = WF<N>::NB, NBQ = WF<N>::NBQ;
    cf O[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) bfly4<N, INV, P>(a + L * b, R[b][0], R[b][1], R[b][2], R[b][3], O[b], tw_s);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int be = 0; be < NBQ; ++be) {
            const int jn = out_pos4<P>(a, r) + 4 * L * be;                 // butterfly of pass P + 1 held by this lane
            cf Pq[4];
            bfly4<N, INV, P + 1>(jn, O[be][r], O[NBQ + be][r], O[2 * NBQ + be][r], O[3 * NBQ + be][r], Pq, tw_s);
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) buf[pad(out_pos4<P + 1>(jn, r2))] = Pq[r2];
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[b][c] = buf[pad(a + L * b + (N / 4) * c)];
    __builtin_amdgcn_wave_barrier();
}

// Full transform of the lane-resident frame R (layout above).  Results: Z[i] = bin pos_final<N>(i, a).
template <int N> __device__ __forceinline__ int pos_final(int i, int a)
{
    if (N == 1024) return a + 64 * (i >> 2) + 256 * (i & 3);                      // i = 4 b + r
    if (N == 2048) return a + 64 * ((i >> 2) & 3) + 256 * (i & 3) + 1024 * (i >> 4);  // i = 16 h + 4 b + r
    return a + 32 * ((i >> 2) + 4 * (i & 1)) + 256 * ((i >> 1) & 1);            // 512: i = 4 b + cl + 2 h
}
template <int N, bool INV>
__device__ __forceinline__ void wave_fft(cf (&R)[WF<N>::NB][4], cf (&Z)[WF<N>::E], cf *buf, const float2 *tw_s, int a)
{
    constexpr int L = WF<N>::L, NB = WF<N>::NB;
    fused_pair<N, INV, 0>(R, buf, tw_s, a, [](int q) { return q + (q >> 4); });
    fused_pair<N, INV, 2>(R, buf, tw_s, a, [](int q) { return q + 16 * (q >> 8); });
    if (N == 1024) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            cf o[4];
            bfly4<N, INV, 4>(a + L * b, R[b][0], R[b][1], R[b][2], R[b][3], o, tw_s);
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[4 * b + r] = o[r];
        }
    } else if (N == 2048) {
        cf O[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) bfly4<N, INV, 4>(a + L * b, R[b][0], R[b][1], R[b][2], R[b][3], O[b], tw_s);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                   // radix-2, Ns = 1024: j = a + 64 b + 256 r
                float2 w = tw_s[a + 64 * b + 256 * r];
                if (INV) w.y = -w.y;
                const cf v = cmulf(O[(b + 4) % NB][r], {w.x, w.y});
                Z[4 * b + r] = caddf(O[b][r], v);
                Z[16 + 4 * b + r] = csubf(O[b][r], v);
            }
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) {                                // radix-2, Ns = 256: j = a + 32 (b + 4 cl)
                float2 w = tw_s[a + 32 * (b + 4 * cl)];
                if (INV) w.y = -w.y;
                const cf v = cmulf(R[b % NB][cl + 2], {w.x, w.y});
                Z[4 * b + cl] = caddf(R[b % NB][cl], v);
                Z[4 * b + cl + 2] = csubf(R[b % NB][cl], v);
            }
    }
}

// tw_s[m] = exp(-2 pi i m / N) from the 2048-point table in global memory, once per workgroup
template <int N>
__device__ __forceinline__ void stage_twiddles(float2 *tw_s, const float2 *__restrict__ tw)
{
    for (int m = threadIdx.x; m < N; m += WF<N>::WAVES * 64) tw_s[m] = tw[m * (MR_MAXN / N)];
    __syncthreads();
}

__device__ __forceinline__ int reflect_index(int s, int T)
{
    if (s < 0) s = -s;
    if (s >= T) s = 2 * (T - 1) - s;
    return s;
}

// frame f of x + i*y, windowed, centre / reflect padded, straight into the stage-A register layout
template <int N>
__device__ __forceinline__ void load_frame(cf (&R)[WF<N>::NB][4], const float *xb, const float *yb, const float *win,
                                           int f, int hop, int T, int a)
{
#pragma unroll
    for (int b = 0; b < WF<N>::NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = a + WF<N>::L * b + (N / 4) * c;
            const int s = reflect_index(f * hop + n - N / 2, T);
            const float w = win[n];
            R[b][c] = {xb[s] * w, yb[s] * w};
        }
}

// spectra of the two real signals from Z = FFT(x + i y):  X[k] = (Z[k] + conj Z[N-k]) / 2,
// Y[k] = (Z[k] - conj Z[N-k]) / (2 i)
template <int N>
__device__ __forceinline__ void split_bins_at(const cf *Z, int k, int km, cf &X, cf &Y)   // km = (N - k) mod N
{
    const cf z = Z[k], zc = Z[km];
    X = {0.5f * (z.re + zc.re), 0.5f * (z.im - zc.im)};
    Y = {0.5f * (z.im + zc.im), -0.5f * (z.re - zc.re)};
}
template <int N>
__device__ __forceinline__ void split_bins(const cf *Z, int k, cf &X, cf &Y)
{
    const cf z = Z[k], zc = Z[(N - k) & (N - 1)];
    X = {0.5f * (z.re + zc.re), 0.5f * (z.im - zc.im)};
    Y = {0.5f * (z.im + zc.im), -0.5f * (z.re - zc.re)};
}

// the lane's frame slot of iteration `it`: frame index within the workgroup's MR_FPG frames
template <int N> __device__ __forceinline__ int frame_slot(int it, int wave, int g)
{
    return (it * WF<N>::WAVES + wave) * WF<N>::FW + g;
}

// ---- pass A -------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WF<N>::WAVES * 64) __attribute__((amdgpu_waves_per_eu(2))) void mr_stats_kernel(const float *__restrict__ x, long long xs,
                                                                     const float *__restrict__ y, long long ys,
                                                                     const float *__restrict__ win,
                                                                     const float2 *__restrict__ tw, int T, int hop,
                                                                     int n_frames, float eps,
                                                                     double *__restrict__ part)
{
    constexpr int L = WF<N>::L, E = WF<N>::E, WAVES = WF<N>::WAVES, FW = WF<N>::FW;
    __shared__ cf xbuf[WAVES * FW][WF<N>::LEN];
    __shared__ float2 tw_s[N];
    __shared__ double red[WAVES][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / L, a = lane % L;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * xs, *yb = y + (size_t)b * ys;
    cf *buf = xbuf[wave * FW + g];
    stage_twiddles<N>(tw_s, tw);
    double s_d = 0.0, s_y = 0.0, s_l = 0.0;
    for (int it = 0; it < MR_FPG / (WAVES * FW); ++it) {
        const int f = blockIdx.x * MR_FPG + frame_slot<N>(it, wave, g);
        if (f - g >= n_frames) break;                                   // wave-uniform (frames of a wave are f-g, f-g+1)
        const bool live = f < n_frames;
        cf R[WF<N>::NB][4], Z[E];
        load_frame<N>(R, xb, yb, win, live ? f : n_frames - 1, hop, T, a);
        wave_fft<N, false>(R, Z, buf, tw_s, a);
#pragma unroll
        for (int i = 0; i < E; ++i) buf[pos_final<N>(i, a)] = Z[i];
        __builtin_amdgcn_wave_barrier();
        if (live) {
            for (int k = a; k <= N / 2; k += L) {
                cf X, Y;
                split_bins<N>(buf, k, X, Y);
                const float xm = sqrtf(fmaxf(X.re * X.re + X.im * X.im, eps));
                const float ym = sqrtf(fmaxf(Y.re * Y.re + Y.im * Y.im, eps));
                const float d = ym - xm;
                s_d += (double)d * (double)d;
                s_y += (double)ym * (double)ym;
                s_l += (double)fabsf(logf(xm) - logf(ym));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    s_d = wave_sum_f64(s_d); s_y = wave_sum_f64(s_y); s_l = wave_sum_f64(s_l);
    if (lane == 0) { red[wave][0] = s_d; red[wave][1] = s_y; red[wave][2] = s_l; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (int w = 0; w < WAVES; ++w) acc += red[w][threadIdx.x];
        part[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = acc;
    }
}

// partial sums of one resolution -> terms[0] = sc, terms[1] = log-mag; coef (2,) for pass B
__global__ __launch_bounds__(256) void mr_finish_kernel(const double *__restrict__ part, int n_part, long long count,
                                                        float w_sc, float w_log, float res_scale,
                                                        float *__restrict__ terms, float *__restrict__ coef)
{
    __shared__ double red[4][3];
    double s_d = 0.0, s_y = 0.0, s_l = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) { s_d += part[i * 3]; s_y += part[i * 3 + 1]; s_l += part[i * 3 + 2]; }
    s_d = wave_sum_f64(s_d); s_y = wave_sum_f64(s_y); s_l = wave_sum_f64(s_l);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s_d; red[wave][1] = s_y; red[wave][2] = s_l; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s_d = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    s_y = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    s_l = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    const double nd = sqrt(s_d), ny = sqrt(s_y);
    terms[0] = (float)(nd / ny);
    terms[1] = (float)(s_l / (double)count);
    // d sc / d Xm = (Xm - Ym) / (||Y-X|| * ||Y||);  d logmag / d Xm = sign(log Xm - log Ym) / (count * Xm)
    coef[0] = nd > 0.0 ? (float)((double)res_scale * w_sc / (nd * ny)) : 0.0f;
    coef[1] = (float)((double)res_scale * w_log / (double)count);
}

// ---- pass B -------------------------------------------------------------------------------------
// N = 2048 holds 32 complex values per lane and fills the 256 registers of two waves per SIMD: 6 loop-invariant
// addresses still live in scratch (18 before the mirror bins became constant offsets, see the bin pass below).  Measured
// with MR_GRAD2048_EU = 1 (512 registers, no scratch, one wave per SIMD): 10.10 instead of 9.95 ms for the three
// resolutions -- occupancy is worth more than the six scratch loads per frame, so 2 stays.
#ifndef MR_GRAD2048_EU
#define MR_GRAD2048_EU 2
#endif
template <int N>
__global__ __launch_bounds__(WF<N>::WAVES * 64) __attribute__((amdgpu_waves_per_eu(N == 2048 ? MR_GRAD2048_EU : 2))) void mr_grad_kernel(const float *__restrict__ x, long long xs,
                                                                    const float *__restrict__ y, long long ys,
                                                                    const float *__restrict__ win,
                                                                    const float2 *__restrict__ tw, int T, int hop,
                                                                    int n_frames, float eps,
                                                                    const float *__restrict__ coef,
                                                                    float *__restrict__ scratch)
{
    constexpr int L = WF<N>::L, E = WF<N>::E, NB = WF<N>::NB, WAVES = WF<N>::WAVES, FW = WF<N>::FW;
    __shared__ cf xbuf[WAVES * FW][WF<N>::LEN];
    __shared__ float2 tw_s[N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / L, a = lane % L;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * xs, *yb = y + (size_t)b * ys;
    const float c_sc = coef[0], c_log = coef[1];
    cf *buf = xbuf[wave * FW + g];
    stage_twiddles<N>(tw_s, tw);
    for (int it = 0; it < MR_FPG / (WAVES * FW); ++it) {
        const int f = blockIdx.x * MR_FPG + frame_slot<N>(it, wave, g);
        if (f - g >= n_frames) break;                                   // wave-uniform
        const bool live = f < n_frames;
        cf R[NB][4], Z[E];
        load_frame<N>(R, xb, yb, win, live ? f : n_frames - 1, hop, T, a);
        wave_fft<N, false>(R, Z, buf, tw_s, a);
#pragma unroll
        for (int i = 0; i < E; ++i) buf[pos_final<N>(i, a)] = Z[i];
        __builtin_amdgcn_wave_barrier();
        // dL/dX at the bins this lane feeds into the inverse transform (positions a + L b + (N/4) c; zero above N/2).
        // Mirror bins N - k as constant offsets from ONE register that is opaque to the optimiser in every iteration:
        // written as (N - k) & (N - 1) each address is loop invariant, gets hoisted out of the frame loop and, for
        // N = 2048 (all 256 registers taken by the transform), lives in scratch: 20 dependent scratch loads per frame.
        int am = N - a;
        asm volatile("" : "+v"(am));
#pragma unroll
        for (int bq = 0; bq < NB; ++bq)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = a + L * bq + (N / 4) * c;
                cf gk = {0.0f, 0.0f};
                if (c < 2 || k == N / 2) {                                  // c >= 2: k >= N/2
                    cf X, Y;
                    split_bins_at<N>(buf, k, (bq == 0 && c == 0) ? (am & (N - 1)) : am - (L * bq + (N / 4) * c), X, Y);
                    const float px = X.re * X.re + X.im * X.im;
                    const float xm = sqrtf(fmaxf(px, eps));
                    const float ym = sqrtf(fmaxf(Y.re * Y.re + Y.im * Y.im, eps));
                    if (px > eps) {                                         // clamp passes no gradient below eps
                        const float dl = logf(xm) - logf(ym);
                        const float dxm =
                            c_sc * (xm - ym) + c_log * (dl > 0.0f ? 1.0f : (dl < 0.0f ? -1.0f : 0.0f)) / xm;
                        const float sc = dxm / xm;
                        gk = {sc * X.re, sc * X.im};                        // dL/dRe X, dL/dIm X
                    }
                }
                R[bq][c] = gk;
            }
        __builtin_amdgcn_wave_barrier();
        // adjoint of the one-sided DFT: dx[n] = Re sum_{k<=N/2} G[k] e^{+2 pi i k n / N}
        wave_fft<N, true>(R, Z, buf, tw_s, a);
        if (live) {
            // (overlap-adding a workgroup's 16 frames in LDS and writing one span instead was measured: the fold pass
            // fell from 1.7 to 0.6 ms but this kernel lost 1.5-2 ms to the read-modify-writes and the lower occupancy)
            float *out = scratch + ((size_t)b * n_frames + f) * N;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int n = pos_final<N>(i, a);
                out[n] = Z[i].re * win[n];
            }
        }
    }
}

// ---- pass C -------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void mr_fold_kernel(const float *__restrict__ scratch, int T, int hop,
                                                      int n_frames, int accumulate, float *__restrict__ dx,
                                                      long long ds)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= T) return;
    const float *sb = scratch + (size_t)b * n_frames * N;
    // padded positions that map to sample n: the direct one and up to two reflected ones
    int pos[3], np = 0;
    pos[np++] = n + N / 2;
    if (n >= 1 && n <= N / 2) pos[np++] = N / 2 - n;
    if (n <= T - 2 && n >= T - 1 - N / 2) pos[np++] = N / 2 + 2 * (T - 1) - n;
    float acc = 0.0f;
    for (int i = 0; i < np; ++i) {
        const int p = pos[i];
        int f_hi = p / hop;
        if (f_hi > n_frames - 1) f_hi = n_frames - 1;
        int f_lo = (p - N + hop) / hop;                                 // ceil((p - N + 1) / hop)
        if (p - N + 1 <= 0) f_lo = 0;
        // same summation order as a plain loop, four loads in flight
        const float *q = sb + (size_t)f_lo * N + (p - f_lo * hop);
        const int step = N - hop;
        int f = f_lo;
        for (; f + 3 <= f_hi; f += 4, q += 4 * (size_t)step) {
            const float a0 = q[0], a1 = q[step], a2 = q[2 * (size_t)step], a3 = q[3 * (size_t)step];
            acc += a0; acc += a1; acc += a2; acc += a3;
        }
        for (; f <= f_hi; ++f, q += step) acc += q[0];
    }
    float *o = dx + (size_t)b * ds + n;
    *o = accumulate ? *o + acc : acc;
}

__global__ void mr_total_kernel(float *__restrict__ terms, int n_res, float w_sc, float w_log)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.0f;
    for (int r = 0; r < n_res; ++r) tot += w_sc * terms[2 * r] + w_log * terms[2 * r + 1];
    terms[2 * n_res] = tot / (float)n_res;
}

template <int N>
static int run_resolution(const float *x, long long xs, const float *y, long long ys, const float *win,
                          const float2 *tw, int B, int T, int hop, float eps, float w_sc, float w_log,
                          float res_scale, double *part, float *terms, float *coef, float *scratch, float *dx,
                          long long ds, int accumulate, hipStream_t st)
{
    const int n_frames = 1 + T / hop;
    const int groups = (n_frames + MR_FPG - 1) / MR_FPG;
    hipLaunchKernelGGL((mr_stats_kernel<N>), dim3(groups, B), dim3(WF<N>::WAVES * 64), 0, st, x, xs, y, ys, win, tw,
                       T, hop, n_frames, eps, part);
    const long long count = (long long)B * n_frames * (N / 2 + 1);
    hipLaunchKernelGGL(mr_finish_kernel, dim3(1), dim3(256), 0, st, part, groups * B, count, w_sc, w_log, res_scale,
                       terms, coef);
    if (dx) {
        hipLaunchKernelGGL((mr_grad_kernel<N>), dim3(groups, B), dim3(WF<N>::WAVES * 64), 0, st, x, xs, y, ys, win, tw,
                           T, hop, n_frames, eps, coef, scratch);
        hipLaunchKernelGGL((mr_fold_kernel<N>), dim3((T + 255) / 256, B), dim3(256), 0, st, scratch, T, hop, n_frames,
                           accumulate, dx, ds);
    }
    return mx_launch_status();
}

// y_hat, y: B rows of T samples (row strides); n_res resolutions with fft_sizes in {512,1024,2048} and hops
// given as HOST arrays (the only host pointers of this ABI: they select kernel instantiations and grids);
// windows (n_res, 2048): row r holds the n_fft-long window of resolution r (win_length hann, centred);
// twiddle (2048,2) = exp(-2 pi i m / 2048).  terms (2*n_res + 1): [sc_0, logmag_0, ..., total].
// dx (B rows, stride dx_stride) = d total / d y_hat, or NULL.  Workspaces: part (doubles) >= 3 * B *
// max_r ceil(frames_r / 8); coef (2,) floats; scratch (floats) >= B * max_r(frames_r * n_fft_r) (only
// when dx != NULL).
MX_EXPORT int mx_mrstft_loss(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride, int64_t B,
                             int64_t T, int32_t n_res, const int32_t *fft_sizes, const int32_t *hops,
                             const float *windows, const float *twiddle, float w_sc, float w_log, float eps,
                             double *part, float *coef, float *scratch, float *terms, float *dx,
                             int64_t dx_stride, void *stream)
{
    if (!y_hat || !y || !fft_sizes || !hops || !windows || !twiddle || !part || !coef || !terms || B <= 0 || T <= 0 ||
        n_res <= 0 || (dx && !scratch))
        return MX_ERR_ARG;
    if (B > 65535 || T >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float res_scale = 1.0f / (float)n_res;
    for (int r = 0; r < n_res; ++r) {
        const int N = fft_sizes[r], hop = hops[r];
        if (T <= N / 2 || hop <= 0) return MX_ERR_UNSUPPORTED;
        const float *win = windows + (size_t)r * MR_MAXN;
        int rc;
#define MR_RUN(NN)                                                                                                   \
    rc = run_resolution<NN>(y_hat, (long long)y_hat_stride, y, (long long)y_stride, win, (const float2 *)twiddle,    \
                            (int)B, (int)T, hop, eps, w_sc, w_log, res_scale, part, terms + 2 * r, coef, scratch, dx, \
                            (long long)dx_stride, r > 0, st)
        if (N == 512) MR_RUN(512);
        else if (N == 1024) MR_RUN(1024);
        else if (N == 2048) MR_RUN(2048);
        else return MX_ERR_UNSUPPORTED;
#undef MR_RUN
        if (rc != MX_OK) return rc;
    }
    hipLaunchKernelGGL(mr_total_kernel, dim3(1), dim3(64), 0, st, terms, (int)n_res, w_sc, w_log);
    return mx_launch_status();
}
