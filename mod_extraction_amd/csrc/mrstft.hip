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
//              (Ym - Xm)^2, Ym^2, |log Xm - log Ym|; when a gradient will follow, (Re X, Im X, Ym^2) of every bin is left in
//              a workspace (12 B per bin: the transform is the expensive part, not the bytes)
//   B  grad  : per-bin dL/dX from the global norms and the parked bins -- no second forward transform --, and ONE
//              inverse FFT per PAIR of frames: the one-sided gradient spectra of frames 2p and 2p + 1 are completed to
//              Hermitian spectra G~ (G~[k] = G[k] / 2, G~[N - k] = conj G[k] / 2, DC and Nyquist real) and transformed
//              as G~_a + i G~_b, whose real / imaginary parts are the two frames' time-domain gradients; window, store to
//              scratch (frames x n_fft)
//   C  fold  : overlap-add as a GATHER (each sample sums the frames that cover it, incl. the reflect-
//              padded positions) -> deterministic, no atomics
// Transforms per frame: 1.5 (it was 3: forward in A, forward again and a zero-padded inverse in B).
// Measured per 256 clips x 4 s (all three resolutions, value + gradient): 19.2 ms (frame spread over 256 threads, an LDS
// round trip and a __syncthreads per pass, twiddles from global memory) -> 14.9 (twiddles staged in LDS) -> 11.4 ms
// (this file).  The FFT passes are VALU-issue bound: ~1 800 vector instructions per 1024-point frame.
#include "common.h"

#define MR_MAXN 2048
#define MR_FPG 16        // frames per workgroup (passes A and B); the partial-sum workspace is sized for >= 8
// parked bins of one frame: Re X[0..N/2), Im X[0..N/2), Ym^2[0..N/2), then Re X[N/2], Ym^2[N/2] (Im X[N/2] == 0), padded to 16 B
#define MR_PARK(N) (3 * (N) / 2 + 4)

struct cf { float re, im; };
// Complex multiply with the second product of each component fused (2 mul + 2 fma instead of 4 mul + 2 add; the file is
// compiled with -ffp-contract=off, so this is the only contraction): 10.95 -> 9.95 ms for the three resolutions when it
// was first measured.  It costs a property the unfused arithmetic had for free: the loss packs the two real signals as
// x + i y into ONE transform, and the spectra separate EXACTLY for x == y only while the arithmetic is symmetric under the
// index mirror k -> N - k, which maps a butterfly's twiddle w to -i conj(w) and thereby SWAPS the two products of a complex
// multiply -- an FMA rounds one of them and not the other, whichever way it is written.  The reference gives loss == 0 and
// gradient == 0 exactly for identical signals (two identical STFTs), so pass A detects frames whose windowed x and y are
// bit-identical and takes Y := X for them (tests/test_gpu_mrstft.py::test_mrstft_identical_signals, also with only some
// clips identical).
__device__ __forceinline__ cf cmulf(cf a, cf b)
{
    return {__builtin_fmaf(a.re, b.re, -(a.im * b.im)), __builtin_fmaf(a.re, b.im, a.im * b.re)};
}
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

// frame f of x + i*y, windowed, centre / reflect padded, straight into the stage-A register layout.  INTERIOR (wave-uniform):
// the frame does not touch either end of the clip, so no position needs the reflection arithmetic (two compares and selects per
// load) and the loads are one base pointer + constant offsets; all but the first and last N / (2 hop) frames of a clip.
template <int N, bool INTERIOR>
__device__ __forceinline__ void load_frame(cf (&R)[WF<N>::NB][4], const float *xb, const float *yb, const float *win,
                                           int f, int hop, int T, int a)
{
    const float *xf = xb + (f * hop - N / 2 + a), *yf = yb + (f * hop - N / 2 + a);     // dereferenced only when INTERIOR
#pragma unroll
    for (int b = 0; b < WF<N>::NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = a + WF<N>::L * b + (N / 4) * c;
            const float w = win[n];
            if (INTERIOR) {
                R[b][c] = {xf[WF<N>::L * b + (N / 4) * c] * w, yf[WF<N>::L * b + (N / 4) * c] * w};
            } else {
                const int s = reflect_index(f * hop + n - N / 2, T);
                R[b][c] = {xb[s] * w, yb[s] * w};
            }
        }
}

// spectra of the two real signals from Z = FFT(x + i y):  X[k] = (Z[k] + conj Z[N-k]) / 2,
// Y[k] = (Z[k] - conj Z[N-k]) / (2 i)
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
template <int N, bool PARK>
__global__ __launch_bounds__(WF<N>::WAVES * 64) __attribute__((amdgpu_waves_per_eu(2))) void mr_stats_kernel(const float *__restrict__ x, long long xs,
                                                                     const float *__restrict__ y, long long ys,
                                                                     const float *__restrict__ win,
                                                                     const float2 *__restrict__ tw, int T, int hop,
                                                                     int n_frames, float eps,
                                                                     double *__restrict__ part, float *__restrict__ park)
{
    constexpr int L = WF<N>::L, E = WF<N>::E, WAVES = WF<N>::WAVES, FW = WF<N>::FW;
    __shared__ cf xbuf[WAVES * FW][WF<N>::LEN];
    __shared__ float2 tw_s[N];
    __shared__ double red[WAVES][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / L, a_ = lane % L;
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * xs, *yb = y + (size_t)b * ys;
    cf *buf = xbuf[wave * FW + g];
    stage_twiddles<N>(tw_s, tw);
    double s_d = 0.0, s_y = 0.0, s_l = 0.0;
    for (int it = 0; it < MR_FPG / (WAVES * FW); ++it) {
        int a = a_;                                                      // opaque per iteration: see mr_grad_kernel
        asm volatile("" : "+v"(a));
        const int f = blockIdx.x * MR_FPG + frame_slot<N>(it, wave, g);
        if (f - g >= n_frames) break;                                   // wave-uniform (frames of a wave are f-g, f-g+1)
        const bool live = f < n_frames;
        cf R[WF<N>::NB][4], Z[E];
        {
            // frames of this wave: f - g (and f - g + 1 for the two-frame layout of 512)
            const int fa = f - g, fz = fa + FW - 1;
            const bool interior = fa * hop - N / 2 >= 0 && fz * hop + N / 2 <= T && fz < n_frames;     // wave-uniform
            if (interior) load_frame<N, true>(R, xb, yb, win, f, hop, T, a);
            else load_frame<N, false>(R, xb, yb, win, live ? f : n_frames - 1, hop, T, a);
        }
        // a frame whose windowed signals are bit-identical has identical spectra in the reference: keep that exact
        bool same_lane = true;
#pragma unroll
        for (int bq = 0; bq < WF<N>::NB; ++bq)
#pragma unroll
            for (int c = 0; c < 4; ++c) same_lane = same_lane && R[bq][c].re == R[bq][c].im;
        const unsigned long long same_mask = __ballot(same_lane);
        const bool same = L == 64 ? same_mask == ~0ull : ((same_mask >> (32 * g)) & 0xffffffffull) == 0xffffffffull;
        wave_fft<N, false>(R, Z, buf, tw_s, a);
#pragma unroll
        for (int i = 0; i < E; ++i) buf[pos_final<N>(i, a)] = Z[i];
        __builtin_amdgcn_wave_barrier();
        if (live) {
            // Bin pass.  Hardware square root / log2 (1 ulp; the library versions' range fix-ups were 40 % of this kernel's
            // instructions), log Xm - log Ym = ln2 / 2 * (log2 max(|X|^2, eps) - log2 max(|Y|^2, eps)), Ym^2 = max(|Y|^2, eps),
            // and the frame's <= 9 terms per lane summed in fp32 before they enter the fp64 accumulators.
            float *pk = PARK ? park + ((size_t)b * n_frames + f) * MR_PARK(N) : nullptr;
            float fd = 0.0f, fy = 0.0f, fl = 0.0f;
            auto bin = [&](int k, bool last) {
                cf X, Y;
                split_bins<N>(buf, k, X, Y);
                if (same) Y = X;
                const float cx = fmaxf(X.re * X.re + X.im * X.im, eps), cy = fmaxf(Y.re * Y.re + Y.im * Y.im, eps);
                const float d = __builtin_amdgcn_sqrtf(cy) - __builtin_amdgcn_sqrtf(cx);
                fd += d * d;
                fy += cy;
                fl += fabsf(__builtin_amdgcn_logf(cx) - __builtin_amdgcn_logf(cy));
                if (PARK) {
                    if (!last) { pk[k] = X.re; pk[N / 2 + k] = X.im; pk[N + k] = cy; }
                    else { pk[3 * N / 2] = X.re; pk[3 * N / 2 + 1] = cy; }
                }
            };
#pragma unroll
            for (int j = 0; j < N / 2 / L; ++j) bin(a + L * j, false);
            if (a == 0) bin(N / 2, true);
            s_d += (double)fd;
            s_y += (double)fy;
            s_l += (double)(0.5f * 0.69314718055994531f * fl);
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
// d loss / d X of one bin from the parked (Re X, Im X, Ym^2):  d sc / d Xm and d logmag / d Xm as in mr_finish_kernel, times
// d Xm / d X = X / Xm; the clamp passes no gradient below eps
__device__ __forceinline__ cf grad_bin(float xr, float xi, float cy, float eps, float c_sc, float c_log)
{
    // branch-free (a divergent branch per bin would also serialise the bins' loads behind one another); hardware square
    // root / reciprocal / log2 (1 ulp each, no range fix-ups: the library sqrtf, two logf and two divisions were 70 of
    // this function's ~90 instructions and most of the kernel); Xm and Ym by the same formula, so that X == Y gives 0
    const float px = xr * xr + xi * xi;
    const float cx = fmaxf(px, eps);
    const float xm = __builtin_amdgcn_sqrtf(cx), ym = __builtin_amdgcn_sqrtf(cy);
    const float rx = __builtin_amdgcn_rcpf(xm);
    const float dl = __builtin_amdgcn_logf(cx) - __builtin_amdgcn_logf(cy);         // sign of log Xm - log Ym
    const float sg = dl > 0.0f ? c_log : (dl < 0.0f ? -c_log : 0.0f);
    const float dxm = c_sc * (xm - ym) + sg * rx;
    const float sc = px > eps ? dxm * rx : 0.0f;
    return {sc * xr, sc * xi};                                          // dL/dRe X, dL/dIm X
}

#ifndef MR_GRAD2048_EU
#define MR_GRAD2048_EU 2
#endif
template <int N>
__global__ __launch_bounds__(WF<N>::WAVES * 64) __attribute__((amdgpu_waves_per_eu(N == 2048 ? MR_GRAD2048_EU : 2))) void mr_grad_kernel(const float *__restrict__ park,
                                                                    const float *__restrict__ win,
                                                                    const float2 *__restrict__ tw, int n_frames, float eps,
                                                                    const float *__restrict__ coef,
                                                                    float *__restrict__ scratch)
{
    constexpr int L = WF<N>::L, E = WF<N>::E, NB = WF<N>::NB, WAVES = WF<N>::WAVES, FW = WF<N>::FW, P = MR_PARK(N);
    __shared__ cf xbuf[WAVES * FW][WF<N>::LEN];
    __shared__ float2 tw_s[N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / L, a_ = lane % L;
    const int b = blockIdx.y;
    const float c_sc = coef[0], c_log = coef[1];
    cf *buf = xbuf[wave * FW + g];
    stage_twiddles<N>(tw_s, tw);
    for (int it = 0; it < MR_FPG / (2 * WAVES * FW); ++it) {
        // the lane's position is opaque to the optimiser in every iteration: otherwise each of the ~100 window / parked-bin /
        // output addresses built from it is loop invariant, gets hoisted out of the frame loop and, for N = 2048 (all 256
        // registers taken by the transform), lives in scratch memory (89 spilled registers)
        int a = a_;
        asm volatile("" : "+v"(a));
        // frames 2 p and 2 p + 1 share a transform
        const int p = blockIdx.x * (MR_FPG / 2) + (it * WAVES + wave) * FW + g;
        if (2 * (p - g) >= n_frames) break;                             // wave-uniform (pairs of a wave are p - g, p - g + 1)
        const int f0 = 2 * p, f1 = 2 * p + 1;
        const bool live0 = f0 < n_frames, live1 = f1 < n_frames;
        const float *s0 = park + ((size_t)b * n_frames + (live0 ? f0 : n_frames - 1)) * P;
        const float *s1 = park + ((size_t)b * n_frames + (live1 ? f1 : n_frames - 1)) * P;
        const float m0 = live0 ? 1.0f : 0.0f, m1 = live1 ? 1.0f : 0.0f;
        cf R[NB][4], Z[E];
        // H = G~_a + i G~_b at the positions this lane feeds into the inverse transform (a + L b + (N/4) c): the lower half
        // (c < 2) from the lane's own bins, which also leave their mirror images (conj G_a + i conj G_b) / 2 in the wave's
        // exchange buffer for the lanes that feed the upper half
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b0 = 0; b0 < NB; b0 += 4) {
                float v0[4][3], v1[4][3];                                // the loads of four bin pairs first: 24 requests in flight
#pragma unroll
                for (int bq = 0; bq < 4; ++bq)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int k = a + L * (b0 + bq) + (N / 4) * c;
                        v0[bq][q] = s0[q * (N / 2) + k];
                        v1[bq][q] = s1[q * (N / 2) + k];
                    }
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) {
                    const int k = a + L * (b0 + bq) + (N / 4) * c;
                    cf ga = grad_bin(v0[bq][0], v0[bq][1], v0[bq][2], eps, c_sc, c_log);
                    cf gb = grad_bin(v1[bq][0], v1[bq][1], v1[bq][2], eps, c_sc, c_log);
                    ga = {m0 * ga.re, m0 * ga.im};
                    gb = {m1 * gb.re, m1 * gb.im};
                    const cf h = {0.5f * (ga.re - gb.im), 0.5f * (ga.im + gb.re)};
                    R[b0 + bq][c] = (b0 + bq == 0 && c == 0 && a == 0) ? cf{ga.re, gb.re} : h;   // DC: real, not halved
                    buf[N - k] = {0.5f * (ga.re + gb.im), 0.5f * (gb.re - ga.im)};  // (k = 0 lands in the pad: never read)
                }
            }
        if (a == 0) {                                                    // Nyquist: real, not halved
            const cf ga = grad_bin(s0[3 * N / 2], 0.0f, s0[3 * N / 2 + 1], eps, c_sc, c_log);
            const cf gb = grad_bin(s1[3 * N / 2], 0.0f, s1[3 * N / 2 + 1], eps, c_sc, c_log);
            buf[N / 2] = {m0 * ga.re, m1 * gb.re};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int bq = 0; bq < NB; ++bq)
#pragma unroll
            for (int c = 2; c < 4; ++c) R[bq][c] = buf[a + L * bq + (N / 4) * c];
        __builtin_amdgcn_wave_barrier();
        // dx_a[n] + i dx_b[n] = sum_k H[k] e^{+2 pi i k n / N}: the adjoint of the one-sided DFT for both frames at once
        wave_fft<N, true>(R, Z, buf, tw_s, a);
        // (overlap-adding a workgroup's 16 frames in LDS and writing one span instead was measured: the fold pass
        // fell from 1.7 to 0.6 ms but this kernel lost 1.5-2 ms to the read-modify-writes and the lower occupancy)
        float *out0 = scratch + ((size_t)b * n_frames + f0) * N, *out1 = out0 + N;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int n = pos_final<N>(i, a);
            const float w = win[n];
            if (live0) out0[n] = Z[i].re * w;
            if (live1) out1[n] = Z[i].im * w;
        }
    }
}

// Pass B for 512 / 1024 with the overlap-add started in LDS.  A wavefront takes MR_SPAN_FRAMES CONSECUTIVE frames and adds
// their windowed gradients into a wave-private span of (MR_SPAN_FRAMES - 1) hop + N samples (read - add - write in frame order:
// no barrier, nothing shared between waves, deterministic); the span is what reaches HBM -- 1 864 instead of 8 192 floats per
// eight 1024-point frames -- and the fold pass gathers one or two span values per sample instead of 8.5 frame values (fold
// 0.35 -> 0.19 ms, this pass 0.78 -> 0.72 ms for 1024).  Two things measured on the way: the LDS fp32 atomic (ds_add_f32) in
// place of the plain read - add - write made this pass 3x slower (2.4 ms); requesting the next pair's parked bins before the
// current pair's transform (52 more registers) changed nothing (7.27 against 7.18 ms for the three resolutions).
// 2048 keeps whole frames (mr_grad_kernel): its span would leave one wavefront per SIMD.
#define MR_SPAN_FRAMES 8
#define MR_SPAN_LEN(N, hop) ((MR_SPAN_FRAMES - 1) * (hop) + (N))
#define MR_SPAN_MAX_HOP 128      // auraloss: 50 and 120; larger hops take the frame version
template <int N>
__global__ __launch_bounds__(WF<N>::WAVES * 64) __attribute__((amdgpu_waves_per_eu(2))) void mr_grad_span_kernel(
    const float *__restrict__ park, const float *__restrict__ win, const float2 *__restrict__ tw, int n_frames, int hop, float eps,
    const float *__restrict__ coef, float *__restrict__ scratch)
{
    constexpr int L = WF<N>::L, E = WF<N>::E, NB = WF<N>::NB, WAVES = WF<N>::WAVES, FW = WF<N>::FW, P = MR_PARK(N);
    constexpr int ITERS = MR_SPAN_FRAMES / (2 * FW);
    static_assert(NB == 4, "512 / 1024 only");
    __shared__ cf xbuf[WAVES * FW][WF<N>::LEN];
    __shared__ float2 tw_s[N];
    extern __shared__ float span_s[];                                   // WAVES x S floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / L, a_ = lane % L;
    const int b = blockIdx.y;
    const float c_sc = coef[0], c_log = coef[1];
    cf *buf = xbuf[wave * FW + g];
    const int S = MR_SPAN_LEN(N, hop);
    float *span = span_s + wave * S;
    stage_twiddles<N>(tw_s, tw);
    const int fs = (blockIdx.x * WAVES + wave) * MR_SPAN_FRAMES;       // the wave's frames: fs .. fs + MR_SPAN_FRAMES - 1
    if (fs >= n_frames) return;                                         // (behind the kernel's only workgroup barrier)
    for (int j = lane; j < S; j += 64) span[j] = 0.0f;

    float pk[2][2][NB][3], ny[2][2];                                    // parked bins of a frame pair: [frame][c][bq][Re X, Im X, Ym^2]
    auto load_pair = [&](int it, int a) {
        const int p = fs / 2 + it * FW + g, f0 = 2 * p, f1 = 2 * p + 1;
        const float *s0 = park + ((size_t)b * n_frames + (f0 < n_frames ? f0 : n_frames - 1)) * P;
        const float *s1 = park + ((size_t)b * n_frames + (f1 < n_frames ? f1 : n_frames - 1)) * P;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int bq = 0; bq < NB; ++bq)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int k = a + L * bq + (N / 4) * c;
                    pk[0][c][bq][q] = s0[q * (N / 2) + k];
                    pk[1][c][bq][q] = s1[q * (N / 2) + k];
                }
        ny[0][0] = s0[3 * N / 2]; ny[0][1] = s0[3 * N / 2 + 1];
        ny[1][0] = s1[3 * N / 2]; ny[1][1] = s1[3 * N / 2 + 1];
    };
    for (int it = 0; it < ITERS; ++it) {
        int a = a_;                                                      // opaque per iteration (see mr_grad_kernel)
        asm volatile("" : "+v"(a));
        load_pair(it, a);
        const int p = fs / 2 + it * FW + g;
        if (2 * (p - g) >= n_frames) break;                             // wave-uniform
        const int f0 = 2 * p, f1 = 2 * p + 1;
        const bool live0 = f0 < n_frames, live1 = f1 < n_frames;
        const float m0 = live0 ? 1.0f : 0.0f, m1 = live1 ? 1.0f : 0.0f;
        cf R[NB][4], Z[E];
        // H = G~_a + i G~_b exactly as in mr_grad_kernel
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int bq = 0; bq < NB; ++bq) {
                const int k = a + L * bq + (N / 4) * c;
                cf ga = grad_bin(pk[0][c][bq][0], pk[0][c][bq][1], pk[0][c][bq][2], eps, c_sc, c_log);
                cf gb = grad_bin(pk[1][c][bq][0], pk[1][c][bq][1], pk[1][c][bq][2], eps, c_sc, c_log);
                ga = {m0 * ga.re, m0 * ga.im};
                gb = {m1 * gb.re, m1 * gb.im};
                const cf h = {0.5f * (ga.re - gb.im), 0.5f * (ga.im + gb.re)};
                R[bq][c] = (bq == 0 && c == 0 && a == 0) ? cf{ga.re, gb.re} : h;
                buf[N - k] = {0.5f * (ga.re + gb.im), 0.5f * (gb.re - ga.im)};
            }
        if (a == 0) {
            const cf ga = grad_bin(ny[0][0], 0.0f, ny[0][1], eps, c_sc, c_log);
            const cf gb = grad_bin(ny[1][0], 0.0f, ny[1][1], eps, c_sc, c_log);
            buf[N / 2] = {m0 * ga.re, m1 * gb.re};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int bq = 0; bq < NB; ++bq)
#pragma unroll
            for (int c = 2; c < 4; ++c) R[bq][c] = buf[a + L * bq + (N / 4) * c];
        __builtin_amdgcn_wave_barrier();
        wave_fft<N, true>(R, Z, buf, tw_s, a);
        // frame order inside the wave: (half 0: f0, f1), then (half 1: f0, f1) -- two lanes of one ds_add never meet in one
        // slot, and every slot receives its frames in ascending order
        // (plain read - add - write, 16 values at a time: LDS operations of a wave execute in order; the LDS fp32 atomic
        // ds_add_f32 is an order of magnitude slower than the three plain instructions)
        float *sp0 = span + (f0 - fs) * hop;
#pragma unroll
        for (int h = 0; h < FW; ++h) {
            if (g == h) {
#pragma unroll
                for (int fr = 0; fr < 2; ++fr) {
                    float *sp = sp0 + fr * hop;
                    const float mm = fr ? m1 : m0;
                    float old[E];
#pragma unroll
                    for (int i = 0; i < E; ++i) old[i] = sp[pos_final<N>(i, a)];
#pragma unroll
                    for (int i = 0; i < E; ++i) {
                        const int n = pos_final<N>(i, a);
                        sp[n] = old[i] + mm * ((fr ? Z[i].im : Z[i].re) * win[n]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // spans that hold a frame: ceil(n_frames / MR_SPAN_FRAMES) per clip -- with T > N / 2 (the reflect padding's condition) they
    // never need more room than the n_frames x N floats of the whole-frame layout, behind which the parked bins start
    const int n_spans = (n_frames + MR_SPAN_FRAMES - 1) / MR_SPAN_FRAMES;
    float *out = scratch + ((size_t)b * n_spans + (blockIdx.x * WAVES + wave)) * S;
    for (int j = lane; j < S; j += 64) out[j] = span[j];
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

// the same gather over the wave spans of mr_grad_span_kernel: span s holds frames [s F, s F + F) of its clip added up at padded
// positions s F hop + j, j < (F - 1) hop + N, F = MR_SPAN_FRAMES
template <int N>
__global__ __launch_bounds__(256) void mr_fold_span_kernel(const float *__restrict__ scratch, int T, int hop,
                                                           int n_frames, int n_spans, int accumulate,
                                                           float *__restrict__ dx, long long ds)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= T) return;
    const int S = MR_SPAN_LEN(N, hop), FH = MR_SPAN_FRAMES * hop;
    const float *sb = scratch + (size_t)b * n_spans * S;
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
        for (int sp = f_lo / MR_SPAN_FRAMES; sp <= f_hi / MR_SPAN_FRAMES; ++sp) acc += sb[(size_t)sp * S + (p - sp * FH)];
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
    // scratch = [ time-domain gradient frames | parked bins ]
    float *park = scratch ? scratch + (size_t)B * n_frames * N : nullptr;
    if (dx)
        hipLaunchKernelGGL((mr_stats_kernel<N, true>), dim3(groups, B), dim3(WF<N>::WAVES * 64), 0, st, x, xs, y, ys, win, tw,
                           T, hop, n_frames, eps, part, park);
    else
        hipLaunchKernelGGL((mr_stats_kernel<N, false>), dim3(groups, B), dim3(WF<N>::WAVES * 64), 0, st, x, xs, y, ys, win,
                           tw, T, hop, n_frames, eps, part, (float *)nullptr);
    const long long count = (long long)B * n_frames * (N / 2 + 1);
    hipLaunchKernelGGL(mr_finish_kernel, dim3(1), dim3(256), 0, st, part, groups * B, count, w_sc, w_log, res_scale,
                       terms, coef);
    if (dx) {
        bool spans = false;
        if constexpr (N != 2048) {
            if (hop <= MR_SPAN_MAX_HOP) {
                spans = true;
                const int ggroups = (n_frames + MR_SPAN_FRAMES * WF<N>::WAVES - 1) / (MR_SPAN_FRAMES * WF<N>::WAVES);
                const size_t span_bytes = (size_t)WF<N>::WAVES * MR_SPAN_LEN(N, hop) * sizeof(float);
                static bool attr_set[64];                               // static + dynamic LDS exceeds 64 KB for 1024
                int dev = 0;
                (void)hipGetDevice(&dev);
                if (dev < 0 || dev >= 64 || !attr_set[dev]) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mr_grad_span_kernel<N>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)(WF<N>::WAVES * MR_SPAN_LEN(N, MR_SPAN_MAX_HOP) * sizeof(float)));
                    if (dev >= 0 && dev < 64) attr_set[dev] = true;
                }
                hipLaunchKernelGGL((mr_grad_span_kernel<N>), dim3(ggroups, B), dim3(WF<N>::WAVES * 64), span_bytes, st, park, win,
                                   tw, n_frames, hop, eps, coef, scratch);
                hipLaunchKernelGGL((mr_fold_span_kernel<N>), dim3((T + 255) / 256, B), dim3(256), 0, st, scratch, T, hop,
                                   n_frames, (n_frames + MR_SPAN_FRAMES - 1) / MR_SPAN_FRAMES, accumulate, dx, ds);
            }
        }
        if (!spans) {
            hipLaunchKernelGGL((mr_grad_kernel<N>), dim3(groups, B), dim3(WF<N>::WAVES * 64), 0, st, park, win, tw, n_frames, eps,
                               coef, scratch);
            hipLaunchKernelGGL((mr_fold_kernel<N>), dim3((T + 255) / 256, B), dim3(256), 0, st, scratch, T, hop, n_frames,
                               accumulate, dx, ds);
        }
    }
    return mx_launch_status();
}

// y_hat, y: B rows of T samples (row strides); n_res resolutions with fft_sizes in {512,1024,2048} and hops
// given as HOST arrays (the only host pointers of this ABI: they select kernel instantiations and grids);
// windows (n_res, 2048): row r holds the n_fft-long window of resolution r (win_length hann, centred);
// twiddle (2048,2) = exp(-2 pi i m / 2048).  terms (2*n_res + 1): [sc_0, logmag_0, ..., total].
// dx (B rows, stride dx_stride) = d total / d y_hat, or NULL.  Workspaces: part (doubles) >= 3 * B *
// max_r ceil(frames_r / 8); coef (2,) floats; scratch (floats) >= B * max_r(frames_r * (5 * n_fft_r / 2 + 4)) (only
// when dx != NULL: the frames' time-domain gradients and the parked bins).
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
