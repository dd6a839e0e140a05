// mrstft.hip -- K11: multi-resolution STFT loss, forward value and gradient w.r.t. the prediction
// (reference: mod_extraction/losses.py:155-156 -> auraloss==0.4.0 MultiResolutionSTFTLoss(reduction=
// "mean"); third-party, source absent from the reference tree: PARITY UNPINNED, checked against
// oracle/losses.py:MultiResolutionSTFTLoss, which restates the published auraloss defaults:
// (n_fft, hop, win) = (1024,120,600), (2048,240,1200), (512,50,240), periodic hann centred in the FFT
// frame, centre=True reflect padding, mag = sqrt(clamp(re^2 + im^2, eps)),
// loss = mean over resolutions of [ ||Y - X||_F / ||Y||_F  +  mean |log X - log Y| ]).
//
// Three passes per resolution, all streaming (12 B/sample algorithmic: x, y in, grad out):
//   A  stats : per frame, ONE complex FFT of x + i*y in LDS (radix-4 Stockham, + one radix-2 pass for
//              512/2048) -> both spectra by Hermitian separation -> per-workgroup partial sums of
//              (Ym - Xm)^2, Ym^2, |log Xm - log Ym|
//   B  grad  : same FFT, per-bin dL/dX from the global norms, inverse FFT, window, store the frame's
//              time-domain gradient to scratch (frames x n_fft)
//   C  fold  : overlap-add as a GATHER (each sample sums the frames that cover it, incl. the reflect-
//              padded positions) -> deterministic, no atomics
#include "common.h"

#define MR_MAXN 2048
#define MR_FR 8          // frames per workgroup in pass A

struct cf { float re, im; };
// LDS index of FFT element i.  Measured: padding one element per 16 (to spread the stride-4 / -16 / -64 scatter of the
// first three Stockham passes over the banks) made the loss 6 % SLOWER -- the passes are VALU-issue bound, not LDS bound
#define MR_PH(i) (i)
#define MR_LEN(N) (N)
__device__ __forceinline__ cf cmulf(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf caddf(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csubf(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }

// In-LDS Stockham FFT of length N (512, 1024 or 2048) by 256 threads; tw = exp(-2 pi i m / 2048).
// INV = true conjugates the twiddles (unnormalised inverse).  Returns the buffer holding the result.
template <int N, bool INV>
__device__ cf *fft_lds(cf *a, cf *b, const float2 *tw)      // tw: the N-point table staged in LDS (stage_twiddles)
{
    constexpr int TWS = 1;
    cf *src = a, *dst = b;
    int Ns = 1;
    // radix-4 passes while 4*Ns <= N (and N/Ns divisible by 4)
    for (; Ns * 4 <= N && ((N / Ns) % 4) == 0; Ns *= 4) {
        for (int j = threadIdx.x; j < N / 4; j += 256) {
            const int k = j & (Ns - 1);
            cf v0 = src[MR_PH(j)], v1 = src[MR_PH(j + N / 4)], v2 = src[MR_PH(j + N / 2)], v3 = src[MR_PH(j + 3 * N / 4)];
            if (Ns > 1) {
                const int step = k * (N / (Ns * 4)) * TWS;
                float2 w1 = tw[step], w2 = tw[2 * step], w3 = tw[3 * step];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v1 = cmulf(v1, {w1.x, w1.y});
                v2 = cmulf(v2, {w2.x, w2.y});
                v3 = cmulf(v3, {w3.x, w3.y});
            }
            const cf a0 = caddf(v0, v2), a1 = csubf(v0, v2), a2 = caddf(v1, v3), d = csubf(v1, v3);
            const cf a3 = INV ? cf{-d.im, d.re} : cf{d.im, -d.re};      // (+/-) i * (v1 - v3)
            const int j0 = ((j - k) << 2) + k;
            dst[MR_PH(j0)] = caddf(a0, a2);
            dst[MR_PH(j0 + Ns)] = caddf(a1, a3);
            dst[MR_PH(j0 + 2 * Ns)] = csubf(a0, a2);
            dst[MR_PH(j0 + 3 * Ns)] = csubf(a1, a3);
        }
        __syncthreads();
        cf *t = src; src = dst; dst = t;
    }
    if (Ns < N) {                                // one radix-2 pass (Ns == N/2)
        for (int j = threadIdx.x; j < N / 2; j += 256) {
            const int k = j & (Ns - 1);
            cf v0 = src[MR_PH(j)], v1 = src[MR_PH(j + N / 2)];
            float2 w = tw[k * (N / (Ns * 2)) * TWS];
            if (INV) w.y = -w.y;
            v1 = cmulf(v1, {w.x, w.y});
            const int j0 = ((j - k) << 1) + k;
            dst[MR_PH(j0)] = caddf(v0, v1);
            dst[MR_PH(j0 + Ns)] = csubf(v0, v1);
        }
        __syncthreads();
        cf *t = src; src = dst; dst = t;
    }
    return src;
}

// tw_s[m] = exp(-2 pi i m / N) from the 2048-point table in global memory: three twiddle loads per butterfly came from
// global memory (L1/L2 round trips in every pass) before the table was staged once per workgroup
template <int N>
__device__ __forceinline__ void stage_twiddles(float2 *tw_s, const float2 *__restrict__ tw)
{
    for (int m = threadIdx.x; m < N; m += 256) tw_s[m] = tw[m * (MR_MAXN / N)];
    __syncthreads();
}

__device__ __forceinline__ int reflect_index(int s, int T)
{
    if (s < 0) s = -s;
    if (s >= T) s = 2 * (T - 1) - s;
    return s;
}

// frame of x + i*y, windowed, centre/reflect padded
template <int N>
__device__ __forceinline__ void load_frame(cf *buf, const float *xb, const float *yb, const float *win, int f,
                                           int hop, int T)
{
    for (int n = threadIdx.x; n < N; n += 256) {
        const int s = reflect_index(f * hop + n - N / 2, T);
        const float w = win[n];
        buf[MR_PH(n)] = {xb[s] * w, yb[s] * w};
    }
    __syncthreads();
}

// spectra of the two real signals from Z = FFT(x + i y):  X[k] = (Z[k] + conj Z[N-k]) / 2,
// Y[k] = (Z[k] - conj Z[N-k]) / (2 i)
template <int N>
__device__ __forceinline__ void split_bins(const cf *Z, int k, cf &X, cf &Y)
{
    const cf z = Z[MR_PH(k)], zc = Z[MR_PH((N - k) & (N - 1))];
    X = {0.5f * (z.re + zc.re), 0.5f * (z.im - zc.im)};
    Y = {0.5f * (z.im + zc.im), -0.5f * (z.re - zc.re)};
}

// ---- pass A -------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void mr_stats_kernel(const float *__restrict__ x, long long xs,
                                                       const float *__restrict__ y, long long ys,
                                                       const float *__restrict__ win,
                                                       const float2 *__restrict__ tw, int T, int hop, int n_frames,
                                                       float eps, double *__restrict__ part)
{
    __shared__ cf bufA[MR_LEN(N)], bufB[MR_LEN(N)];
    __shared__ float2 tw_s[N];
    __shared__ double red[4][3];
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * xs, *yb = y + (size_t)b * ys;
    stage_twiddles<N>(tw_s, tw);
    double s_d = 0.0, s_y = 0.0, s_l = 0.0;
    for (int fl = 0; fl < MR_FR; ++fl) {
        const int f = blockIdx.x * MR_FR + fl;
        if (f >= n_frames) break;                                      // block-uniform
        load_frame<N>(bufA, xb, yb, win, f, hop, T);
        const cf *Z = fft_lds<N, false>(bufA, bufB, tw_s);
        for (int k = threadIdx.x; k <= N / 2; k += 256) {
            cf X, Y;
            split_bins<N>(Z, k, X, Y);
            const float xm = sqrtf(fmaxf(X.re * X.re + X.im * X.im, eps));
            const float ym = sqrtf(fmaxf(Y.re * Y.re + Y.im * Y.im, eps));
            const float d = ym - xm;
            s_d += (double)d * (double)d;
            s_y += (double)ym * (double)ym;
            s_l += (double)fabsf(logf(xm) - logf(ym));
        }
        __syncthreads();
    }
    s_d = wave_sum_f64(s_d); s_y = wave_sum_f64(s_y); s_l = wave_sum_f64(s_l);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s_d; red[wave][1] = s_y; red[wave][2] = s_l; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const size_t o = ((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x;
        part[o] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
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
template <int N>
__global__ __launch_bounds__(256) void mr_grad_kernel(const float *__restrict__ x, long long xs,
                                                      const float *__restrict__ y, long long ys,
                                                      const float *__restrict__ win,
                                                      const float2 *__restrict__ tw, int T, int hop, int n_frames,
                                                      float eps, const float *__restrict__ coef,
                                                      float *__restrict__ scratch)
{
    __shared__ cf bufA[MR_LEN(N)], bufB[MR_LEN(N)];
    __shared__ float2 tw_s[N];
    const int b = blockIdx.y;
    const float *xb = x + (size_t)b * xs, *yb = y + (size_t)b * ys;
    const float c_sc = coef[0], c_log = coef[1];
    stage_twiddles<N>(tw_s, tw);
    for (int fl = 0; fl < MR_FR; ++fl) {
        const int f = blockIdx.x * MR_FR + fl;
        if (f >= n_frames) break;                                      // block-uniform
        load_frame<N>(bufA, xb, yb, win, f, hop, T);
        cf *Z = fft_lds<N, false>(bufA, bufB, tw_s);
        cf *G = (Z == bufA) ? bufB : bufA;
        for (int k = threadIdx.x; k < N; k += 256) {
            cf g = {0.0f, 0.0f};
            if (k <= N / 2) {
                cf X, Y;
                split_bins<N>(Z, k, X, Y);
                const float px = X.re * X.re + X.im * X.im;
                const float xm = sqrtf(fmaxf(px, eps));
                const float ym = sqrtf(fmaxf(Y.re * Y.re + Y.im * Y.im, eps));
                if (px > eps) {                                             // clamp passes no gradient below eps
                    const float dl = logf(xm) - logf(ym);
                    const float dxm = c_sc * (xm - ym) + c_log * (dl > 0.0f ? 1.0f : (dl < 0.0f ? -1.0f : 0.0f)) / xm;
                    const float s = dxm / xm;
                    g = {s * X.re, s * X.im};                               // dL/dRe X, dL/dIm X
                }
            }
            G[MR_PH(k)] = g;
        }
        __syncthreads();
        // adjoint of the one-sided DFT: dx[n] = Re sum_{k<=N/2} G[k] e^{+2 pi i k n / N}
        cf *gt = fft_lds<N, true>(G, Z, tw_s);
        float *out = scratch + ((size_t)b * n_frames + f) * N;
        for (int n = threadIdx.x; n < N; n += 256) out[n] = gt[MR_PH(n)].re * win[n];
        __syncthreads();                                               // the next frame reuses both buffers
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
        for (int f = f_lo; f <= f_hi; ++f) acc += sb[(size_t)f * N + (p - f * hop)];
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
    const int groups = (n_frames + MR_FR - 1) / MR_FR;
    hipLaunchKernelGGL((mr_stats_kernel<N>), dim3(groups, B), dim3(256), 0, st, x, xs, y, ys, win, tw, T, hop,
                       n_frames, eps, part);
    const long long count = (long long)B * n_frames * (N / 2 + 1);
    hipLaunchKernelGGL(mr_finish_kernel, dim3(1), dim3(256), 0, st, part, groups * B, count, w_sc, w_log, res_scale,
                       terms, coef);
    if (dx) {
        hipLaunchKernelGGL((mr_grad_kernel<N>), dim3(groups, B), dim3(256), 0, st, x, xs, y, ys, win, tw, T, hop,
                           n_frames, eps, coef, scratch);
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
