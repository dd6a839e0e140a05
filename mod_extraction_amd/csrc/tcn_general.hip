// tcn_general.hip -- the TCN variants of mod_extraction/tcn.py that SpectralTCN / SpectralDSTCN do not use: explicit padding
// with the causal / centre crop of the residual branch (tcn.py:14-29,188-191), the cached streaming convolution (tcn.py:31-79)
// and FiLM conditioning with or without its affine-free BatchNorm1d (tcn.py:82-103).  Dense (B, C, T) fp32 tensors of any
// length; the convolutions are mx_im2col2d (one bin row) + mx_sgemm_f32 with a strided operand for the stride, LayerNorm is
// mx_rowln_fwd / _bwd over a clip's (C, T) block; what is left are the per-channel pieces below.  Deterministic (fixed-order
// fp64 reductions, no atomics).
//   * mx_chan_stats        mean and biased variance of every channel over (clips, frames)  -- BatchNorm1d in training mode;
//   * mx_chan_norm_fwd/bwd xhat = (z - mean_c) rstd_c and its backward with batch statistics (train) or constants (eval);
//   * mx_film_fwd / _bwd   a = xhat gain[b][c] + shift[b][c] (the adaptor's output, (B, 2 C)), gradients of both;
//   * mx_prelu_res_fwd/bwd y = PReLU_c(a) + res and da, the slope-gradient partial of every (clip, channel) row.
#include "common.h"

// one workgroup per channel: stats[c] = {mean, biased variance} over b, t (two passes, fp64)
__global__ __launch_bounds__(256) void chan_stats_kernel(const float *__restrict__ z, int B, int C, int T, float *__restrict__ stats)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const long long n = (long long)B * T;
    double s = 0.0;
    for (long long e = threadIdx.x; e < n; e += 256) s += (double)z[((e / T) * C + c) * T + e % T];
    const double mean = block256_sum_f64(s, red) / (double)n;
    double q = 0.0;
    for (long long e = threadIdx.x; e < n; e += 256) {
        const double d = (double)z[((e / T) * C + c) * T + e % T] - mean;
        q += d * d;
    }
    q = block256_sum_f64(q, red) / (double)n;
    if (threadIdx.x == 0) {
        stats[2 * c] = (float)mean;
        stats[2 * c + 1] = (float)q;
    }
}

// xhat = (z - norm[c][0]) norm[c][1]
__global__ __launch_bounds__(256) void chan_norm_fwd_kernel(const float *__restrict__ z, const float *__restrict__ norm, int C, int T,
                                                            float *__restrict__ xhat, size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)((e / T) % C);
    xhat[e] = (z[e] - norm[2 * c]) * norm[2 * c + 1];
}

// one workgroup per channel.  train: dz = rstd (g - mean(g) - xhat mean(g xhat)) over (b, t);  eval: dz = rstd g
__global__ __launch_bounds__(256) void chan_norm_bwd_kernel(const float *__restrict__ g, const float *__restrict__ xhat,
                                                            const float *__restrict__ norm, int B, int C, int T, int train,
                                                            float *__restrict__ dz)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const long long n = (long long)B * T;
    const float rstd = norm[2 * c + 1];
    float m1 = 0.0f, m2 = 0.0f;
    if (train) {
        double s1 = 0.0, s2 = 0.0;
        for (long long e = threadIdx.x; e < n; e += 256) {
            const size_t i = ((e / T) * C + c) * T + e % T;
            s1 += (double)g[i];
            s2 += (double)g[i] * (double)xhat[i];
        }
        m1 = (float)(block256_sum_f64(s1, red) / (double)n);
        m2 = (float)(block256_sum_f64(s2, red) / (double)n);
    }
    for (long long e = threadIdx.x; e < n; e += 256) {
        const size_t i = ((e / T) * C + c) * T + e % T;
        dz[i] = rstd * (g[i] - m1 - xhat[i] * m2);
    }
}

// a = xhat gain + shift; gb (B, 2 C) = [gain | shift] (torch.chunk of the adaptor's output, tcn.py:96-97)
__global__ __launch_bounds__(256) void film_fwd_kernel(const float *__restrict__ xhat, const float *__restrict__ gb, int C, int T,
                                                       float *__restrict__ a, size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t row = e / T;                                   // b C + c
    const size_t b = row / C, c = row % C;
    a[e] = xhat[e] * gb[b * 2 * C + c] + gb[b * 2 * C + C + c];
}

// one workgroup per (b, c) row: dxhat = da gain; dgb[b][c] = sum da xhat, dgb[b][C + c] = sum da
__global__ __launch_bounds__(256) void film_bwd_kernel(const float *__restrict__ da, const float *__restrict__ xhat,
                                                       const float *__restrict__ gb, int C, int T, float *__restrict__ dxhat,
                                                       float *__restrict__ dgb)
{
    __shared__ double red[4];
    const size_t row = blockIdx.x, b = row / C, c = row % C;
    const float gain = gb[b * 2 * C + c];
    double sg = 0.0, sb = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const size_t i = row * T + t;
        const float d = da[i];
        sg += (double)d * (double)xhat[i];
        sb += (double)d;
        dxhat[i] = d * gain;
    }
    sg = block256_sum_f64(sg, red);
    sb = block256_sum_f64(sb, red);
    if (threadIdx.x == 0) {
        dgb[b * 2 * C + c] = (float)sg;
        dgb[b * 2 * C + C + c] = (float)sb;
    }
}

// y = (slope ? PReLU(a; slope[c]) : a) + (res ? res : 0)
__global__ __launch_bounds__(256) void prelu_res_fwd_kernel(const float *__restrict__ a, const float *__restrict__ slope,
                                                            const float *__restrict__ res, int C, int T, float *__restrict__ y,
                                                            size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float v = a[e];
    if (slope) v = v > 0.0f ? v : slope[(e / T) % C] * v;
    if (res) v += res[e];
    y[e] = v;
}

// one workgroup per (b, c) row: da = dy PReLU'(a), part[row] = sum of dy a where a <= 0 (slope gradient partial)
__global__ __launch_bounds__(256) void prelu_res_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ a,
                                                            const float *__restrict__ slope, int C, int T, float *__restrict__ da,
                                                            float *__restrict__ part)
{
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    const float s = slope[row % C];
    double acc = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const size_t i = row * T + t;
        const float d = dy[i], v = a[i];
        if (!(v > 0.0f)) acc += (double)d * (double)v;
        da[i] = v > 0.0f ? d : s * d;
    }
    acc = block256_sum_f64(acc, red);
    if (threadIdx.x == 0) part[row] = (float)acc;
}

static int tg_dims(int64_t B, int64_t C, int64_t T)
{
    if (B <= 0 || C <= 0 || T <= 0) return MX_ERR_ARG;
    if (B * C > 0x7fffffffll || T > (1ll << 30) || (B * C * T + 255) / 256 > 0x7fffffffll) return MX_ERR_UNSUPPORTED;
    return MX_OK;
}

MX_EXPORT int mx_chan_stats(const float *z, int64_t B, int64_t C, int64_t T, float *stats, void *stream)
{
    if (!z || !stats) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, z, (int)B, (int)C, (int)T, stats);
    return mx_launch_status();
}

MX_EXPORT int mx_chan_norm_fwd(const float *z, const float *norm, int64_t B, int64_t C, int64_t T, float *xhat, void *stream)
{
    if (!z || !norm || !xhat) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    const size_t total = (size_t)B * C * T;
    hipLaunchKernelGGL(chan_norm_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, norm, (int)C,
                       (int)T, xhat, total);
    return mx_launch_status();
}

MX_EXPORT int mx_chan_norm_bwd(const float *g, const float *xhat, const float *norm, int64_t B, int64_t C, int64_t T, int32_t train,
                               float *dz, void *stream)
{
    if (!g || !xhat || !norm || !dz) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    hipLaunchKernelGGL(chan_norm_bwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, g, xhat, norm, (int)B, (int)C, (int)T,
                       (int)train, dz);
    return mx_launch_status();
}

MX_EXPORT int mx_film_fwd(const float *xhat, const float *gb, int64_t B, int64_t C, int64_t T, float *a, void *stream)
{
    if (!xhat || !gb || !a) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    const size_t total = (size_t)B * C * T;
    hipLaunchKernelGGL(film_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xhat, gb, (int)C, (int)T, a,
                       total);
    return mx_launch_status();
}

MX_EXPORT int mx_film_bwd(const float *da, const float *xhat, const float *gb, int64_t B, int64_t C, int64_t T, float *dxhat, float *dgb,
                          void *stream)
{
    if (!da || !xhat || !gb || !dxhat || !dgb) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    hipLaunchKernelGGL(film_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, da, xhat, gb, (int)C, (int)T, dxhat, dgb);
    return mx_launch_status();
}

MX_EXPORT int mx_prelu_res_fwd(const float *a, const float *slope, const float *res, int64_t B, int64_t C, int64_t T, float *y,
                               void *stream)
{
    if (!a || !y) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    const size_t total = (size_t)B * C * T;
    hipLaunchKernelGGL(prelu_res_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, slope, res, (int)C,
                       (int)T, y, total);
    return mx_launch_status();
}

MX_EXPORT int mx_prelu_res_bwd(const float *dy, const float *a, const float *slope, int64_t B, int64_t C, int64_t T, float *da, float *part,
                               void *stream)
{
    if (!dy || !a || !slope || !da || !part) return MX_ERR_ARG;
    const int rc = tg_dims(B, C, T);
    if (rc != MX_OK) return rc;
    hipLaunchKernelGGL(prelu_res_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, dy, a, slope, (int)C, (int)T, da,
                       part);
    return mx_launch_status();
}
