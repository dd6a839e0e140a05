// effect_loss.hip -- K11a: the effect-model losses of mod_extraction/losses.py:14-67 and nn.L1Loss:
// per-clip sums from which L1, ESR (error-to-signal ratio) and DC loss follow.
//   part[b] = ( sum |y - y_hat|, sum (y - y_hat)^2, sum y^2, sum (y - y_hat) )   over the T samples
// One workgroup per clip, fp64 accumulation, one coalesced pass: 8 B/sample, HBM-bound.
#include "common.h"

__global__ __launch_bounds__(256) void effect_loss_kernel(const float *__restrict__ y_hat, long long hs,
                                                          const float *__restrict__ y, long long ys, int T,
                                                          float *__restrict__ part)
{
    __shared__ double sh[4][4];
    const int b = blockIdx.x;
    const float *a = y_hat + (size_t)b * hs, *t = y + (size_t)b * ys;
    double s_abs = 0, s_sq = 0, s_yy = 0, s_e = 0;
    for (int i = threadIdx.x; i < T; i += 256) {
        const float e = t[i] - a[i];
        s_abs += (double)fabsf(e);
        s_sq += (double)e * (double)e;
        s_yy += (double)t[i] * (double)t[i];
        s_e += (double)e;
    }
    s_abs = wave_sum_f64(s_abs); s_sq = wave_sum_f64(s_sq); s_yy = wave_sum_f64(s_yy); s_e = wave_sum_f64(s_e);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave][0] = s_abs; sh[wave][1] = s_sq; sh[wave][2] = s_yy; sh[wave][3] = s_e; }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(size_t)b * 4 + threadIdx.x] =
            (float)(sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// y_hat, y: B rows of T samples with row strides; part (B, 4).
MX_EXPORT int mx_effect_loss_sums(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride,
                                  int64_t B, int64_t T, float *part, void *stream)
{
    if (!y_hat || !y || !part || B <= 0 || T <= 0) return MX_ERR_ARG;
    hipLaunchKernelGGL(effect_loss_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, y_hat,
                       (long long)y_hat_stride, y, (long long)y_stride, (int)T, part);
    return mx_launch_status();
}


// d (w_l1 L1 + w_mse MSE + w_esr ESR + w_dc DC) / d y_hat, all with 'mean' reduction over the B clips (one channel):
//   L1, MSE : mean over B T samples          -> w/(B T) sign(a - t),  2 w/(B T) (a - t)
//   ESR     : mean_b sum_t (t-a)^2 / (sum_t t^2 + eps)                       (losses.py:33-38)  -> 2 w/B (a - t) / (S_tt + eps)
//   DC      : mean_b (mean_t (t-a))^2 / (mean_t t^2 + eps)                   (losses.py:61-66)  -> -2 w/(B T) mean_t(t-a) / (mean_t t^2 + eps)
// One workgroup per clip: a reduction sweep (fp64), then the write sweep (the row is L1 / L2 resident).  accumulate != 0 adds
// onto dy (e.g. the MR-STFT gradient already there).
__global__ __launch_bounds__(256) void effect_loss_grad_kernel(const float *__restrict__ y_hat, long long hs,
                                                               const float *__restrict__ y, long long ys, int B, int T,
                                                               float w_l1, float w_mse, float w_esr, float w_dc, float eps,
                                                               int accumulate, float *__restrict__ dy, long long ds)
{
    __shared__ double sh[4][2];
    const int b = blockIdx.x;
    const float *a = y_hat + (size_t)b * hs, *t = y + (size_t)b * ys;
    float *o = dy + (size_t)b * ds;
    double s_yy = 0, s_e = 0;
    if (w_esr != 0.0f || w_dc != 0.0f) {
        for (int i = threadIdx.x; i < T; i += 256) {
            s_yy += (double)t[i] * (double)t[i];
            s_e += (double)(t[i] - a[i]);
        }
        s_yy = wave_sum_f64(s_yy); s_e = wave_sum_f64(s_e);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { sh[wave][0] = s_yy; sh[wave][1] = s_e; }
        __syncthreads();
        s_yy = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        s_e = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
    }
    const double n = (double)B * (double)T;
    const float c_l1 = (float)((double)w_l1 / n), c_mse = (float)(2.0 * (double)w_mse / n);
    const float c_esr = (float)(2.0 * (double)w_esr / (double)B / (s_yy + (double)eps));
    const float c_dc = (float)(-2.0 * (double)w_dc / n * (s_e / (double)T) / (s_yy / (double)T + (double)eps));
    for (int i = threadIdx.x; i < T; i += 256) {
        const float d = a[i] - t[i];
        float g = c_l1 * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) + (c_mse + c_esr) * d + c_dc;
        if (accumulate) g += o[i];
        o[i] = g;
    }
}

MX_EXPORT int mx_effect_loss_grad(const float *y_hat, int64_t y_hat_stride, const float *y, int64_t y_stride, int64_t B,
                                  int64_t T, float w_l1, float w_mse, float w_esr, float w_dc, float eps,
                                  int32_t accumulate, float *dy, int64_t dy_stride, void *stream)
{
    if (!y_hat || !y || !dy || B <= 0 || T <= 0 || dy_stride < T) return MX_ERR_ARG;
    if (T >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(effect_loss_grad_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, y_hat,
                       (long long)y_hat_stride, y, (long long)y_stride, (int)B, (int)T, w_l1, w_mse, w_esr, w_dc, eps,
                       (int)accumulate, dy, (long long)dy_stride);
    return mx_launch_status();
}
