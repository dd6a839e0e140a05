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
