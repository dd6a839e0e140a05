// optim.hip -- K12: AdamW step on one flat fp32 parameter buffer
// (reference: torch.optim.AdamW as configured by configs/opt/adam_w.yml: lr 1e-4, betas (0.8, 0.99),
// eps 1e-8, weight_decay 0.01; single-tensor update order of torch/optim/adamw.py).
// Also the per-plane sum used for the bias gradients and a fused grad-scale used by the DDP loop.
// Pure streaming kernels: 16 B read + 12 B written per parameter.
#include "conv_common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, long long n,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bias_c1, float bias_c2_sqrt, float grad_scale)
{
    const float step_size = lr / bias_c1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.0f - lr * wd);                    // param.mul_(1 - lr * weight_decay)
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        pi -= step_size * (mi / denom);                        // param.addcdiv_(exp_avg, denom, value=-step_size)
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// step: 1-based step count after increment.  grad_scale multiplies the gradient first (1/world_size
// after a sum all-reduce).
MX_EXPORT int mx_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                            float grad_scale, void *stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return MX_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return mx_launch_status();
}

// Column sums of a small (R, C) matrix of per-clip gradient rows AND the AdamW step on them in one launch: the truncated-BPTT loop
// of the effect model takes 83 optimizer steps per batch on 17 473 parameters (lightning.py:355-384), where mx_reduce_rows and
// mx_adamw_step were two ~8 us launches each.  Same arithmetic in the same order as the two entry points (fp64 column sums over
// rows r, r + 16, ... combined in a fixed order; then adamw_kernel's update of element c): bit-identical results.
__global__ __launch_bounds__(256) void reduce_rows_adamw_kernel(const float *__restrict__ part, int R, int C, float *__restrict__ p,
                                                                float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                                float lr, float beta1, float beta2, float eps, float wd,
                                                                float bias_c1, float bias_c2_sqrt, float grad_scale)
{
    __shared__ double sh[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C)
        for (int r = rl; r < R; r += 16) s += (double)part[(size_t)r * C + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        const float gsum = (float)t;
        g[c] = gsum;                                            // the flat gradient stays observable (logging, tests)
        const float step_size = lr / bias_c1;
        const float gi = gsum * grad_scale;
        float pi = p[c] * (1.0f - lr * wd);
        const float mi = m[c] + (gi - m[c]) * (1.0f - beta1);
        const float vi = v[c] * beta2 + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[c] = pi; m[c] = mi; v[c] = vi;
    }
}

// part (R, n): one gradient row per clip; grad (n,) receives the column sums; then the AdamW step of mx_adamw_step on them.
MX_EXPORT int mx_reduce_rows_adamw_step(const float *part, int64_t R, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                        int64_t n, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        float grad_scale, void *stream)
{
    if (!part || !param || !grad || !exp_avg || !exp_avg_sq || R <= 0 || n <= 0 || step <= 0) return MX_ERR_ARG;
    if (n >= (1ll << 31) || R >= (1ll << 31)) return MX_ERR_UNSUPPORTED;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(reduce_rows_adamw_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, part, (int)R,
                       (int)n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                       grad_scale);
    return mx_launch_status();
}

// out[plane] = sum of the H x Wv valid region of plane (B*C planes of (H, 352)); fp64 accumulate.
__global__ __launch_bounds__(256) void plane_sum_kernel(const float *__restrict__ x, int H, int Wv,
                                                        float *__restrict__ out)
{
    __shared__ double sh[4];
    const floatx4 *p = reinterpret_cast<const floatx4 *>(x + (size_t)blockIdx.x * H * CV_PITCH);
    double s = 0.0;
    const int n4 = H * (CV_PITCH / 4);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int w0 = (i % (CV_PITCH / 4)) * 4;
        floatx4 v = p[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (w0 + e < Wv) s += (double)v[e];
    }
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

MX_EXPORT int mx_plane_sum(const float *x, int64_t planes, int64_t H, int64_t Wv, float *out, void *stream)
{
    if (!x || !out || planes <= 0 || H <= 0 || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_ARG;
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, (int)H, (int)Wv,
                       out);
    return mx_launch_status();
}
