// head_loss.hip -- K7 + K8: the head of Spectral2DCNN and the LFO-extraction loss.
//   K7 (models.py:209-215): y6 = PReLU(p6); latent = mean over bins; out = sigmoid(Conv1d 64->L, k=1)
//   K8 (lightning.py:33-62, losses.py:70-102): sum_k w_k * loss_k with l1, fdl1 (central first
//       difference), sdl1 (second difference), mse -- forward values and d/d(y_hat) in one pass.
// All of it is O(B * 64 * 4 * 345) elementwise work: one workgroup per clip, HBM-bound, trivial.
#include "conv_common.h"

#define HL_MAXL 4       // latent_dim supported by the head kernels
#define LOSS_MAXN 2048  // frames per row supported by the loss kernel

// ---- head forward ----------------------------------------------------------------------------
// Workgroup = (64-column tile, clip); thread = (column, group of C / 4 channels): every thread reduces its channels' rows,
// writes their latents and leaves a partial logit per latent dimension in LDS; the first channel group adds the four
// partials in a fixed order.  (One workgroup per clip with a serial loop over all 64 x 4 rows per thread took 0.22 ms for
// 92 MB: a chain of dependent load batches.)
#define HF_TW 64
__global__ __launch_bounds__(256) void head_fwd_kernel(const float *__restrict__ p6,
                                                       const float *__restrict__ slope,
                                                       const float *__restrict__ wout,
                                                       const float *__restrict__ bout, int C, int Hl, int Wv,
                                                       int L, float *__restrict__ latent, float *__restrict__ out)
{
    __shared__ float part[4][HL_MAXL][HF_TW];
    const int b = blockIdx.y, wl = threadIdx.x & (HF_TW - 1), cg = threadIdx.x / HF_TW;
    const int w = blockIdx.x * HF_TW + wl;
    const int cpg = (C + 3) / 4, c0 = cg * cpg, c1 = min(C, c0 + cpg);
    const float inv_h = 1.0f / (float)Hl;
    float logit[HL_MAXL];
#pragma unroll
    for (int l = 0; l < HL_MAXL; ++l) logit[l] = 0.0f;
    if (w < Wv) {
        for (int c = c0; c < c1; ++c) {
            const float sl = slope[c];
            const float *pc = p6 + (((size_t)b * C + c) * Hl) * CV_PITCH + w;
            float acc = 0.0f;
            for (int h = 0; h < Hl; ++h) {
                float v = pc[(size_t)h * CV_PITCH];
                acc += v > 0.0f ? v : sl * v;
            }
            const float lat = acc * inv_h;
            latent[((size_t)b * C + c) * Wv + w] = lat;
#pragma unroll
            for (int l = 0; l < HL_MAXL; ++l)
                if (l < L) logit[l] = fmaf(wout[l * C + c], lat, logit[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < HL_MAXL; ++l) part[cg][l][wl] = logit[l];
    __syncthreads();
    if (cg == 0 && w < Wv) {
#pragma unroll
        for (int l = 0; l < HL_MAXL; ++l)
            if (l < L) {
                const float z = ((part[0][l][wl] + part[1][l][wl]) + part[2][l][wl]) + part[3][l][wl];
                out[((size_t)b * L + l) * Wv + w] = 1.0f / (1.0f + expf(-(z + bout[l])));
            }
    }
}

MX_EXPORT int mx_head_fwd(const float *p6, const float *slope, const float *wout, const float *bout, int64_t B,
                          int64_t C, int64_t Hl, int64_t Wv, int64_t L, float *latent, float *out, void *stream)
{
    if (!p6 || !slope || !wout || !bout || !latent || !out || B <= 0) return MX_ERR_ARG;
    if (L < 1 || L > HL_MAXL || Wv > CV_PITCH || C <= 0 || Hl <= 0 || B > 65535) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((Wv + HF_TW - 1) / HF_TW), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, p6, slope, wout, bout, (int)C, (int)Hl, (int)Wv, (int)L, latent, out);
    return mx_launch_status();
}

// ---- head backward ---------------------------------------------------------------------------
// d_out (B,L,Wv): grad w.r.t. the sigmoid output; out: saved sigmoid; d_latent (B,C,Wv) optional
// extra grad on the latent.  Produces G6 (B,C,Hl,352) = dL/dp6 and per-clip partials of the
// parameter grads: dwout_part (B, L*C), dbout_part (B, L), dslope_part (B*C).
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ p6,
                                                       const float *__restrict__ slope,
                                                       const float *__restrict__ wout,
                                                       const float *__restrict__ latent,
                                                       const float *__restrict__ out,
                                                       const float *__restrict__ d_out,
                                                       const float *__restrict__ d_latent, int C, int Hl, int Wv,
                                                       int L, float *__restrict__ G6,
                                                       float *__restrict__ dwout_part,
                                                       float *__restrict__ dbout_part,
                                                       float *__restrict__ dslope_part, unsigned *__restrict__ gmax_bits)
{
    __shared__ float dlogit[HL_MAXL][CV_PITCH];
    float gmax = 0.0f;                                          // max |G6| of this thread (the f16x3 scale of the last block)
    // grid (clip, channel slice): a workgroup takes C / gridDim.y channels, one wave per channel at a time (a single workgroup
    // per clip walked its 64 channels in 16 rounds: 0.19 ms for 92 MB read + 92 MB written)
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cps = (C + gridDim.y - 1) / gridDim.y, cbeg = blockIdx.y * cps, cend = min(C, cbeg + cps);
    for (int i = threadIdx.x; i < L * CV_PITCH; i += 256) {
        const int l = i / CV_PITCH, w = i % CV_PITCH;
        float v = 0.0f;
        if (w < Wv) {
            const float s = out[((size_t)b * L + l) * Wv + w];
            v = d_out[((size_t)b * L + l) * Wv + w] * s * (1.0f - s);
        }
        dlogit[l][w] = v;
    }
    __syncthreads();
    if (wave < L && blockIdx.y == 0) {     // bias grads: one wave per latent dim
        float s = 0.0f;
        for (int w = lane; w < Wv; w += 64) s += dlogit[wave][w];
        s = wave_sum_f32(s);
        if (lane == 0) dbout_part[(size_t)b * L + wave] = s;
    }
    const float inv_h = 1.0f / (float)Hl;
    for (int c = cbeg + wave; c < cend; c += 4) {
        const float sl = slope[c];
        float dw[HL_MAXL], ds = 0.0f;
#pragma unroll
        for (int l = 0; l < HL_MAXL; ++l) dw[l] = 0.0f;
        for (int w = lane; w < CV_PITCH; w += 64) {
            float dlat = 0.0f;
            if (w < Wv) {
                const float lat = latent[((size_t)b * C + c) * Wv + w];
                if (d_latent) dlat = d_latent[((size_t)b * C + c) * Wv + w];
#pragma unroll
                for (int l = 0; l < HL_MAXL; ++l)
                    if (l < L) {
                        dlat = fmaf(dlogit[l][w], wout[l * C + c], dlat);
                        dw[l] = fmaf(dlogit[l][w], lat, dw[l]);
                    }
            }
            const float dy = dlat * inv_h;
            for (int h = 0; h < Hl; ++h) {
                const size_t off = (((size_t)b * C + c) * Hl + h) * CV_PITCH + w;
                float g = 0.0f;
                if (w < Wv) {
                    const float pv = p6[off];
                    g = pv > 0.0f ? dy : sl * dy;
                    if (!(pv > 0.0f)) ds = fmaf(dy, pv, ds);
                }
                gmax = fmaxf(gmax, fabsf(g));
                G6[off] = g;
            }
        }
        ds = wave_sum_f32(ds);
        if (lane == 0) dslope_part[(size_t)b * C + c] = ds;
#pragma unroll
        for (int l = 0; l < HL_MAXL; ++l)
            if (l < L) {
                float t = wave_sum_f32(dw[l]);
                if (lane == 0) dwout_part[((size_t)b * L + l) * C + c] = t;
            }
    }
    if (gmax_bits) {                                            // (order of non-negative floats = order of their bits)
        gmax = wave_max_f32(gmax);
        if (lane == 0) atomicMax(gmax_bits, __float_as_uint(gmax));
    }
}

MX_EXPORT int mx_head_bwd(const float *p6, const float *slope, const float *wout, const float *latent,
                          const float *out, const float *d_out, const float *d_latent, int64_t B, int64_t C,
                          int64_t Hl, int64_t Wv, int64_t L, float *G6, float *dwout_part, float *dbout_part,
                          float *dslope_part, uint32_t *gmax_bits, void *stream)
{
    if (!p6 || !slope || !wout || !latent || !out || !d_out || !G6 || !dwout_part || !dbout_part || !dslope_part ||
        B <= 0)
        return MX_ERR_ARG;
    if (L < 1 || L > HL_MAXL || Wv > CV_PITCH || C <= 0 || Hl <= 0) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)B, 4), dim3(256), 0, (hipStream_t)stream, p6, slope, wout,
                       latent, out, d_out, d_latent, (int)C, (int)Hl, (int)Wv, (int)L, G6, dwout_part, dbout_part,
                       dslope_part, gmax_bits);
    return mx_launch_status();
}

// ---- LFO loss (forward terms + gradient) -------------------------------------------------------
// y_hat, y: (B, n).  part (B, 4): per-clip sums of |e|, |d1 e|, |d2 e|, e^2 (un-normalised).
// grad (B, n) optional: d(sum_k w_k loss_k)/d y_hat with 'mean' reductions over B*n, B*(n-2), B*(n-4).
__global__ __launch_bounds__(256) void lfo_loss_kernel(const float *__restrict__ y_hat,
                                                       const float *__restrict__ y, int n, int B, float w_l1,
                                                       float w_fd, float w_sd, float w_mse,
                                                       float *__restrict__ part, float *__restrict__ grad)
{
    __shared__ float a[LOSS_MAXN], t[LOSS_MAXN];      // y_hat row, target row
    __shared__ float d1a[LOSS_MAXN], d1t[LOSS_MAXN];  // first central differences
    __shared__ float s1[LOSS_MAXN], s2[LOSS_MAXN];    // sign(d1 error), sign(d2 error)
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) {
        a[i] = y_hat[(size_t)b * n + i];
        t[i] = y[(size_t)b * n + i];
    }
    __syncthreads();
    double sum_l1 = 0.0, sum_mse = 0.0, sum_fd = 0.0, sum_sd = 0.0;
    for (int i = tid; i < n; i += 256) {
        const float e = a[i] - t[i];
        sum_l1 += (double)fabsf(e);
        sum_mse += (double)(e * e);
        if (i < n - 2) {                                   // losses.py:81-84
            d1a[i] = (a[i + 2] - a[i]) / 2.0f;
            d1t[i] = (t[i + 2] - t[i]) / 2.0f;
            const float e1 = d1a[i] - d1t[i];
            sum_fd += (double)fabsf(e1);
            s1[i] = e1 > 0.0f ? 1.0f : (e1 < 0.0f ? -1.0f : 0.0f);
        }
    }
    __syncthreads();
    for (int i = tid; i < n - 4; i += 256) {               // losses.py:97-102
        const float e2 = (d1a[i + 2] - d1a[i]) / 2.0f - (d1t[i + 2] - d1t[i]) / 2.0f;
        sum_sd += (double)fabsf(e2);
        s2[i] = e2 > 0.0f ? 1.0f : (e2 < 0.0f ? -1.0f : 0.0f);
    }
    __syncthreads();
    {
        double v0 = sum_l1, v1 = sum_fd, v2 = sum_sd, v3 = sum_mse;
        v0 = wave_sum_f64(v0); v1 = wave_sum_f64(v1); v2 = wave_sum_f64(v2); v3 = wave_sum_f64(v3);
        __shared__ double r4[4][4];
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) { r4[wave][0] = v0; r4[wave][1] = v1; r4[wave][2] = v2; r4[wave][3] = v3; }
        __syncthreads();
        if (tid < 4) part[(size_t)b * 4 + tid] = (float)(r4[0][tid] + r4[1][tid] + r4[2][tid] + r4[3][tid]);
    }
    if (grad) {
        const float c_l1 = w_l1 / ((float)B * (float)n);
        const float c_fd = n > 2 ? w_fd / ((float)B * (float)(n - 2)) : 0.0f;
        const float c_sd = n > 4 ? w_sd / ((float)B * (float)(n - 4)) : 0.0f;
        const float c_mse = 2.0f * w_mse / ((float)B * (float)n);
        for (int k = tid; k < n; k += 256) {
            const float e = a[k] - t[k];
            float g = c_l1 * (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f)) + c_mse * e;
            // fdl1: d1[i] = (x[i+2] - x[i]) / 2
            if (k >= 2 && k - 2 < n - 2) g += c_fd * 0.5f * s1[k - 2];
            if (k < n - 2) g -= c_fd * 0.5f * s1[k];
            // sdl1: d2[i] = (x[i+4] - 2 x[i+2] + x[i]) / 4
            if (k >= 4 && k - 4 < n - 4) g += c_sd * 0.25f * s2[k - 4];
            if (k >= 2 && k - 2 < n - 4) g -= c_sd * 0.5f * s2[k - 2];
            if (k < n - 4) g += c_sd * 0.25f * s2[k];
            grad[(size_t)b * n + k] = g;
        }
    }
}

// losses (5,): l1, fdl1, sdl1, mse means and the weighted total (only weights > 0 are added,
// lightning.py:48-52).
__global__ void lfo_loss_finish_kernel(const float *__restrict__ part, int B, int n, float w_l1, float w_fd,
                                       float w_sd, float w_mse, float *__restrict__ losses)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k) s[k] += (double)part[(size_t)b * 4 + k];
    const float l1 = (float)(s[0] / ((double)B * n));
    const float fd = n > 2 ? (float)(s[1] / ((double)B * (n - 2))) : 0.0f;
    const float sd = n > 4 ? (float)(s[2] / ((double)B * (n - 4))) : 0.0f;
    const float mse = (float)(s[3] / ((double)B * n));
    losses[0] = l1; losses[1] = fd; losses[2] = sd; losses[3] = mse;
    float tot = 0.0f;
    if (w_l1 > 0.0f) tot += w_l1 * l1;
    if (w_fd > 0.0f) tot += w_fd * fd;
    if (w_sd > 0.0f) tot += w_sd * sd;
    if (w_mse > 0.0f) tot += w_mse * mse;
    losses[4] = tot;
}

MX_EXPORT int mx_lfo_loss(const float *y_hat, const float *y, int64_t B, int64_t n, float w_l1, float w_fdl1,
                          float w_sdl1, float w_mse, float *part, float *losses, float *grad, void *stream)
{
    if (!y_hat || !y || !part || !losses || B <= 0 || n <= 0) return MX_ERR_ARG;
    if (n > LOSS_MAXN) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(lfo_loss_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, y_hat, y, (int)n,
                       (int)B, w_l1, w_fdl1, w_sdl1, w_mse, part, grad);
    hipLaunchKernelGGL(lfo_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)B, (int)n, w_l1,
                       w_fdl1, w_sdl1, w_mse, losses);
    return mx_launch_status();
}
