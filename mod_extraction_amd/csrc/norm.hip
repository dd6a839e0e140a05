// norm.hip -- K5: LayerNorm([bins, frames], elementwise_affine=False) statistics and backward,
// fused with the PReLU that precedes the norm (reference: mod_extraction/models.py:186-189,
// torch.nn.LayerNorm eps=1e-5 biased variance; torch.nn.PReLU per-channel slope).
//
// The normalised tensor itself is never materialised: the conv kernels apply (x - mean) * rstd
// while staging.  These kernels are pure HBM streams (one read of the plane for the statistics;
// two reads + one write for the backward); sums are accumulated in fp64 so the result does not
// depend on the reduction order.
#include "conv_common.h"

__device__ __forceinline__ void block_sum2(double &a, double &b, double *sh)
{
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { sh[wave * 2] = a; sh[wave * 2 + 1] = b; }
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sh[i * 2]; b += sh[i * 2 + 1]; }
}

// stats[plane] = (mean, rstd) of f(x) over the H x Wv valid region, f = PReLU(slope[c]) or identity.
#ifndef PST_THREADS
#define PST_THREADS 256
#endif
__global__ __launch_bounds__(PST_THREADS) void plane_stats_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ slope, int C, int H,
                                                          int Wv, float eps, float *__restrict__ stats)
{
    __shared__ double sh[8];
    const int plane = blockIdx.x;
    const float sl = slope ? slope[plane % C] : 1.0f;
    const floatx4 *p = reinterpret_cast<const floatx4 *>(x + (size_t)plane * H * CV_PITCH);
    double s = 0.0, ss = 0.0;
    const int n4 = H * (CV_PITCH / 4);
    // branch-free (pad columns and the tail are masked with selects), four 16-byte loads in flight per thread; the
    // fp64 accumulators take the same terms in the same order as a one-vector-at-a-time loop
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * PST_THREADS) {
        floatx4 v[4];
        int col[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PST_THREADS;
            const bool ok = i < n4;
            const int ic = ok ? i : (int)threadIdx.x;
            col[u] = ok ? (ic % (CV_PITCH / 4)) * 4 : CV_PITCH;
            v[u] = p[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[u][e];
                if (slope) t = t > 0.0f ? t : sl * t;
                const bool valid = col[u] + e < Wv;
                const double td = valid ? (double)t : 0.0;
                s += td;
                ss += td * td;
            }
    }
    block_sum2(s, ss, sh);
    if (threadIdx.x == 0) {
        const double n = (double)H * (double)Wv;
        const double mean = s / n;
        double var = ss / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[plane * 2] = (float)mean;
        stats[plane * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

MX_EXPORT int mx_plane_stats(const float *x, const float *slope, int64_t B, int64_t C, int64_t H, int64_t Wv,
                             float eps, float *stats, void *stream)
{
    if (!x || !stats || B <= 0 || C <= 0 || H <= 0 || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_ARG;
    hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)(B * C)), dim3(PST_THREADS), 0, (hipStream_t)stream, x, slope,
                       (int)C, (int)H, (int)Wv, eps, stats);
    return mx_launch_status();
}

// The same statistics from the per-row partial sums a forward convolution's epilogue leaves (conv_f16.hip, stats_part
// (B, H, C, 2): {sum, sum of squares} of PReLU(out) over the valid columns of one pooled row, fp32 over <= 352 terms);
// fp64 across the rows.
// 256 threads = PSF_ROWS row groups x 64 planes (consecutive channels: 8-byte loads coalesce): every thread sums its share of
// the rows, the groups are combined through LDS in a fixed order.  (One thread per plane over all H rows ran on 64 of the
// 256 CUs with a dependent chain of H loads: 24 us per launch, five launches per step.)
#define PSF_ROWS 4
__global__ __launch_bounds__(256) void plane_stats_finish_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                                 const float *__restrict__ slope, int n_planes, int C, int H,
                                                                 int Wv, float eps, float *__restrict__ stats)
{
    __shared__ double sh[PSF_ROWS][64][2];
    const int pl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int plane = blockIdx.x * 64 + pl;
    const bool live = plane < n_planes;
    const int b = live ? plane / C : 0, c = live ? plane - b * C : 0;
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    const floatx2 *p = reinterpret_cast<const floatx2 *>(part) + (size_t)b * H * C + c;
    double s = 0.0, ss = 0.0;
    if (live)
        for (int h = rg; h < H; h += PSF_ROWS) {
            const floatx2 v = p[(size_t)h * C];
            s += (double)v[0];
            ss += (double)v[1];
        }
    sh[rg][pl][0] = s;
    sh[rg][pl][1] = ss;
    __syncthreads();
    if (rg != 0 || !live) return;
#pragma unroll
    for (int g = 1; g < PSF_ROWS; ++g) { s += sh[g][pl][0]; ss += sh[g][pl][1]; }
    // the sums are those of t - shift_c, shift_c = PReLU(bias_c) evaluated exactly as in the convolution's epilogue: the
    // variance is shift invariant, the mean gets the shift back
    const float bc = bias[c], shift = bc > 0.0f ? bc : slope[c] * bc;
    const double n = (double)H * (double)Wv;
    const double md = s / n;
    double var = ss / n - md * md;
    var = var > 0.0 ? var : 0.0;
    stats[plane * 2] = (float)((double)shift + md);
    stats[plane * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

MX_EXPORT int mx_plane_stats_finish(const float *part, const float *bias, const float *slope, int64_t B, int64_t C, int64_t H,
                                    int64_t Wv, float eps, float *stats, void *stream)
{
    if (!part || !bias || !slope || !stats || B <= 0 || C <= 0 || H <= 0 || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_ARG;
    const int n = (int)(B * C);
    hipLaunchKernelGGL(plane_stats_finish_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, part, bias,
                       slope, n, (int)C, (int)H, (int)Wv, eps, stats);
    return mx_launch_status();
}

// LayerNorm backward fused with the backward of the PReLU in front of it.
//   p      (B,C,H,352): pooled pre-activations of the previous block (input of PReLU)
//   dxhat  (B,C,H,352): gradient w.r.t. the normalised tensor (from mx_conv_block_dgrad); overwritten
//                       IN PLACE with G = dL/dp
//   stats  (B,C,2), slope (C,)
//   dslope_part (B*C,): per-plane partial of dL/dslope (summed over B by mx_reduce_rows)
// LN backward:  dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
// PReLU bwd  :  G = dx * (p > 0 ? 1 : slope);  dslope += dx * (p > 0 ? 0 : p)
#ifndef LNB_UNROLL
#define LNB_UNROLL 4
#endif
#ifndef LNB_THREADS
#define LNB_THREADS 64     // one wavefront per plane: many planes in flight per CU hide the two sweeps' latency and the reduction between them (256: +1 ms per step, 1024: +10 ms)
#endif
__global__ __launch_bounds__(LNB_THREADS) void ln_prelu_bwd_kernel(const float *__restrict__ p, float *__restrict__ dxhat,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ slope, int C, int H, int Wv,
                                                           float *__restrict__ dslope_part, float *__restrict__ gsum_part,
                                                           unsigned *__restrict__ gmax_bits, const float *__restrict__ ln_part,
                                                           const float *__restrict__ pair_scale)
{
    __shared__ double sh[32];
    const int plane = blockIdx.x;
    const float sl = slope[plane % C];
    const float mean = stats[plane * 2], rstd = stats[plane * 2 + 1];
    const floatx4 *pp = reinterpret_cast<const floatx4 *>(p + (size_t)plane * H * CV_PITCH);
    floatx4 *gp = reinterpret_cast<floatx4 *>(dxhat + (size_t)plane * H * CV_PITCH);
    const int n4 = H * (CV_PITCH / 4);
    // fp64 accumulators are fed once per 4-element vector (the 4 terms are summed in fp32): the pass is
    // otherwise limited by the fp64 add rate, not by HBM
    double s1 = 0.0, s2 = 0.0;
    if (ln_part) {
        // the producer of dxhat (mx_conv_block_dgrad_sp_f16) left {sum dxhat, sum dxhat * xhat} per (row, position
        // half): 2H pairs per plane instead of a sweep over the plane's two tensors
        typedef float floatx2 __attribute__((ext_vector_type(2)));
        const floatx2 *lp = reinterpret_cast<const floatx2 *>(ln_part) + (size_t)plane * (2 * H);
        for (int i = threadIdx.x; i < 2 * H; i += LNB_THREADS) {
            const floatx2 v = lp[i];
            s1 += (double)v[0];
            s2 += (double)v[1];
        }
    } else
    for (int i = threadIdx.x; i < n4; i += LNB_THREADS) {
        const int w0 = (i % (CV_PITCH / 4)) * 4;
        floatx4 pv = pp[i], gv = gp[i];
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (w0 + e < Wv) {
                float x = pv[e] > 0.0f ? pv[e] : sl * pv[e];
                float xh = (x - mean) * rstd;
                t1 += gv[e];
                t2 += gv[e] * xh;
            }
        }
        s1 += (double)t1;
        s2 += (double)t2;
    }
    block_sum2(s1, s2, sh);
    const double n = (double)H * (double)Wv;
    const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
    double ds = 0.0, gs = 0.0;
    float gmax = 0.0f;
    // branch-free (pad columns and the tail are masked with selects), LNB_UNROLL vectors of each tensor per trip: the
    // wave keeps 2 x LNB_UNROLL 16-byte loads in flight; same terms in the same order as one vector per trip
    constexpr int U = LNB_UNROLL;
    for (int i0 = threadIdx.x; i0 < n4; i0 += U * LNB_THREADS) {
        floatx4 pv[U], gv[U];
        int col[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * LNB_THREADS;
            const bool ok = i < n4;
            const int ic = ok ? i : (int)threadIdx.x;
            col[u] = ok ? (ic % (CV_PITCH / 4)) * 4 : CV_PITCH;          // tail: every element masked
            pv[u] = __builtin_nontemporal_load(pp + ic);        // both tensors are read exactly once
            gv[u] = __builtin_nontemporal_load(gp + ic);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            floatx4 o;
            float tds = 0.0f, tgs = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool valid = col[u] + e < Wv;
                const bool pos = pv[u][e] > 0.0f;
                const float x = pos ? pv[u][e] : sl * pv[u][e];
                const float xh = (x - mean) * rstd;
                const float dx = rstd * (gv[u][e] - m1 - xh * m2);
                const float r = valid ? (pos ? dx : sl * dx) : 0.0f;
                tds += (valid && !pos) ? dx * pv[u][e] : 0.0f;
                tgs += r;
                gmax = fmaxf(gmax, fabsf(r));
                o[e] = r;
            }
            ds += (double)tds;
            gs += (double)tgs;
            const int i = i0 + u * LNB_THREADS;
            if (pair_scale) {
                // the first block's weight gradient takes G as f16x3 pairs: written here, in place, as (hi | lo << 16) of G * S per
                // element -- the same 4 bytes -- so that its staging pass only routes and unpacks (mx_ln_prelu_bwd_pair)
                const float S = pair_scale[0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = o[e] * S;
                    const _Float16 hh = (_Float16)v;
                    const _Float16 ll = (_Float16)(v - (float)hh);
                    o[e] = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, hh) |
                                           ((unsigned)__builtin_bit_cast(unsigned short, ll) << 16));
                }
            }
            if (i < n4) __builtin_nontemporal_store(o, gp + i);
        }
    }
    block_sum2(ds, gs, sh);
    if (threadIdx.x == 0) {
        dslope_part[plane] = (float)ds;
        if (gsum_part) gsum_part[plane] = (float)gs;          // = mx_plane_sum of the G just written (bias gradient)
    }
    if (gmax_bits) {                                          // = max |G|, the input of the f16x3 gradient scale
        gmax = wave_max_f32(gmax);
        if ((threadIdx.x & 63) == 0) atomicMax(gmax_bits, __float_as_uint(gmax));
    }
}

MX_EXPORT int mx_ln_prelu_bwd(const float *p, float *dxhat_inout, const float *stats, const float *slope,
                              int64_t B, int64_t C, int64_t H, int64_t Wv, float *dslope_part, float *gsum_part,
                              uint32_t *gmax_bits, const float *ln_part, void *stream)
{
    if (!p || !dxhat_inout || !stats || !slope || !dslope_part || B <= 0 || C <= 0 || H <= 0 || Wv <= 0 ||
        Wv > CV_PITCH)
        return MX_ERR_ARG;
    hipLaunchKernelGGL(ln_prelu_bwd_kernel, dim3((unsigned)(B * C)), dim3(LNB_THREADS), 0, (hipStream_t)stream, p,
                       dxhat_inout, stats, slope, (int)C, (int)H, (int)Wv, dslope_part, gsum_part, gmax_bits, ln_part, nullptr);
    return mx_launch_status();
}

// The same pass leaving G as f16x3 PAIRS in place: element -> (hi | lo << 16) of G * scale[0] (fp16 bit patterns in one 32-bit
// word; pad columns 0).  scale: {S, 1/S} from mx_ln_bwd_finish (a bound on max |G| that exists before the pass; ln_part is
// therefore required).  Consumer: mx_conv_block1_wgrad_pair_f16.
MX_EXPORT int mx_ln_prelu_bwd_pair(const float *p, float *dxhat_inout, const float *stats, const float *slope,
                                   int64_t B, int64_t C, int64_t H, int64_t Wv, float *dslope_part, float *gsum_part,
                                   const float *ln_part, const float *scale, void *stream)
{
    if (!p || !dxhat_inout || !stats || !slope || !dslope_part || !ln_part || !scale || B <= 0 || C <= 0 || H <= 0 || Wv <= 0 ||
        Wv > CV_PITCH)
        return MX_ERR_ARG;
    hipLaunchKernelGGL(ln_prelu_bwd_kernel, dim3((unsigned)(B * C)), dim3(LNB_THREADS), 0, (hipStream_t)stream, p,
                       dxhat_inout, stats, slope, (int)C, (int)H, (int)Wv, dslope_part, gsum_part, nullptr, ln_part, scale);
    return mx_launch_status();
}

// out[c] (+)= sum over r of part[r*C + c]   (deterministic column sum of a small (R, C) matrix)
// 256 threads = 16 row lanes x 16 columns: a column is summed by 16 threads (rows r, r+16, ...) in fp64 and
// combined in a fixed order through LDS, so that the few-column cases (C = 64 bias / slope gradients over R =
// batch rows) are not one long dependent load chain per thread.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *__restrict__ part, int R, int C,
                                                          int accumulate, float *__restrict__ out)
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
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}

MX_EXPORT int mx_reduce_rows(const float *part, int64_t R, int64_t C, int32_t accumulate, float *out, void *stream)
{
    if (!part || !out || R <= 0 || C <= 0) return MX_ERR_ARG;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       part, (int)R, (int)C, (int)accumulate, out);
    return mx_launch_status();
}
