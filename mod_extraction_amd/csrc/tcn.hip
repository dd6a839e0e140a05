// tcn.hip -- the 1-D temporal convolution network of mod_extraction/tcn.py:106-302 (TCNBlock / TCN) as used by
// SpectralTCN / SpectralDSTCN (models.py:72-125,218-289): per block
//     LayerNorm([C, T]) (no affine) -> Conv1d(C -> Cout, k, dilation d, stride s, padding (k / 2) d) -> PReLU(Cout)
//     -> + Conv1d(C -> Cout, 1, stride s, no bias)(block input)
// The extractors are not on the headline path (no shipped training config uses them) and cost < 1 GFLOP per clip, so
// the design favours few, general kernels over peak rate -- and spends HBM, which this part has in abundance:
//   * mx_tcn_im2col     gathers the (normalised, zero-padded, strided, dilated) taps of every output position into a
//                       K-major matrix col[(ci, k)][(clip, t')] (coalesced along t' on both sides); at 256 clips the
//                       first block's matrix is 2.4 GB -- 0.8 % of the HBM
//   * mx_sgemm_f32      one general fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation) with
//                       arbitrary row / column strides for A, B and C, a batch dimension and an optional in-kernel
//                       reduction over batches: forward (W col), weight gradient (dz col^T, batches reduced), data
//                       gradient (W^T dz), the 1x1 residual convolutions and their gradients are all calls of it
//   * mx_tcn_col2im     the transposed gather (gradient w.r.t. the normalised input)
//   * mx_tcn_act_fwd / mx_tcn_act_bwd   bias + PReLU + residual add and their gradients (+ per-row partial sums for
//                       the bias and slope gradients); mx_tcn_ln_bwd: LayerNorm backward per clip.
// Activations use the CNN's plane layout (clip, channel, 352-float rows; mx_plane_stats provides the LayerNorm
// statistics with "C = 1, H = channels").
#include "conv_common.h"

#define TG_TM 64
#define TG_TN 64
#define TG_TK 16
#define TG_LD 68            // LDS row pitch of the staged tiles (floats)

struct SgemmArgs {
    const float *a, *b;
    float *c;
    long long a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;   // element strides: A(m,k) = a[m a_rs + k a_cs], ...
    long long a_bs, b_bs, c_bs;                     // batch strides (c_bs applies per batch GROUP when reducing)
    int M, N, K;
    int n_batch, batches_per_group;                 // grid.z = ceil(n_batch / batches_per_group); a group sums its batches
    int accumulate;                                 // C += result instead of C = result
};

__global__ __launch_bounds__(256) void sgemm_f32_kernel(SgemmArgs g)
{
    __shared__ float As[TG_TK][TG_LD], Bs[TG_TK][TG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.y * TG_TM, n0 = blockIdx.x * TG_TN;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    const int c32 = lane & 31, kpar = lane >> 5;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int b_beg = blockIdx.z * g.batches_per_group, b_end = min(g.n_batch, b_beg + g.batches_per_group);
    // staging roles: thread -> (k row, 4 consecutive m / n) of the 16 x 64 tiles
    const int sk = tid >> 4, sm = (tid & 15) * 4;
    // per-thread element offsets of its four A / four B items at k = sk (out-of-range rows / columns: item 0 of the operand,
    // masked below); a k-tile later they are 16 a_cs / 16 b_rs further -- one 64-bit add per item instead of two multiplies
    long long oa[4], ob[4];
    bool ma[4], mb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + sm + j, n = n0 + sm + j;
        ma[j] = m < g.M;
        mb[j] = n < g.N;
        oa[j] = ma[j] ? (long long)m * g.a_rs + (long long)sk * g.a_cs : 0;
        ob[j] = mb[j] ? (long long)sk * g.b_rs + (long long)n * g.b_cs : 0;
    }
    const long long da = (long long)TG_TK * g.a_cs, db = (long long)TG_TK * g.b_rs;
    for (int bi = b_beg; bi < b_end; ++bi) {
        const float *A = g.a + (long long)bi * g.a_bs, *Bm = g.b + (long long)bi * g.b_bs;
        long long pa[4], pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j] = oa[j]; pb[j] = ob[j]; }
        float av[4], bv[4];
        auto fetch = [&](int k0) {                             // the tile at k0 -> registers (its loads stay in flight)
            const bool kin = k0 + sk < g.K;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = (kin && ma[j]) ? A[pa[j]] : 0.0f;
                bv[j] = (kin && mb[j]) ? Bm[pb[j]] : 0.0f;
                pa[j] += ma[j] ? da : 0;
                pb[j] += mb[j] ? db : 0;
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < g.K; k0 += TG_TK) {
            __syncthreads();                                   // the previous tile has been consumed
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                As[sk][sm + j] = av[j];
                Bs[sk][sm + j] = bv[j];
            }
            __syncthreads();
            if (k0 + TG_TK < g.K) fetch(k0 + TG_TK);           // the next tile's global loads run under this tile's matrix instructions
#pragma unroll
            for (int kk = 0; kk < TG_TK; kk += 2)
                acc = mfma32(As[kk + kpar][wm + c32], Bs[kk + kpar][wn + c32], acc);
        }
    }
    float *C = g.c + (long long)blockIdx.z * g.c_bs;
    const int n = n0 + wn + c32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + mfma_row(r, lane);
        if (m < g.M && n < g.N) {
            float *p = C + (long long)m * g.c_rs + (long long)n * g.c_cs;
            *p = g.accumulate ? *p + acc[r] : acc[r];
        }
    }
}

// C[m][n] (+)= sum_k A(m,k) B(k,n) per batch; with batches_per_group > 1 a workgroup sums that many consecutive batches
// (C then has ceil(n_batch / batches_per_group) slices, c_bs apart).
MX_EXPORT int mx_sgemm_f32(const float *a, int64_t a_rs, int64_t a_cs, int64_t a_bs, const float *b, int64_t b_rs,
                           int64_t b_cs, int64_t b_bs, float *c, int64_t c_rs, int64_t c_cs, int64_t c_bs, int64_t M,
                           int64_t N, int64_t K, int64_t n_batch, int64_t batches_per_group, int32_t accumulate,
                           void *stream)
{
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || n_batch <= 0 || batches_per_group <= 0) return MX_ERR_ARG;
    if (M >= (1ll << 30) || N >= (1ll << 30) || K >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    SgemmArgs g;
    g.a = a; g.b = b; g.c = c;
    g.a_rs = a_rs; g.a_cs = a_cs; g.b_rs = b_rs; g.b_cs = b_cs; g.c_rs = c_rs; g.c_cs = c_cs;
    g.a_bs = a_bs; g.b_bs = b_bs; g.c_bs = c_bs;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.n_batch = (int)n_batch; g.batches_per_group = (int)batches_per_group; g.accumulate = accumulate;
    const int64_t groups = (n_batch + batches_per_group - 1) / batches_per_group;
    if (groups > 65535 || (M + TG_TM - 1) / TG_TM > 65535) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sgemm_f32_kernel, dim3((unsigned)((N + TG_TN - 1) / TG_TN), (unsigned)((M + TG_TM - 1) / TG_TM), (unsigned)groups),
                       dim3(256), 0, (hipStream_t)stream, g);
    return mx_launch_status();
}

// col[(ci * ksz + k)][b * To + t'] = xhat[b][ci][t' s + (k - ksz/2) d]  (0 outside [0, T)); xhat = (x - mean_b) rstd_b
// when stats != NULL (tcn.py:169-178: LayerNorm, then the zero padding of Conv1d), else x
__global__ __launch_bounds__(256) void tcn_im2col_kernel(const float *__restrict__ x, const float *__restrict__ stats,
                                                         int C, int T, int To, int ksz, int dil, int stride, int B,
                                                         float *__restrict__ col)
{
    const int row = blockIdx.y;                       // ci * ksz + k
    const int ci = row / ksz, k = row - ci * ksz;
    const long long ncol = (long long)B * To;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < ncol; j += (long long)gridDim.x * 256) {
        const int b = (int)(j / To), t = (int)(j - (long long)b * To);
        const int src = t * stride + (k - ksz / 2) * dil;
        float v = 0.0f;
        if (src >= 0 && src < T) {
            v = x[((size_t)b * C + ci) * CV_PITCH + src];
            if (stats) v = (v - stats[2 * b]) * stats[2 * b + 1];
        }
        col[(size_t)row * ncol + j] = v;
    }
}

MX_EXPORT int mx_tcn_im2col(const float *x, const float *stats, int64_t B, int64_t C, int64_t T, int64_t To,
                            int64_t ksz, int64_t dilation, int64_t stride, float *col, void *stream)
{
    if (!x || !col || B <= 0 || C <= 0 || T <= 0 || To <= 0 || T > CV_PITCH || ksz <= 0 || dilation <= 0 || stride <= 0)
        return MX_ERR_ARG;
    if (C * ksz > 65535) return MX_ERR_UNSUPPORTED;
    const int64_t ncol = B * To;
    const unsigned gx = (unsigned)((ncol + 255) / 256 < 1024 ? (ncol + 255) / 256 : 1024);
    hipLaunchKernelGGL(tcn_im2col_kernel, dim3(gx, (unsigned)(C * ksz)), dim3(256), 0, (hipStream_t)stream, x, stats, (int)C,
                       (int)T, (int)To, (int)ksz, (int)dilation, (int)stride, (int)B, col);
    return mx_launch_status();
}

// dxhat[b][ci][t] = sum_k [ (t - (k - ksz/2) d) divisible by s, quotient t' in [0, To) ] dcol[(ci, k)][b, t']
__global__ __launch_bounds__(256) void tcn_col2im_kernel(const float *__restrict__ dcol, int C, int T, int To, int ksz,
                                                         int dil, int stride, int B, float *__restrict__ dx)
{
    const int ci = blockIdx.y, b = blockIdx.z;
    const long long ncol = (long long)B * To;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        float acc = 0.0f;
        for (int k = 0; k < ksz; ++k) {
            const int num = t - (k - ksz / 2) * dil;
            if (num >= 0 && num % stride == 0) {
                const int to = num / stride;
                if (to < To) acc += dcol[(size_t)(ci * ksz + k) * ncol + (size_t)b * To + to];
            }
        }
        dx[((size_t)b * C + ci) * CV_PITCH + t] = acc;
    }
}

MX_EXPORT int mx_tcn_col2im(const float *dcol, int64_t B, int64_t C, int64_t T, int64_t To, int64_t ksz, int64_t dilation,
                            int64_t stride, float *dx, void *stream)
{
    if (!dcol || !dx || B <= 0 || C <= 0 || T <= 0 || To <= 0 || T > CV_PITCH || ksz <= 0 || dilation <= 0 || stride <= 0)
        return MX_ERR_ARG;
    if (C > 65535 || B > 65535) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tcn_col2im_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)C, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, dcol, (int)C, (int)T, (int)To, (int)ksz, (int)dilation, (int)stride, (int)B, dx);
    return mx_launch_status();
}

// z (B, C, 352) holds the convolution WITHOUT bias on entry and z + bias (the PReLU input, kept for the backward) on
// exit; y = PReLU(z + bias) (+ res).  slope == NULL: no activation (tcn.py use_act = False); res == NULL: no residual.
__global__ __launch_bounds__(256) void tcn_act_fwd_kernel(float *__restrict__ z, const float *__restrict__ bias,
                                                          const float *__restrict__ slope, const float *__restrict__ res,
                                                          int C, int T, float *__restrict__ y)
{
    const int row = blockIdx.x, c = row % C;
    const float bc = bias ? bias[c] : 0.0f, a = slope ? slope[c] : 1.0f;
    const size_t o = (size_t)row * CV_PITCH;
    for (int t = threadIdx.x; t < CV_PITCH; t += 256) {
        float out = 0.0f;
        if (t < T) {
            const float v = z[o + t] + bc;
            z[o + t] = v;
            out = v > 0.0f ? v : a * v;                       // nn.PReLU
            if (res) out += res[o + t];                       // tcn.py:188-191
        }
        y[o + t] = out;
    }
}

MX_EXPORT int mx_tcn_act_fwd(float *z, const float *bias, const float *slope, const float *res, int64_t B, int64_t C,
                             int64_t T, float *y, void *stream)
{
    if (!z || !y || B <= 0 || C <= 0 || T <= 0 || T > CV_PITCH) return MX_ERR_ARG;
    hipLaunchKernelGGL(tcn_act_fwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, z, bias, slope, res,
                       (int)C, (int)T, y);
    return mx_launch_status();
}

// dz = dy * (zb > 0 ? 1 : slope[c]);  part (B*C, 2) = per-row sums of dz (bias gradient) and of dy * zb [zb <= 0]
// (slope gradient), summed over clips by mx_reduce_rows.  dz may alias dy.
__global__ __launch_bounds__(256) void tcn_act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ zb,
                                                          const float *__restrict__ slope, int C, int T,
                                                          float *__restrict__ dz, float *__restrict__ part)
{
    __shared__ double red[2][4];
    const int row = blockIdx.x, c = row % C;
    const float a = slope ? slope[c] : 1.0f;
    const size_t o = (size_t)row * CV_PITCH;
    double s_b = 0.0, s_a = 0.0;
    for (int t = threadIdx.x; t < CV_PITCH; t += 256) {
        float g = 0.0f;
        if (t < T) {
            const float d = dy[o + t], v = zb[o + t];
            g = v > 0.0f ? d : a * d;
            if (!(v > 0.0f) && slope) s_a += (double)d * (double)v;
            s_b += (double)g;
        }
        dz[o + t] = g;
    }
    s_b = wave_sum_f64(s_b);
    s_a = wave_sum_f64(s_a);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_b; red[1][threadIdx.x >> 6] = s_a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * row] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        part[2 * row + 1] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

MX_EXPORT int mx_tcn_act_bwd(const float *dy, const float *zb, const float *slope, int64_t B, int64_t C, int64_t T,
                             float *dz, float *part, void *stream)
{
    if (!dy || !zb || !dz || !part || B <= 0 || C <= 0 || T <= 0 || T > CV_PITCH) return MX_ERR_ARG;
    hipLaunchKernelGGL(tcn_act_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, dy, zb, slope, (int)C,
                       (int)T, dz, part);
    return mx_launch_status();
}

// LayerNorm([C, T]) backward for one clip per workgroup:  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) + add
// (add = gradient that reaches x around the normalisation -- the residual branch -- or NULL).  dx may alias dxhat.
__global__ __launch_bounds__(1024) void tcn_ln_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dxhat,
                                                          const float *__restrict__ stats, const float *__restrict__ add,
                                                          int C, int T, float *__restrict__ dx)
{
    __shared__ double red[2][16];
    __shared__ float m12[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const size_t base = (size_t)b * C * CV_PITCH;
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < C * T; i += 1024) {
        const int c = i / T, t = i - c * T;
        const size_t o = base + (size_t)c * CV_PITCH + t;
        const float g = dxhat[o], xh = (x[o] - mean) * rstd;
        s1 += (double)g;
        s2 += (double)g * (double)xh;
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int i = 0; i < 16; ++i) { a1 += red[0][i]; a2 += red[1][i]; }
        m12[0] = (float)(a1 / ((double)C * T));
        m12[1] = (float)(a2 / ((double)C * T));
    }
    __syncthreads();
    const float m1 = m12[0], m2 = m12[1];
    for (int i = tid; i < C * CV_PITCH; i += 1024) {
        const int c = i / CV_PITCH, t = i - c * CV_PITCH;
        const size_t o = base + i;
        float v = 0.0f;
        if (t < T) {
            const float xh = (x[o] - mean) * rstd;
            v = rstd * (dxhat[o] - m1 - xh * m2);
            if (add) v += add[o];
        }
        dx[o] = v;
    }
}

MX_EXPORT int mx_tcn_ln_bwd(const float *x, const float *dxhat, const float *stats, const float *add, int64_t B, int64_t C,
                            int64_t T, float *dx, void *stream)
{
    if (!x || !dxhat || !stats || !dx || B <= 0 || C <= 0 || T <= 0 || T > CV_PITCH) return MX_ERR_ARG;
    hipLaunchKernelGGL(tcn_ln_bwd_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, x, dxhat, stats, add, (int)C,
                       (int)T, dx);
    return mx_launch_status();
}
