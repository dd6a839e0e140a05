// cnn_generic.hip -- Spectral2DCNN (mod_extraction/models.py:127-215) OUTSIDE the family the f16x3 kernels are built for:
// any kernel size, channel list, bin / frame dilations, MaxPool2d((p, 1)) with any p, with or without LayerNorm, any number
// of input channels, frames and latent dimensions -- the class's own defaults (pool (3, 1), five blocks, temp dilations
// 1..16) are such a configuration.  No shipped YAML uses one, so the design is the TCN extractors' (tcn.hip): few, general
// kernels on dense NCHW fp32 tensors, exact fp32 arithmetic, HBM spent freely:
//   * mx_im2col2d / mx_col2im2d   gather the dilated, zero-padded ("same": total = d (k - 1), before = total / 2, the rest
//                                 after -- aten's rule, asymmetric for even kernels) taps of a chunk of clips into the
//                                 K-major matrix col[(ci, i, j)][(clip, h, w)] and the transposed gather;
//   * mx_sgemm_f32 (tcn.hip)      the three convolution products (forward, weight gradient, data gradient) on the fp32
//                                 matrix instructions;
//   * mx_rowln_fwd / _bwd         LayerNorm([bins, frames]) without affine of one (clip, channel) plane = one contiguous row;
//   * mx_pool_prelu_fwd / _bwd    MaxPool2d((p, 1)) (floor: the last H mod p rows are dropped; first maximum wins) + PReLU;
//   * mx_binmean_head_fwd / _bwd  mean over bins -> Conv1d(C, L, 1) -> sigmoid (models.py:209-215);
//   * mx_row_sums                 fp64-accumulated sums of contiguous rows (bias gradients).
// Every kernel is deterministic (gathers and fixed-order reductions, no atomics).
#include "common.h"

// ---- im2col / col2im --------------------------------------------------------------------------------------------------
struct Im2colGeom {
    int nb, Cin, H, W, kh, kw, dh, dw, pt, pl;
};

__global__ __launch_bounds__(256) void im2col2d_kernel(const float *__restrict__ x, Im2colGeom g, float *__restrict__ col)
{
    const int k = blockIdx.y;                                   // (ci, i, j): workgroup-uniform
    const int j = k % g.kw, i = (k / g.kw) % g.kh, ci = k / (g.kw * g.kh);
    const int HW = g.H * g.W;
    const size_t P = (size_t)g.nb * HW;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int b = (int)(p / HW), r = (int)(p - (size_t)b * HW);
    const int h = r / g.W, w = r - h * g.W;
    const int hs = h + i * g.dh - g.pt, ws = w + j * g.dw - g.pl;
    float v = 0.0f;
    if (hs >= 0 && hs < g.H && ws >= 0 && ws < g.W) v = x[(((size_t)b * g.Cin + ci) * g.H + hs) * g.W + ws];
    col[(size_t)k * P + p] = v;
}

// dx[b][ci][y][x] = sum over taps of dcol[(ci, i, j)][(b, y - i dh + pt, x - j dw + pl)]   (fixed tap order)
__global__ __launch_bounds__(256) void col2im2d_kernel(const float *__restrict__ dcol, Im2colGeom g, float *__restrict__ dx)
{
    const int HW = g.H * g.W;
    const size_t P = (size_t)g.nb * HW, total = P * g.Cin;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int xw = (int)(e % g.W), y = (int)((e / g.W) % g.H), ci = (int)((e / HW) % g.Cin), b = (int)(e / ((size_t)HW * g.Cin));
    float acc = 0.0f;
    for (int i = 0; i < g.kh; ++i) {
        const int h = y - i * g.dh + g.pt;
        if (h < 0 || h >= g.H) continue;
        for (int j = 0; j < g.kw; ++j) {
            const int w = xw - j * g.dw + g.pl;
            if (w < 0 || w >= g.W) continue;
            acc += dcol[(size_t)((ci * g.kh + i) * g.kw + j) * P + (size_t)b * HW + h * g.W + w];
        }
    }
    dx[e] = acc;
}

static int im2col_geom(int64_t nb, int64_t Cin, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t dh, int64_t dw, int64_t pt,
                       int64_t pl, Im2colGeom *g)
{
    if (nb <= 0 || Cin <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || dh <= 0 || dw <= 0 || pt < 0 || pl < 0) return MX_ERR_ARG;
    if (Cin * kh * kw > 65535 || H * W > (1ll << 30) || nb * H * W > (1ll << 40) / 256 || dh * kh > (1 << 20) || dw * kw > (1 << 20))
        return MX_ERR_UNSUPPORTED;
    *g = Im2colGeom{(int)nb, (int)Cin, (int)H, (int)W, (int)kh, (int)kw, (int)dh, (int)dw, (int)pt, (int)pl};
    return MX_OK;
}

MX_EXPORT int mx_im2col2d(const float *x, int64_t nb, int64_t Cin, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t dh, int64_t dw,
                          int64_t pt, int64_t pl, float *col, void *stream)
{
    if (!x || !col) return MX_ERR_ARG;
    Im2colGeom g;
    const int rc = im2col_geom(nb, Cin, H, W, kh, kw, dh, dw, pt, pl, &g);
    if (rc != MX_OK) return rc;
    const size_t P = (size_t)nb * H * W;
    if ((P + 255) / 256 > 0x7fffffffull) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(im2col2d_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(Cin * kh * kw)), dim3(256), 0, (hipStream_t)stream, x, g, col);
    return mx_launch_status();
}

MX_EXPORT int mx_col2im2d(const float *dcol, int64_t nb, int64_t Cin, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t dh,
                          int64_t dw, int64_t pt, int64_t pl, float *dx, void *stream)
{
    if (!dcol || !dx) return MX_ERR_ARG;
    Im2colGeom g;
    const int rc = im2col_geom(nb, Cin, H, W, kh, kw, dh, dw, pt, pl, &g);
    if (rc != MX_OK) return rc;
    const size_t total = (size_t)nb * Cin * H * W;
    if ((total + 255) / 256 > 0x7fffffffull) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(col2im2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcol, g, dx);
    return mx_launch_status();
}

// ---- LayerNorm over one contiguous row (nn.LayerNorm([bins, frames], elementwise_affine=False), models.py:186) ----------
// biased variance, eps inside the square root; fp64 sums (two passes: mean, then centred squares)
__global__ __launch_bounds__(256) void rowln_fwd_kernel(const float *__restrict__ x, int64_t n, float eps, float *__restrict__ y,
                                                        float *__restrict__ stats)
{
    __shared__ double red[4];
    const float *xr = x + (size_t)blockIdx.x * n;
    float *yr = y + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)xr[i];
    const double mean = block256_sum_f64(s, red) / (double)n;
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double d = (double)xr[i] - mean;
        q += d * d;
    }
    const double var = block256_sum_f64(q, red) / (double)n;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int64_t i = threadIdx.x; i < n; i += 256) yr[i] = (xr[i] - mu) * rstd;
    if (threadIdx.x == 0) {
        stats[2 * (size_t)blockIdx.x] = mu;
        stats[2 * (size_t)blockIdx.x + 1] = rstd;
    }
}

// dx = rstd (dy - mean(dy) - y mean(dy y))
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ stats, int64_t n, float *__restrict__ dx)
{
    __shared__ double red[4];
    const size_t row = blockIdx.x;
    const float *gr = dy + row * n, *yr = y + row * n;
    float *dr = dx + row * n;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double g = (double)gr[i];
        s1 += g;
        s2 += g * (double)yr[i];
    }
    const float m1 = (float)(block256_sum_f64(s1, red) / (double)n);
    const float m2 = (float)(block256_sum_f64(s2, red) / (double)n);
    const float rstd = stats[2 * row + 1];
    for (int64_t i = threadIdx.x; i < n; i += 256) dr[i] = rstd * (gr[i] - m1 - yr[i] * m2);
}

MX_EXPORT int mx_rowln_fwd(const float *x, int64_t rows, int64_t n, float eps, float *y, float *stats, void *stream)
{
    if (!x || !y || !stats || rows <= 0 || n <= 0) return MX_ERR_ARG;
    if (rows > 0x7fffffffll) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rowln_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, n, eps, y, stats);
    return mx_launch_status();
}

MX_EXPORT int mx_rowln_bwd(const float *dy, const float *y, const float *stats, int64_t rows, int64_t n, float *dx, void *stream)
{
    if (!dy || !y || !stats || !dx || rows <= 0 || n <= 0) return MX_ERR_ARG;
    if (rows > 0x7fffffffll) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rowln_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, y, stats, n, dx);
    return mx_launch_status();
}

__global__ __launch_bounds__(256) void row_sums_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out)
{
    __shared__ double red[4];
    const float *xr = x + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)xr[i];
    s = block256_sum_f64(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}

MX_EXPORT int mx_row_sums(const float *x, int64_t rows, int64_t n, float *out, void *stream)
{
    if (!x || !out || rows <= 0 || n <= 0) return MX_ERR_ARG;
    if (rows > 0x7fffffffll) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(row_sums_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return mx_launch_status();
}

// ---- MaxPool2d((p, 1)) + PReLU (models.py:188-189) ---------------------------------------------------------------------
// z (planes, H, W) = the convolution products, + bias[c] here (before the comparison, as aten's pooling sees them) -> v = pooled
// pre-activation, out = v > 0 ? v : slope[c] v (aten's PReLU), amax = row offset of the FIRST maximum inside the window
// (aten's max_pool2d: a later row replaces the maximum only if it is greater, or NaN).
__global__ __launch_bounds__(256) void pool_prelu_fwd_kernel(const float *__restrict__ z, const float *__restrict__ bias, int C, int H,
                                                             int W, int p,
                                                             const float *__restrict__ slope, float *__restrict__ v,
                                                             float *__restrict__ out, uint8_t *__restrict__ amax, size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int Hp = H / p;
    const int w = (int)(e % W), hp = (int)((e / W) % Hp);
    const size_t plane = e / ((size_t)W * Hp);
    const float *zp = z + (plane * H + (size_t)hp * p) * W + w;
    const float bc = bias[plane % C];
    float m = zp[0] + bc;
    int am = 0;
    for (int r = 1; r < p; ++r) {
        const float t = zp[(size_t)r * W] + bc;
        if (t > m || t != t) {
            m = t;
            am = r;
        }
    }
    const float a = slope[plane % C];
    v[e] = m;
    out[e] = m > 0.0f ? m : a * m;
    amax[e] = (uint8_t)am;
}

// one workgroup per plane: dz (zero outside the routed rows), part[plane] = {sum of dz (bias gradient), sum of g v [v <= 0]
// (slope gradient)} in fp64
__global__ __launch_bounds__(256) void pool_prelu_bwd_kernel(const float *__restrict__ g, const float *__restrict__ v,
                                                             const uint8_t *__restrict__ amax, int C, int H, int W, int p,
                                                             const float *__restrict__ slope, float *__restrict__ dz,
                                                             float *__restrict__ part)
{
    __shared__ double red[4];
    const size_t plane = blockIdx.x;
    const int Hp = H / p;
    const float a = slope[plane % C];
    const float *gp = g + plane * Hp * W, *vp = v + plane * Hp * W;
    const uint8_t *ap = amax + plane * Hp * W;
    float *dp = dz + plane * H * W;
    double sb = 0.0, ss = 0.0;
    const int n_out = Hp * W;
    for (int e = threadIdx.x; e < n_out; e += 256) {
        const int hp = e / W, w = e - hp * W;
        const float gv = gp[e], vv = vp[e];
        const float d = vv > 0.0f ? gv : a * gv;
        if (!(vv > 0.0f)) ss += (double)gv * (double)vv;
        sb += (double)d;
        const int am = ap[e];
        for (int r = 0; r < p; ++r) dp[((size_t)hp * p + r) * W + w] = r == am ? d : 0.0f;
    }
    for (int e = Hp * p * W + threadIdx.x; e < H * W; e += 256) dp[e] = 0.0f;          // rows the floor-mode pooling dropped
    sb = block256_sum_f64(sb, red);
    ss = block256_sum_f64(ss, red);
    if (threadIdx.x == 0) {
        part[2 * plane] = (float)sb;
        part[2 * plane + 1] = (float)ss;
    }
}

MX_EXPORT int mx_pool_prelu_fwd(const float *z, const float *bias, int64_t planes, int64_t C, int64_t H, int64_t W, int64_t p,
                                const float *slope, float *v, float *out, uint8_t *amax, void *stream)
{
    if (!z || !bias || !slope || !v || !out || !amax || planes <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0) return MX_ERR_ARG;
    if (p > 255 || H / p < 1 || planes % C != 0 || H * W > (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const size_t total = (size_t)planes * (H / p) * W;
    if ((total + 255) / 256 > 0x7fffffffull) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_prelu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, bias, (int)C,
                       (int)H, (int)W, (int)p, slope, v, out, amax, total);
    return mx_launch_status();
}

MX_EXPORT int mx_pool_prelu_bwd(const float *g, const float *v, const uint8_t *amax, int64_t planes, int64_t C, int64_t H, int64_t W,
                                int64_t p, const float *slope, float *dz, float *part, void *stream)
{
    if (!g || !v || !amax || !slope || !dz || !part || planes <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0) return MX_ERR_ARG;
    if (p > 255 || H / p < 1 || planes % C != 0 || H * W > (1ll << 30) || planes > 0x7fffffffll) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_prelu_bwd_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, g, v, amax, (int)C, (int)H, (int)W,
                       (int)p, slope, dz, part);
    return mx_launch_status();
}

// ---- head: mean over bins -> Conv1d(C, L, 1) -> sigmoid (models.py:209-215) ---------------------------------------------
// x (B, C, H, W) -> latent (B, C, W) = sum over h / H (torch.mean: sum, then one division), out (B, L, W)
__global__ __launch_bounds__(256) void binmean_head_fwd_kernel(const float *__restrict__ x, int B, int C, int H, int W,
                                                               const float *__restrict__ wout, const float *__restrict__ bout, int L,
                                                               float *__restrict__ latent, float *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)B * W) return;
    const int w = (int)(e % W), b = (int)(e / W);
    for (int c = 0; c < C; ++c) {
        const float *xp = x + (((size_t)b * C + c) * H) * W + w;
        float s = 0.0f;
        for (int h = 0; h < H; ++h) s += xp[(size_t)h * W];
        latent[((size_t)b * C + c) * W + w] = s / (float)H;
    }
    for (int l = 0; l < L; ++l) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc = fmaf(wout[l * C + c], latent[((size_t)b * C + c) * W + w], acc);     // (this thread's own stores)
        acc += bout[l];
        out[((size_t)b * L + l) * W + w] = 1.0f / (1.0f + expf(-acc));
    }
}

// ds (B, L, W) = d_out out (1 - out); dx (B, C, H, W) = (d_latent + sum_l wout[l][c] ds[l]) / H on every bin
__global__ __launch_bounds__(256) void binmean_head_bwd_kernel(const float *__restrict__ d_out, const float *__restrict__ d_latent,
                                                               const float *__restrict__ out, const float *__restrict__ wout, int B,
                                                               int C, int H, int W, int L, float *__restrict__ ds,
                                                               float *__restrict__ dx)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)B * W) return;
    const int w = (int)(e % W), b = (int)(e / W);
    for (int l = 0; l < L; ++l) {
        const size_t i = ((size_t)b * L + l) * W + w;
        const float o = out[i];
        ds[i] = d_out ? d_out[i] * o * (1.0f - o) : 0.0f;
    }
    for (int c = 0; c < C; ++c) {
        float dl = d_latent ? d_latent[((size_t)b * C + c) * W + w] : 0.0f;
        for (int l = 0; l < L; ++l) dl = fmaf(wout[l * C + c], ds[((size_t)b * L + l) * W + w], dl);
        dl /= (float)H;
        float *dp = dx + (((size_t)b * C + c) * H) * W + w;
        for (int h = 0; h < H; ++h) dp[(size_t)h * W] = dl;
    }
}

MX_EXPORT int mx_binmean_head_fwd(const float *x, int64_t B, int64_t C, int64_t H, int64_t W, const float *wout, const float *bout,
                                  int64_t L, float *latent, float *out, void *stream)
{
    if (!x || !wout || !bout || !latent || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || L <= 0) return MX_ERR_ARG;
    if (B * W > (1ll << 38) || C > (1 << 20) || L > (1 << 20) || H * W > (1ll << 30)) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(binmean_head_fwd_kernel, dim3((unsigned)((B * W + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (int)B, (int)C,
                       (int)H, (int)W, wout, bout, (int)L, latent, out);
    return mx_launch_status();
}

MX_EXPORT int mx_binmean_head_bwd(const float *d_out, const float *d_latent, const float *out, const float *wout, int64_t B, int64_t C,
                                  int64_t H, int64_t W, int64_t L, float *ds, float *dx, void *stream)
{
    if (!out || !wout || !ds || !dx || B <= 0 || C <= 0 || H <= 0 || W <= 0 || L <= 0) return MX_ERR_ARG;
    if (B * W > (1ll << 38) || C > (1 << 20) || L > (1 << 20) || H * W > (1ll << 30)) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(binmean_head_bwd_kernel, dim3((unsigned)((B * W + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out, d_latent,
                       out, wout, (int)B, (int)C, (int)H, (int)W, (int)L, ds, dx);
    return mx_launch_status();
}
