// corners.hip -- K9: LFO post-processing (reference: mod_extraction/modulations.py:219-363):
// moving average, corner detection, corner stretching and the LFO validity filter.
//
// Rows are at most a few hundred frames (345 / 342 / 338), so each row is handled by ONE thread
// that walks it in index order -- exactly the order of the reference's per-item python loops -- with
// the same fp32 operations in the same sequence: the results are bit-identical to the reference
// (tests compare with ==).  A 256-row batch is ~0.1 MFLOP; these kernels are bookkeeping, not a
// roofline target.
#include "common.h"

// top/bottom corner value at interior index i (modulations.py:224-231):
//   -floor( (d_l > 0 ? d_l : 0) * (d_r + 1e-16) )  and the same with d_l < 0
__device__ __forceinline__ void corner_values(const float *m, int i, float &top, float &bot)
{
    const float d_l = __fsub_rn(m[i], m[i - 1]);
    const float d_r = __fsub_rn(m[i + 1], m[i]);
    const float nudged = __fadd_rn(d_r, 1e-16f);
    const float rising = d_l > 0.0f ? d_l : 0.0f;
    const float falling = d_l < 0.0f ? d_l : 0.0f;
    top = (float)(-(long long)floorf(__fmul_rn(rising, nudged)));
    bot = (float)(-(long long)floorf(__fmul_rn(falling, nudged)));
}

// modulations.py:359-363 (x.unfold(-1, k, 1).mean(-1)): left-to-right fp32 sum, then / k.
__global__ void smoothen_kernel(const float *__restrict__ x, int R, int n, int k, float *__restrict__ out)
{
    const int n_out = n - k + 1;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * n_out) return;
    const int r = (int)(idx / n_out), i = (int)(idx % n_out);
    const float *row = x + (size_t)r * n;
    float acc = 0.0f;
    for (int j = 0; j < k; ++j) acc = __fadd_rn(acc, row[i + j]);
    out[idx] = __fdiv_rn(acc, (float)k);
}

MX_EXPORT int mx_smoothen(const float *x, int64_t rows, int64_t n, int64_t k, float *out, void *stream)
{
    if (!x || !out || rows <= 0 || n <= 0 || k < 1 || k > n) return MX_ERR_ARG;
    const long long total = rows * (n - k + 1);
    hipLaunchKernelGGL(smoothen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (int)rows, (int)n, (int)k, out);
    return mx_launch_status();
}

// modulations.py:219-238: float 0/1(+) maps, zero at both ends.
__global__ void find_corners_kernel(const float *__restrict__ x, int R, int n, float *__restrict__ top,
                                    float *__restrict__ bot)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * n) return;
    const int i = (int)(idx % n);
    float t = 0.0f, b = 0.0f;
    if (i >= 1 && i <= n - 2) corner_values(x + (idx - i), i, t, b);
    top[idx] = t;
    bot[idx] = b;
}

MX_EXPORT int mx_find_corners(const float *x, int64_t rows, int64_t n, float *top, float *bot, void *stream)
{
    if (!x || !top || !bot || rows <= 0 || n < 3) return MX_ERR_ARG;
    const long long total = rows * n;
    hipLaunchKernelGGL(find_corners_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (int)rows, (int)n, top, bot);
    return mx_launch_status();
}

// modulations.py:260-307 (_stretch_corners / stretch_corners after smoothing): every segment between
// consecutive anchors (corners, then the last sample) is shifted to start at 0, scaled so that its
// span matches |previous target - target| and re-anchored on the target (1.0 for a top corner, 0.0
// for a bottom corner, the original value for the last sample).  Rows with more than max_n_corners
// corners are copied unchanged.
__global__ void stretch_corners_kernel(const float *__restrict__ x, int R, int n, int max_n_corners,
                                       float *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *m = x + (size_t)r * n;
    float *o = out + (size_t)r * n;
    float n_corners = 0.0f;
    for (int i = 1; i <= n - 2; ++i) {
        float t, b;
        corner_values(m, i, t, b);
        n_corners = __fadd_rn(n_corners, __fadd_rn(t, b));
    }
    for (int i = 0; i < n; ++i) o[i] = m[i];
    if (n_corners > (float)max_n_corners) return;
    int prev_i = 0;
    float prev_target = m[0];
    for (int i = 1; i <= n - 1; ++i) {
        float target;
        if (i == n - 1) {
            target = m[n - 1];
        } else {
            float t, b;
            corner_values(m, i, t, b);
            if (t == 1.0f) target = 1.0f;
            else if (b == 1.0f) target = 0.0f;
            else continue;
        }
        if (prev_target != target) {
            const float have = fabsf(__fsub_rn(m[prev_i], m[i]));
            const float want = fabsf(__fsub_rn(prev_target, target));
            const float gain = __fdiv_rn(want, have);
            float mn = o[prev_i + 1];
            for (int j = prev_i + 2; j <= i; ++j) mn = fminf(mn, o[j]);
            for (int j = prev_i + 1; j <= i; ++j) o[j] = __fmul_rn(__fsub_rn(o[j], mn), gain);
            const float shift = __fsub_rn(target, o[i]);
            for (int j = prev_i + 1; j <= i; ++j) o[j] = __fadd_rn(o[j], shift);
        }
        prev_i = i;
        prev_target = target;
    }
}

MX_EXPORT int mx_stretch_corners(const float *x, int64_t rows, int64_t n, int64_t max_n_corners, float *out,
                                 void *stream)
{
    if (!x || !out || rows <= 0 || n < 3) return MX_ERR_ARG;
    hipLaunchKernelGGL(stretch_corners_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x,
                       (int)rows, (int)n, (int)max_n_corners, out);
    return mx_launch_status();
}

// Gradient of stretch_corners w.r.t. its (smoothed) input -- the reference's _stretch_corners is written with differentiable
// torch ops (modulations.py:260-291), and an UNFROZEN LFO model inside the TBPTT step back-propagates through it
// (lightning.py:258,294-296,344-349).  The corner positions and targets are discrete (no gradient); a stretched segment
// S = (p, i] with anchors p = previous anchor, i = this anchor is, in exact arithmetic (the segment minimum cancels),
//     out_j = (m_j - m_i) g + target,      g = |A - target| / |m_p - m_i|,
// A = the previous target (the tensor m_0 for the first segment), target = 1 / 0 at a corner, the tensor m_(n-1) at the end.
// With the upstream gradient G:  s1 = sum_S G_j,  s2 = sum_S G_j (m_j - m_i):
//     dm_j += G_j g (j in S);   dm_i -= g s1;   dg = s2:  dm_p -= s2 g sign(m_p - m_i) / |m_p - m_i|,  dm_i += the same;
//     first segment: dm_0 += s2 sign(A - target) / |m_p - m_i|;   last segment: dm_i += s1 - s2 sign(A - target) / |m_p - m_i|.
// Unstretched segments (previous target == target), sample 0 and rows with too many corners pass the gradient through.
__global__ void stretch_corners_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout, int R, int n,
                                           int max_n_corners, float *__restrict__ dx)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *m = x + (size_t)r * n, *G = dout + (size_t)r * n;
    float *d = dx + (size_t)r * n;
    float n_corners = 0.0f;
    for (int i = 1; i <= n - 2; ++i) {
        float t, b;
        corner_values(m, i, t, b);
        n_corners = __fadd_rn(n_corners, __fadd_rn(t, b));
    }
    for (int i = 0; i < n; ++i) d[i] = G[i];
    if (n_corners > (float)max_n_corners) return;
    int prev_i = 0;
    float prev_target = m[0];
    for (int i = 1; i <= n - 1; ++i) {
        float target;
        const bool last = i == n - 1;
        if (last) {
            target = m[n - 1];
        } else {
            float t, b;
            corner_values(m, i, t, b);
            if (t == 1.0f) target = 1.0f;
            else if (b == 1.0f) target = 0.0f;
            else continue;
        }
        if (prev_target != target) {
            const float D = m[prev_i] - m[i], have = fabsf(D);
            const float E = prev_target - target, want = fabsf(E);
            const float gain = want / have;
            float s1 = 0.0f, s2 = 0.0f;
            for (int j = prev_i + 1; j <= i; ++j) {
                s1 += G[j];
                s2 = fmaf(G[j], m[j] - m[i], s2);
                d[j] = G[j] * gain;                         // (overwrites the pass-through value set above)
            }
            const float sD = D > 0.0f ? 1.0f : (D < 0.0f ? -1.0f : 0.0f), sE = E > 0.0f ? 1.0f : (E < 0.0f ? -1.0f : 0.0f);
            const float dH = -s2 * gain / have;             // d loss / d |m_p - m_i|
            const float dW = s2 / have;                     // d loss / d |A - target|
            d[i] -= gain * s1 + dH * sD;
            d[prev_i] += dH * sD;
            if (prev_i == 0) d[0] += dW * sE;               // A = m_0 only for the first segment (later ones: the constant 1 / 0)
            if (last) d[i] += s1 - dW * sE;                 // target = m_(n-1)
        }
        prev_i = i;
        prev_target = target;
    }
}

// x: the (rows, n) input of mx_stretch_corners, dout: gradient w.r.t. its output -> dx (rows, n)
MX_EXPORT int mx_stretch_corners_bwd(const float *x, const float *dout, int64_t rows, int64_t n, int64_t max_n_corners,
                                     float *dx, void *stream)
{
    if (!x || !dout || !dx || rows <= 0 || n < 3) return MX_ERR_ARG;
    hipLaunchKernelGGL(stretch_corners_bwd_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, dout,
                       (int)rows, (int)n, (int)max_n_corners, dx);
    return mx_launch_status();
}

// modulations.py:311-356 (check_mod_sig / find_valid_mod_sig_indices): valid[r] = 1 iff the row has
// min_top..max_top top corners, min_bot..max_bot bottom corners and consecutive like corners are at
// least min_gap frames apart (min_gap = int(min_fraction_between_corners * n), computed by the host).
__global__ void check_mod_sig_kernel(const float *__restrict__ x, int R, int n, int min_top, int max_top,
                                     int min_bot, int max_bot, int min_gap, int *__restrict__ valid)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *m = x + (size_t)r * n;
    float n_top = 0.0f, n_bot = 0.0f;
    int last_top = -1, last_bot = -1, gap_top = 0x7fffffff, gap_bot = 0x7fffffff;
    for (int i = 1; i <= n - 2; ++i) {
        float t, b;
        corner_values(m, i, t, b);
        n_top = __fadd_rn(n_top, t);
        n_bot = __fadd_rn(n_bot, b);
        if (t == 1.0f) {
            if (last_top >= 0) gap_top = min(gap_top, i - last_top);
            last_top = i;
        }
        if (b == 1.0f) {
            if (last_bot >= 0) gap_bot = min(gap_bot, i - last_bot);
            last_bot = i;
        }
    }
    bool ok = !(n_top < (float)min_top) && !(n_bot < (float)min_bot) && !(n_top > (float)max_top) &&
              !(n_bot > (float)max_bot);
    if (gap_top < min_gap || gap_bot < min_gap) ok = false;
    valid[r] = ok ? 1 : 0;
}

MX_EXPORT int mx_check_mod_sig(const float *x, int64_t rows, int64_t n, int32_t min_top, int32_t max_top,
                               int32_t min_bot, int32_t max_bot, int32_t min_gap, int32_t *valid, void *stream)
{
    if (!x || !valid || rows <= 0 || n < 3) return MX_ERR_ARG;
    hipLaunchKernelGGL(check_mod_sig_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x,
                       (int)rows, (int)n, min_top, max_top, min_bot, max_bot, min_gap, valid);
    return mx_launch_status();
}
