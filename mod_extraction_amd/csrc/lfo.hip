// lfo.hip -- K1: LFO synthesis (reference: mod_extraction/modulations.py:16-57).
//
// One thread per output point.  The reference builds the phase with torch.cumsum of a constant
// fp32 step; on the CPU that accumulates in fp64 and rounds each partial sum to fp32, and
// (k+1)*step is exact in fp64, so the phase has the closed form
//     arg[k] = fl32( fl64(k+1+start) * fl64(step) ) + fl32(phase)
// which this kernel evaluates directly (bit-exact phase bookkeeping, no scan).
// `start` supports the phaser ground-truth LFO, which is a crop of a longer signal
// (datasets.py:442-449); `n_out != n_src` applies the align_corners=True resampling of
// util.py:15-29 on the fly (two closed-form evaluations per output point).
// HBM traffic: 4 B written per output point -- trivially HBM-bound, 3.5 KB per 882-point LFO.
#include "common.h"

#define LFO_COS 0
#define LFO_RECT_COS 1
#define LFO_INV_RECT_COS 2
#define LFO_TRI 3
#define LFO_SAW 4
#define LFO_RSAW 5
#define LFO_SQR 6

__device__ __forceinline__ float lfo_value(int k, int start, float step, float ph, int shape, float ex)
{
    const float TWO_PI_F = 6.283185307179586f;
    const float PI_F = 3.141592653589793f;
    const float HALF_PI_F = 1.5707963267948966f;
    double run = (double)((long long)k + 1 + (long long)start) * (double)step;
    float arg = __fadd_rn(__double2float_rn(run), ph);
    float v;
    if (shape == LFO_COS) {
        v = __fmul_rn(__fadd_rn(cosf(__fadd_rn(arg, PI_F)), 1.0f), 0.5f);
    } else if (shape == LFO_RECT_COS) {
        v = fabsf(cosf(__fadd_rn(arg, HALF_PI_F)));
    } else if (shape == LFO_INV_RECT_COS) {
        v = __fadd_rn(-fabsf(cosf(arg)), 1.0f);
    } else if (shape == LFO_SQR) {
        float c = cosf(__fadd_rn(arg, PI_F));
        float s = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
        v = __fmul_rn(__fadd_rn(s, 1.0f), 0.5f);
    } else {
        float saw = __fdiv_rn(torch_remainderf(arg, TWO_PI_F), TWO_PI_F);
        if (shape == LFO_SAW) {
            v = saw;
        } else if (shape == LFO_RSAW) {
            v = __fsub_rn(1.0f, saw);
        } else {  // LFO_TRI
            float tri = __fmul_rn(2.0f, saw);
            v = tri > 1.0f ? __fsub_rn(2.0f, tri) : tri;
        }
    }
    if (ex != 1.0f) {
        // torch.pow(tensor, scalar) fast paths (aten PowKernel.cpp), then the generic powf
        if (ex == 2.0f) v = __fmul_rn(v, v);
        else if (ex == 3.0f) v = __fmul_rn(__fmul_rn(v, v), v);
        else if (ex == 0.5f) v = sqrtf(v);
        else v = powf(v, ex);
    }
    return v;
}

__global__ __launch_bounds__(256) void lfo_synth_kernel(const float *__restrict__ freq,
                                                        const float *__restrict__ phase,
                                                        const int *__restrict__ shape,
                                                        const float *__restrict__ ex,
                                                        const int *__restrict__ start,
                                                        int n_src, int n_out, float sr, float scale,
                                                        float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const float TWO_PI_F = 6.283185307179586f;
    float f = freq[b], ph = phase[b];
    const int sh = shape ? shape[b] : LFO_COS;
    const float e = ex ? ex[b] : 1.0f;
    const int st = start ? start[b] : 0;
    if (sh == LFO_RECT_COS || sh == LFO_INV_RECT_COS) {  // modulations.py:26-29 (exact halving)
        f = __fmul_rn(f, 0.5f);
        ph = __fmul_rn(ph, 0.5f);
    }
    const float step = __fdiv_rn(__fmul_rn(TWO_PI_F, f), sr);  // modulations.py:31
    float v;
    if (n_out == n_src) {
        v = lfo_value(i, st, step, ph, sh, e);
    } else {
        InterpTap t = interp_tap(scale, i, n_src);
        float v0 = lfo_value(t.i0, st, step, ph, sh, e);
        float v1 = t.i1 == t.i0 ? v0 : lfo_value(t.i1, st, step, ph, sh, e);
        v = interp_combine(t, v0, v1);
    }
    out[(size_t)b * n_out + i] = v;
}

// C ABI ---------------------------------------------------------------------------------------
MX_EXPORT int mx_lfo_synth(const float *freq, const float *phase, const int32_t *shape,
                           const float *exp, const int32_t *start, int64_t B, int64_t n_src,
                           int64_t n_out, float sr, float *out, void *stream)
{
    if (!freq || !phase || !out || B <= 0 || n_src <= 0 || n_out <= 0 || sr <= 0.0f) return MX_ERR_ARG;
    if (B > 65535 || n_src >= (1ll << 29) || n_out >= (1ll << 29)) return MX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(lfo_synth_kernel, grid, dim3(256), 0, (hipStream_t)stream, freq, phase, shape,
                       exp, start, (int)n_src, (int)n_out, sr, interp_scale_host(n_src, n_out), out);
    return mx_launch_status();
}

// util.py:15-29 as a stand-alone op: rows (R, n_in) -> (R, n_out), align_corners=True.
__global__ __launch_bounds__(256) void interp_rows_kernel(const float *__restrict__ x, int n_in,
                                                          int n_out, float scale, float *__restrict__ y)
{
    const int r = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const float *row = x + (size_t)r * n_in;
    float v;
    if (n_in == n_out) {
        v = row[i];
    } else {
        InterpTap t = interp_tap(scale, i, n_in);
        v = interp_combine(t, row[t.i0], row[t.i1]);
    }
    y[(size_t)r * n_out + i] = v;
}

MX_EXPORT int mx_interp_linear(const float *x, int64_t rows, int64_t n_in, int64_t n_out, float *y,
                               void *stream)
{
    if (!x || !y || rows <= 0 || n_in <= 0 || n_out <= 0) return MX_ERR_ARG;
    if (rows > 65535 || n_in >= (1ll << 30) || n_out >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(interp_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (int)n_in,
                       (int)n_out, interp_scale_host(n_in, n_out), y);
    return mx_launch_status();
}

// Transpose of mx_interp_linear for a WINDOW of the output axis: dy (rows, j_len) = d loss / d y[:, j0 : j0 + j_len] (zero outside the
// window: a TBPTT step only back-propagates through its own chunk of the resampled LFO, lightning.py:361-366) ->
//   dx[r][i] = sum over j in the window of  [i0(j) == i] lam0(j) dy[j] + [i1(j) == i] lam1(j) dy[j]
// with the taps of util.py:15 / aten UpSample.h exactly as the forward evaluates them (interp_tap).  One thread per (row, i): it
// walks the j that can touch point i (scale * j within (i - 1, i + 1), widened by one on either side against rounding) in
// ascending order -- a gather: deterministic, no atomics.
__global__ __launch_bounds__(256) void interp_rows_bwd_kernel(const float *__restrict__ dy, long long dy_stride, int n_in, int n_out,
                                                              float scale, int j0, int j_len, float *__restrict__ dx)
{
    const int r = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const float *row = dy + (size_t)r * dy_stride;
    float acc = 0.0f;
    if (n_in == n_out) {
        if (i >= j0 && i < j0 + j_len) acc = row[i - j0];
    } else {
        int lo, hi;
        if (scale > 0.0f) {
            lo = (int)floorf((float)(i - 1) / scale) - 1;
            hi = (int)ceilf((float)(i + 1) / scale) + 1;
        } else {                                        // n_out == 1: every output reads point 0
            lo = 0;
            hi = n_out - 1;
        }
        lo = lo < j0 ? j0 : lo;
        hi = hi > j0 + j_len - 1 ? j0 + j_len - 1 : hi;
        for (int j = lo; j <= hi; ++j) {
            const InterpTap t = interp_tap(scale, j, n_in);
            const float g = row[j - j0];
            if (t.i0 == i) acc = fmaf(t.lam0, g, acc);
            if (t.i1 == i) acc = fmaf(t.lam1, g, acc);       // (i0 == i1 at the last point: lam1 is 0 there)
        }
    }
    dx[(size_t)r * n_in + i] = acc;
}

MX_EXPORT int mx_interp_linear_bwd(const float *dy, int64_t dy_stride, int64_t rows, int64_t n_in, int64_t n_out, int64_t j0,
                                   int64_t j_len, float *dx, void *stream)
{
    if (!dy || !dx || rows <= 0 || n_in <= 0 || n_out <= 0 || j0 < 0 || j_len <= 0 || j0 + j_len > n_out || dy_stride < j_len)
        return MX_ERR_ARG;
    if (rows > 65535 || n_in >= (1ll << 30) || n_out >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((n_in + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(interp_rows_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, (long long)dy_stride, (int)n_in,
                       (int)n_out, interp_scale_host(n_in, n_out), (int)j0, (int)j_len, dx);
    return mx_launch_status();
}
