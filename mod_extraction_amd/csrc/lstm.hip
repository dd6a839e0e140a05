// lstm.hip -- K10: the sample-wise LSTM-64 effect model (reference: mod_extraction/models.py:311-339,
// nn.LSTM(2, 64, batch_first) -> Linear(64, 1) -> + x -> tanh) and its truncated BPTT with the L1
// loss fused (lightning.py:355-384: one 1024-sample chunk = one forward, one backward, one
// optimizer step, hidden state carried and detached between chunks).
//
// One 256-thread workgroup per clip; thread j owns gate row j (gate order i, f, g, o like torch,
// unit u = j & 63).  The recurrent state never leaves the CU: h lives in LDS, c in registers of
// wave 0, the 256x64 recurrent matrix in registers (64 per thread).  Each time step is
//   all threads : pre[j] = W_ih[j]·(lfo_t, x_t) + b_ih[j] + W_hh[j]·h + b_hh[j]; activation -> LDS
//   wave 0      : c, h update, y_t = tanh(fc·h + b + x_t)
// with two workgroup barriers.  Inputs are staged 256 samples at a time (coalesced), outputs
// likewise.  For BPTT the forward stores (i, f, g, o, c, h) per step (1536 B/sample, the figure
// SURVEY.md section 8d quotes); the backward walks the chunk in reverse, accumulating the
// 17 473 parameter gradients of its clip in registers and writing one partial row per clip,
// which mx_reduce_rows sums over the batch (deterministic, no atomics).
// The kernel is bound by the serial dependency chain (1024 dependent steps per launch), not by
// HBM or MFMA: per step it moves 12 B of audio and 1.5 KB of stash against ~33 kFLOP.
#include "common.h"

#define LS_H 64
#define LS_G 256
#define LS_STASH 384        // floats per time step: gates 256 + c 64 + h 64
#define LS_BLK 256          // samples staged per block
#define LS_NPARAM 17473     // 512 + 16384 + 256 + 256 + 64 + 1

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float *__restrict__ x, long long xs,
                                                       const float *__restrict__ lfo, long long ls,
                                                       const float *__restrict__ w_ih,
                                                       const float *__restrict__ w_hh,
                                                       const float *__restrict__ b_ih,
                                                       const float *__restrict__ b_hh,
                                                       const float *__restrict__ fc_w,
                                                       const float *__restrict__ fc_b, float *__restrict__ h_io,
                                                       float *__restrict__ c_io, float *__restrict__ y, long long ys,
                                                       float *__restrict__ stash, int T)
{
    __shared__ float hbuf[LS_H], gates[LS_G], xin[LS_BLK], lin[LS_BLK], ybuf[LS_BLK];
    const int b = blockIdx.x, j = threadIdx.x, q = j >> 6, u = j & 63;
    float w[LS_H];
#pragma unroll
    for (int k = 0; k < LS_H; ++k) w[k] = w_hh[j * LS_H + k];
    const float wi0 = w_ih[j * 2], wi1 = w_ih[j * 2 + 1], bi = b_ih[j], bh = b_hh[j];
    const float fcw = fc_w[u], fcb = fc_b[0];
    float c_reg = 0.0f;
    if (j < LS_H) {
        hbuf[j] = h_io[(size_t)b * LS_H + j];
        c_reg = c_io[(size_t)b * LS_H + j];
    }
    const float *xb = x + (size_t)b * xs, *lb = lfo + (size_t)b * ls;
    float *yb = y + (size_t)b * ys;
    float *sb = stash ? stash + (size_t)b * T * LS_STASH : nullptr;
    for (int t0 = 0; t0 < T; t0 += LS_BLK) {
        const int cnt = min(LS_BLK, T - t0);
        __syncthreads();
        if (j < cnt) { xin[j] = xb[t0 + j]; lin[j] = lb[t0 + j]; }
        __syncthreads();
        for (int tt = 0; tt < cnt; ++tt) {
            const float xv = xin[tt], lv = lin[tt];
            float ih = fmaf(wi1, xv, fmaf(wi0, lv, bi));       // input order: (lfo, audio), models.py:328
            float hh = bh;
#pragma unroll
            for (int k = 0; k < LS_H; ++k) hh = fmaf(w[k], hbuf[k], hh);
            const float pre = ih + hh;
            const float act = q == 2 ? tanhf(pre) : sigmoidf_(pre);
            gates[j] = act;
            if (sb) sb[(size_t)(t0 + tt) * LS_STASH + j] = act;
            __syncthreads();
            if (j < LS_H) {
                const float ig = gates[u], fg = gates[64 + u], gg = gates[128 + u], og = gates[192 + u];
                c_reg = fmaf(fg, c_reg, ig * gg);
                const float hv = og * tanhf(c_reg);
                if (sb) {
                    sb[(size_t)(t0 + tt) * LS_STASH + 256 + u] = c_reg;
                    sb[(size_t)(t0 + tt) * LS_STASH + 320 + u] = hv;
                }
                const float s = wave_sum_f32(fcw * hv);
                if (u == 0) ybuf[tt] = tanhf(s + fcb + xv);  // models.py:335-337
                hbuf[u] = hv;
            }
            __syncthreads();
        }
        if (j < cnt) yb[t0 + j] = ybuf[j];
    }
    if (j < LS_H) {
        h_io[(size_t)b * LS_H + j] = hbuf[j];
        c_io[(size_t)b * LS_H + j] = c_reg;
    }
}

// x (B rows, stride xs) audio, lfo (B rows, stride ls), T samples each; parameters in torch layout:
// w_ih (256,2), w_hh (256,64), b_ih (256), b_hh (256), fc_w (64), fc_b (1); h_io / c_io (B,64) are
// read as the initial state and overwritten with the final state; y (B rows, stride ys);
// stash (B, T, 384) or NULL when no backward follows.
MX_EXPORT int mx_lstm_fwd(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                          const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w,
                          const float *fc_b, float *h_io, float *c_io, float *y, int64_t y_stride, float *stash,
                          int64_t B, int64_t T, void *stream)
{
    if (!x || !lfo || !w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b || !h_io || !c_io || !y || B <= 0 || T <= 0)
        return MX_ERR_ARG;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, (long long)x_stride,
                       lfo, (long long)lfo_stride, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, h_io, c_io, y,
                       (long long)y_stride, stash, (int)T);
    return mx_launch_status();
}

// ---- truncated BPTT of one chunk with the L1 loss fused -----------------------------------------
// loss = loss_scale_total * sum_{b,t} |y - wet|  (loss_scale = w_l1 / (B*T) for nn.L1Loss 'mean').
// part (B, 17473): per-clip gradient rows in state-dict order
//   [lstm.weight_ih_l0 (256,2) | lstm.weight_hh_l0 (256,64) | lstm.bias_ih_l0 | lstm.bias_hh_l0 | fc.weight | fc.bias]
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float *__restrict__ x, long long xs,
                                                       const float *__restrict__ lfo, long long ls,
                                                       const float *__restrict__ y, long long ys,
                                                       const float *__restrict__ wet, long long ws,
                                                       const float *__restrict__ stash,
                                                       const float *__restrict__ w_hh,
                                                       const float *__restrict__ fc_w,
                                                       const float *__restrict__ h_init,
                                                       const float *__restrict__ c_init, float loss_scale,
                                                       float *__restrict__ part, int T)
{
    __shared__ float dgat[LS_G], hprev[LS_H], pdh[4][LS_H];
    __shared__ float xin[LS_BLK], lin[LS_BLK], yin[LS_BLK], win[LS_BLK];
    const int b = blockIdx.x, j = threadIdx.x, q = j >> 6, u = j & 63;
    float wT[LS_H], dW[LS_H];
#pragma unroll
    for (int k = 0; k < LS_H; ++k) {
        wT[k] = w_hh[(q * LS_H + k) * LS_H + u];     // column u of gate block q
        dW[k] = 0.0f;
    }
    float dwi0 = 0.0f, dwi1 = 0.0f, db = 0.0f;
    const float fcw = fc_w[u];
    float dfcw = 0.0f, dfcb = 0.0f, dh_next = 0.0f, dc_next = 0.0f;
    const float *xb = x + (size_t)b * xs, *lb = lfo + (size_t)b * ls, *yb = y + (size_t)b * ys,
                *wb = wet + (size_t)b * ws;
    const float *sb = stash + (size_t)b * T * LS_STASH;
    const float h0 = j < LS_H ? h_init[(size_t)b * LS_H + u] : 0.0f;
    const float c0 = j < LS_H ? c_init[(size_t)b * LS_H + u] : 0.0f;

    // wave-0 register pipeline over the stash: values of step t are loaded during step t+1
    float n_i = 0.f, n_f = 0.f, n_g = 0.f, n_o = 0.f, n_c = 0.f, n_h = 0.f;      // step t (current)
    float cm1 = 0.f, hm1 = 0.f;                                                  // step t-1
    if (j < LS_H) {
        const float *s = sb + (size_t)(T - 1) * LS_STASH;
        n_i = s[u]; n_f = s[64 + u]; n_g = s[128 + u]; n_o = s[192 + u]; n_c = s[256 + u]; n_h = s[320 + u];
        if (T > 1) { cm1 = s[256 + u - LS_STASH]; hm1 = s[320 + u - LS_STASH]; } else { cm1 = c0; hm1 = h0; }
    }
    const int n_blocks = (T + LS_BLK - 1) / LS_BLK;
    for (int blk = n_blocks - 1; blk >= 0; --blk) {
        const int t0 = blk * LS_BLK, cnt = min(LS_BLK, T - t0);
        __syncthreads();
        if (j < cnt) { xin[j] = xb[t0 + j]; lin[j] = lb[t0 + j]; yin[j] = yb[t0 + j]; win[j] = wb[t0 + j]; }
        __syncthreads();
        for (int tt = cnt - 1; tt >= 0; --tt) {
            const int t = t0 + tt;
            if (j < LS_H) {
                // prefetch step t-1 (gates) and t-2 (c, h) while step t is processed
                float p_i = 0.f, p_f = 0.f, p_g = 0.f, p_o = 0.f, p_c = c0, p_h = h0;
                if (t >= 1) {
                    const float *s = sb + (size_t)(t - 1) * LS_STASH;
                    p_i = s[u]; p_f = s[64 + u]; p_g = s[128 + u]; p_o = s[192 + u];
                    if (t >= 2) { p_c = s[256 + u - LS_STASH]; p_h = s[320 + u - LS_STASH]; }
                }
                const float yv = yin[tt];
                const float e = yv - win[tt];
                const float dy = loss_scale * (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f));
                const float dzy = dy * (1.0f - yv * yv);
                dfcw = fmaf(dzy, n_h, dfcw);
                dfcb += dzy;
                const float dh = fmaf(dzy, fcw, dh_next);
                const float tc = tanhf(n_c);
                const float d_o = dh * tc;
                const float dc = fmaf(dh * n_o, 1.0f - tc * tc, dc_next);
                dgat[u] = dc * n_g * n_i * (1.0f - n_i);
                dgat[64 + u] = dc * cm1 * n_f * (1.0f - n_f);
                dgat[128 + u] = dc * n_i * (1.0f - n_g * n_g);
                dgat[192 + u] = d_o * n_o * (1.0f - n_o);
                dc_next = dc * n_f;
                hprev[u] = hm1;
                // rotate the pipeline: step t-1 becomes current
                n_i = p_i; n_f = p_f; n_g = p_g; n_o = p_o; n_c = cm1; n_h = hm1; cm1 = p_c; hm1 = p_h;
            }
            __syncthreads();
            {
                const float d = dgat[j];
                const float xv = xin[tt], lv = lin[tt];
                dwi0 = fmaf(d, lv, dwi0);
                dwi1 = fmaf(d, xv, dwi1);
                db += d;
                float p = 0.0f;
#pragma unroll
                for (int k = 0; k < LS_H; ++k) {
                    dW[k] = fmaf(d, hprev[k], dW[k]);
                    p = fmaf(wT[k], dgat[q * LS_H + k], p);
                }
                pdh[q][u] = p;
            }
            __syncthreads();
            if (j < LS_H) dh_next = (pdh[0][u] + pdh[1][u]) + (pdh[2][u] + pdh[3][u]);
        }
    }
    float *pb = part + (size_t)b * LS_NPARAM;
    pb[j * 2] = dwi0;
    pb[j * 2 + 1] = dwi1;
#pragma unroll
    for (int k = 0; k < LS_H; ++k) pb[512 + j * LS_H + k] = dW[k];
    pb[512 + 16384 + j] = db;
    pb[512 + 16384 + 256 + j] = db;
    if (j < LS_H) {
        pb[512 + 16384 + 512 + u] = dfcw;
        if (u == 0) pb[LS_NPARAM - 1] = dfcb;
    }
}

MX_EXPORT int mx_lstm_bwd_l1(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                             int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                             const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                             float loss_scale, float *part, int64_t B, int64_t T, void *stream)
{
    if (!x || !lfo || !y || !wet || !stash || !w_hh || !fc_w || !h_init || !c_init || !part || B <= 0 || T <= 0)
        return MX_ERR_ARG;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, (long long)x_stride,
                       lfo, (long long)lfo_stride, y, (long long)y_stride, wet, (long long)wet_stride, stash, w_hh,
                       fc_w, h_init, c_init, loss_scale, part, (int)T);
    return mx_launch_status();
}
