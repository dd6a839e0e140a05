// lstm_generic.hip -- LSTMEffectModel (mod_extraction/models.py:311-339) for ANY in_ch / out_ch / n_hidden / latent_dim:
// nn.LSTM(in_ch + latent_dim, n_hidden, batch_first) -> nn.Linear(n_hidden, out_ch) -> + x -> tanh.  The shipped LSTM-64 with one
// audio and one LFO channel has its own kernels (lstm.hip); no shipped config uses another size, and a param_model
// (lightning.py:344-347,371-375) widens latent_dim.  So this file holds only what is sequential:
//   * mx_lstmg_fwd   the recurrence of one clip per workgroup: gate pre-activations = zin[t] (the input projection
//                    W_ih u_t, an mx_sgemm_f32 over all steps beforehand) + bias + W_hh h_{t-1}; W_hh lives in LDS when it
//                    fits (row pitch n_hidden + 1: a lane per gate row reads its row without bank conflicts), else it is
//                    read from L2 every step; (i, f, g, o, c, h) of every step go to the stash (B, T, 6, n_hidden);
//   * mx_lstmg_bwd   the BPTT recurrence: d loss / d h_t from the output layer (an mx_sgemm_f32) + the recurrent
//                    term W_hh^T dgate_{t+1}, gate derivatives from the stash, dgate (B, T, 4 n_hidden) written out;
//   * mx_lstmg_out_fwd / _bwd   y = tanh(fc + bias + x) with torch's channel broadcast and its derivative.
// Every parameter gradient and d loss / d input is then ONE general GEMM over the stash / dgate (host side: lstm_generic.py).
// Gate order i, f, g, o (aten); accurate expf / tanhf (this path is not the measured one).
#include "common.h"

__device__ __forceinline__ float sigmoidf_acc(float v) { return 1.0f / (1.0f + expf(-v)); }

struct LstmgArgs {
    const float *zin;        // (B, T, 4 Hn)  fwd: W_ih u;     bwd: unused
    const float *bias_ih, *bias_hh;
    const float *w_hh;       // (4 Hn, Hn)
    const float *h0, *c0;    // (B, Hn)
    float *stash;            // (B, T, 6, Hn)
    float *h1, *c1;          // fwd: final state (B, Hn)
    const float *dhfc;       // bwd: (B, T, Hn) gradient from the output layer
    float *dgate;            // bwd: (B, T, 4 Hn)
    int T, Hn, w_in_lds;
};

// LDS: [w: 4 Hn x (Hn + 1) if w_in_lds] [h / dh: Hn] [c / dc: Hn] [gate / dgate: 4 Hn]
__global__ __launch_bounds__(256) void lstmg_fwd_kernel(LstmgArgs a)
{
    extern __shared__ float lg_smem[];
    const int Hn = a.Hn, G = 4 * Hn, WP = Hn + 1, tid = threadIdx.x;
    float *wl = lg_smem;
    float *h = lg_smem + (a.w_in_lds ? (size_t)G * WP : 0), *c = h + Hn, *gate = c + Hn;
    const size_t b = blockIdx.x;
    if (a.w_in_lds)
        for (int e = tid; e < G * Hn; e += 256) wl[(e / Hn) * WP + (e % Hn)] = a.w_hh[e];
    for (int k = tid; k < Hn; k += 256) {
        h[k] = a.h0[b * Hn + k];
        c[k] = a.c0[b * Hn + k];
    }
    __syncthreads();
    for (int t = 0; t < a.T; ++t) {
        const float *z = a.zin + (b * a.T + t) * G;
        for (int r = tid; r < G; r += 256) {
            float acc = z[r] + a.bias_ih[r] + a.bias_hh[r];
            if (a.w_in_lds) {
                const float *wr = wl + (size_t)r * WP;
                for (int k = 0; k < Hn; ++k) acc = fmaf(wr[k], h[k], acc);
            } else {
                const float *wr = a.w_hh + (size_t)r * Hn;
                for (int k = 0; k < Hn; ++k) acc = fmaf(wr[k], h[k], acc);
            }
            const int q = r / Hn;
            gate[r] = q == 2 ? tanhf(acc) : sigmoidf_acc(acc);
        }
        __syncthreads();
        float *st = a.stash + (b * a.T + t) * 6 * Hn;
        for (int k = tid; k < Hn; k += 256) {
            const float gi = gate[k], gf = gate[Hn + k], gg = gate[2 * Hn + k], go = gate[3 * Hn + k];
            const float cn = gf * c[k] + gi * gg;
            const float hn = go * tanhf(cn);
            c[k] = cn;
            h[k] = hn;
            st[k] = gi;
            st[Hn + k] = gf;
            st[2 * Hn + k] = gg;
            st[3 * Hn + k] = go;
            st[4 * Hn + k] = cn;
            st[5 * Hn + k] = hn;
        }
        __syncthreads();
    }
    for (int k = tid; k < Hn; k += 256) {
        a.h1[b * Hn + k] = h[k];
        a.c1[b * Hn + k] = c[k];
    }
}

__global__ __launch_bounds__(256) void lstmg_bwd_kernel(LstmgArgs a)
{
    extern __shared__ float lg_smem[];
    const int Hn = a.Hn, G = 4 * Hn, WP = Hn + 1, tid = threadIdx.x;
    float *wl = lg_smem;
    float *dh = lg_smem + (a.w_in_lds ? (size_t)G * WP : 0), *dc = dh + Hn, *dg = dc + Hn;
    const size_t b = blockIdx.x;
    if (a.w_in_lds)
        for (int e = tid; e < G * Hn; e += 256) wl[(e / Hn) * WP + (e % Hn)] = a.w_hh[e];
    for (int k = tid; k < Hn; k += 256) dh[k] = dc[k] = 0.0f;
    __syncthreads();
    for (int t = a.T - 1; t >= 0; --t) {
        const float *st = a.stash + (b * a.T + t) * 6 * Hn;
        float *out = a.dgate + (b * a.T + t) * G;
        for (int k = tid; k < Hn; k += 256) {
            const float gi = st[k], gf = st[Hn + k], gg = st[2 * Hn + k], go = st[3 * Hn + k], cn = st[4 * Hn + k];
            const float cp = t > 0 ? st[4 * Hn + k - 6 * Hn] : a.c0[b * Hn + k];
            const float dht = a.dhfc[(b * a.T + t) * Hn + k] + dh[k];
            const float tc = tanhf(cn);
            const float dct = dc[k] + dht * go * (1.0f - tc * tc);
            const float dai = dct * gg * gi * (1.0f - gi), daf = dct * cp * gf * (1.0f - gf);
            const float dag = dct * gi * (1.0f - gg * gg), dao = dht * tc * go * (1.0f - go);
            dc[k] = dct * gf;
            dg[k] = dai;
            dg[Hn + k] = daf;
            dg[2 * Hn + k] = dag;
            dg[3 * Hn + k] = dao;
            out[k] = dai;
            out[Hn + k] = daf;
            out[2 * Hn + k] = dag;
            out[3 * Hn + k] = dao;
        }
        __syncthreads();
        for (int k = tid; k < Hn; k += 256) {
            float acc = 0.0f;
            if (a.w_in_lds)
                for (int r = 0; r < G; ++r) acc = fmaf(wl[(size_t)r * WP + k], dg[r], acc);
            else
                for (int r = 0; r < G; ++r) acc = fmaf(a.w_hh[(size_t)r * Hn + k], dg[r], acc);
            dh[k] = acc;
        }
        __syncthreads();
    }
}

static int lstmg_launch(bool bwd, LstmgArgs &a, int64_t B, int64_t T, int64_t Hn, void *stream)
{
    if (B <= 0 || T <= 0 || Hn <= 0) return MX_ERR_ARG;
    if (B > 0x7fffffffll || T > (1ll << 30) || Hn > 4096) return MX_ERR_UNSUPPORTED;
    const size_t small = (size_t)6 * Hn * sizeof(float), wbytes = (size_t)4 * Hn * (Hn + 1) * sizeof(float);
    a.T = (int)T;
    a.Hn = (int)Hn;
    a.w_in_lds = wbytes + small <= 150 * 1024;
    const size_t lds = small + (a.w_in_lds ? wbytes : 0);
    static MxLdsLatch latch[2] = {};
    const void *fn = bwd ? (const void *)lstmg_bwd_kernel : (const void *)lstmg_fwd_kernel;
    if (lds > 64 * 1024 && mx_set_dyn_lds(latch[bwd], fn, 160 * 1024) != MX_OK) return MX_ERR_LAUNCH;
    if (bwd)
        hipLaunchKernelGGL(lstmg_bwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(lstmg_fwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
    return mx_launch_status();
}

MX_EXPORT int mx_lstmg_fwd(const float *zin, const float *bias_ih, const float *bias_hh, const float *w_hh, const float *h0,
                           const float *c0, int64_t B, int64_t T, int64_t Hn, float *stash, float *h1, float *c1, void *stream)
{
    if (!zin || !bias_ih || !bias_hh || !w_hh || !h0 || !c0 || !stash || !h1 || !c1) return MX_ERR_ARG;
    LstmgArgs a{zin, bias_ih, bias_hh, w_hh, h0, c0, stash, h1, c1, nullptr, nullptr, 0, 0, 0};
    return lstmg_launch(false, a, B, T, Hn, stream);
}

MX_EXPORT int mx_lstmg_bwd(const float *stash, const float *dhfc, const float *w_hh, const float *c0, int64_t B, int64_t T, int64_t Hn,
                           float *dgate, void *stream)
{
    if (!stash || !dhfc || !w_hh || !c0 || !dgate) return MX_ERR_ARG;
    LstmgArgs a{nullptr, nullptr, nullptr, w_hh, nullptr, c0, const_cast<float *>(stash), nullptr, nullptr, dhfc, dgate, 0, 0, 0};
    return lstmg_launch(true, a, B, T, Hn, stream);
}

// y (B, Co, T) = tanh(fc[b][t][c or 0] + bias[c or 0] + x[b][c or 0][t]), Co = max(out_ch, in_ch) (models.py:337-338: torch
// broadcasting of (B, out_ch, T) + (B, in_ch, T): equal, or one of them 1)
__global__ __launch_bounds__(256) void lstmg_out_fwd_kernel(const float *__restrict__ fc, const float *__restrict__ bias,
                                                            const float *__restrict__ x, int T, int out_ch, int in_ch, int Co,
                                                            float *__restrict__ y, size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int t = (int)(e % T), c = (int)((e / T) % Co);
    const size_t b = e / ((size_t)T * Co);
    const int co = out_ch == 1 ? 0 : c, ci = in_ch == 1 ? 0 : c;
    y[e] = tanhf(fc[(b * T + t) * out_ch + co] + bias[co] + x[(b * in_ch + ci) * T + t]);
}

// dpre (B, T, out_ch) = sum over the channels c that broadcast onto o of dy[b][c][t] (1 - y[b][c][t]^2)
__global__ __launch_bounds__(256) void lstmg_out_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, int T, int out_ch,
                                                            int Co, float *__restrict__ dpre, size_t total)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int o = (int)(e % out_ch), t = (int)((e / out_ch) % T);
    const size_t b = e / ((size_t)out_ch * T);
    float s = 0.0f;
    if (out_ch == Co) {
        const size_t i = (b * Co + o) * T + t;
        s = dy[i] * (1.0f - y[i] * y[i]);
    } else {
        for (int c = 0; c < Co; ++c) {
            const size_t i = (b * Co + c) * T + t;
            s += dy[i] * (1.0f - y[i] * y[i]);
        }
    }
    dpre[e] = s;
}

MX_EXPORT int mx_lstmg_out_fwd(const float *fc, const float *bias, const float *x, int64_t B, int64_t T, int64_t out_ch, int64_t in_ch,
                               float *y, void *stream)
{
    if (!fc || !bias || !x || !y || B <= 0 || T <= 0 || out_ch <= 0 || in_ch <= 0) return MX_ERR_ARG;
    if (out_ch != in_ch && out_ch != 1 && in_ch != 1) return MX_ERR_ARG;          // not broadcastable
    const int64_t Co = out_ch > in_ch ? out_ch : in_ch;
    const size_t total = (size_t)B * Co * T;
    if ((total + 255) / 256 > 0x7fffffffull || T > (1ll << 30) || Co > (1 << 20)) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(lstmg_out_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fc, bias, x, (int)T,
                       (int)out_ch, (int)in_ch, (int)Co, y, total);
    return mx_launch_status();
}

MX_EXPORT int mx_lstmg_out_bwd(const float *dy, const float *y, int64_t B, int64_t T, int64_t out_ch, int64_t Co, float *dpre, void *stream)
{
    if (!dy || !y || !dpre || B <= 0 || T <= 0 || out_ch <= 0 || Co <= 0) return MX_ERR_ARG;
    if (out_ch != Co && out_ch != 1) return MX_ERR_ARG;
    const size_t total = (size_t)B * T * out_ch;
    if ((total + 255) / 256 > 0x7fffffffull || T > (1ll << 30) || Co > (1 << 20)) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(lstmg_out_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, (int)T,
                       (int)out_ch, (int)Co, dpre, total);
    return mx_launch_status();
}
