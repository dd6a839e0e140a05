// conv2d.hip -- K5+K6: LayerNorm -> Conv2d(5x13, dil (1,T), same) -> (+bias, MaxPool(2,1)) of one
// Spectral2DCNN block as a single implicit-GEMM kernel on the fp32 matrix cores, and the data
// gradient of the same convolution (reference: mod_extraction/models.py:183-195; torch.nn
// LayerNorm / Conv2d / MaxPool2d / PReLU semantics).
//
// GEMM view per workgroup:  D[co (64)] [w (352)]  for 2 adjacent output rows h0, h0+1
//     D = sum over (ci, kh, kw) of  Wt[ci][kh][kw][co] * Xhat[ci][h + kh - 2][w + (kw - 6) T]
// 4 waves = (co tile of 32) x (output row); each wave owns 11 accumulators of 32x32
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain -- no reduced precision anywhere).
// The K loop runs over pairs of input channels ("stages"): per stage the 2 x 6 x (352 + 2 HALO)
// input patch and the 2 x 65 x 64 weight slab are staged in LDS (51-59 KB, two workgroups per CU so
// one stages while the other multiplies); the two lane halves of a wave read the two channels of
// the pair, so every LDS fragment read is 32 consecutive floats (conflict free).
//
// The input transform is fused into the staging step, so normalised activations never exist in
// HBM:   forward l>=2:  xhat = (prelu(p_prev) - mean) * rstd   (PReLU of the previous block and
//                       LayerNorm of this block; zero outside the image = "same" padding of xhat)
//        forward l=1 :  xhat = (logmel - mean) * rstd
//        dgrad       :  dz[h][w] = (argmax[h/2][w] == h&1) ? G[h/2][w] : 0   (max-pool routing)
// Epilogues: forward adds the bias, max-pools the two rows (first-max wins, like torch) and stores
// the pooled PRE-activation p plus a 1-byte argmax; dgrad stores the two rows of dxhat.
//
// Roofline: MFMA-bound (fp32 matrix peak 157 TFLOP/s).  Useful flops per launch are
// 2*64*Cin*65*H*345 per clip; padded columns 345..351 cost 2 %.
#include "conv_common.h"

enum { IN_LOGMEL = 0, IN_PRELU = 1, IN_ROUTE = 2 };
enum { OUT_POOL = 0, OUT_PLAIN = 1 };

struct ConvArgs {
    const float *in;            // (B, Cin, Hin, PITCH): Hin = H (forward) or H/2 (dgrad: pooled grads G)
    const unsigned char *amax;  // dgrad: (B, Cin, H/2, PITCH) argmax row of the forward pool
    const float *stats;         // forward: (B, Cin, 2) = mean, rstd of the (transformed) input plane
    const float *slope;         // forward l>=2: (Cin,) PReLU slopes of the previous block
    const float *wt;            // packed weights (Cin, 5, 13, 64): [ci][kh][kw][co]
    const float *bias;          // forward: (64,)
    float *out;                 // forward: (B, 64, H/2, PITCH) pooled pre-activations; dgrad: (B, 64, H, PITCH)
    unsigned char *out_amax;    // forward: (B, 64, H/2, PITCH)
    int Cin, H, Wv;             // H = conv-domain height (rows of xhat / dz), Wv = valid width (345)
};

template <int T, int INMODE, int OUTMODE>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvArgs a)
{
    constexpr int HALO = cv_halo(T);
    constexpr int PW = CV_PITCH + 2 * HALO;      // patch row length (floats)
    constexpr int PW4 = PW / 4;
    constexpr int PROWS = 6;
    constexpr int PATCH = 2 * PROWS * PW;
    constexpr int WSLAB = 2 * CV_TAPS * CV_CO;   // 8320 floats
    __shared__ __attribute__((aligned(16))) float lds[PATCH + WSLAB];
    float *patch = lds;
    float *wl = lds + PATCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave & 1, row = wave >> 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int b = blockIdx.y, h0 = blockIdx.x * 2;
    const int Hin = (INMODE == IN_ROUTE) ? (a.H >> 1) : a.H;

    floatx16 acc[CV_WT];
#pragma unroll
    for (int i = 0; i < CV_WT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int n_stage = a.Cin >> 1;
    for (int s = 0; s < n_stage; ++s) {
        __syncthreads();
        // ---- stage the weight slab of channels (2s, 2s+1): contiguous 33 KB ----
        // All global loads of a batch are issued before the first LDS write (one memory round trip per
        // batch instead of one per vector): 8+1 weight vectors, then NP patch vectors per thread.
        {
            const floatx4 *src = reinterpret_cast<const floatx4 *>(a.wt + (size_t)s * WSLAB);
            floatx4 *dst = reinterpret_cast<floatx4 *>(wl);
            constexpr int NWV = (WSLAB / 4) / 256;             // 8 full rounds, + 32 vectors
            floatx4 wv[NWV], wtail = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < NWV; ++q) wv[q] = src[tid + q * 256];
            if (tid < WSLAB / 4 - NWV * 256) wtail = src[NWV * 256 + tid];
#pragma unroll
            for (int q = 0; q < NWV; ++q) dst[tid + q * 256] = wv[q];
            if (tid < WSLAB / 4 - NWV * 256) dst[NWV * 256 + tid] = wtail;
        }
        // ---- stage the input patch with the fused transform ----
        {
            constexpr int NITEM = 2 * PROWS * PW4;
            constexpr int NP = (NITEM + 255) / 256;
            floatx4 pv[NP];
            uchar4 pam[NP];
            float st_mean[2], st_rstd[2], st_sl[2];
            if (INMODE != IN_ROUTE) {
#pragma unroll
                for (int cl = 0; cl < 2; ++cl) {
                    const int ci = 2 * s + cl;
                    st_mean[cl] = a.stats[((size_t)b * a.Cin + ci) * 2];
                    st_rstd[cl] = a.stats[((size_t)b * a.Cin + ci) * 2 + 1];
                    st_sl[cl] = (INMODE == IN_PRELU) ? a.slope[ci] : 1.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int i = tid + q * 256;
                const int rowid = i / PW4, c4 = i - rowid * PW4;
                const int cl = rowid / PROWS, r = rowid - cl * PROWS;
                const int h = h0 - 2 + r, w0 = c4 * 4 - HALO;
                pv[q] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
                pam[q] = uchar4{2, 2, 2, 2};
                if (i < NITEM && h >= 0 && h < a.H && w0 >= 0 && w0 < CV_PITCH) {
                    const int hin = (INMODE == IN_ROUTE) ? (h >> 1) : h;
                    const size_t off = (((size_t)b * a.Cin + 2 * s + cl) * Hin + hin) * CV_PITCH + w0;
                    pv[q] = *reinterpret_cast<const floatx4 *>(a.in + off);
                    if (INMODE == IN_ROUTE) pam[q] = *reinterpret_cast<const uchar4 *>(a.amax + off);
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int i = tid + q * 256;
                const int rowid = i / PW4, c4 = i - rowid * PW4;
                const int cl = rowid / PROWS, r = rowid - cl * PROWS;
                const int h = h0 - 2 + r, w0 = c4 * 4 - HALO;
                floatx4 v = pv[q];
                const bool inside = h >= 0 && h < a.H && w0 >= 0 && w0 < CV_PITCH;
                if (INMODE == IN_ROUTE) {
                    const unsigned want = (unsigned)(h & 1);
                    v[0] = pam[q].x == want ? v[0] : 0.0f;
                    v[1] = pam[q].y == want ? v[1] : 0.0f;
                    v[2] = pam[q].z == want ? v[2] : 0.0f;
                    v[3] = pam[q].w == want ? v[3] : 0.0f;
                } else if (inside) {
                    const float mean = cl ? st_mean[1] : st_mean[0], rstd = cl ? st_rstd[1] : st_rstd[0];
                    const float sl = cl ? st_sl[1] : st_sl[0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e];
                        if (INMODE == IN_PRELU) x = x > 0.0f ? x : sl * x;
                        v[e] = (x - mean) * rstd;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!inside || w0 + e >= a.Wv) v[e] = 0.0f;
                if (i < NITEM) *reinterpret_cast<floatx4 *>(patch + rowid * PW + c4 * 4) = v;
            }
        }
        __syncthreads();
        // ---- 65 k-steps (one tap each, both channels of the pair), 11 MFMAs per k-step ----
        // Software pipelined by hand with two fragment register sets (ping-pong, loop unrolled by 2): the
        // 12 LDS fragment reads of tap k+1 are issued BEFORE the 11 MFMAs of tap k, and sched_barriers keep
        // the compiler from rotating the loads back in front of their own MFMAs, so the matrix pipe never
        // waits on an LDS round trip.
        {
            const float *wp = wl + half * (CV_TAPS * CV_CO) + mt * 32 + l32;
            const float *pp = patch + half * (PROWS * PW) + row * PW + HALO + l32 - 6 * T;
            int kw = 0;
            float a0, a1, b0[CV_WT], b1[CV_WT];
#define CV_ADVANCE()                                                                      \
    wp += CV_CO;                                                                          \
    pp += (kw == CV_KW - 1) ? (PW - (CV_KW - 1) * T) : T; /* kw+1, or first tap of next row */ \
    kw = (kw == CV_KW - 1) ? 0 : kw + 1;
#define CV_LOAD(A, B)                                                                     \
    A = wp[0];                                                                            \
    _Pragma("unroll") for (int i = 0; i < CV_WT; ++i) B[i] = pp[i * 32];
#define CV_MMA(A, B) _Pragma("unroll") for (int i = 0; i < CV_WT; ++i) acc[i] = mfma32(A, B[i], acc[i]);
            __builtin_amdgcn_s_setprio(1);              // matrix phase outranks the partner workgroup's staging VALU
            CV_LOAD(a0, b0)
#pragma unroll 1
            for (int tap = 0; tap < CV_TAPS - 1; tap += 2) {
                CV_ADVANCE()
                CV_LOAD(a1, b1)
                __builtin_amdgcn_sched_barrier(0);
                CV_MMA(a0, b0)
                __builtin_amdgcn_sched_barrier(0);
                CV_ADVANCE()
                CV_LOAD(a0, b0)
                __builtin_amdgcn_sched_barrier(0);
                CV_MMA(a1, b1)
                __builtin_amdgcn_sched_barrier(0);
            }
            CV_MMA(a0, b0)           // tap 64 (65 taps: 32 pairs + 1)
            __builtin_amdgcn_s_setprio(0);
#undef CV_ADVANCE
#undef CV_LOAD
#undef CV_MMA
        }
    }

    // ---- epilogue ----
    if (OUTMODE == OUT_PLAIN) {
        // store address = (wave-uniform base of the channel row: scalar registers) + (one 32-bit lane offset): written as a
        // 64-bit offset per (tile, register) the 176 addresses spilled 37-53 registers to scratch memory
        const int h = h0 + row;
        const unsigned vo = (unsigned)(half * 4 * a.H * CV_PITCH + l32);
#pragma unroll
        for (int i = 0; i < CV_WT; ++i) {
            const int w = i * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co_u = mt * 32 + (r & 3) + 8 * (r >> 2);             // + 4 * half: in vo
                float *rowp = a.out + (((size_t)b * CV_CO + co_u) * a.H + h) * CV_PITCH + i * 32;
                rowp[vo] = w < a.Wv ? acc[i][r] : 0.0f;
            }
        }
    } else {
        // max-pool the two rows: row-1 waves hand their accumulators to the row-0 waves through LDS,
        // 4 w-tiles at a time (2 co tiles x 4 x 16 regs x 64 lanes x 4 B = 32 KB, fits the staging LDS)
        float *xch = lds;
        const int hp = h0 >> 1, Hp = a.H >> 1;
        const unsigned vo = (unsigned)(half * 4 * Hp * CV_PITCH + l32);
        // this lane's 16 bias values, fetched before the store loop (a load between stores would make every
        // iteration wait for the previous stores: vmcnt retires in order)
        float bias_r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = a.bias[mt * 32 + mfma_row(r, lane)];
#pragma unroll
        for (int c0 = 0; c0 < CV_WT; c0 += 4) {
            __syncthreads();
            if (row == 1) {
#pragma unroll
                for (int i = c0; i < c0 + 4 && i < CV_WT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((mt * 4 + (i - c0)) * 16 + r) * 64 + lane] = acc[i][r];
            }
            __syncthreads();
            if (row == 0) {
#pragma unroll
                for (int i = c0; i < c0 + 4 && i < CV_WT; ++i) {
                    const int w = i * 32 + l32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co_u = mt * 32 + (r & 3) + 8 * (r >> 2);     // + 4 * half: in vo (scalar base + lane offset, as above)
                        const float top = acc[i][r];
                        const float bot = xch[((mt * 4 + (i - c0)) * 16 + r) * 64 + lane];
                        const bool take_bot = bot > top;                 // ties keep the first row (torch)
                        const float m = (take_bot ? bot : top) + bias_r[r];
                        const size_t ro = (((size_t)b * CV_CO + co_u) * Hp + hp) * CV_PITCH + i * 32;
                        (a.out + ro)[vo] = w < a.Wv ? m : 0.0f;
                        (a.out_amax + ro)[vo] = take_bot ? 1 : 0;
                    }
                }
            }
        }
    }
}

template <int T>
static int launch_conv(int inmode, int outmode, const ConvArgs &a, int B, hipStream_t st)
{
    dim3 grid(a.H / 2, B), block(256);
    if (inmode == IN_LOGMEL && outmode == OUT_POOL)
        hipLaunchKernelGGL((conv_kernel<T, IN_LOGMEL, OUT_POOL>), grid, block, 0, st, a);
    else if (inmode == IN_PRELU && outmode == OUT_POOL)
        hipLaunchKernelGGL((conv_kernel<T, IN_PRELU, OUT_POOL>), grid, block, 0, st, a);
    else if (inmode == IN_ROUTE && outmode == OUT_PLAIN)
        hipLaunchKernelGGL((conv_kernel<T, IN_ROUTE, OUT_PLAIN>), grid, block, 0, st, a);
    else
        return MX_ERR_ARG;
    return mx_launch_status();
}

static int dispatch_conv(int T, int inmode, int outmode, const ConvArgs &a, int B, hipStream_t st)
{
    switch (T) {
    case 1: return launch_conv<1>(inmode, outmode, a, B, st);
    case 2: return launch_conv<2>(inmode, outmode, a, B, st);
    case 4: return launch_conv<4>(inmode, outmode, a, B, st);
    case 8: return launch_conv<8>(inmode, outmode, a, B, st);
    case 16: return launch_conv<16>(inmode, outmode, a, B, st);
    default: return MX_ERR_UNSUPPORTED;
    }
}

// ---- weight packing: torch (Cout, Cin, 5, 13) -> kernel layout --------------------------------
//   flip == 0 (forward):  wt[ci][kh][kw][co] = W[co][ci][kh][kw]                    (Cin x 65 x Cout)
//   flip == 1 (dgrad):    wt[co][kh][kw][ci] = W[co][ci][4-kh][12-kw]                (Cout x 65 x Cin)
__global__ void pack_weights_kernel(const float *__restrict__ W, int Cout, int Cin, int flip,
                                    float *__restrict__ wt)
{
    const int total = Cout * Cin * CV_TAPS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int kw = i % CV_KW, kh = (i / CV_KW) % CV_KH, ci = (i / CV_TAPS) % Cin, co = i / (CV_TAPS * Cin);
        const float v = W[i];
        if (!flip)
            wt[((size_t)(ci * CV_KH + kh) * CV_KW + kw) * Cout + co] = v;
        else
            wt[((size_t)(co * CV_KH + (CV_KH - 1 - kh)) * CV_KW + (CV_KW - 1 - kw)) * Cin + ci] = v;
    }
}

MX_EXPORT int mx_conv_pack_weights(const float *W, int64_t Cout, int64_t Cin, int32_t flip, float *wt,
                                   void *stream)
{
    if (!W || !wt || Cout <= 0 || Cin <= 0) return MX_ERR_ARG;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, W, (int)Cout, (int)Cin,
                       (int)flip, wt);
    return mx_launch_status();
}

// ---- forward block: LayerNorm (stats given) -> conv -> +bias -> maxpool(2,1) ------------------
// in (B,Cin,H,352): log-mel (first_layer=1) or the previous block's pooled pre-activations (PReLU
// with `slope` applied on the fly); stats (B,Cin,2) from mx_plane_stats; wt = packed forward
// weights; out (B,64,H/2,352) pooled pre-activations; out_amax (B,64,H/2,352) uint8.
MX_EXPORT int mx_conv_block_fwd(const float *in, const float *stats, const float *slope, const float *wt,
                                const float *bias, int64_t B, int64_t Cin, int64_t H, int64_t Wv,
                                int32_t dilation, int32_t first_layer, float *out, uint8_t *out_amax,
                                void *stream)
{
    if (!in || !stats || !wt || !bias || !out || !out_amax || (!first_layer && !slope)) return MX_ERR_ARG;
    if (B <= 0 || B > 65535 || Cin < 2 || (Cin & 1) || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH)
        return MX_ERR_UNSUPPORTED;
    ConvArgs a{in, nullptr, stats, slope, wt, bias, out, out_amax, (int)Cin, (int)H, (int)Wv};
    return dispatch_conv(dilation, first_layer ? IN_LOGMEL : IN_PRELU, OUT_POOL, a, (int)B, (hipStream_t)stream);
}

// ---- data gradient: G (B,64,H/2,352) pooled grads + argmax -> dxhat (B,Cin_orig=64,H,352) ------
// wt = weights packed with flip=1.  (Not needed for the first block: the log-mel input needs no grad.)
MX_EXPORT int mx_conv_block_dgrad(const float *G, const uint8_t *amax, const float *wt_flipped, int64_t B,
                                  int64_t H, int64_t Wv, int32_t dilation, float *dxhat, void *stream)
{
    if (!G || !amax || !wt_flipped || !dxhat) return MX_ERR_ARG;
    if (B <= 0 || B > 65535 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_UNSUPPORTED;
    ConvArgs a{G, amax, nullptr, nullptr, wt_flipped, nullptr, dxhat, nullptr, CV_CO, (int)H, (int)Wv};
    return dispatch_conv(dilation, IN_ROUTE, OUT_PLAIN, a, (int)B, (hipStream_t)stream);
}
