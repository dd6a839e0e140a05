// conv_f16.hip -- the 64->64 channel Spectral2DCNN convolutions (forward and data gradient) on the
// fp16 matrix cores with fp32-equivalent accuracy ("f16x3" split arithmetic).
// Reference semantics are unchanged: mod_extraction/models.py:183-195, LayerNorm -> Conv2d(5x13,
// dilation (1,T), same) -> +bias -> MaxPool(2,1), and the conv's data gradient.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 157 TFLOP/s, v_mfma_f32_32x32x16_f16 at 16x that rate.  Every fp32
// operand x is split once into two fp16 numbers, x*S ~= hi + lo (hi = fp16(x*S), lo = fp16(x*S - hi),
// S a power of two), and the product of two fp32 numbers is evaluated as hi*hi + hi*lo + lo*hi -- three
// fp16 MFMAs accumulating in fp32.  hi+lo carries 22 mantissa bits, the dropped lo*lo term is 2^-22
// relative: measured end to end (6 blocks) the sigmoid output / latent differ from an fp64 evaluation
// by 4e-7 / 2.4e-6, the same as true fp32 arithmetic (4.9e-7 / 3.1e-6) and well inside the 1e-5 gate.
// Three MFMAs per fp32 MAC group = 5.3x the fp32-MFMA rate.
//
// Operands are prepared once per layer by streaming kernels in channels-last fp16 pairs
//   X_hi, X_lo : (B, H, 352, 64)   forward: (prelu(p_prev) - mean) * rstd        (S = 1)
//                                  dgrad  : max-pool routed gradient * S_dz      (S_dz = 2^k from max|G|)
//   W_hi, W_lo : [ci/16][kh][kw][co][16]  weights * 256
// so that the conv kernel stages plain 16-byte vectors and every MFMA fragment (8 consecutive channels of
// one position / one output channel) is ONE aligned ds_read_b128.
//
// Kernel: one 256-thread workgroup per (clip, output-row pair), 1 workgroup per CU (LDS 100-155 KB,
// <= 512 VGPRs per wave).  K loop = 4 channel blocks x 5 kernel rows = 20 stages; per stage the 13 taps of
// one kernel row for 16 input channels: weights 53 KB + the two needed input rows (46-51 KB) in LDS.
// The next stage's global loads are issued into registers before the current stage's MFMAs (26 vectors per
// thread), fragments are software pipelined in two half-tap sets.  4 waves = (co tile) x (output row),
// 11 accumulators of 32x32 each; epilogue identical to the fp32 kernel (bias + max-pool + argmax, or plain).
// Used for the five 64->64 blocks (all dilations; LDS 100-123 KB); the 2-channel first block and the weight
// gradients stay on the exact-fp32 kernels (conv2d.hip, wgrad.hip).
#include "conv_common.h"
#include <hip/hip_fp16.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define F16_WSCALE 256.0f

__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ---- operand preparation -----------------------------------------------------------------------------
// scale[0] = S (power of two with max|x| * S in [512, 1024)), scale[1] = 1 / S; amax_bits = bit pattern of max|x|
__global__ void absmax_kernel(const float *__restrict__ x, long long n4, unsigned *__restrict__ amax_bits)
{
    float m = 0.0f;
    const floatx4 *p = reinterpret_cast<const floatx4 *>(x);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        floatx4 v = p[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));   // order of non-negative floats = order of bits
}
__global__ void pow2_scale_kernel(const unsigned *__restrict__ amax_bits, float *__restrict__ scale)
{
    const float m = __uint_as_float(*amax_bits);
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) {
        frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
        e = 10 - e;                    // m * 2^(10 - e) in [512, 1024)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    scale[0] = ldexpf(1.0f, e);
    scale[1] = ldexpf(1.0f, -e);
}

// NCHW fp32 planes -> channels-last fp16 pairs.  One workgroup per (b, h, 32-column tile): 64 channels x
// 32 columns are read along w (coalesced), transformed, transposed through LDS and written as 128-byte
// channel vectors per position.  MODE 0: xhat = (prelu(x) - mean) * rstd.  MODE 1: dz = routed G * S.
template <int MODE>
__global__ __launch_bounds__(256) void split_prep_kernel(const float *__restrict__ x,
                                                         const unsigned char *__restrict__ amax,
                                                         const float *__restrict__ stats,
                                                         const float *__restrict__ slope,
                                                         const float *__restrict__ scale, int H, int Wv,
                                                         _Float16 *__restrict__ out_hi, _Float16 *__restrict__ out_lo)
{
    __shared__ float tile[64][33];
    const int wt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int Hin = MODE == 1 ? (H >> 1) : H, hin = MODE == 1 ? (h >> 1) : h;
    const float S = MODE == 1 ? scale[0] : 1.0f;
    for (int i = tid; i < 64 * 8; i += 256) {
        const int c = i >> 3, c4 = i & 7;
        const int w0 = wt * 32 + c4 * 4;
        const size_t off = (((size_t)b * 64 + c) * Hin + hin) * CV_PITCH + w0;
        floatx4 v = *reinterpret_cast<const floatx4 *>(x + off);
        if (MODE == 1) {
            const uchar4 am = *reinterpret_cast<const uchar4 *>(amax + off);
            const unsigned want = (unsigned)(h & 1);
            v[0] = am.x == want ? v[0] * S : 0.0f;
            v[1] = am.y == want ? v[1] * S : 0.0f;
            v[2] = am.z == want ? v[2] * S : 0.0f;
            v[3] = am.w == want ? v[3] * S : 0.0f;
        } else {
            const float mean = stats[((size_t)b * 64 + c) * 2], rstd = stats[((size_t)b * 64 + c) * 2 + 1], sl = slope[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e] > 0.0f ? v[e] : sl * v[e];
                v[e] = (t - mean) * rstd;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[c][c4 * 4 + e] = (w0 + e < Wv) ? v[e] : 0.0f;
    }
    __syncthreads();
    // 32 positions x 64 channels: thread -> (position, 8-channel group)
    const int pos = tid >> 3, cg = tid & 7;
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = tile[cg * 8 + j][pos];
        const _Float16 hh = (_Float16)v;
        hi[j] = hh;
        lo[j] = (_Float16)(v - (float)hh);
    }
    const size_t o = ((((size_t)b * H + h) * CV_PITCH + wt * 32 + pos) * 64 + cg * 8);
    *reinterpret_cast<half8 *>(out_hi + o) = hi;
    *reinterpret_cast<half8 *>(out_lo + o) = lo;
}

// torch (64, 64, 5, 13) fp32 -> [ci/16][kh][kw][co][16] fp16 pairs of W * 256
//   flip = 0 (forward): in = ci, out = co;  flip = 1 (dgrad): in = co, out = ci, taps mirrored
__global__ void pack_weights_f16_kernel(const float *__restrict__ W, int flip, _Float16 *__restrict__ w_hi,
                                        _Float16 *__restrict__ w_lo)
{
    const int total = 64 * 64 * CV_TAPS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int kw = i % CV_KW, kh = (i / CV_KW) % CV_KH, ci = (i / CV_TAPS) % 64, co = i / (CV_TAPS * 64);
        const float v = W[i] * F16_WSCALE;
        const _Float16 hh = (_Float16)v;
        const _Float16 ll = (_Float16)(v - (float)hh);
        int cin, cout, kh2, kw2;
        if (!flip) { cin = ci; cout = co; kh2 = kh; kw2 = kw; }
        else { cin = co; cout = ci; kh2 = CV_KH - 1 - kh; kw2 = CV_KW - 1 - kw; }
        const size_t o = ((((size_t)(cin >> 4) * CV_KH + kh2) * CV_KW + kw2) * 64 + cout) * 16 + (cin & 15);
        w_hi[o] = hh;
        w_lo[o] = ll;
    }
}

// ---- the convolution ------------------------------------------------------------------------------------
struct ConvF16Args {
    const _Float16 *x_hi, *x_lo;   // (B, H, 352, 64)
    const _Float16 *w_hi, *w_lo;   // [4][5][13][64][16]
    const float *bias;             // forward: (64,)
    const float *scale;            // dgrad: {S_dz, 1/S_dz} on the device; forward: nullptr
    float *out;                    // forward: (B, 64, H/2, 352) pooled pre-activations; dgrad: (B, 64, H, 352)
    unsigned char *out_amax;       // forward
    int H, Wv;
};

template <int T, int OUTMODE>      // OUTMODE 0: bias + maxpool + argmax, 1: plain rows
__global__ __launch_bounds__(256, 1) void conv_f16x3_kernel(ConvF16Args a)
{
    constexpr int PWP = CV_PITCH + 12 * T;            // patch positions per row (w = q - 6T)
    constexpr int WSL = CV_KW * 64 * 16;              // halfs per weight slab and split (13312)
    constexpr int PSL = 2 * PWP * 16;                 // halfs per patch and split (2 rows)
    constexpr int NWV = (2 * WSL / 8 + 255) / 256;    // 16-byte vectors per thread: weights (13)
    constexpr int NPI = 2 * 2 * PWP * 2;              // patch vectors: split x row x position x 2 halves
    constexpr int NPV = (NPI + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *wl = reinterpret_cast<_Float16 *>(smem);            // [split][kw][co][16]
    _Float16 *pl = wl + 2 * WSL;                                   // [split][row][q][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave & 1, row = wave >> 1, half = lane >> 5, l32 = lane & 31;
    const int b = blockIdx.y, h0 = blockIdx.x * 2;

    floatx16 acc[CV_WT];
#pragma unroll
    for (int i = 0; i < CV_WT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    floatx4 wv[NWV], pv[NPV];
    auto issue = [&](int s) {                                   // global loads of stage s = (cb, kh)
        const int cb = s / CV_KH, kh = s - cb * CV_KH;
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            const int i = tid + q * 256;
            wv[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (i < 2 * (WSL / 8)) {
                const int split = i / (WSL / 8), j = i - split * (WSL / 8);
                const _Float16 *src = (split ? a.w_lo : a.w_hi) + (size_t)(cb * CV_KH + kh) * WSL + (size_t)j * 8;
                wv[q] = *reinterpret_cast<const floatx4 *>(src);
            }
        }
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int i = tid + q * 256;
            pv[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (i < NPI) {
                const int part = i & 1, pos = (i >> 1) % PWP, sr = (i >> 1) / PWP;     // sr = split * 2 + row
                const int split = sr >> 1, r = sr & 1;
                const int hx = h0 + r + kh - 2, w = pos - 6 * T;
                if (hx >= 0 && hx < a.H && w >= 0 && w < CV_PITCH) {
                    const _Float16 *src = (split ? a.x_lo : a.x_hi) +
                                          ((((size_t)b * a.H + hx) * CV_PITCH + w) * 64 + cb * 16 + part * 8);
                    pv[q] = *reinterpret_cast<const floatx4 *>(src);
                }
            }
        }
    };
    auto commit = [&]() {                                       // registers -> LDS (layouts are linear in i)
        floatx4 *wd = reinterpret_cast<floatx4 *>(wl);
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            const int i = tid + q * 256;
            if (i < 2 * (WSL / 8)) wd[i] = wv[q];
        }
        floatx4 *pd = reinterpret_cast<floatx4 *>(pl);
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int i = tid + q * 256;
            if (i < NPI) pd[i] = pv[q];
        }
    };

    constexpr int N_STAGE = 4 * CV_KH;
    issue(0);
    for (int s = 0; s < N_STAGE; ++s) {
        __syncthreads();                                        // previous stage's fragments are all read
        commit();
        if (s + 1 < N_STAGE) issue(s + 1);                      // in flight during the MFMAs below
        __syncthreads();
        // fragment base addresses (halfs)
        const _Float16 *a_hi_p = wl + (mt * 32 + l32) * 16 + half * 8;
        const _Float16 *a_lo_p = a_hi_p + WSL;
        const _Float16 *b_hi_p = pl + ((size_t)row * PWP + l32) * 16 + half * 8;
        const _Float16 *b_lo_p = b_hi_p + PSL;
        // two half-tap fragment sets: tiles 0..5 and 6..10
        half8 ah, al, ah_n, al_n, bh0[6], bl0[6], bh1[5], bl1[5];
#define F16_LOAD_A(AH, AL, KW)                                              \
    AH = *reinterpret_cast<const half8 *>(a_hi_p + (KW) * (64 * 16));        \
    AL = *reinterpret_cast<const half8 *>(a_lo_p + (KW) * (64 * 16));
#define F16_LOAD_S0(KW)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                              \
        bh0[i] = *reinterpret_cast<const half8 *>(b_hi_p + (i * 32 + (KW) * T) * 16);            \
        bl0[i] = *reinterpret_cast<const half8 *>(b_lo_p + (i * 32 + (KW) * T) * 16);            \
    }
#define F16_LOAD_S1(KW)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                              \
        bh1[i] = *reinterpret_cast<const half8 *>(b_hi_p + ((i + 6) * 32 + (KW) * T) * 16);      \
        bl1[i] = *reinterpret_cast<const half8 *>(b_lo_p + ((i + 6) * 32 + (KW) * T) * 16);      \
    }
#define F16_MMA(ACC, AH, AL, BH, BL)        \
    ACC = mfma16(AL, BH, ACC);              \
    ACC = mfma16(AH, BL, ACC);              \
    ACC = mfma16(AH, BH, ACC);
        F16_LOAD_A(ah, al, 0)
        F16_LOAD_S0(0)
        F16_LOAD_S1(0)
#pragma unroll 1
        for (int kw = 0; kw < CV_KW; ++kw) {
            const int kn = kw + 1 < CV_KW ? kw + 1 : kw;         // last tap reloads itself (discarded)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) { F16_MMA(acc[i], ah, al, bh0[i], bl0[i]) }
            __builtin_amdgcn_sched_barrier(0);
            F16_LOAD_A(ah_n, al_n, kn)
            F16_LOAD_S0(kn)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) { F16_MMA(acc[i + 6], ah, al, bh1[i], bl1[i]) }
            __builtin_amdgcn_sched_barrier(0);
            F16_LOAD_S1(kn)
            ah = ah_n;
            al = al_n;
        }
#undef F16_LOAD_A
#undef F16_LOAD_S0
#undef F16_LOAD_S1
#undef F16_MMA
    }

    // ---- epilogue (same data layout as the fp32 kernel) ----
    const float inv = (OUTMODE == 1 ? a.scale[1] : 1.0f) * (1.0f / F16_WSCALE);
    if (OUTMODE == 1) {
        const int h = h0 + row;
#pragma unroll
        for (int i = 0; i < CV_WT; ++i) {
            const int w = i * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mt * 32 + mfma_row(r, lane);
                a.out[(((size_t)b * CV_CO + co) * a.H + h) * CV_PITCH + w] = w < a.Wv ? acc[i][r] * inv : 0.0f;
            }
        }
    } else {
        float *xch = reinterpret_cast<float *>(smem);
        const int hp = h0 >> 1, Hp = a.H >> 1;
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
                        const int co = mt * 32 + mfma_row(r, lane);
                        const float top = acc[i][r];
                        const float bot = xch[((mt * 4 + (i - c0)) * 16 + r) * 64 + lane];
                        const bool take_bot = bot > top;                 // ties keep the first row (torch)
                        const float m = (take_bot ? bot : top) * inv + a.bias[co];
                        const size_t off = (((size_t)b * CV_CO + co) * Hp + hp) * CV_PITCH + w;
                        a.out[off] = w < a.Wv ? m : 0.0f;
                        a.out_amax[off] = take_bot ? 1 : 0;
                    }
                }
            }
        }
    }
}

template <int T, int OUTMODE>
static int launch_f16(const ConvF16Args &a, int B, hipStream_t st)
{
    constexpr int PWP = CV_PITCH + 12 * T;
    const size_t lds = (size_t)(2 * CV_KW * 64 * 16 + 2 * 2 * PWP * 16) * sizeof(_Float16);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)conv_f16x3_kernel<T, OUTMODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MX_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_f16x3_kernel<T, OUTMODE>), dim3(a.H / 2, B), dim3(256), lds, st, a);
    return mx_launch_status();
}

static int dispatch_f16(int T, int outmode, const ConvF16Args &a, int B, hipStream_t st)
{
    if (outmode == 0) {
        if (T == 1) return launch_f16<1, 0>(a, B, st);
        if (T == 2) return launch_f16<2, 0>(a, B, st);
        if (T == 4) return launch_f16<4, 0>(a, B, st);
        if (T == 8) return launch_f16<8, 0>(a, B, st);
        if (T == 16) return launch_f16<16, 0>(a, B, st);
    } else {
        if (T == 1) return launch_f16<1, 1>(a, B, st);
        if (T == 2) return launch_f16<2, 1>(a, B, st);
        if (T == 4) return launch_f16<4, 1>(a, B, st);
        if (T == 8) return launch_f16<8, 1>(a, B, st);
        if (T == 16) return launch_f16<16, 1>(a, B, st);
    }
    return MX_ERR_UNSUPPORTED;
}

// ---- C ABI ------------------------------------------------------------------------------------------------
// weights (64,64,5,13) fp32 -> w_hi, w_lo: 4*5*13*64*16 halfs each (flip = 1 for the data gradient)
MX_EXPORT int mx_conv_pack_weights_f16(const float *W, int32_t flip, void *w_hi, void *w_lo, void *stream)
{
    if (!W || !w_hi || !w_lo) return MX_ERR_ARG;
    hipLaunchKernelGGL(pack_weights_f16_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, W, (int)flip,
                       (_Float16 *)w_hi, (_Float16 *)w_lo);
    return mx_launch_status();
}

// forward operand: x (B,64,H,352) previous block's pooled pre-activations, stats (B,64,2), slope (64,)
// -> x_hi, x_lo (B,H,352,64) fp16 = split of (prelu(x) - mean) * rstd
MX_EXPORT int mx_conv_prep_fwd_f16(const float *x, const float *stats, const float *slope, int64_t B, int64_t H,
                                   int64_t Wv, void *x_hi, void *x_lo, void *stream)
{
    if (!x || !stats || !slope || !x_hi || !x_lo || B <= 0 || H <= 0 || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_ARG;
    if (B > 65535 || H > 65535) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((split_prep_kernel<0>), dim3(CV_PITCH / 32, (unsigned)H, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, x, nullptr, stats, slope, nullptr, (int)H, (int)Wv, (_Float16 *)x_hi,
                       (_Float16 *)x_lo);
    return mx_launch_status();
}

// dgrad operand: G, amax (B,64,H/2,352) -> dz_hi, dz_lo (B,H,352,64) fp16 = split of routed G * S_dz;
// scale (2,) device floats receives {S_dz, 1/S_dz} (S_dz = power of two from max|G|); amax_ws: 1 uint workspace
MX_EXPORT int mx_conv_prep_dgrad_f16(const float *G, const uint8_t *amax, int64_t B, int64_t H, int64_t Wv,
                                     uint32_t *amax_ws, float *scale, void *dz_hi, void *dz_lo, void *stream)
{
    if (!G || !amax || !amax_ws || !scale || !dz_hi || !dz_lo || B <= 0 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH)
        return MX_ERR_ARG;
    if (B > 65535 || H > 65535) return MX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(amax_ws, 0, sizeof(uint32_t), st) != hipSuccess) return MX_ERR_LAUNCH;
    const long long n4 = (long long)B * 64 * (H / 2) * CV_PITCH / 4;
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, st, G, n4, amax_ws);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, amax_ws, scale);
    hipLaunchKernelGGL((split_prep_kernel<1>), dim3(CV_PITCH / 32, (unsigned)H, (unsigned)B), dim3(256), 0, st, G, amax,
                       nullptr, nullptr, scale, (int)H, (int)Wv, (_Float16 *)dz_hi, (_Float16 *)dz_lo);
    return mx_launch_status();
}

// forward conv (64 -> 64 channels, dilation in {1,2,4}) from prepared operands
MX_EXPORT int mx_conv_block_fwd_f16(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo,
                                    const float *bias, int64_t B, int64_t H, int64_t Wv, int32_t dilation, float *out,
                                    uint8_t *out_amax, void *stream)
{
    if (!x_hi || !x_lo || !w_hi || !w_lo || !bias || !out || !out_amax) return MX_ERR_ARG;
    if (B <= 0 || B > 65535 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_UNSUPPORTED;
    ConvF16Args a{(const _Float16 *)x_hi, (const _Float16 *)x_lo, (const _Float16 *)w_hi, (const _Float16 *)w_lo, bias,
                  nullptr, out, out_amax, (int)H, (int)Wv};
    return dispatch_f16(dilation, 0, a, (int)B, (hipStream_t)stream);
}

// data gradient from prepared operands (weights packed with flip = 1); scale = the {S_dz, 1/S_dz} pair
MX_EXPORT int mx_conv_block_dgrad_f16(const void *dz_hi, const void *dz_lo, const void *w_hi, const void *w_lo,
                                      const float *scale, int64_t B, int64_t H, int64_t Wv, int32_t dilation,
                                      float *dxhat, void *stream)
{
    if (!dz_hi || !dz_lo || !w_hi || !w_lo || !scale || !dxhat) return MX_ERR_ARG;
    if (B <= 0 || B > 65535 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH) return MX_ERR_UNSUPPORTED;
    ConvF16Args a{(const _Float16 *)dz_hi, (const _Float16 *)dz_lo, (const _Float16 *)w_hi, (const _Float16 *)w_lo,
                  nullptr, scale, dxhat, nullptr, (int)H, (int)Wv};
    return dispatch_f16(dilation, 1, a, (int)B, (hipStream_t)stream);
}
