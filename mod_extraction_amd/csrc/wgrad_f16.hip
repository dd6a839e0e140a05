// wgrad_f16.hip -- weight gradient of the 64->64 channel Spectral2DCNN convolutions on the fp16 matrix
// cores with fp32-equivalent accuracy ("f16x3", see conv_f16.hip for the arithmetic and its error budget).
// Reference semantics: torch.nn.Conv2d backward w.r.t. weight (mod_extraction/models.py:187):
//   dW[co][ci][kh][kw] = sum over (b, h, w) of  dz[b][co][h][w] * xhat[b][ci][h + kh - 2][w + (kw - 6) T]
//
// Operands are the channels-last fp16 pairs the f16x3 path prepares anyway: dz_hi/lo (the routed, scaled
// gradient, shared with the data-gradient kernel) and x_hi/x_lo (the normalised block input), both
// (B, H, 4, 352, 16) -- channel-block major, see conv_f16.hip.  GEMM view per kernel row kh: D[co][ci] (13 taps) += A[co][k] * B[k][ci] with k = position.
// The MFMA wants 8 consecutive k per lane, i.e. 8 consecutive POSITIONS of one channel, while memory and LDS
// hold 64 consecutive CHANNELS per position: the fragments are fetched with ds_read_b64_tr_b16 (gfx950's
// transposing LDS read: a 16-lane group reads 4 rows x 16 columns of halfs and each lane receives one column),
// so no transposition pass exists anywhere and the tap shift (kw - 6) T is a plain ROW offset (always aligned).
// LDS rows are 128 B (64 channels); their two 64-byte halves are swapped on rows with bit 1 set, which makes the
// 4-row transposed reads bank-conflict free.
//
// Work split: workgroup = (kh, slab of (b, h) rows); 4 waves = (ci tile, tap group g), each holding 13 accumulators
// (208 registers): both co tiles of its 6 taps 7g .. 7g+5 plus co tile g of the middle tap 6 -- so a k-step reads
// 4 dz and 14 x fragments per wave instead of 2 and 26 (the LDS read pipe is shared by the four waves);
// 1 workgroup per CU.  Per row two chunks of 176 positions (11 MFMA k-steps);
// the next chunk's vectors are fetched into registers during the current chunk's MFMAs.  Workgroup ids are
// remapped so that the 5 kh-workgroups of one slab run on the same XCD (they re-read the same rows from L2).
// Partial results per slab go through the same deterministic fp64 slab reduction as the fp32 kernel.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ floatx16 mfma16w(half8 a, half8 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#define WF_CH 176                      // positions per chunk (352 = 2 x 176, 11 k-steps of 16)
#define WF_KS (WF_CH / 16)

struct WgradF16Args {
    const _Float16 *dz_hi, *dz_lo;     // (B, H, 4, 352, 16)
    const _Float16 *x_hi, *x_lo;       // (B, H, 4, 352, 16)
    float *part;                       // (n_slabs, 5, 13, 64, 64)
    int B, H, rows_per_slab, n_slabs;
};

// One transposed fragment = 8 consecutive rows (positions) r0 .. r0+7 of a 32-channel tile, for this lane's
// channel: two ds_read_b64_tr_b16 (rows +0..3 and +4..7 of the lane half's 8 rows).
// Address of lane (q, p, g1, h2) for read t:  row = r0 + 8 h2 + 4 t + q,  byte = row*128 + (tile ^ swz)*64 +
// 32 g1 + 8 p  with swz = (row >> 1) & 1.  r0 = 16 ks + kw T is wave-uniform, so swz = (((r0 & 3) + q) >> 1) & 1
// takes one of 4 lane patterns selected by r0 & 3: the 4 lane offsets are computed once per kernel and every
// read is `image + lane_off[r0 & 3] + r0*128 + t*512` (t*512 and the kw T part fold into the instruction's
// immediate offset), leaving the VALU free for nothing but the MFMAs' neighbours.
struct TrLane { int off[4]; };
__device__ __forceinline__ TrLane tr_lane_offsets(int tile, int lane)
{
    const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1, h2 = lane >> 5;
    TrLane L;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int swz = ((ph + q) >> 1) & 1;
        L.off[ph] = (8 * h2 + q) * 128 + ((tile ^ swz) * 64) + 32 * g1 + 8 * p;
    }
    return L;
}
// img_bytes: LDS byte address of the image; r0 = first row (wave-uniform); ph = r0 & 3 (compile-time where r0 is)
__device__ __forceinline__ half8 tr_frag(const unsigned char *img_bytes, int lane_off, int r0)
{
    half8 out;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned char *ptr = img_bytes + lane_off + r0 * 128 + t * 512;
        short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)ptr);
        union { short4v s; _Float16 h[4]; } u;
        u.s = v;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * t + j] = u.h[j];
    }
    return out;
}

template <int T>
__global__ __launch_bounds__(256, 1) void wgrad_f16x3_kernel(WgradF16Args a)
{
    constexpr int WIN = WF_CH + 12 * T;                 // staged x positions per chunk (origin at w0 - 6T)
    constexpr int NDZ = 2 * WF_CH * 8;                  // 16-byte vectors: split x position x 8
    constexpr int NX = 2 * WIN * 8;
    constexpr bool PREF = T < 16;                       // T = 16 would need 34 prefetch vectors per thread
    constexpr int NV = PREF ? (NDZ + NX + 255) / 256 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *dzl = reinterpret_cast<_Float16 *>(smem);               // [split][WF_CH][64]
    _Float16 *xl = dzl + 2 * WF_CH * 64;                               // [split][WIN][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & 1, nt = wave >> 1;
    // workgroup id -> (kh, slab): the 5 kh of a slab get ids that differ by 8 (same XCD under round-robin
    // dispatch; a speed-only assumption, correctness does not depend on placement)
    const int id = blockIdx.x;
    const int kh = (id >> 3) % CV_KH;
    const int slab = (id & 7) + 8 * (id / (8 * CV_KH));
    if (slab >= a.n_slabs) return;

    floatx16 acc[CV_KW];
#pragma unroll
    for (int i = 0; i < CV_KW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int row_begin = slab * a.rows_per_slab;
    int row_end = row_begin + a.rows_per_slab;
    if (row_end > a.B * a.H) row_end = a.B * a.H;
    auto row_valid = [&](int rid) {
        const int hx = rid % a.H + kh - 2;
        return hx >= 0 && hx < a.H;
    };
    auto next_iter = [&](int &rid, int &ch) {
        if (++ch < 2) return;
        ch = 0;
        do { ++rid; } while (rid < row_end && !row_valid(rid));
    };

    floatx4 pv[NV];
    // vector i of a chunk: i < NDZ -> dz (split, pos, v); else x window (split, pos, v)
    auto load_vec = [&](int i, int rid, int ch) -> floatx4 {
        const int b = rid / a.H, h = rid - b * a.H, w0 = ch * WF_CH;
        floatx4 z = {0.f, 0.f, 0.f, 0.f};
        if (i < NDZ) {
            const int v = i & 7, pos = (i >> 3) % WF_CH, split = (i >> 3) / WF_CH;
            const _Float16 *src = (split ? a.dz_lo : a.dz_hi) +
                                  ((((size_t)b * a.H + h) * 4 + (v >> 1)) * CV_PITCH + w0 + pos) * 16 + (v & 1) * 8;
            return *reinterpret_cast<const floatx4 *>(src);
        } else if (i < NDZ + NX) {
            const int k = i - NDZ;
            const int v = k & 7, pos = (k >> 3) % WIN, split = (k >> 3) / WIN;
            const int w = w0 - 6 * T + pos, hx = h + kh - 2;
            if (w >= 0 && w < CV_PITCH) {
                const _Float16 *src = (split ? a.x_lo : a.x_hi) +
                                      ((((size_t)b * a.H + hx) * 4 + (v >> 1)) * CV_PITCH + w) * 16 + (v & 1) * 8;
                return *reinterpret_cast<const floatx4 *>(src);
            }
        }
        return z;
    };
    auto store_vec = [&](int i, floatx4 val) {
        if (i < NDZ) {
            const int v = i & 7, pos = (i >> 3) % WF_CH, split = (i >> 3) / WF_CH;
            const int vs = v ^ (((pos >> 1) & 1) << 2);                 // swap the 64-byte halves on rows with bit 1 set
            *reinterpret_cast<floatx4 *>(dzl + ((size_t)split * WF_CH + pos) * 64 + vs * 8) = val;
        } else if (i < NDZ + NX) {
            const int k = i - NDZ;
            const int v = k & 7, pos = (k >> 3) % WIN, split = (k >> 3) / WIN;
            const int vs = v ^ (((pos >> 1) & 1) << 2);
            *reinterpret_cast<floatx4 *>(xl + ((size_t)split * WIN + pos) * 64 + vs * 8) = val;
        }
    };

    int rid = row_begin, ch = 0;
    while (rid < row_end && !row_valid(rid)) ++rid;
    if (PREF && rid < row_end) {
#pragma unroll
        for (int q = 0; q < NV; ++q) pv[q] = load_vec(tid + q * 256, rid, ch);
    }
    const unsigned char *dz_h = reinterpret_cast<const unsigned char *>(dzl);
    const unsigned char *dz_l = dz_h + WF_CH * 128;
    const unsigned char *x_h = reinterpret_cast<const unsigned char *>(xl);
    const unsigned char *x_l = x_h + WIN * 128;
    // accumulators: acc[2k + j] = tap 7g + k (k < 6), co tile j;   acc[12] = tap 6, co tile g
    const TrLane la0 = tr_lane_offsets(0, lane), la1 = tr_lane_offsets(1, lane), lb = tr_lane_offsets(nt, lane);
    // the wave's taps start at row 7gT of the x window: fold that into the base pointer and rotate the swizzle
    // phase table by (7gT) & 3, so that every read below keeps a compile-time phase index and immediate offset
    TrLane lbg;
#pragma unroll
    for (int j = 0; j < 4; ++j) lbg.off[j] = g ? lb.off[(j + 7 * T) & 3] : lb.off[j];
    const int xg_off = g * 7 * T * 128;
    while (rid < row_end) {
        __syncthreads();                                // everyone is done reading the previous tiles
        if (PREF) {
#pragma unroll
            for (int q = 0; q < NV; ++q) store_vec(tid + q * 256, pv[q]);
        } else {
            for (int i = tid; i < NDZ + NX; i += 256) store_vec(i, load_vec(i, rid, ch));
        }
        int nrid = rid, nch = ch;
        next_iter(nrid, nch);
        if (PREF && nrid < row_end) {
#pragma unroll
            for (int q = 0; q < NV; ++q) pv[q] = load_vec(tid + q * 256, nrid, nch);   // in flight during the MFMAs
        }
        __syncthreads();
        // ---- 11 k-steps of 16 positions, 39 MFMAs each; fragments pipelined in two half-sets:
        //      S0 = taps k = 0..2, S1 = taps k = 3..5 + the middle tap
        half8 a0h, a0l, a1h, a1l, n0h, n0l, n1h, n1l, bh0[3], bl0[3], bh1[4], bl1[4];
        // (KS)*16 is a multiple of 4, so the swizzle phase of a read is its tap offset & 3: a compile-time constant
#define WF_LOAD_A(A0H, A0L, A1H, A1L, KS)                \
    A0H = tr_frag(dz_h + (KS) * 2048, la0.off[0], 0);    \
    A0L = tr_frag(dz_l + (KS) * 2048, la0.off[0], 0);    \
    A1H = tr_frag(dz_h + (KS) * 2048, la1.off[0], 0);    \
    A1L = tr_frag(dz_l + (KS) * 2048, la1.off[0], 0);
#define WF_LOAD_S0(KS)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
        bh0[i] = tr_frag(x_h + xg_off + (KS) * 2048, lbg.off[(i * T) & 3], i * T);              \
        bl0[i] = tr_frag(x_l + xg_off + (KS) * 2048, lbg.off[(i * T) & 3], i * T);              \
    }
#define WF_LOAD_S1(KS)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
        bh1[i] = tr_frag(x_h + xg_off + (KS) * 2048, lbg.off[((i + 3) * T) & 3], (i + 3) * T);  \
        bl1[i] = tr_frag(x_l + xg_off + (KS) * 2048, lbg.off[((i + 3) * T) & 3], (i + 3) * T);  \
    }                                                                                           \
    bh1[3] = tr_frag(x_h + (KS) * 2048, lb.off[(6 * T) & 3], 6 * T);                            \
    bl1[3] = tr_frag(x_l + (KS) * 2048, lb.off[(6 * T) & 3], 6 * T);
        WF_LOAD_A(a0h, a0l, a1h, a1l, 0)
        WF_LOAD_S0(0)
        WF_LOAD_S1(0)
#pragma unroll 1
        for (int ks = 0; ks < WF_KS; ++ks) {
            const int kn = ks + 1 < WF_KS ? ks + 1 : ks;       // last step reloads itself (discarded)
            __builtin_amdgcn_sched_barrier(0);
            // the three split products of one accumulator are issued six MFMAs apart
#pragma unroll
            for (int i = 0; i < 3; ++i) { acc[2 * i] = mfma16w(a0l, bh0[i], acc[2 * i]); acc[2 * i + 1] = mfma16w(a1l, bh0[i], acc[2 * i + 1]); }
#pragma unroll
            for (int i = 0; i < 3; ++i) { acc[2 * i] = mfma16w(a0h, bl0[i], acc[2 * i]); acc[2 * i + 1] = mfma16w(a1h, bl0[i], acc[2 * i + 1]); }
#pragma unroll
            for (int i = 0; i < 3; ++i) { acc[2 * i] = mfma16w(a0h, bh0[i], acc[2 * i]); acc[2 * i + 1] = mfma16w(a1h, bh0[i], acc[2 * i + 1]); }
            __builtin_amdgcn_sched_barrier(0);
            WF_LOAD_A(n0h, n0l, n1h, n1l, kn)
            WF_LOAD_S0(kn)
            __builtin_amdgcn_sched_barrier(0);
            {
                const half8 amh = g ? a1h : a0h, aml = g ? a1l : a0l;      // middle tap: co tile g
#pragma unroll
                for (int i = 0; i < 3; ++i) { acc[6 + 2 * i] = mfma16w(a0l, bh1[i], acc[6 + 2 * i]); acc[7 + 2 * i] = mfma16w(a1l, bh1[i], acc[7 + 2 * i]); }
                acc[12] = mfma16w(aml, bh1[3], acc[12]);
#pragma unroll
                for (int i = 0; i < 3; ++i) { acc[6 + 2 * i] = mfma16w(a0h, bl1[i], acc[6 + 2 * i]); acc[7 + 2 * i] = mfma16w(a1h, bl1[i], acc[7 + 2 * i]); }
                acc[12] = mfma16w(amh, bl1[3], acc[12]);
#pragma unroll
                for (int i = 0; i < 3; ++i) { acc[6 + 2 * i] = mfma16w(a0h, bh1[i], acc[6 + 2 * i]); acc[7 + 2 * i] = mfma16w(a1h, bh1[i], acc[7 + 2 * i]); }
                acc[12] = mfma16w(amh, bh1[3], acc[12]);
            }
            __builtin_amdgcn_sched_barrier(0);
            WF_LOAD_S1(kn)
            a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
        }
#undef WF_LOAD_A
#undef WF_LOAD_S0
#undef WF_LOAD_S1
        rid = nrid;
        ch = nch;
    }
    // partial tiles: part[slab][kh][kw][co][ci]   (D rows = co, columns = ci)
    const int l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < CV_KW; ++i) {
        const int kw = i == 12 ? 6 : 7 * g + (i >> 1), mt = i == 12 ? g : (i & 1);
        float *dst = a.part + ((((size_t)slab * CV_KH + kh) * CV_KW + kw) * 64) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(mt * 32 + mfma_row(r, lane)) * 64 + nt * 32 + l32] = acc[i][r];
    }
}

// dW[co][ci][kh][kw] = inv_scale * sum over slabs of part[slab][kh][kw][co][ci]   (fp64 accumulate)
__global__ __launch_bounds__(256) void wgrad_f16_reduce_kernel(const float *__restrict__ part, int n_slabs,
                                                               const float *__restrict__ scale, float *__restrict__ dW)
{
    const int total = CV_TAPS * 64 * 64;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    double s = 0.0;
    for (int k = 0; k < n_slabs; ++k) s += (double)part[(size_t)k * total + j];
    const int ci = j % 64, co = (j / 64) % 64, tap = j / (64 * 64);
    dW[((size_t)co * 64 + ci) * CV_TAPS + tap] = (float)(s * (double)scale[1]);
}

template <int T>
static int launch_wgrad_f16(const WgradF16Args &a, hipStream_t st)
{
    constexpr int WIN = WF_CH + 12 * T;
    const size_t lds = (size_t)(2 * WF_CH * 64 + 2 * WIN * 64) * sizeof(_Float16);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)wgrad_f16x3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return MX_ERR_LAUNCH;
        attr_done = true;
    }
    const int groups = (a.n_slabs + 7) / 8;
    hipLaunchKernelGGL((wgrad_f16x3_kernel<T>), dim3(groups * 8 * CV_KH), dim3(256), lds, st, a);
    return mx_launch_status();
}

// dz_hi/lo, x_hi/lo: (B,H,4,352,16) fp16 pairs from mx_conv_prep_dgrad_f16 / mx_conv_prep_fwd_f16 of the same block;
// scale: the {S_dz, 1/S_dz} pair; part: workspace of ceil(B*H/rows_per_slab)*65*64*64 floats; dW (64,64,5,13).
MX_EXPORT int mx_conv_block_wgrad_f16(const void *dz_hi, const void *dz_lo, const void *x_hi, const void *x_lo,
                                      const float *scale, int64_t B, int64_t H, int32_t dilation,
                                      int64_t rows_per_slab, float *part, float *dW, void *stream)
{
    if (!dz_hi || !dz_lo || !x_hi || !x_lo || !scale || !part || !dW || B <= 0 || H <= 0 || rows_per_slab <= 0)
        return MX_ERR_ARG;
    const int64_t n_slabs = (B * H + rows_per_slab - 1) / rows_per_slab;
    if (n_slabs > 1000000) return MX_ERR_UNSUPPORTED;
    WgradF16Args a{(const _Float16 *)dz_hi, (const _Float16 *)dz_lo, (const _Float16 *)x_hi, (const _Float16 *)x_lo, part,
                   (int)B, (int)H, (int)rows_per_slab, (int)n_slabs};
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dilation) {
    case 1: rc = launch_wgrad_f16<1>(a, st); break;
    case 2: rc = launch_wgrad_f16<2>(a, st); break;
    case 4: rc = launch_wgrad_f16<4>(a, st); break;
    case 8: rc = launch_wgrad_f16<8>(a, st); break;
    case 16: rc = launch_wgrad_f16<16>(a, st); break;
    default: return MX_ERR_UNSUPPORTED;
    }
    if (rc != MX_OK) return rc;
    const int total = CV_TAPS * 64 * 64;
    hipLaunchKernelGGL(wgrad_f16_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, (int)n_slabs, scale, dW);
    return mx_launch_status();
}
