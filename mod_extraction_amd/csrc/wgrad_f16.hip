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
    const unsigned char *ptr = img_bytes + lane_off + r0 * 128;
    const short4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)ptr);
    const short4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)(ptr + 512));
    // pure register naming: the two 64-bit results are the low and high half of the fragment
    typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
    const short8v both = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(half8, both);
}

template <int T>
__global__ __launch_bounds__(256, 1) void wgrad_f16x3_kernel(WgradF16Args a)
{
    constexpr int WIN = WF_CH + 12 * T;                 // staged x positions per chunk (origin at w0 - 6T)
    // staging map: thread = (position pos0 + 32 q, 16-byte vector v of the position's 8); every address below is a
    // per-thread constant + a compile-time multiple of q, and the swizzle bit of a position is that of pos0
    constexpr int QDZ = (WF_CH + 31) / 32, QX = (WIN + 31) / 32;       // iterations per split
    constexpr bool PREF = true;                         // the next chunk travels through registers during the MFMAs
    constexpr int NV = PREF ? 2 * (QDZ + QX) : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *dzl = reinterpret_cast<_Float16 *>(smem);               // [split][WF_CH][64]
    _Float16 *xl = dzl + 2 * WF_CH * 64;                               // [split][WIN][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform address parts stay off the vector unit
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
    const int sv = tid & 7, pos0 = tid >> 3;
    const int g_thr = ((sv >> 1) * CV_PITCH + pos0) * 16 + (sv & 1) * 8;                 // halfs, within an operand row
    const int l_thr = pos0 * 128 + ((sv ^ (((pos0 >> 1) & 1) << 2)) * 16);                 // bytes, within an LDS image
    // vector k of a chunk, k = (kind * 2 + split) * Q + q with kind 0 = dz, 1 = x window
    auto load_vec = [&](int k, int rid, int ch) -> floatx4 {
        const int b = rid / a.H, h = rid - b * a.H, w0 = ch * WF_CH;
        floatx4 z = {0.f, 0.f, 0.f, 0.f};
        if (k < 2 * QDZ) {
            const int split = k / QDZ, q = k - split * QDZ;
            if (pos0 + 32 * q < WF_CH) {
                const _Float16 *row = (split ? a.dz_lo : a.dz_hi) + ((size_t)b * a.H + h) * (4 * CV_PITCH * 16) + w0 * 16;
                return *reinterpret_cast<const floatx4 *>(row + g_thr + q * (32 * 16));
            }
        } else {
            const int kk = k - 2 * QDZ, split = kk / QX, q = kk - split * QX;
            const int w = w0 - 6 * T + pos0 + 32 * q;
            if (pos0 + 32 * q < WIN && w >= 0 && w < CV_PITCH) {
                const _Float16 *row = (split ? a.x_lo : a.x_hi) + ((size_t)b * a.H + (h + kh - 2)) * (4 * CV_PITCH * 16) +
                                      (w0 - 6 * T) * 16;
                return *reinterpret_cast<const floatx4 *>(row + g_thr + q * (32 * 16));
            }
        }
        return z;
    };
    auto store_vec = [&](int k, floatx4 val) {
        if (k < 2 * QDZ) {
            const int split = k / QDZ, q = k - split * QDZ;
            if (pos0 + 32 * q < WF_CH)
                *reinterpret_cast<floatx4 *>(reinterpret_cast<unsigned char *>(dzl) + split * (WF_CH * 128) + q * 4096 + l_thr) = val;
        } else {
            const int kk = k - 2 * QDZ, split = kk / QX, q = kk - split * QX;
            if (pos0 + 32 * q < WIN)
                *reinterpret_cast<floatx4 *>(reinterpret_cast<unsigned char *>(xl) + split * (WIN * 128) + q * 4096 + l_thr) = val;
        }
    };

    int rid = row_begin, ch = 0;
    while (rid < row_end && !row_valid(rid)) ++rid;
    if (PREF && rid < row_end) {
#pragma unroll
        for (int q = 0; q < NV; ++q) pv[q] = load_vec(q, rid, ch);
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
    const int lag_off = g ? la1.off[0] : la0.off[0];
    while (rid < row_end) {
        __syncthreads();                                // everyone is done reading the previous tiles
        if (PREF) {
#pragma unroll
            for (int q = 0; q < NV; ++q) store_vec(q, pv[q]);
        } else if (!PREF) {
#pragma unroll 4
            for (int k = 0; k < 2 * (QDZ + QX); ++k) store_vec(k, load_vec(k, rid, ch));
        }
        int nrid = rid, nch = ch;
        next_iter(nrid, nch);
        if (PREF && nrid < row_end) {
#pragma unroll
            for (int q = 0; q < NV; ++q) pv[q] = load_vec(q, nrid, nch);   // in flight during the MFMAs
        }
        __syncthreads();
        // ---- 11 k-steps of 16 positions, 39 MFMAs each, in three phases of 13 (one per split product, every
        //      accumulator once per phase).  Fragment lifetimes are staggered so that ONE register set suffices and
        //      every transposed read is issued a phase (>= 13 MFMAs) before its first use, one read per MFMA:
        //        phase 1  lo(dz) * hi(x)   while lo(x) of this k-step arrives        (14 reads)
        //        phase 2  hi(dz) * hi(x)   while the next k-step's dz arrives        (12 reads; hi(dz) double buffered)
        //        phase 3  hi(dz) * lo(x)   while the next k-step's hi(x) arrives     (14 reads)
        //      The wave issues in order and the LDS pipe serves four waves: reads clustered between MFMA groups
        //      would idle the matrix pipe.
        // A fragments: [co tile 0, co tile 1, co tile g (the middle tap's; fetched separately so that no register
        // select -- which the optimizer would turn into a dynamically indexed private array -- is needed)]
        half8 AL[3], AH[2][3], BH[7], BL[7];      // AH double buffered by k-step parity; B: taps 7g..7g+5 + the middle tap
        // (KS)*16 is a multiple of 4, so the swizzle phase of a read is its tap offset & 3: a compile-time constant
        auto rd_a = [&](const unsigned char *img, int ks, int j) {
            return tr_frag(img + ks * 2048, j == 0 ? la0.off[0] : (j == 1 ? la1.off[0] : lag_off), 0);
        };
        auto rd_b = [&](const unsigned char *img, int ks, int tap) {
            return tap < 6 ? tr_frag(img + ks * 2048 + xg_off, lbg.off[(tap * T) & 3], tap * T)
                           : tr_frag(img + ks * 2048, lb.off[(6 * T) & 3], 6 * T);
        };
#define WF_PIN(N_PAIR, N_MFMA_TAIL, N_DS_TAIL)                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < (N_PAIR); ++q_) {                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                              \
    if ((N_MFMA_TAIL) > 0) __builtin_amdgcn_sched_group_barrier(0x008, (N_MFMA_TAIL), 0);          \
    if ((N_DS_TAIL) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (N_DS_TAIL), 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) { AL[j] = rd_a(dz_l, 0, j); AH[0][j] = rd_a(dz_h, 0, j); }
#pragma unroll
        for (int t = 0; t < 7; ++t) BH[t] = rd_b(x_h, 0, t);
#pragma unroll
        for (int ks = 0; ks < WF_KS; ++ks) {
            const int p = ks & 1;
            const bool last = ks + 1 == WF_KS;
            __builtin_amdgcn_sched_barrier(0);
            {   // phase 1
#pragma unroll
                for (int u = 0; u < 13; ++u) acc[u] = mfma16w(AL[u < 12 ? (u & 1) : 2], BH[u < 12 ? (u >> 1) : 6], acc[u]);
#pragma unroll
                for (int t = 0; t < 7; ++t) BL[t] = rd_b(x_l, ks, t);
                WF_PIN(13, 0, 1)
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // phase 2
#pragma unroll
                for (int u = 0; u < 13; ++u) acc[u] = mfma16w(AH[p][u < 12 ? (u & 1) : 2], BH[u < 12 ? (u >> 1) : 6], acc[u]);
                if (!last) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { AL[j] = rd_a(dz_l, ks + 1, j); AH[p ^ 1][j] = rd_a(dz_h, ks + 1, j); }
                    WF_PIN(12, 1, 0)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // phase 3
#pragma unroll
                for (int u = 0; u < 13; ++u) acc[u] = mfma16w(AH[p][u < 12 ? (u & 1) : 2], BL[u < 12 ? (u >> 1) : 6], acc[u]);
                if (!last) {
#pragma unroll
                    for (int t = 0; t < 7; ++t) BH[t] = rd_b(x_h, ks + 1, t);
                    WF_PIN(13, 0, 1)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WF_PIN
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
    // 256 threads = 64 groups of 4 consecutive outputs x 4 quarters of the slab range: 16-byte loads, four of them in
    // flight per thread, quarters combined through LDS in a fixed order (one output per thread over all slabs was a
    // single 4-byte load stream per lane: 1.5 TB/s)
    __shared__ double sh[4][64][4];
    const int total = CV_TAPS * 64 * 64;                      // a multiple of 256
    const int jq = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int j = (blockIdx.x * 64 + jq) * 4;
    const int k0 = (int)((long long)n_slabs * kq / 4), k1 = (int)((long long)n_slabs * (kq + 1) / 4);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const float *src = part + (size_t)k0 * total + j;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        floatx4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const floatx4 *>(src + (size_t)u * total);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (double)v[u][e];
        src += (size_t)4 * total;
    }
    for (; k < k1; ++k) {
        const floatx4 v = *reinterpret_cast<const floatx4 *>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
        src += total;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[kq][jq][e] = s[e];
    __syncthreads();
    {
        const int e = kq, jj = j + e;                         // thread (jq, kq) finishes output 4 (block group) + kq
        const double t = ((sh[0][jq][e] + sh[1][jq][e]) + sh[2][jq][e]) + sh[3][jq][e];
        const int ci = jj % 64, co = (jj / 64) % 64, tap = jj / (64 * 64);
        dW[((size_t)co * 64 + ci) * CV_TAPS + tap] = (float)(t * (double)scale[1]);
    }
}

template <int T>
static int launch_wgrad_f16(const WgradF16Args &a, hipStream_t st)
{
    constexpr int WIN = WF_CH + 12 * T;
    const size_t lds = (size_t)(2 * WF_CH * 64 + 2 * WIN * 64) * sizeof(_Float16);
    static MxLdsLatch latch = {};                             // per device (common.h)
    if (mx_set_dyn_lds(latch, (const void *)wgrad_f16x3_kernel<T>, lds) != MX_OK) return MX_ERR_LAUNCH;
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
