// dgrad_sp_f16.hip -- data gradient of the 64->64 channel Spectral2DCNN convolutions on the SPARSE fp16 matrix
// instruction v_smfmac_f32_32x32x32_f16, with fp32-equivalent accuracy ("f16x3", see conv_f16.hip).
// Reference semantics: torch.nn.Conv2d backward w.r.t. its input, behind MaxPool2d((2,1)) (mod_extraction/models.py:187-188):
//   dxhat[ci][h][w] = sum over (co, kh, kw) of  dz[co][h - kh + 2][w - (kw - 6) T] * W[co][ci][kh][kw]
// with dz = the pooled gradient G routed to the row of each pooling pair that won the max (see wgrad_sp_f16.hip).
//
// The sparse instruction wants the sparse matrix as its A operand, so the tile is computed transposed:
//   D^T[position][ci] += A[position][k] * B[k][ci],    k = 2 * co + row parity of the pooling pair   (K = 32 logical)
//   A  = the gradient patch: compressed = G at pooled resolution (16 output channels of one position = 32 bytes of a
//        channels-last image), index bits = the pooling argmax; lane l: row (position) l & 31, half hh = l >> 5 holds
//        co 4hh..4hh+3 and 8+4hh..8+4hh+3 of the block (two 8-byte LDS reads; layout derived in tools/probe)
//   B  = the weights of the TWO kernel rows that the pair's rows meet for this output row (kh = 4 + r - 2m - parity for
//        output row h0 + r and pair m = -1, 0, +1 around it; rows outside 0..4 are zero), packed per (channel block, pair,
//        output row, tap, ci tile) in fragment order and loaded straight from global memory (L2-resident, 32 bytes per
//        lane) -- they would not fit the LDS beside the patch for both output rows.
// K loop = 4 channel blocks x 3 pooling pairs = 12 stages of 13 taps (the dense kernel: 20 stages), 33 sparse MFMAs per
// tap and wave.  Round 5: a STAGE of the staging pipeline covers DS_NCB = 2 channel blocks (26 taps) for dilations <= 8 -- the
// commit to LDS, the workgroup barrier and the first tap's exposed reads (2.2 k of a 13-tap stage's 22.9 k cycles, s_memtime
// stamps of round 4) are paid 6 times per workgroup instead of 12; the two 31 KB images of a stage sit side by side in each of
// the two buffers (124-143 KB of LDS; dilation 16 would need 174 KB and keeps one block per stage).  Workgroup = (clip, output row pair), waves = (output row, position half) with the accumulator split of
// conv_f16.hip; the patch (31 KB per stage) is register-staged into double-buffered LDS.  The epilogue transposes the
// [position][ci] tiles through wave-private LDS and writes dxhat (B, 64, H, 352) rows coalesced.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
#define DS_WSCALE 256.0f
#define DS_ROWB 40                      // bytes per patch row: 16 co x 2 B + 8 pad (conflict-free 8-byte reads by 32 rows)

struct DgradSpArgs {
    const _Float16 *g_hi, *g_lo;        // (B, Hp, 4, 352, 16): fp16 pair of G * S, pooled resolution, channels last
    const unsigned *g_idx;              // (B, Hp, 4, 352): index words of lane half 0 (low 16 bits) and 1 (high)
    const _Float16 *w_hi, *w_lo;        // [4 cb][3 m][2 r][13 kwf][2 ci tile][64 lanes][16]
    const float *scale;                 // {S, 1/S}
    float *out;                         // (B, 64, H, 352)
    int H, Wv;
    const _Float16 *x_hi, *x_lo;        // LN only: (B, H, 4, 352, 16) operand pair of the block's forward pass (= xhat)
    float *ln_part;                     // LN only: (B, 64, H, 2, 2) partial sums {dxhat, dxhat * xhat} per (row, position half)
    unsigned *gx_bits;                  // LN only, optional: bit patterns of max |dxhat| and max |xhat| over the launch (atomicMax;
                                        // zeroed by the caller) -- the inputs of mx_ln_bwd_finish's bound on max |G|
};

__device__ __forceinline__ floatx16 ds_smfmac(half8 a, half16 b, floatx16 c, int idx)
{
    return __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c, idx, 0, 0);
}
__device__ __forceinline__ half8 ds_a_frag(const unsigned char *ptr)
{
    const half4 lo = *reinterpret_cast<const half4 *>(ptr);
    const half4 hi = *reinterpret_cast<const half4 *>(ptr + 16);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ half16 ds_w_frag(const _Float16 *p)
{
    const half8 lo = *reinterpret_cast<const half8 *>(p);
    const half8 hi = *reinterpret_cast<const half8 *>(p + 8);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

#ifndef DS_NCB
#define DS_NCB 2                        // channel blocks per staging stage for dilations <= 8 (1: the round-4 pipeline; same-box A/B knob)
#endif
template <int T> struct DsGeom {
    static constexpr int NCB = T <= 8 ? DS_NCB : 1;
    static constexpr int PWP = CV_PITCH + 12 * T;
    static constexpr size_t IMG = ((2 * (size_t)PWP * DS_ROWB + PWP * 4 + 15) / 16) * 16;   // one channel block: hi, lo, index words
    static constexpr size_t BUF = NCB * IMG;
};

template <int T, bool LN>
__global__ __launch_bounds__(256, 1) void dgrad_sp_f16x3_kernel(DgradSpArgs a)
{
    constexpr int NCB = DsGeom<T>::NCB;                     // channel blocks per stage
    constexpr int PWP = CV_PITCH + 12 * T;                  // patch rows (position w = row - 6T)
    constexpr int PA_SPLIT = PWP * DS_ROWB;                 // bytes per split
    constexpr int IMG_BYTES = (int)DsGeom<T>::IMG;          // one channel block's image
    constexpr int BUF_BYTES = (int)DsGeom<T>::BUF;
    constexpr int QP = (PWP + 127) / 128;                   // patch iterations: thread = (position (tid >> 1) + 128 q, 16-byte half)
    constexpr int QI = (PWP + 255) / 256;
    constexpr int N_STAGE = 12 / NCB;                       // stage st = cbg * 3 + m: channel blocks cbg * NCB .. + NCB - 1, pooled row hp + m - 1
    constexpr int N_TAPS = CV_KW * NCB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // (wave index pinned to a scalar register: row / c are then wave-uniform for the compiler and the weight-fragment addresses
    // -- base + a per-tap constant beyond the loads' immediate range -- become scalar adds instead of a 64-bit vector add per load)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = wave >> 1, c = wave & 1, hh = lane >> 5, l32 = lane & 31;
    int tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const int n_tiles = gridDim.x * gridDim.y;
        if ((n_tiles & 7) == 0) tile_id = (tile_id & 7) * (n_tiles >> 3) + (tile_id >> 3);
    }
    const int b = tile_id / gridDim.x, hp = tile_id - b * gridDim.x, h0 = hp * 2;
    const int H = a.H, Hp = H >> 1;

    floatx16 acc[CV_WT];
#pragma unroll
    for (int i = 0; i < CV_WT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // ---- patch staging (registers -> LDS), stage st = cbg * 3 + m: pooled row hp + m - 1, channel blocks cbg * NCB + s
    floatx4 pv[NCB][2 * QP];
    unsigned pidx[NCB][QI];
    const int spos = tid >> 1, spart = tid & 1;
    auto stage_row = [&](int st) { return hp + (st % 3) - 1; };
    auto issue = [&](int st) {
        const int hq = stage_row(st);
        const int hq_eff = hq < 0 ? 0 : (hq >= Hp ? Hp - 1 : hq);       // out-of-image pairs are skipped by the caller
#pragma unroll
        for (int s = 0; s < NCB; ++s) {
            const int cb = (st / 3) * NCB + s;
            const size_t rbase = (((size_t)b * Hp + hq_eff) * 4 + cb) * CV_PITCH;
#pragma unroll
            for (int k = 0; k < 2 * QP; ++k) {
                const int split = k / QP, q = k - split * QP;
                const int pos = spos + 128 * q, w = pos - 6 * T;
                const int w_eff = (w >= 0 && w < CV_PITCH) ? w : CV_PITCH - 1;      // column 351 is zero (Wv <= 351)
                pv[s][k] = *reinterpret_cast<const floatx4 *>((split ? a.g_lo : a.g_hi) + (rbase + w_eff) * 16 + spart * 8);
            }
#pragma unroll
            for (int q = 0; q < QI; ++q) {
                const int pos = tid + 256 * q, w = pos - 6 * T;
                const int w_eff = (w >= 0 && w < CV_PITCH) ? w : CV_PITCH - 1;
                pidx[s][q] = a.g_idx[rbase + w_eff];
            }
        }
    };
    auto commit = [&](unsigned char *buf0) {
        typedef float floatx2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int s = 0; s < NCB; ++s) {
            unsigned char *buf = buf0 + s * IMG_BYTES;
#pragma unroll
            for (int k = 0; k < 2 * QP; ++k) {
                const int split = k / QP, q = k - split * QP;
                const int pos = spos + 128 * q;
                if (pos < PWP) {
                    unsigned char *dst = buf + split * PA_SPLIT + pos * DS_ROWB + spart * 16;
                    reinterpret_cast<floatx2 *>(dst)[0] = floatx2{pv[s][k][0], pv[s][k][1]};
                    reinterpret_cast<floatx2 *>(dst)[1] = floatx2{pv[s][k][2], pv[s][k][3]};
                }
            }
#pragma unroll
            for (int q = 0; q < QI; ++q) {
                const int pos = tid + 256 * q;
                if (pos < PWP) reinterpret_cast<unsigned *>(buf + 2 * PA_SPLIT)[pos] = pidx[s][q];
            }
        }
    };
    auto stage_live = [&](int st) { const int hq = stage_row(st); return hq >= 0 && hq < Hp; };

    // weights of (stage, tap kk of its NCB x 13, this wave's output row): fragment f = 0: ci tile c, f = 1: ci tile c ^ 1
    // (packed per (channel block, pair): index ws = cb * 3 + m)
    auto w_ptr = [&](const _Float16 *base, int st, int kk, int f) {
        const int tile = f ? (c ^ 1) : c;
        const int ws = ((st / 3) * NCB + kk / CV_KW) * 3 + st % 3;
        int kwf = kk % CV_KW;
#ifdef DS_ABL_WFIX        // ablation (wrong results): every tap reads tap 0's fragments -> the weight stream stays in the CU's L1
        kwf = 0;
#endif
        // (wave-uniform part) + (32-bit lane offset): the loads take the scalar-base form, no vector address arithmetic
        const _Float16 *ub = base + (((size_t)(ws * 2 + row) * CV_KW + kwf) * 2 + tile) * (64 * 16);
        return ub + (unsigned)(lane * 16);
    };

    int st = 0;
    while (st < N_STAGE && !stage_live(st)) ++st;
    if (st < N_STAGE) issue(st);
    int bufsel = 0;
    // weight fragments of the NEXT stage's first tap: fetched from global memory during the current stage's last tap
    // (they do not depend on the barrier; an L2 round trip at the head of every stage was 15 % of the kernel)
    half16 WNH[2], WNL[2];
    if (st < N_STAGE) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { WNH[j] = ds_w_frag(w_ptr(a.w_hi, st, 0, j)); WNL[j] = ds_w_frag(w_ptr(a.w_lo, st, 0, j)); }
    }
    while (st < N_STAGE) {
        unsigned char *const buf = smem + bufsel * BUF_BYTES;
        commit(buf);                                    // the other buffer may still be read by slower waves
        int nst = st + 1;
        while (nst < N_STAGE && !stage_live(nst)) ++nst;
        if (nst < N_STAGE) issue(nst);                  // in flight during this stage's MFMAs
        __syncthreads();
        // ---- NCB x 13 taps: A fragments (6 position tiles x hi/lo + index words) one tap ahead, weight fragments one tap ahead
        // position tile t < 5: tile c*6 + t; t = 5: the middle tile 5
        const int a_lane = (l32 + c * 6 * 32) * DS_ROWB + hh * 8;
        const int m_lane = (l32 + 5 * 32) * DS_ROWB + hh * 8;
        const int ai_lane = (l32 + c * 6 * 32) * 4 + hh * 2, mi_lane = (l32 + 5 * 32) * 4 + hh * 2;
        half8 AH[2][6], AL[2][6];
        int IX[2][6];
        half16 WH[2][2], WL[2][2];
        auto rd_tap = [&](int f, int kk, int t0, int t1) {          // tap kk of the stage: image kk / 13, kernel column kk % 13
            const unsigned char *img = buf + (kk / CV_KW) * IMG_BYTES;
            const unsigned char *pa_h = img, *pa_l = img + PA_SPLIT, *pidx_img = img + 2 * PA_SPLIT;
            const int kwf = kk % CV_KW;
#pragma unroll
            for (int t = t0; t < t1; ++t) {
                const int off = (t < 5 ? a_lane + t * 32 * DS_ROWB : m_lane) + kwf * T * DS_ROWB;
                const int ioff = (t < 5 ? ai_lane + t * 32 * 4 : mi_lane) + kwf * T * 4;
                AH[f][t] = ds_a_frag(pa_h + off);
                AL[f][t] = ds_a_frag(pa_l + off);
                IX[f][t] = (int)*reinterpret_cast<const unsigned short *>(pidx_img + ioff);
            }
        };
        auto ld_w = [&](int f, int kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                WH[f][j] = ds_w_frag(w_ptr(a.w_hi, st, kk, j));
                WL[f][j] = ds_w_frag(w_ptr(a.w_lo, st, kk, j));
            }
        };
        rd_tap(0, 0, 0, 6);
#pragma unroll
        for (int j = 0; j < 2; ++j) { WH[0][j] = WNH[j]; WL[0][j] = WNL[j]; }
#pragma unroll
        for (int kwf = 0; kwf < N_TAPS; ++kwf) {                     // (kwf counts the stage's taps: 13 per channel block)
            const int f = kwf & 1;
            const bool more_taps = kwf + 1 < N_TAPS;
            __builtin_amdgcn_sched_barrier(0);
            // accumulator u = 2t + j (t < 5): position tile c*6 + t, weight fragment j; u = 10: middle tile, fragment 0
            // Segment 1 (11 MFMAs): the next tap's 8 weight loads are fenced into it -- left to itself the scheduler
            // sinks them to the END of the tap, reuses this tap's registers and waits an L2 round trip at the head of
            // every tap -- together with the first half of the next tap's LDS reads, one per MFMA.
#pragma unroll
            for (int u = 0; u < CV_WT; ++u) {
                const int t = u < 10 ? (u >> 1) : 5, j = u < 10 ? (u & 1) : 0;
                acc[u] = ds_smfmac(AL[f][t], WH[f][j], acc[u], IX[f][t]);
            }
            if (more_taps) {
                ld_w(f ^ 1, kwf + 1);
                rd_tap(f ^ 1, kwf + 1, 0, 3);
#pragma unroll
                for (int q_ = 0; q_ < 9; ++q_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else if (nst < N_STAGE) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { WNH[j] = ds_w_frag(w_ptr(a.w_hi, nst, 0, j)); WNL[j] = ds_w_frag(w_ptr(a.w_lo, nst, 0, j)); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // Segment 2 (22 MFMAs) with the other half of the reads
#pragma unroll
            for (int u = 0; u < CV_WT; ++u) {
                const int t = u < 10 ? (u >> 1) : 5, j = u < 10 ? (u & 1) : 0;
                acc[u] = ds_smfmac(AH[f][t], WL[f][j], acc[u], IX[f][t]);
            }
#pragma unroll
            for (int u = 0; u < CV_WT; ++u) {
                const int t = u < 10 ? (u >> 1) : 5, j = u < 10 ? (u & 1) : 0;
                acc[u] = ds_smfmac(AH[f][t], WH[f][j], acc[u], IX[f][t]);
            }
            if (more_taps) {
                rd_tap(f ^ 1, kwf + 1, 3, 6);
#pragma unroll
                for (int q_ = 0; q_ < 9; ++q_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        st = nst;
        bufsel ^= 1;
    }

    // ---- epilogue: [position][ci] tiles -> dxhat rows, transposed through wave-private LDS (32 ci x 33 floats).
    // LN: the LayerNorm backward that consumes dxhat needs sum(dxhat) and sum(dxhat * xhat) per (clip, channel) plane
    // before it can touch an element -- a whole extra sweep over two tensors if it computes them itself.  Here every
    // value is in a register next to its position's 16 channels of xhat (one 32-byte vector of each operand half), so
    // the wave accumulates both sums per channel and writes one partial per (plane, row, position half).
    __syncthreads();                                            // the patch buffers are dead
    // two wave-private transposition images (32 ci x 33 floats each): tile u + 1 is written while tile u's rows are in flight
    float *scr = reinterpret_cast<float *>(smem) + wave * (2 * 32 * 33);
    const float inv = a.scale[1] * (1.0f / DS_WSCALE);
    const int h = h0 + row;
    float s1a[2][16], s2a[2][16];                               // [weight fragment j: ci tile j ^ c][channel hh*16 + i]
    float mxd = 0.0f, mxx = 0.0f;                               // running max |dxhat|, max |xhat| of this lane
    const unsigned out_vo = (unsigned)(hh * 16 * H * CV_PITCH + l32);   // lane part of a dxhat element offset
    // the wave's xhat vectors: a ring of XSLOTS tiles, the first XSLOTS requested up front, tile u + XSLOTS as soon as tile u
    // is done (its data is then XSLOTS - 1 tiles of work away) -- one exposed memory round trip per workgroup instead of
    // eleven, and 16 XSLOTS registers instead of 176 (all eleven tiles at once spilled 56 registers to scratch memory
    // beside the 176 accumulators)
#ifndef DS_XSLOTS
#define DS_XSLOTS 3
#endif
    constexpr int XSLOTS = DS_XSLOTS;
    half8 xh[LN ? XSLOTS : 1][2], xl[LN ? XSLOTS : 1][2];
    auto load_x = [&](int u) {
        const int ptile = u < 10 ? c * 6 + (u >> 1) : 5, cit = (u < 10 ? (u & 1) : 0) ^ c;
        const size_t xo = ((((size_t)b * H + h) * 4 + cit * 2 + hh) * CV_PITCH + ptile * 32 + l32) * 16;
        xh[u % XSLOTS][0] = *reinterpret_cast<const half8 *>(a.x_hi + xo);
        xh[u % XSLOTS][1] = *reinterpret_cast<const half8 *>(a.x_hi + xo + 8);
        xl[u % XSLOTS][0] = *reinterpret_cast<const half8 *>(a.x_lo + xo);
        xl[u % XSLOTS][1] = *reinterpret_cast<const half8 *>(a.x_lo + xo + 8);
    };
    if (LN) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1a[j][i] = 0.0f; s2a[j][i] = 0.0f; }
#pragma unroll
        for (int u = 0; u < XSLOTS; ++u) load_x(u);
    }
    // A wave's LDS operations execute in order, so its reads see its own earlier writes without a wait; the reads of tile u
    // are issued BEFORE the writes of tile u + 1 (other image), whose 16 multiplications + writes hide their latency.
    auto put_tile = [&](int u) {
        float *d = scr + (u & 1) * (32 * 33);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[l32 * 33 + mfma_row(r, lane)] = acc[u][r] * inv;      // [ci local][position local]
    };
    put_tile(0);
#pragma unroll
    for (int u = 0; u < CV_WT; ++u) {
        const int ptile = u < 10 ? c * 6 + (u >> 1) : 5;
        const int jf = u < 10 ? (u & 1) : 0;
        const int cit = jf ^ c;
        const int w = ptile * 32 + l32;
        float tv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tv[i] = (scr + (u & 1) * (32 * 33))[(hh * 16 + i) * 33 + l32];
        asm volatile("" ::: "memory");
        if (u + 1 < CV_WT) put_tile(u + 1);
        // pad columns exist in at most one column tile: the masks sit behind ONE wave-uniform test per tile (as selects per
        // element they were an exec-mask save / restore around every LDS read)
        const bool valid = w < a.Wv;
        auto rows16 = [&](auto partial_tag) {
            constexpr bool PARTIAL = decltype(partial_tag)::value;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = tv[i];
                if (PARTIAL) v = valid ? v : 0.0f;
                // (wave-uniform row base: scalar registers) + (32-bit lane offset): no 64-bit vector address arithmetic per store
                (a.out + ((((size_t)b * CV_CO + cit * 32 + i) * H + h) * CV_PITCH + ptile * 32))[out_vo] = v;
                if (LN) {
                    const float xv = (float)xh[u % XSLOTS][i >> 3][i & 7] + (float)xl[u % XSLOTS][i >> 3][i & 7];
                    s1a[jf][i] += v;
                    s2a[jf][i] += v * xv;
                    mxd = fmaxf(mxd, fabsf(v));
                    mxx = fmaxf(mxx, (!PARTIAL || valid) ? fabsf(xv) : 0.0f);
                }
            }
        };
        if (ptile * 32 + 32 > a.Wv) rows16(std::true_type{});
        else rows16(std::false_type{});
        if (LN && u + XSLOTS < CV_WT) load_x(u + XSLOTS);
    }
    if (LN && a.gx_bits) {                                      // (order of non-negative floats = order of their bits)
        mxd = wave_max_f32(mxd);
        mxx = wave_max_f32(mxx);
        if (lane == 0) {
            atomicMax(a.gx_bits, __float_as_uint(mxd));
            atomicMax(a.gx_bits + 1, __float_as_uint(mxx));
        }
    }
    if (LN) {
        // sum over the 32 positions held by the lanes of each half (same hh), in a fixed order, through wave-private
        // LDS (the wave's two transposition images, free now): lane (hh, l32) writes its 32 partials [q = j*16 + i] as
        // column l32, then sums row q = l32
        float *red = scr + hh * (32 * 33);
        float tot[2];
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
#pragma unroll
            for (int q = 0; q < 32; ++q) red[q * 33 + l32] = kind ? s2a[q >> 4][q & 15] : s1a[q >> 4][q & 15];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += red[l32 * 33 + k];
            tot[kind] = t;
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        const int j = l32 >> 4, i = l32 & 15;
        const int ci = (j ^ c) * 32 + hh * 16 + i;
        typedef float floatx2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<floatx2 *>(a.ln_part + ((((size_t)b * CV_CO + ci) * H + h) * 2 + c) * 2) = floatx2{tot[0], tot[1]};
    }
}

// torch (64 co, 64 ci, 5, 13) fp32 -> [cb][m][r][kwf][ci tile][lane][16] fp16 pairs of W * 256: lane l = (column n = l & 31,
// k half s = l >> 5), element j: k = 16 s + j -> co = 16 cb + (k >> 1), parity = k & 1, kernel row kh = 4 + r - 2m - parity
// (zero outside 0..4), kernel column kw = 12 - kwf, ci = 32 tile + n
__global__ void pack_weights_sp_f16_kernel(const float *__restrict__ W, _Float16 *__restrict__ w_hi, _Float16 *__restrict__ w_lo)
{
    const int total = 4 * 3 * 2 * CV_KW * 2 * 64 * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 15, l = (i >> 4) & 63, tile = (i >> 10) & 1;
        int rest = i >> 11;
        const int kwf = rest % CV_KW; rest /= CV_KW;
        const int r = rest & 1; rest >>= 1;
        const int m = rest % 3, cb = rest / 3;
        const int k = 16 * (l >> 5) + j, co = 16 * cb + (k >> 1), par = k & 1, kh = 4 + r - 2 * m - par;
        const int ci = 32 * tile + (l & 31), kw = 12 - kwf;
        float v = 0.0f;
        if (kh >= 0 && kh < CV_KH) v = W[(((size_t)co * 64 + ci) * CV_KH + kh) * CV_KW + kw] * DS_WSCALE;
        const _Float16 hv = (_Float16)v;
        w_hi[i] = hv;
        w_lo[i] = (_Float16)(v - (float)hv);
    }
}

// G, amax (B,64,Hp,352) -> channels-last pooled operand: g_hi, g_lo (B,Hp,4,352,16) = split of G * S (columns >= Wv: 0)
// and g_idx (B,Hp,4,352) uint32: low / high 16 bits = index word of lane half 0 / 1, element j of half hh = output
// channel 4hh + j (j < 4) or 8 + 4hh + j - 4 of the block, field = 2 (j & 1) + argmax.
// Workgroup = (64-position tile, pooled row, clip): per channel it reads 256 B of G and 64 B of argmax and writes 128 B of
// each planar half -- whole cache lines (a 32-position tile moved half lines: 3.7 TB/s); the last of the six tiles of a
// row is half empty (352 = 5.5 x 64).
#define GP_TW 64
__global__ __launch_bounds__(256) void gpool_cl_prep_kernel(const float *__restrict__ G, const unsigned char *__restrict__ amax,
                                                            const float *__restrict__ scale, int Hp, int Wv,
                                                            _Float16 *__restrict__ g_hi, _Float16 *__restrict__ g_lo,
                                                            unsigned *__restrict__ g_idx, unsigned char *__restrict__ gidx)
{
    __shared__ float tile[64][GP_TW + 1];
    __shared__ unsigned char tam[64][GP_TW + 4];
    __shared__ __attribute__((aligned(16))) unsigned char idxb[64][16];     // the tile's 16 planar index bytes per channel
    const int wt = blockIdx.x, hp = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const float S = scale[0];
    constexpr int NIT = 64 * (GP_TW / 4) / 256;                 // (channel, 4 positions) items per thread
    floatx4 gv[NIT];
    uchar4 av[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {                             // all loads first: NIT x (16 + 4) bytes in flight per thread
        const int i = tid + 256 * k, ch = i / (GP_TW / 4), c4 = i % (GP_TW / 4), w0 = wt * GP_TW + c4 * 4;
        const size_t off = (((size_t)b * 64 + ch) * Hp + hp) * CV_PITCH + (w0 < CV_PITCH ? w0 : 0);
        gv[k] = *reinterpret_cast<const floatx4 *>(G + off);
        av[k] = *reinterpret_cast<const uchar4 *>(amax + off);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k, ch = i / (GP_TW / 4), c4 = i % (GP_TW / 4), w0 = wt * GP_TW + c4 * 4;
        const floatx4 v = gv[k];
        const uchar4 am = av[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ch][c4 * 4 + e] = (w0 + e < Wv) ? v[e] * S : 0.0f;
        tam[ch][c4 * 4 + 0] = am.x & 1; tam[ch][c4 * 4 + 1] = am.y & 1; tam[ch][c4 * 4 + 2] = am.z & 1; tam[ch][c4 * 4 + 3] = am.w & 1;
        if (gidx) {
            // this thread's byte of the PLANAR index words of the sparse weight-gradient kernel (wgrad_sp_f16.hip; the
            // pair itself is shared): (k-step, lane half, word byte) = (4 wt + (c4 >> 2), c4 & 1, (c4 >> 1) & 1)
            const unsigned byte = (am.x & 1u) | ((2u + (am.y & 1u)) << 2) | ((am.z & 1u) << 4) | ((2u + (am.w & 1u)) << 6);
            idxb[ch][(((c4 >> 2) * 2 + (c4 & 1)) * 2) + ((c4 >> 1) & 1)] = (unsigned char)byte;   // gathered, stored 8 at a time below
        }
    }
    __syncthreads();
    if (gidx && tid < 128) {       // per channel and k-step pair: 2 k-steps x 2 lane halves x 2 bytes = 8 contiguous bytes of gidx
        const int ch = tid >> 1, pr = tid & 1, ks0 = wt * (GP_TW / 16) + pr * 2;
        if (ks0 < 22)
            *reinterpret_cast<unsigned long long *>(gidx + ((((size_t)b * 64 + ch) * Hp + hp) * 22 + ks0) * 4) =
                *reinterpret_cast<const unsigned long long *>(&idxb[ch][pr * 8]);
    }
    // 64 positions x 64 channels: thread -> (position, 8-channel group), two positions per thread
#pragma unroll
    for (int it = 0; it < GP_TW / 32; ++it) {
        const int pos = (tid >> 3) + 32 * it, cg = tid & 7, w = wt * GP_TW + pos;
        if (w >= CV_PITCH) continue;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile[cg * 8 + j][pos];
            const _Float16 hv = (_Float16)v;
            hi[j] = hv;
            lo[j] = (_Float16)(v - (float)hv);
        }
        const size_t o = ((((size_t)b * Hp + hp) * 4 + (cg >> 1)) * CV_PITCH + w) * 16 + (cg & 1) * 8;
        *reinterpret_cast<half8 *>(g_hi + o) = hi;
        *reinterpret_cast<half8 *>(g_lo + o) = lo;
        if (cg < 4) {                                               // one thread per (position, channel block): both index words
            unsigned word = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = (j < 4 ? 4 * hf + j : 8 + 4 * hf + (j - 4));
                    const unsigned f = 2u * (j & 1) + tam[cg * 16 + col][pos];
                    word |= f << (16 * hf + 2 * j);
                }
            g_idx[(((size_t)b * Hp + hp) * 4 + cg) * CV_PITCH + w] = word;
        }
    }
}

// ---- LayerNorm / PReLU backward fused INTO the pooled-operand pass ---------------------------------------------------
// mx_ln_prelu_bwd writes G = dL/dp (fp32, 4 B per element) only for gpool_cl_prep_kernel to read it back, scale it, split it
// and transpose it.  When the consumer block takes nothing but the pooled channels-last pair (blocks 2-4: both gradients on
// the sparse instruction), G never needs to exist: this kernel is gpool_cl_prep_kernel with the LayerNorm / PReLU backward as
// its load stage.  What stood in the way is the f16x3 scale, a power of two from max|G| over the whole tensor, which is only
// known after the pass: mx_ln_bwd_finish replaces it by an UPPER BOUND from quantities that exist before --
//   |G| <= rstd_p max(1, |slope_c|) (max|dxhat| + |m1_p| + max|xhat| |m2_p|)       maximised over the planes p = (clip, channel),
// max|dxhat| and max|xhat| from the data-gradient epilogue (dgrad_sp_f16x3_kernel, gx_bits), m1_p, m2_p = the plane means of
// dxhat and dxhat * xhat it already leaves (ln_part).  A bound 2^k above the true maximum costs precision only on elements
// below 2^(k-18) of the maximum (their fp16 `lo` half turns subnormal); it can never overflow.
//   p, dxhat (B,64,Hp,352) fp32, stats (B,64,2), slope (64,), m12 (B,64,2) = {m1, m2} (mx_ln_bwd_finish), scale {S, 1/S}
//   -> g_hi, g_lo, g_idx, gidx as gpool_cl_prep_kernel; part (B,64,Hp,6,2) = {d loss / d slope, sum of G} per workgroup and
//      channel (summed by mx_plane_partials_sum: the PReLU-slope and bias gradients)
__global__ __launch_bounds__(256) void lnbwd_gpool_kernel(const float *__restrict__ p, const float *__restrict__ dxhat,
                                                          const unsigned char *__restrict__ amax,
                                                          const float *__restrict__ stats, const float *__restrict__ slope,
                                                          const float *__restrict__ m12, const float *__restrict__ scale,
                                                          int Hp, int Wv, _Float16 *__restrict__ g_hi,
                                                          _Float16 *__restrict__ g_lo, unsigned *__restrict__ g_idx,
                                                          unsigned char *__restrict__ gidx, float *__restrict__ part)
{
    __shared__ float tile[64][GP_TW + 1];
    __shared__ unsigned char tam[64][GP_TW + 4];
    __shared__ __attribute__((aligned(16))) unsigned char idxb[64][16];
    const int wt = blockIdx.x, hp = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const float S = scale[0];
    constexpr int NIT = 64 * (GP_TW / 4) / 256;                 // (channel, 4 positions) items per thread: channels tid/16 + 16 k
    floatx4 pv[NIT], gv[NIT];
    uchar4 av[NIT];
    float mean[NIT], rstd[NIT], sl[NIT], m1[NIT], m2[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {                             // all loads first
        const int i = tid + 256 * k, ch = i / (GP_TW / 4), c4 = i % (GP_TW / 4), w0 = wt * GP_TW + c4 * 4;
        const size_t off = (((size_t)b * 64 + ch) * Hp + hp) * CV_PITCH + (w0 < CV_PITCH ? w0 : 0);
        pv[k] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(p + off));          // read exactly once
        gv[k] = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(dxhat + off));
        av[k] = *reinterpret_cast<const uchar4 *>(amax + off);
        const size_t pl = (size_t)b * 64 + ch;
        mean[k] = stats[pl * 2]; rstd[k] = stats[pl * 2 + 1]; sl[k] = slope[ch];
        m1[k] = m12[pl * 2]; m2[k] = m12[pl * 2 + 1];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k, ch = i / (GP_TW / 4), c4 = i % (GP_TW / 4), w0 = wt * GP_TW + c4 * 4;
        const uchar4 am = av[k];
        float tds = 0.0f, tgs = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                           // the arithmetic of ln_prelu_bwd_kernel (norm.hip), term by term
            const bool valid = w0 + e < Wv;
            const bool pos = pv[k][e] > 0.0f;
            const float x = pos ? pv[k][e] : sl[k] * pv[k][e];
            const float xh = (x - mean[k]) * rstd[k];
            const float dx = rstd[k] * (gv[k][e] - m1[k] - xh * m2[k]);
            const float r = valid ? (pos ? dx : sl[k] * dx) : 0.0f;
            tds += (valid && !pos) ? dx * pv[k][e] : 0.0f;
            tgs += r;
            tile[ch][c4 * 4 + e] = r * S;
        }
        // the 16 threads of a channel are 16 consecutive lanes: fixed-order sum, one partial per (channel, workgroup)
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            tds += __shfl_xor(tds, d, 16);
            tgs += __shfl_xor(tgs, d, 16);
        }
        if (c4 == 0) {
            float *q = part + (((((size_t)b * 64 + ch) * Hp + hp) * gridDim.x) + wt) * 2;
            q[0] = tds;
            q[1] = tgs;
        }
        tam[ch][c4 * 4 + 0] = am.x & 1; tam[ch][c4 * 4 + 1] = am.y & 1; tam[ch][c4 * 4 + 2] = am.z & 1; tam[ch][c4 * 4 + 3] = am.w & 1;
        if (gidx) {
            const unsigned byte = (am.x & 1u) | ((2u + (am.y & 1u)) << 2) | ((am.z & 1u) << 4) | ((2u + (am.w & 1u)) << 6);
            idxb[ch][(((c4 >> 2) * 2 + (c4 & 1)) * 2) + ((c4 >> 1) & 1)] = (unsigned char)byte;
        }
    }
    __syncthreads();
    if (gidx && tid < 128) {
        const int ch = tid >> 1, pr = tid & 1, ks0 = wt * (GP_TW / 16) + pr * 2;
        if (ks0 < 22)
            *reinterpret_cast<unsigned long long *>(gidx + ((((size_t)b * 64 + ch) * Hp + hp) * 22 + ks0) * 4) =
                *reinterpret_cast<const unsigned long long *>(&idxb[ch][pr * 8]);
    }
#pragma unroll
    for (int it = 0; it < GP_TW / 32; ++it) {
        const int pos = (tid >> 3) + 32 * it, cg = tid & 7, w = wt * GP_TW + pos;
        if (w >= CV_PITCH) continue;
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile[cg * 8 + j][pos];
            const _Float16 hv = (_Float16)v;
            hi[j] = hv;
            lo[j] = (_Float16)(v - (float)hv);
        }
        const size_t o = ((((size_t)b * Hp + hp) * 4 + (cg >> 1)) * CV_PITCH + w) * 16 + (cg & 1) * 8;
        __builtin_nontemporal_store(hi, reinterpret_cast<half8 *>(g_hi + o));
        __builtin_nontemporal_store(lo, reinterpret_cast<half8 *>(g_lo + o));
        if (cg < 4) {
            unsigned word = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = (j < 4 ? 4 * hf + j : 8 + 4 * hf + (j - 4));
                    const unsigned f = 2u * (j & 1) + tam[cg * 16 + col][pos];
                    word |= f << (16 * hf + 2 * j);
                }
            g_idx[(((size_t)b * Hp + hp) * 4 + cg) * CV_PITCH + w] = word;
        }
    }
}

// plane means of dxhat and dxhat * xhat from the data gradient's partial sums, and the bound on max |G| (above)
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float *__restrict__ ln_part, const float *__restrict__ stats,
                                                            const float *__restrict__ slope,
                                                            const unsigned *__restrict__ gx_bits, int n_planes, int C, int H,
                                                            int Wv, float *__restrict__ m12, unsigned *__restrict__ bound_bits)
{
    const int plane = blockIdx.x * 256 + threadIdx.x;
    float bound = 0.0f;
    if (plane < n_planes) {
        typedef float floatx2 __attribute__((ext_vector_type(2)));
        const floatx2 *lp = reinterpret_cast<const floatx2 *>(ln_part) + (size_t)plane * (2 * H);
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < 2 * H; ++i) {
            const floatx2 v = lp[i];
            s1 += (double)v[0];
            s2 += (double)v[1];
        }
        const double n = (double)H * (double)Wv;
        const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
        m12[plane * 2] = m1;
        m12[plane * 2 + 1] = m2;
        const float mxd = __uint_as_float(gx_bits[0]), mxx = __uint_as_float(gx_bits[1]);
        const float sl = fabsf(slope[plane % C]);
        // (1 + 2^-20): the bound is evaluated in fp32 like the values it bounds
        bound = stats[plane * 2 + 1] * fmaxf(1.0f, sl) * (mxd + fabsf(m1) + mxx * fabsf(m2)) * 1.000001f;
    }
    bound = wave_max_f32(bound);
    if ((threadIdx.x & 63) == 0) atomicMax(bound_bits, __float_as_uint(bound));
}

__global__ void pow2_scale_from_bits_kernel(const unsigned *__restrict__ bits, float *__restrict__ scale)
{
    const float m = __uint_as_float(*bits);
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) {
        frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
        e = 10 - e;                    // m * 2^(10 - e) in [512, 1024)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    scale[0] = ldexpf(1.0f, e);
    scale[1] = ldexpf(1.0f, -e);
}

// out_a[plane], out_b[plane] = sums of the K {a, b} pairs of a plane (fp64, fixed order)
__global__ __launch_bounds__(256) void plane_partials_sum_kernel(const float *__restrict__ part, int n_planes, int K,
                                                                 float *__restrict__ out_a, float *__restrict__ out_b)
{
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= n_planes) return;
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    const floatx2 *q = reinterpret_cast<const floatx2 *>(part) + (size_t)plane * K;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < K; i += 64) {
        const floatx2 v = q[i];
        a += (double)v[0];
        b += (double)v[1];
    }
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if (lane == 0) {
        out_a[plane] = (float)a;
        out_b[plane] = (float)b;
    }
}

template <int T, bool LN>
static int launch_dgrad_sp(const DgradSpArgs &a, int B, hipStream_t st)
{
    constexpr size_t buf = DsGeom<T>::BUF;                        // NCB channel-block images per buffer
    constexpr size_t scratch = 4 * 3 * 32 * 33 * 4;              // epilogue: transposition tiles + the LN partial reduction
    constexpr size_t lds = 2 * buf > scratch ? 2 * buf : scratch;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static MxLdsLatch latch = {};                             // per device (common.h)
    if (mx_set_dyn_lds(latch, (const void *)dgrad_sp_f16x3_kernel<T, LN>, lds) != MX_OK) return MX_ERR_LAUNCH;
    hipLaunchKernelGGL((dgrad_sp_f16x3_kernel<T, LN>), dim3(a.H / 2, B), dim3(256), lds, st, a);
    return mx_launch_status();
}

// W (64,64,5,13) -> w_hi, w_lo: 4*3*2*13*2*64*16 halfs each
MX_EXPORT int mx_conv_pack_weights_sp_f16(const float *W, void *w_hi, void *w_lo, void *stream)
{
    if (!W || !w_hi || !w_lo) return MX_ERR_ARG;
    hipLaunchKernelGGL(pack_weights_sp_f16_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, W, (_Float16 *)w_hi,
                       (_Float16 *)w_lo);
    return mx_launch_status();
}

// G, amax: (B,64,H/2,352); scale: the {S, 1/S} pair -> g_hi, g_lo (B,H/2,4,352,16) halfs = the pooled operand of BOTH
// sparse gradient kernels, g_idx (B,H/2,4,352) uint32 index words of the data gradient; gidx (optional): (B,64,H/2,22,2)
// uint16 index words of the weight gradient (mx_conv_block_wgrad_sp_f16)
MX_EXPORT int mx_conv_prep_gpool_cl_f16(const float *G, const uint8_t *amax, const float *scale, int64_t B, int64_t H,
                                        int64_t Wv, void *g_hi, void *g_lo, void *g_idx, void *gidx, void *stream)
{
    if (!G || !amax || !scale || !g_hi || !g_lo || !g_idx || B <= 0 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH)
        return MX_ERR_ARG;
    if (B > 65535 || H > 131070) return MX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gpool_cl_prep_kernel, dim3((CV_PITCH + GP_TW - 1) / GP_TW, (unsigned)(H / 2), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, G, amax, scale, (int)(H / 2), (int)Wv, (_Float16 *)g_hi, (_Float16 *)g_lo,
                       (unsigned *)g_idx, (unsigned char *)gidx);
    return mx_launch_status();
}

// ln_part (B,64,H,2,2), stats (B,64,2), slope (64,), gx_bits (2,) from mx_conv_block_dgrad_sp_f16 -> m12 (B,64,2) plane means
// {dxhat, dxhat * xhat} and scale (2,) = {S, 1/S}, S the power of two that puts the BOUND on max |G| into [512, 1024);
// bound_ws: 1 uint workspace
MX_EXPORT int mx_ln_bwd_finish(const float *ln_part, const float *stats, const float *slope, const uint32_t *gx_bits,
                               int64_t B, int64_t C, int64_t H, int64_t Wv, float *m12, uint32_t *bound_ws, float *scale,
                               void *stream)
{
    if (!ln_part || !stats || !slope || !gx_bits || !m12 || !bound_ws || !scale || B <= 0 || C <= 0 || H <= 0 || Wv <= 0 ||
        Wv > CV_PITCH)
        return MX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(bound_ws, 0, sizeof(uint32_t), st) != hipSuccess) return MX_ERR_LAUNCH;
    const int n = (int)(B * C);
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ln_part, stats, slope, gx_bits, n,
                       (int)C, (int)H, (int)Wv, m12, bound_ws);
    hipLaunchKernelGGL(pow2_scale_from_bits_kernel, dim3(1), dim3(1), 0, st, bound_ws, scale);
    return mx_launch_status();
}

// LayerNorm / PReLU backward straight into the pooled operand of the block below (see lnbwd_gpool_kernel): p, dxhat, amax
// (B,64,Hp,352) -> g_hi, g_lo (B,Hp,4,352,16), g_idx (B,Hp,4,352), gidx (B,64,Hp,22,2) (optional), part (B,64,Hp,6,2);
// dslope_part, gsum_part (B*64,): the per-plane sums of mx_ln_prelu_bwd
MX_EXPORT int mx_ln_prelu_bwd_gpool_f16(const float *p, const float *dxhat, const uint8_t *amax, const float *stats,
                                        const float *slope, const float *m12, const float *scale, int64_t B, int64_t Hp,
                                        int64_t Wv, void *g_hi, void *g_lo, void *g_idx, void *gidx, float *part,
                                        float *dslope_part, float *gsum_part, void *stream)
{
    if (!p || !dxhat || !amax || !stats || !slope || !m12 || !scale || !g_hi || !g_lo || !g_idx || !part || !dslope_part ||
        !gsum_part || B <= 0 || Hp <= 0 || Wv <= 0 || Wv > CV_PITCH)
        return MX_ERR_ARG;
    if (B > 65535 || Hp > 65535) return MX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const unsigned tiles = (CV_PITCH + GP_TW - 1) / GP_TW;
    hipLaunchKernelGGL(lnbwd_gpool_kernel, dim3(tiles, (unsigned)Hp, (unsigned)B), dim3(256), 0, st, p, dxhat, amax, stats, slope,
                       m12, scale, (int)Hp, (int)Wv, (_Float16 *)g_hi, (_Float16 *)g_lo, (unsigned *)g_idx,
                       (unsigned char *)gidx, part);
    const int n = (int)(B * 64);
    hipLaunchKernelGGL(plane_partials_sum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, part, n, (int)(Hp * tiles),
                       dslope_part, gsum_part);
    return mx_launch_status();
}

// data gradient from the pooled channels-last operand and the fragment-packed weights; dxhat (B,64,H,352).
// x_hi, x_lo, ln_part (all or none): the block's forward operand pair (B,H,4,352,16) and the (B,64,H,2,2) partial sums
// {sum dxhat, sum dxhat * xhat} that mx_ln_prelu_bwd takes in place of its own statistics sweep.
MX_EXPORT int mx_conv_block_dgrad_sp_f16(const void *g_hi, const void *g_lo, const void *g_idx, const void *w_hi,
                                         const void *w_lo, const float *scale, int64_t B, int64_t H, int64_t Wv,
                                         int32_t dilation, float *dxhat, const void *x_hi, const void *x_lo,
                                         float *ln_part, uint32_t *gx_bits, void *stream)
{
    if (!g_hi || !g_lo || !g_idx || !w_hi || !w_lo || !scale || !dxhat) return MX_ERR_ARG;
    if ((x_hi || x_lo || ln_part) && !(x_hi && x_lo && ln_part)) return MX_ERR_ARG;
    if (gx_bits && !ln_part) return MX_ERR_ARG;
    if (B <= 0 || B > 65535 || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH - 1) return MX_ERR_UNSUPPORTED;
    DgradSpArgs a{(const _Float16 *)g_hi, (const _Float16 *)g_lo, (const unsigned *)g_idx, (const _Float16 *)w_hi,
                  (const _Float16 *)w_lo, scale, dxhat, (int)H, (int)Wv, (const _Float16 *)x_hi, (const _Float16 *)x_lo,
                  ln_part, gx_bits};
    hipStream_t st = (hipStream_t)stream;
    const bool ln = ln_part != nullptr;
    switch (dilation) {
    case 1: return ln ? launch_dgrad_sp<1, true>(a, (int)B, st) : launch_dgrad_sp<1, false>(a, (int)B, st);
    case 2: return ln ? launch_dgrad_sp<2, true>(a, (int)B, st) : launch_dgrad_sp<2, false>(a, (int)B, st);
    case 4: return ln ? launch_dgrad_sp<4, true>(a, (int)B, st) : launch_dgrad_sp<4, false>(a, (int)B, st);
    case 8: return ln ? launch_dgrad_sp<8, true>(a, (int)B, st) : launch_dgrad_sp<8, false>(a, (int)B, st);
    case 16: return ln ? launch_dgrad_sp<16, true>(a, (int)B, st) : launch_dgrad_sp<16, false>(a, (int)B, st);
    default: return MX_ERR_UNSUPPORTED;
    }
}
