// wgrad_sp_f16.hip -- weight gradient of the 64->64 channel Spectral2DCNN convolutions on the SPARSE fp16 matrix
// instruction v_smfmac_f32_32x32x32_f16, with fp32-equivalent accuracy ("f16x3", see conv_f16.hip).
// Reference semantics: torch.nn.Conv2d backward w.r.t. weight behind MaxPool2d((2,1)) (mod_extraction/models.py:187-188):
//   dW[co][ci][kh][kw] = sum over (b, h, w) of  dz[b][co][h][w] * xhat[b][ci][h + kh - 2][w + (kw - 6) T]
// where dz is the pooled gradient G routed to the row of each pooling pair that won the max: of dz[.][2hp][w] and
// dz[.][2hp+1][w] exactly one is G[.][hp][w], the other is 0.  Ordering the GEMM's K dimension as
//   k = 2 * position + row parity
// makes that a 2:4 structured-sparse A operand: every group of 4 consecutive k (two positions x two parities) holds two
// non-zeros, one in {0,1} and one in {2,3}.  The sparse MFMA takes A compressed -- here simply G itself, at pooled
// resolution -- plus 2-bit positions (the pooling argmax), multiplies it with a dense K = 32 B operand in the time
// of a dense K = 16 instruction, and so does BOTH rows of a pooling pair at once: half the matrix instructions (and
// half their energy: these kernels are power-limited) of the dense kernel in wgrad_f16.hip.
//
// Operand layouts (derived with tools/probe/probe_smfmac.py, checked numerically by tools/probe/check_smfmac.py; this
// image has no ISA manual):
//   A  lane l: row m = l & 31, half hh = l >> 5; compressed element j (0..7) lies in the logical group of 4 starting at
//      k = 16 (j >> 2) + 8 hh + 4 ((j >> 1) & 1), at position (idx >> 2j) & 3 of it  ->  with k = 2 pos + parity the lane
//      holds G at positions 4hh .. 4hh+3 and 8+4hh .. 8+4hh+3 of the 16-position k-step (two transposed 4-row reads from
//      a [position][channel] image: the channels-last pooled pair that the sparse data gradient consumes as well) and idx
//      field j = 2 (j & 1) + argmax(position j).
//   B  lane l: column n = l & 31, k = 16 (l >> 5) + j, j = 0..15: 16 consecutive rows of a [2 pos + parity][channel]
//      image = four ds_read_b64_tr_b16 (x rows h + kh - 2 for the two parities are staged interleaved).
//   D  as the dense 32x32 MFMA.
// Everything else follows wgrad_f16.hip: workgroup = (kernel row kh, slab of POOLED rows), 4 waves = (ci tile, tap
// group), 13 accumulators, three-phase k-steps with staggered fragment lifetimes, register prefetch of the next chunk,
// deterministic fp64 slab reduction.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

#define WS_ROW_KS 22                    // 352 positions = 22 k-steps of 16 positions (32 logical k)

struct WgradSpArgs {
    const _Float16 *gp_hi, *gp_lo;      // (B, Hp, 4, 352, 16): fp16 pair of G * S at pooled resolution, channels last
                                        // (the operand of the sparse data gradient: one pooled pair feeds both)
    const unsigned short *gidx;         // (B, 64, Hp, 22, 2): index word of (k-step, lane half)
    const _Float16 *x_hi, *x_lo;        // (B, H, 4, 352, 16)
    float *part;                        // (n_slabs, 5, 13, 64, 64)
    int B, H, rows_per_slab, n_slabs;   // rows = pooled rows (b, hp)
};

__device__ __forceinline__ floatx16 smfmac(half8 a, half16 b, floatx16 c, int idx)
{
    return __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c, idx, 0, 0);
}

// Transposed-read lane offsets (see wgrad_f16.hip:tr_lane_offsets), for a lane half that owns 16 consecutive image rows
struct TrLane32 { int off[4]; };
__device__ __forceinline__ TrLane32 tr_lane_offsets32(int tile, int lane)
{
    const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1, h2 = lane >> 5;
    TrLane32 L;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int swz = ((ph + q) >> 1) & 1;
        L.off[ph] = (16 * h2 + q) * 128 + ((tile ^ swz) * 64) + 32 * g1 + 8 * p;
    }
    return L;
}
// 16 consecutive rows r0 + 16 h2 .. + 15 of this lane's channel: four transposed reads (r0 wave-uniform, ph = r0 & 3)
__device__ __forceinline__ half16 tr_frag32(const unsigned char *img_bytes, int lane_off, int r0)
{
    const unsigned char *ptr = img_bytes + lane_off + r0 * 128;
    const short4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)ptr);
    const short4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)(ptr + 512));
    const short4v v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)(ptr + 1024));
    const short4v v3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)(ptr + 1536));
    typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
    typedef short short16v __attribute__((__vector_size__(16 * sizeof(short))));
    const short8v lo = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    const short8v hi = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
    const short16v all = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    return __builtin_bit_cast(half16, all);
}

// ---- x fragments as SLIDING WINDOWS over register chains.  The seven taps of a wave read 16-row fragments whose first
// rows differ by 2T image rows; with 4-row transposed blocks, fragments that differ by a multiple of 4 rows share blocks.
// A fragment must be 8 consecutive VGPRs, and left to itself the compiler gives every window its own 8 registers and
// COPIES the shared blocks into them (~30 v_accvgpr_mov clumped at the head of every 13-MFMA phase, behind a wait for
// nearly all of the phase's LDS reads: the matrix pipe idled for a third of the kernel).  Holding a chain of blocks in
// ONE wide vector that passes through an empty asm ("pin") makes it a single virtual register; the windows are then
// sub-register ranges of it and cost nothing.
typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int16v __attribute__((ext_vector_type(16)));
typedef int int32v __attribute__((ext_vector_type(32)));

template <int S, typename V>
__device__ __forceinline__ half16 chain_window(const V &c)
{
    if constexpr (2 * S + 7 < (int)(sizeof(V) / 4)) {
        const int8v w = __builtin_shufflevector(c, c, 2 * S, 2 * S + 1, 2 * S + 2, 2 * S + 3, 2 * S + 4, 2 * S + 5, 2 * S + 6, 2 * S + 7);
        return __builtin_bit_cast(half16, w);
    } else {
        return half16{};
    }
}

//   T = 1: two chains (even taps: rows 0,4,..,24 = 7 blocks; odd taps: rows 2,6,..,22 = 6 blocks)    13 reads, not 28
//   T = 2: the ten blocks of rows 0,4,..,36 as two 512-bit chains (taps 0..4: blocks 0..7; taps 5,6: blocks 5..9 -- three
//          blocks are read twice; one 1024-bit chain with 12 unused registers made the compiler shuffle)   13 reads
//   T = 4: one chain of 16 blocks                                                                      16 reads
template <int T>
struct SlidingB {
    static constexpr int NBLK = T == 4 ? 16 : 13;
    static constexpr int NC = T == 4 ? 1 : 2;
    static constexpr int N0 = T == 1 ? 7 : (T == 2 ? 8 : 16);          // blocks in chain 0
    using V = std::conditional_t<T == 4, int32v, int16v>;
    V c[NC];
    static __device__ __forceinline__ int row_of(int n)
    {
        return n < N0 ? 4 * n : (T == 1 ? 2 + 4 * (n - N0) : 4 * (n - N0 + 5));
    }
    __device__ __forceinline__ void put(int n, short4v v)
    {
        const int2v b = __builtin_bit_cast(int2v, v);
        const int ci = n >= N0 ? 1 : 0, slot = n >= N0 ? n - N0 : n;
        c[ci][2 * slot] = b[0];
        c[ci][2 * slot + 1] = b[1];
    }
    __device__ __forceinline__ void pin()
    {
#pragma unroll
        for (int i = 0; i < NC; ++i) asm volatile("" : "+v"(c[i]));
    }
    // window of tap i (0..6 within the wave's tap group)
    __device__ __forceinline__ half16 frag(int i) const
    {
        const V &v = c[T == 1 ? (i & 1) : (T == 2 && i >= 5 ? 1 : 0)];
        const int s = T == 1 ? (i >> 1) : (T == 2 ? (i >= 5 ? i - 5 : i) : 2 * i);
        switch (s) {
        case 0: return chain_window<0>(v);
        case 1: return chain_window<1>(v);
        case 2: return chain_window<2>(v);
        case 3: return chain_window<3>(v);
        case 4: return chain_window<4>(v);
        case 5: return chain_window<5>(v);
        case 6: return chain_window<6>(v);
        case 8: return chain_window<8>(v);
        case 10: return chain_window<10>(v);
        default: return chain_window<12>(v);
        }
    }
};
// T >= 8: the taps' fragments do not overlap (7 x 4 blocks)
template <int T>
struct FlatB {
    static constexpr int NBLK = 28;
    short4v blk[28];
    static __device__ __forceinline__ int row_of(int n) { return (n >> 2) * 2 * T + 4 * (n & 3); }
    __device__ __forceinline__ void put(int n, short4v v) { blk[n] = v; }
    __device__ __forceinline__ void pin() {}
    __device__ __forceinline__ half16 frag(int i) const
    {
        typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
        typedef short short16v __attribute__((__vector_size__(16 * sizeof(short))));
        const int n = 4 * i;
        const short8v lo = __builtin_shufflevector(blk[n], blk[n + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        const short8v hi = __builtin_shufflevector(blk[n + 2], blk[n + 3], 0, 1, 2, 3, 4, 5, 6, 7);
        const short16v all = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        return __builtin_bit_cast(half16, all);
    }
};

template <int T>
__global__ __launch_bounds__(256, 1) void wgrad_sp_f16x3_kernel(WgradSpArgs a)
{
    constexpr int KC = T >= 16 ? 3 : 6;                                  // k-steps per chunk
    constexpr int NCH = (WS_ROW_KS + KC - 1) / KC;                       // chunks per row: 8,7,7 / 6,6,5,5 / 3,3,3,3,3,3,2,2
    constexpr int KS_LO = WS_ROW_KS / NCH, KS_REM = WS_ROW_KS % NCH;
    constexpr int CHP = KC * 16;                                         // positions per chunk (max)
    constexpr int WINP = CHP + 12 * T;                                   // x window positions (origin w0 - 6T)
    constexpr int A_SPLIT = CHP * 128, B_SPLIT = 2 * WINP * 128;         // bytes per split
    constexpr int IDX_BYTES = 64 * KC * 2 * 2;
    constexpr int QI = (KC * 2 + 3) / 4;                                 // index words per thread
    constexpr int QX = (WINP + 15) / 16;                                 // x iterations per split
    constexpr bool PREF = T <= 4;                                        // register prefetch of the next chunk
    constexpr int NVI = PREF ? QI : 1, NVB = PREF ? 2 * QX : 1;
    static_assert(2 * A_SPLIT + IDX_BYTES + 2 * B_SPLIT <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const Aimg = smem;                                    // [split][position][64 co], 64-B halves swizzled
    unsigned char *const Iimg = smem + 2 * A_SPLIT;                      // [co][KC][2] u16
    unsigned char *const Bimg = Iimg + ((IDX_BYTES + 15) / 16) * 16;     // [split][2 pos + parity][64 ch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & 1, nt = wave >> 1;
    const int id = blockIdx.x;
    const int kh = (id >> 3) % CV_KH;
    const int slab = (id & 7) + 8 * (id / (8 * CV_KH));
    if (slab >= a.n_slabs) return;
    const int H = a.H, Hp = H >> 1;

    floatx16 acc[CV_KW];
#pragma unroll
    for (int i = 0; i < CV_KW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int row_begin = slab * a.rows_per_slab;
    int row_end = row_begin + a.rows_per_slab;
    if (row_end > a.B * Hp) row_end = a.B * Hp;
    // a pooled row contributes unless BOTH x rows of the pair fall outside the image
    auto row_valid = [&](int rid) {
        const int hx0 = 2 * (rid % Hp) + kh - 2;
        return hx0 + 1 >= 0 && hx0 < H;
    };
    auto next_iter = [&](int &rid, int &ch) {
        if (++ch < NCH) return;
        ch = 0;
        do { ++rid; } while (rid < row_end && !row_valid(rid));
    };
    auto chunk_ks = [&](int ch) { return KS_LO + (ch < KS_REM ? 1 : 0); };
    auto chunk_k0 = [&](int ch) { return ch * KS_LO + (ch < KS_REM ? ch : KS_REM); };      // first k-step

    // ---- staging: global -> registers -> LDS (next chunk in flight during the current chunk's MFMAs).  Regular maps:
    //      every address is a per-thread constant + a compile-time multiple of the iteration index.
    //   A   thread = (co = tid >> 2, vector c8 = (tid & 3) + 4 q of 8 positions), q < CHP / 32, per split
    //   idx thread = (co = tid >> 2, word e = (tid & 3) + 4 q of the chunk's 2 KC), q < QI
    //   x   thread = (position pos0 + 16 q, row parity, 16-byte vector sv), per split: a wave stores 8 CONSECUTIVE image
    //       rows (alternating LDS bank halves; a fixed parity would put a whole wave on the even rows = half the banks)
    static_assert(CHP % 32 == 0 || CHP == 48, "A staging map");
    floatx4 pa[PREF ? 2 * ((CHP + 31) / 32) : 1], pb[NVB];
    unsigned short pi[NVI];
    const int sv = tid & 7, xpar = (tid >> 3) & 1, pos0 = tid >> 4;
    const int a_co = tid >> 2, a_c8 = tid & 3;                           // index words: (co, word)
    const int a_pos = tid >> 3, a_sv = tid & 7;                          // G: (position a_pos + 32 q, 16-byte vector = 8 channels)
    constexpr int QA2 = (CHP + 31) / 32;                                 // G iterations per split
    // Loads are branch-free (a predicated load costs an exec-mask branch each, 36 per chunk): out-of-range items read
    // a clamped address whose value is either never used (k-steps past the chunk's end) or known to be zero (the
    // pad column 351 of an operand row stands in for the halo and for rows outside the image; Wv <= 351 is checked
    // by the entry point).  Every address is a wave-uniform base (scalar arithmetic on (row, chunk)) plus a 32-bit
    // lane offset that is constant over the kernel up to one select: the loads sit between the MFMAs of phase 2, where
    // 64-bit lane arithmetic (a dozen VALU operations per load) delays the matrix instructions behind it.
    struct ChunkBase {
        const unsigned char *a_hi, *a_lo, *idx, *x_hi, *x_lo;   // uniform
        int npos, nks2, w_first;                                // positions / index words in the chunk; window origin
        unsigned row_off;                                       // lane: byte offset of its x row (parity) from the base row
        bool row_ok;                                            // lane: its x row is inside the image
    };
    const unsigned a_thr = (unsigned)(a_sv >> 1) * (CV_PITCH * 32) + (unsigned)(a_sv & 1) * 16 + (unsigned)a_pos * 32;   // gp: (B,Hp,4,352,16) halfs
    const unsigned i_thr = (unsigned)a_co * (unsigned)Hp * (WS_ROW_KS * 4);         // gidx: (B,64,Hp,22,2) u16
    const unsigned x_thr = (unsigned)(sv >> 1) * (CV_PITCH * 32) + (unsigned)(sv & 1) * 16;   // x: (B,H,4,352,16) halfs
    auto chunk_base = [&](int rid, int ch) {
        ChunkBase c;
        const int b = rid / Hp, hp = rid - b * Hp, ks0 = chunk_k0(ch);
        const size_t arow = ((size_t)b * Hp + hp) * (4 * CV_PITCH * 32) + (size_t)ks0 * (16 * 32);
        c.a_hi = reinterpret_cast<const unsigned char *>(a.gp_hi) + arow;
        c.a_lo = reinterpret_cast<const unsigned char *>(a.gp_lo) + arow;
        c.idx = reinterpret_cast<const unsigned char *>(a.gidx) + ((size_t)b * 64 * Hp + hp) * (WS_ROW_KS * 4) + ks0 * 4;
        c.npos = chunk_ks(ch) * 16;
        c.nks2 = chunk_ks(ch) * 2;
        c.w_first = ks0 * 16 - 6 * T;
        // x rows 2hp + kh - 2 (parity 0) and + 1 (parity 1), clamped into the image; the base is the parity-0 row
        const int hx0 = 2 * hp + kh - 2, hx1 = hx0 + 1;
        const int r0 = hx0 < 0 ? 0 : (hx0 >= H ? H - 1 : hx0), r1 = hx1 < 0 ? 0 : (hx1 >= H ? H - 1 : hx1);
        const size_t xrow = ((size_t)b * H + r0) * (4 * CV_PITCH * 32);
        c.x_hi = reinterpret_cast<const unsigned char *>(a.x_hi) + xrow;
        c.x_lo = reinterpret_cast<const unsigned char *>(a.x_lo) + xrow;
        const unsigned d1 = (unsigned)(r1 - r0) * (4 * CV_PITCH * 32);
        const bool ok0 = hx0 >= 0 && hx0 < H, ok1 = hx1 >= 0 && hx1 < H;
        c.row_off = xpar ? d1 : 0u;
        c.row_ok = xpar ? ok1 : ok0;
        return c;
    };
    // G vector k = split * QA2 + q: position a_pos + 32 q of the chunk, channels 8 a_sv .. + 7
    auto load_a = [&](int k, const ChunkBase &c) -> floatx4 {
        const int split = k / QA2, q = k - split * QA2;
        const int pos = a_pos + 32 * q;
        const unsigned off = a_thr + (pos < c.npos ? (unsigned)q * 1024 : 0u - (unsigned)a_pos * 32);   // past the chunk: position 0 (unused)
        return *reinterpret_cast<const floatx4 *>((split ? c.a_lo : c.a_hi) + off);
    };
    auto store_a = [&](int k, floatx4 v) {
        const int split = k / QA2, q = k - split * QA2, pos = a_pos + 32 * q;
        if (pos < CHP)
            *reinterpret_cast<floatx4 *>(Aimg + split * A_SPLIT + pos * 128 + ((a_sv ^ (((pos >> 1) & 1) << 2)) * 16)) = v;
    };
    auto load_i = [&](int q, const ChunkBase &c) -> unsigned short {
        const int e = a_c8 + 4 * q;
        const unsigned off = i_thr + (e < c.nks2 ? (unsigned)e * 2 : 0u);
        return *reinterpret_cast<const unsigned short *>(c.idx + off);
    };
    auto store_i = [&](int q, unsigned short v) {
        const int e = a_c8 + 4 * q;
        if (e < KC * 2) reinterpret_cast<unsigned short *>(Iimg)[a_co * (KC * 2) + e] = v;
    };
    // x vector k = split * QX + q: position pos0 + 16 q of the window, row parity xpar, 16-byte vector sv
    auto load_b = [&](int k, const ChunkBase &c) -> floatx4 {
        const int split = k / QX, q = k - split * QX;
        const int w = c.w_first + pos0 + 16 * q;
        const bool ok = (unsigned)w < (unsigned)CV_PITCH && c.row_ok;
        const unsigned off = ok ? c.row_off + x_thr + (unsigned)w * 32 : x_thr + (CV_PITCH - 1) * 32;
        return *reinterpret_cast<const floatx4 *>((split ? c.x_lo : c.x_hi) + off);
    };
    auto store_b = [&](int k, floatx4 v) {
        const int split = k / QX, q = k - split * QX, par = xpar;
        const int pos = pos0 + 16 * q;
        if (pos < WINP)
            *reinterpret_cast<floatx4 *>(Bimg + split * B_SPLIT + (2 * pos + par) * 128 + ((sv ^ ((pos & 1) << 2)) * 16)) = v;
    };
    constexpr int NA = 2 * QA2, NI = (KC * 2 + 3) / 4;

    int rid = row_begin, ch = 0;
    while (rid < row_end && !row_valid(rid)) ++rid;
    if (PREF && rid < row_end) {
        const ChunkBase c0 = chunk_base(rid, ch);
#pragma unroll
        for (int q = 0; q < NA; ++q) pa[q] = load_a(q, c0);
#pragma unroll
        for (int q = 0; q < NI; ++q) pi[q] = load_i(q, c0);
#pragma unroll
        for (int q = 0; q < NVB; ++q) pb[q] = load_b(q, c0);
    }
    // accumulators: acc[2k + j] = tap 7g + k (k < 6), co tile j;   acc[12] = tap 6, co tile g
    const int m32 = lane & 31, hh = lane >> 5;
    // The wave's seven taps sit at image-row offsets 2T i, i = 0..6, from row 12 g T of the window (g = 0: taps 0..5 then
    // the middle tap 6; g = 1: the middle tap then taps 7..12); see SlidingB for how their fragments share blocks.
    using BFrags = std::conditional_t<(T <= 4), SlidingB<T>, FlatB<T>>;
    constexpr int NBLK = BFrags::NBLK;
    const TrLane32 lb = tr_lane_offsets32(nt, lane);
    const int xg_off = g * 12 * T * 128;
    // A / index addressing: co tile j -> rows j*32 + m32; the middle tap uses co tile g
    // G fragment of co tile j: positions 4hh .. 4hh+3 and 8+4hh .. of the k-step for channel j*32 + m32 = two transposed
    // 4-row blocks of the [position][channel] image (rows 4 hh + q, lane (q, p, g1) as in tr_lane_offsets32; the swizzle
    // phase of a block starting on a multiple of 4 rows is (q >> 1) & 1)
    int a_lane[2];
    {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1, swz = (q4 >> 1) & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) a_lane[j] = (4 * hh + q4) * 128 + ((j ^ swz) * 64) + 32 * g1 + 8 * p4;
    }
    const int i_lane = m32 * (KC * 4) + hh * 2;
    const unsigned char *Ah = Aimg, *Al = Aimg + A_SPLIT;
    const unsigned char *x_h = Bimg, *x_l = Bimg + B_SPLIT;

    // the k-step loop, specialised on the tap group (which tap of the window is the middle one)
    // next chunk's prefetch item q (A vectors, index words, x vectors): issued a few per k-step INSIDE the MFMA loop --
    // as one burst between the barriers their address arithmetic alone cost 18 % of the kernel
    constexpr int NPF = PREF ? NA + NI + NVB : 0;
    constexpr int PF_PER = PREF ? (NPF + (KS_LO - 1) - 1) / (KS_LO - 1) : 0;      // all issued within the first KS_LO - 1 k-steps
    auto prefetch_item = [&](int q, const ChunkBase &c) {
        if (q < NA) pa[q] = load_a(q, c);
        else if (q < NA + NI) pi[q - NA] = load_i(q - NA, c);
        else if (q < NPF) pb[q - NA - NI] = load_b(q - NA - NI, c);
    };
    auto run_chunk = [&](auto gc, int nks, const ChunkBase &nxt, bool more) {
        constexpr int G = decltype(gc)::value;
        half8 AL[2], AH[2][2];                      // [co tile]; the middle tap uses co tile G
        int IX[2][2];
        BFrags BH, BL;                              // 4-row blocks of hi(x) / lo(x)
        auto rd_a = [&](const unsigned char *img, int ks, int j) {
            const unsigned char *ptr = img + ks * 2048 + a_lane[j];
            const short4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)ptr);
            const short4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)(ptr + 1024));
            typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
            const short8v both = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(half8, both);
        };
        auto rd_i = [&](int ks, int j) {
            return (int)*reinterpret_cast<const unsigned short *>(Iimg + j * (32 * KC * 4) + i_lane + ks * 4);
        };
        auto rd_blk = [&](const unsigned char *img, int ks, int n) {
            const int r0 = BFrags::row_of(n);
            const unsigned char *ptr = img + ks * 4096 + xg_off + lb.off[(r0 + 12 * T * G) & 3] + r0 * 128;
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3))) *)ptr);
        };
        // accumulator u -> (window tap i, co tile): u = 2k + j (k < 6): i = k + G, co tile j; u = 12: middle tap, co tile G
        auto tap_of = [](int u) { return u < 12 ? (u >> 1) + G : (G ? 0 : 6); };
        auto ct_of = [](int u) { return u < 12 ? (u & 1) : G; };
        // N_DS LDS reads spread over the phase's 13 MFMAs, PER after each MFMA until they are all issued
#define WS_PIN(N_DS, PER)                                                                          \
    {                                                                                              \
        constexpr int pairs_ = (N_DS) / (PER) < 13 ? (N_DS) / (PER) : 13;                          \
        _Pragma("unroll") for (int q_ = 0; q_ < pairs_; ++q_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, (PER), 0);                                 \
        }                                                                                          \
        if ((N_DS) - pairs_ * (PER) > 0) {                                                         \
            if (pairs_ < 13) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, (N_DS) - pairs_ * (PER), 0);               \
        }                                                                                          \
        if (13 - pairs_ - ((N_DS) - pairs_ * (PER) > 0 ? 1 : 0) > 0)                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 13 - pairs_ - ((N_DS) - pairs_ * (PER) > 0 ? 1 : 0), 0); \
    }
#pragma unroll
        for (int j = 0; j < 2; ++j) { AL[j] = rd_a(Al, 0, j); AH[0][j] = rd_a(Ah, 0, j); IX[0][j] = rd_i(0, j); }
#pragma unroll
        for (int n = 0; n < NBLK; ++n) BH.put(n, rd_blk(x_h, 0, n));
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            if (ks < nks) {
                const int p = ks & 1;
                const bool last = ks + 1 >= nks;
                __builtin_amdgcn_sched_barrier(0);
                BH.pin();
                {   // phase 1: lo(G) * hi(x), while lo(x) arrives
#pragma unroll
                    for (int u = 0; u < 13; ++u) acc[u] = smfmac(AL[ct_of(u)], BH.frag(tap_of(u)), acc[u], IX[p][ct_of(u)]);
#pragma unroll
                    for (int n = 0; n < NBLK; ++n) BL.put(n, rd_blk(x_l, ks, n));
                    WS_PIN(NBLK, NBLK > 13 ? 2 : 1)
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // phase 2: hi(G) * hi(x), while the next k-step's G and index words arrive
#pragma unroll
                    for (int u = 0; u < 13; ++u) acc[u] = smfmac(AH[p][ct_of(u)], BH.frag(tap_of(u)), acc[u], IX[p][ct_of(u)]);
                    if (PREF) {         // unconditional (after the slab's last chunk it re-reads that chunk): a branch here
                                        // would put the loads in their own block BEHIND the phase's MFMAs
#pragma unroll
                        for (int q = ks * PF_PER; q < (ks + 1) * PF_PER; ++q) prefetch_item(q, nxt);
                    }
                    if (!last) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            AL[j] = rd_a(Al, ks + 1, j);
                            AH[p ^ 1][j] = rd_a(Ah, ks + 1, j);
                            IX[p ^ 1][j] = rd_i(ks + 1, j);
                        }
                        WS_PIN(10, 1)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                BL.pin();
                {   // phase 3: hi(G) * lo(x); the next k-step's hi(x) is requested in the FIRST half of the phase (phase 1
                    // of the next k-step needs all of it at once)
#pragma unroll
                    for (int u = 0; u < 13; ++u) acc[u] = smfmac(AH[p][ct_of(u)], BL.frag(tap_of(u)), acc[u], IX[p][ct_of(u)]);
                    if (!last) {
#pragma unroll
                        for (int n = 0; n < NBLK; ++n) BH.put(n, rd_blk(x_h, ks + 1, n));
                        WS_PIN(NBLK, 2)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef WS_PIN
    };

    while (rid < row_end) {
        __syncthreads();                                // everyone is done reading the previous images
        if (PREF) {
#pragma unroll
            for (int q = 0; q < NA; ++q) store_a(q, pa[q]);
#pragma unroll
            for (int q = 0; q < NI; ++q) store_i(q, pi[q]);
#pragma unroll
            for (int q = 0; q < NVB; ++q) store_b(q, pb[q]);
        } else {
            const ChunkBase cc = chunk_base(rid, ch);
#pragma unroll
            for (int q = 0; q < NA; ++q) store_a(q, load_a(q, cc));
#pragma unroll
            for (int q = 0; q < NI; ++q) store_i(q, load_i(q, cc));
#pragma unroll 4
            for (int q = 0; q < 2 * QX; ++q) store_b(q, load_b(q, cc));
        }
        const int nks = chunk_ks(ch);
        int nrid = rid, nch = ch;
        next_iter(nrid, nch);
        const bool more = nrid < row_end;
        __syncthreads();
        const ChunkBase nxt = chunk_base(more ? nrid : rid, more ? nch : ch);
        if (g) run_chunk(std::integral_constant<int, 1>{}, nks, nxt, more);
        else run_chunk(std::integral_constant<int, 0>{}, nks, nxt, more);
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
__global__ __launch_bounds__(256) void wgrad_sp_reduce_kernel(const float *__restrict__ part, int n_slabs,
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
static int launch_wgrad_sp(const WgradSpArgs &a, hipStream_t st)
{
    constexpr int KC = T >= 16 ? 3 : 6, CHP = KC * 16, WINP = CHP + 12 * T;
    const size_t lds = 2 * (size_t)(CHP * 128) + ((64 * KC * 4 + 15) / 16) * 16 + 2 * (size_t)(2 * WINP * 128);
    static MxLdsLatch latch = {};                             // per device (common.h)
    if (mx_set_dyn_lds(latch, (const void *)wgrad_sp_f16x3_kernel<T>, lds) != MX_OK) return MX_ERR_LAUNCH;
    const int groups = (a.n_slabs + 7) / 8;
    hipLaunchKernelGGL((wgrad_sp_f16x3_kernel<T>), dim3(groups * 8 * CV_KH), dim3(256), lds, st, a);
    return mx_launch_status();
}

// gp_hi/lo: (B,H/2,4,352,16) channels-last pooled gradient pair, gidx: (B,64,H/2,22,2) planar index words -- both from
// mx_conv_prep_gpool_cl_f16;  x_hi/lo: (B,H,4,352,16) operand pair of the block's forward pass (Wv <= 351 valid columns, the rest zero); part: workspace of ceil(B*(H/2)/rows_per_slab)*65*64*64
// floats (rows = pooled rows); dW (64,64,5,13).
MX_EXPORT int mx_conv_block_wgrad_sp_f16(const void *gp_hi, const void *gp_lo, const void *gidx, const void *x_hi,
                                         const void *x_lo, const float *scale, int64_t B, int64_t H, int64_t Wv,
                                         int32_t dilation, int64_t rows_per_slab, float *part, float *dW, void *stream)
{
    if (!gp_hi || !gp_lo || !gidx || !x_hi || !x_lo || !scale || !part || !dW || B <= 0 || H < 2 || (H & 1) ||
        rows_per_slab <= 0 || Wv <= 0)
        return MX_ERR_ARG;
    if (Wv > CV_PITCH - 1) return MX_ERR_UNSUPPORTED;      // the kernel reads operand column 351 as its zero source
    const int64_t n_slabs = (B * (H / 2) + rows_per_slab - 1) / rows_per_slab;
    if (n_slabs > 1000000) return MX_ERR_UNSUPPORTED;
    WgradSpArgs a{(const _Float16 *)gp_hi, (const _Float16 *)gp_lo, (const unsigned short *)gidx, (const _Float16 *)x_hi,
                  (const _Float16 *)x_lo, part, (int)B, (int)H, (int)rows_per_slab, (int)n_slabs};
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dilation) {
    case 1: rc = launch_wgrad_sp<1>(a, st); break;
    case 2: rc = launch_wgrad_sp<2>(a, st); break;
    case 4: rc = launch_wgrad_sp<4>(a, st); break;
    case 8: rc = launch_wgrad_sp<8>(a, st); break;
    case 16: rc = launch_wgrad_sp<16>(a, st); break;
    default: return MX_ERR_UNSUPPORTED;
    }
    if (rc != MX_OK) return rc;
    const int total = CV_TAPS * 64 * 64;
    hipLaunchKernelGGL(wgrad_sp_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, (int)n_slabs, scale, dW);
    return mx_launch_status();
}
