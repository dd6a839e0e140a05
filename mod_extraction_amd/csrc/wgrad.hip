// wgrad.hip -- K6 (weight gradient) of the Spectral2DCNN convolutions
// (reference: torch.nn.Conv2d backward w.r.t. weight, mod_extraction/models.py:187).
//
//   dW[co][ci][kh][kw] = sum over (b, h, w) of  dz[b][co][h][w] * xhat[b][ci][h + kh - 2][w + (kw - 6) T]
//
// GEMM view: M = co (64), N = (ci, kw) for one kh, K = (b, h, w) positions -- the reduction runs
// over the whole batch, so K is split into "slabs" of consecutive (b, h) rows; each workgroup
// accumulates one slab for one (kh, kw-half) in registers (fp32 MFMA 32x32x2, 6-7 accumulators
// per wave) and writes a partial [kh][kw][co][ci] tile set; a second kernel sums the slabs in
// fp64 (deterministic, no atomics) and transposes to torch's (Cout, Cin, 5, 13) layout.
// dz and xhat are built on the fly while staging 88-position chunks (32 for T = 16) into LDS, transposed to
// [position][channel] (pitch 65 floats) so that both MFMA operand reads are 32 consecutive floats:
//   dz   = max-pool routing of the pooled gradient G via the stored argmax
//   xhat = (prelu(p_prev) - mean) * rstd  (or (logmel - mean) * rstd for the first block)
// MFMA-bound like the forward pass (same flop count).
#include "conv_common.h"

#define WG_CHUNK 32
#define WG_LP 65          // LDS pitch (floats) of the [position][channel] tiles
// -DWG_DIAG builds a diagnostic variant (tools/diag_wgrad.py) that stamps s_memtime around the phases of
// wave 0 of every workgroup; the shipped library never defines it.
#ifdef WG_DIAG
#define WG_STAMP(slot)                                                                           \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        if (tid == 0) { diag_acc[slot] += t_ - diag_last; }                                      \
        diag_last = t_;                                                                          \
    } while (0)
#else
#define WG_STAMP(slot)
#endif

struct WgradArgs {
    const float *G;             // (B, 64, H/2, PITCH) gradient w.r.t. pooled pre-activations
    const unsigned char *amax;  // (B, 64, H/2, PITCH)
    const float *x;             // (B, Cin, H, PITCH) block input before PReLU/LayerNorm
    const float *stats;         // (B, Cin, 2)
    const float *slope;         // (Cin,) or nullptr (first block)
    float *part;                // (n_slabs, 5, 13, 64, Cin_pad) partial sums
    int B, H, Wv, rows_per_slab;
#ifdef WG_DIAG
    unsigned long long *diag;   // (n_workgroups, 4) cycles: barrier-wait, store, issue+barrier, mfma
#endif
};

// ---- blocks 2..6: Cin = 64 ---------------------------------------------------------------------
// grid (10, n_slabs): blockIdx.x = kh * 2 + kw_half (kw 0..6 | 7..12); 4 waves = (co tile, ci tile).
// Per 32-position chunk: global loads of the NEXT chunk are issued into registers before the MFMAs of
// the current one (latency hidden behind ~7k cycles of matrix work), the LDS tiles are rewritten
// between two barriers, and the MFMA loop is ping-pong pipelined like the forward kernel.
template <int T>
__global__ __launch_bounds__(256, 2) void wgrad64_kernel(WgradArgs a)
{
    constexpr int HALO = cv_halo(T);
    constexpr int CH = T <= 2 ? 88 : (T <= 8 ? 44 : 32);   // positions per chunk (352 = 4 x 88 = 8 x 44 = 11 x 32):
                                                   // longer MFMA phases between barriers where registers / LDS allow
    constexpr int WIN = CH + 2 * HALO;             // staged xhat positions per chunk
    constexpr int WIN4 = WIN / 4;
    constexpr int NXV = (64 * WIN4 + 255) / 256;   // xhat float4 loads per thread per chunk
    constexpr int NCH = CV_PITCH / CH;             // chunks per row
    constexpr int NDZ = (64 * (CH / 4) + 255) / 256; // dz float4 loads per thread per chunk
    constexpr bool PREF = T < 16;                  // T = 16 would need 14 prefetch vectors per thread (spills)
    __shared__ float dzl[CH * WG_LP];
    __shared__ float xl[WIN * WG_LP];
    __shared__ float st_l[3 * 64];                 // mean | rstd | slope of the clip being staged
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave & 1, cb = wave >> 1, half = lane >> 5, l32 = lane & 31;
    const int kh = blockIdx.x >> 1, kwh = blockIdx.x & 1;
    const int kw0 = kwh ? 7 : 0, nkw = kwh ? 6 : 7;
    const int Hp = a.H >> 1;

    floatx16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int row_begin = blockIdx.y * a.rows_per_slab;
    int row_end = row_begin + a.rows_per_slab;
    if (row_end > a.B * a.H) row_end = a.B * a.H;

    // iteration = (row id, chunk); rows whose xhat row (h + kh - 2) is padding contribute nothing
    auto row_valid = [&](int rid) {
        const int h = rid % a.H, hx = h + kh - 2;
        return hx >= 0 && hx < a.H;
    };
    auto next_iter = [&](int &rid, int &ch) {          // advance to the next valid (rid, chunk)
        if (++ch < NCH) return;
        ch = 0;
        do { ++rid; } while (rid < row_end && !row_valid(rid));
    };

    floatx4 pg[NDZ], px[PREF ? NXV : 1];
    uchar4 pa[NDZ];
    auto issue = [&](int rid, int ch) {                 // global loads of one chunk into registers
        if (!PREF) return;
        const int b = rid / a.H, h = rid - b * a.H, hx = h + kh - 2, w0 = ch * CH;
#pragma unroll
        for (int q = 0; q < NDZ; ++q) {
            const int i = tid + q * 256, co = i / (CH / 4), c4 = i - co * (CH / 4);
            if (i < 64 * (CH / 4)) {
                const size_t off = (((size_t)b * 64 + co) * Hp + (h >> 1)) * CV_PITCH + w0 + c4 * 4;
                pg[q] = *reinterpret_cast<const floatx4 *>(a.G + off);
                pa[q] = *reinterpret_cast<const uchar4 *>(a.amax + off);
            }
        }
#pragma unroll
        for (int q = 0; q < (PREF ? NXV : 0); ++q) {
            const int i = tid + q * 256, ci = i / WIN4, c4 = i - ci * WIN4;
            const int wq = w0 - HALO + c4 * 4;
            floatx4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (i < 64 * WIN4 && wq >= 0 && wq < CV_PITCH)
                v = *reinterpret_cast<const floatx4 *>(a.x + (((size_t)b * 64 + ci) * a.H + hx) * CV_PITCH + wq);
            px[q] = v;
        }
    };
    auto store = [&](int rid, int ch) {                 // transform + transposed LDS writes
        if (!PREF) return;
        const int b = rid / a.H, h = rid - b * a.H, w0 = ch * CH;
        const unsigned want = (unsigned)(h & 1);
#pragma unroll
        for (int q = 0; q < NDZ; ++q) {
            const int i = tid + q * 256, co = i / (CH / 4), c4 = i - co * (CH / 4);
            if (i < 64 * (CH / 4)) {
                const int wq = w0 + c4 * 4;
                float *d = dzl + (c4 * 4) * WG_LP + co;
                d[0] = (pa[q].x == want && wq + 0 < a.Wv) ? pg[q][0] : 0.0f;
                d[WG_LP] = (pa[q].y == want && wq + 1 < a.Wv) ? pg[q][1] : 0.0f;
                d[2 * WG_LP] = (pa[q].z == want && wq + 2 < a.Wv) ? pg[q][2] : 0.0f;
                d[3 * WG_LP] = (pa[q].w == want && wq + 3 < a.Wv) ? pg[q][3] : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < (PREF ? NXV : 0); ++q) {
            const int i = tid + q * 256, ci = i / WIN4, c4 = i - ci * WIN4;
            if (i < 64 * WIN4) {
                const int wq = w0 - HALO + c4 * 4;
                floatx4 v = px[q];
                if (wq >= 0 && wq < CV_PITCH) {
                    const float mean = st_l[ci], rstd = st_l[64 + ci], sl = st_l[128 + ci];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e] > 0.0f ? v[e] : sl * v[e];
                        v[e] = wq + e < a.Wv ? (t - mean) * rstd : 0.0f;
                    }
                }
                float *d = xl + (c4 * 4) * WG_LP + ci;
                d[0] = v[0]; d[WG_LP] = v[1]; d[2 * WG_LP] = v[2]; d[3 * WG_LP] = v[3];
            }
        }
    };

    int rid = row_begin, ch = 0;
    while (rid < row_end && !row_valid(rid)) ++rid;
    if (PREF && rid < row_end) issue(rid, ch);
    const float *ab = dzl + half * WG_LP + mt * 32 + l32;
    const float *bb = xl + (HALO + half + (kw0 - 6) * T) * WG_LP + cb * 32 + l32;
    int b_cached = -1;
#ifdef WG_DIAG
    unsigned long long diag_acc[4] = {0, 0, 0, 0}, diag_last = 0;
    WG_STAMP(3);
    diag_acc[3] = 0;
#endif
    while (rid < row_end) {
        __syncthreads();                                // everyone is done reading the previous tiles
        WG_STAMP(0);
        if (rid / a.H != b_cached) {                    // new clip: its LayerNorm statistics / slopes -> LDS
            b_cached = rid / a.H;
            if (tid < 64) {
                st_l[tid] = a.stats[((size_t)b_cached * 64 + tid) * 2];
                st_l[64 + tid] = a.stats[((size_t)b_cached * 64 + tid) * 2 + 1];
                st_l[128 + tid] = a.slope[tid];
            }
            __syncthreads();
        }
        if (PREF) {
            store(rid, ch);
        } else {
            // large dilation: the staged window is 7x the chunk, keep the loads in a loop (no register arrays)
            const int b = rid / a.H, h = rid - b * a.H, hx = h + kh - 2, w0 = ch * CH;
            const unsigned want = (unsigned)(h & 1);
            for (int i = tid; i < 64 * (CH / 4); i += 256) {
                const int co = i / (CH / 4), c4 = i - co * (CH / 4);
                const size_t off = (((size_t)b * 64 + co) * Hp + (h >> 1)) * CV_PITCH + w0 + c4 * 4;
                const floatx4 g = *reinterpret_cast<const floatx4 *>(a.G + off);
                const uchar4 am = *reinterpret_cast<const uchar4 *>(a.amax + off);
                const int wq = w0 + c4 * 4;
                float *d = dzl + (c4 * 4) * WG_LP + co;
                d[0] = (am.x == want && wq + 0 < a.Wv) ? g[0] : 0.0f;
                d[WG_LP] = (am.y == want && wq + 1 < a.Wv) ? g[1] : 0.0f;
                d[2 * WG_LP] = (am.z == want && wq + 2 < a.Wv) ? g[2] : 0.0f;
                d[3 * WG_LP] = (am.w == want && wq + 3 < a.Wv) ? g[3] : 0.0f;
            }
            for (int i = tid; i < 64 * WIN4; i += 256) {
                const int ci = i / WIN4, c4 = i - ci * WIN4;
                const int wq = w0 - HALO + c4 * 4;
                floatx4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (wq >= 0 && wq < CV_PITCH) {
                    v = *reinterpret_cast<const floatx4 *>(a.x + (((size_t)b * 64 + ci) * a.H + hx) * CV_PITCH + wq);
                    const float mean = st_l[ci], rstd = st_l[64 + ci], sl = st_l[128 + ci];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e] > 0.0f ? v[e] : sl * v[e];
                        v[e] = wq + e < a.Wv ? (t - mean) * rstd : 0.0f;
                    }
                }
                float *d = xl + (c4 * 4) * WG_LP + ci;
                d[0] = v[0]; d[WG_LP] = v[1]; d[2 * WG_LP] = v[2]; d[3 * WG_LP] = v[3];
            }
        }
        WG_STAMP(1);
        int nrid = rid, nch = ch;
        next_iter(nrid, nch);
        if (PREF && nrid < row_end) issue(nrid, nch);   // in flight during the MFMAs below
        __syncthreads();
        WG_STAMP(2);
        __builtin_amdgcn_s_setprio(1);                  // matrix phase outranks the partner workgroup's staging VALU
        float a0, a1, b0[7], b1[7];
#define WG_LOAD(A, B, KS)                                                                    \
    A = ab[2 * (KS) * WG_LP];                                                                \
    _Pragma("unroll") for (int i = 0; i < 7; ++i) B[i] = i < nkw ? bb[(2 * (KS) + i * T) * WG_LP] : 0.0f;
#define WG_MMA(A, B) _Pragma("unroll") for (int i = 0; i < 7; ++i) if (i < nkw) acc[i] = mfma32(A, B[i], acc[i]);
        WG_LOAD(a0, b0, 0)
#pragma unroll 1
        for (int ks = 0; ks < CH / 2; ks += 2) {
            WG_LOAD(a1, b1, ks + 1)
            __builtin_amdgcn_sched_barrier(0);
            WG_MMA(a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < CH / 2) { WG_LOAD(a0, b0, ks + 2) }
            __builtin_amdgcn_sched_barrier(0);
            WG_MMA(a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WG_LOAD
#undef WG_MMA
        __builtin_amdgcn_s_setprio(0);
        WG_STAMP(3);
        rid = nrid;
        ch = nch;
    }
#ifdef WG_DIAG
    if (tid == 0 && a.diag) {
        unsigned long long *o = a.diag + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        o[0] = diag_acc[0]; o[1] = diag_acc[1]; o[2] = diag_acc[2]; o[3] = diag_acc[3];
    }
#endif
    // partial tiles: part[slab][kh][kw][co][ci]
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (i < nkw) {
            float *dst = a.part + ((((size_t)blockIdx.y * CV_KH + kh) * CV_KW + kw0 + i) * 64) * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(mt * 32 + mfma_row(r, lane)) * 64 + cb * 32 + l32] = acc[i][r];
        }
    }
}

// ---- block 1: Cin = 2 --------------------------------------------------------------------------
// N tile = (ci, kw) -> 26 of 32 columns; one workgroup covers all 5 kh for its slab, so G is read
// once.  4 waves = (co tile, k-step parity); the two parities are summed through LDS at the end.
// part layout: (n_slabs, 5, 13, 64, 2).
#define W1_CHUNK 128
template <int T>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(WgradArgs a)
{
    constexpr int HALO = cv_halo(T);
    constexpr int WIN = W1_CHUNK + 2 * HALO;
    constexpr int WIN4 = WIN / 4;
    __shared__ float dzl[W1_CHUNK * WG_LP];            // 33 KB
    __shared__ float xl[2 * CV_KH * WIN];              // [ci][kh][pos]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave & 1, par = wave >> 1, half = lane >> 5, l32 = lane & 31;
    const int Hp = a.H >> 1;
    const int jci = l32 / CV_KW, jkw = l32 - jci * CV_KW;      // column j -> (ci, kw); j >= 26 unused
    const bool jvalid = l32 < 2 * CV_KW;

    floatx16 acc[CV_KH];
#pragma unroll
    for (int i = 0; i < CV_KH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int row_begin = blockIdx.x * a.rows_per_slab;
    int row_end = row_begin + a.rows_per_slab;
    if (row_end > a.B * a.H) row_end = a.B * a.H;
    for (int rid = row_begin; rid < row_end; ++rid) {
        const int b = rid / a.H, h = rid - b * a.H;
        for (int w0 = 0; w0 < CV_PITCH; w0 += W1_CHUNK) {
            __syncthreads();
            for (int i = tid; i < 64 * (W1_CHUNK / 4); i += 256) {
                const int co = i / (W1_CHUNK / 4), c4 = i - co * (W1_CHUNK / 4);
                const int wq = w0 + c4 * 4;
                floatx4 g = {0.0f, 0.0f, 0.0f, 0.0f};
                uchar4 am = {2, 2, 2, 2};
                if (wq < CV_PITCH) {
                    const size_t off = (((size_t)b * 64 + co) * Hp + (h >> 1)) * CV_PITCH + wq;
                    g = *reinterpret_cast<const floatx4 *>(a.G + off);
                    am = *reinterpret_cast<const uchar4 *>(a.amax + off);
                }
                const unsigned want = (unsigned)(h & 1);
                float *d = dzl + (c4 * 4) * WG_LP + co;
                d[0] = (am.x == want && wq + 0 < a.Wv) ? g[0] : 0.0f;
                d[WG_LP] = (am.y == want && wq + 1 < a.Wv) ? g[1] : 0.0f;
                d[2 * WG_LP] = (am.z == want && wq + 2 < a.Wv) ? g[2] : 0.0f;
                d[3 * WG_LP] = (am.w == want && wq + 3 < a.Wv) ? g[3] : 0.0f;
            }
            for (int i = tid; i < 2 * CV_KH * WIN4; i += 256) {
                const int rowid = i / WIN4, c4 = i - rowid * WIN4;
                const int ci = rowid / CV_KH, kh = rowid - ci * CV_KH;
                const int hx = h + kh - 2, wq = w0 - HALO + c4 * 4;
                floatx4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (hx >= 0 && hx < a.H && wq >= 0 && wq < CV_PITCH) {
                    v = *reinterpret_cast<const floatx4 *>(a.x + (((size_t)b * 2 + ci) * a.H + hx) * CV_PITCH + wq);
                    const float mean = a.stats[((size_t)b * 2 + ci) * 2], rstd = a.stats[((size_t)b * 2 + ci) * 2 + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e];
                        if (a.slope) t = t > 0.0f ? t : a.slope[ci] * t;
                        v[e] = wq + e < a.Wv ? (t - mean) * rstd : 0.0f;
                    }
                }
                *reinterpret_cast<floatx4 *>(xl + rowid * WIN + c4 * 4) = v;
            }
            __syncthreads();
            const float *ab = dzl + half * WG_LP + mt * 32 + l32;
            const float *bb = xl + jci * (CV_KH * WIN) + HALO + half + (jkw - 6) * T;
            for (int ks = par; ks < W1_CHUNK / 2; ks += 2) {
                const float av = ab[2 * ks * WG_LP];
#pragma unroll
                for (int kh = 0; kh < CV_KH; ++kh) {
                    const float bv = jvalid ? bb[kh * WIN + 2 * ks] : 0.0f;
                    acc[kh] = mfma32(av, bv, acc[kh]);
                }
            }
        }
    }
    // sum the two k-step parities through LDS, then write part[slab][kh][kw][co][ci]
    __syncthreads();
    float *xch = dzl;                                  // 2 mt x 5 kh x 16 regs x 64 lanes = 40 KB > dzl: do it per kh
#pragma unroll
    for (int kh = 0; kh < CV_KH; ++kh) {
        __syncthreads();
        if (par == 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(mt * 16 + r) * 64 + lane] = acc[kh][r];
        __syncthreads();
        if (par == 0 && jvalid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[kh][r] + xch[(mt * 16 + r) * 64 + lane];
                const int co = mt * 32 + mfma_row(r, lane);
                a.part[((((size_t)blockIdx.x * CV_KH + kh) * CV_KW + jkw) * 64 + co) * 2 + jci] = v;
            }
        }
    }
}

// dW[co][ci][kh][kw] = sum over slabs of part[slab][kh][kw][co][ci]   (fp64 accumulate)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int n_slabs, int Cin,
                                                           float *__restrict__ dW)
{
    const int total = CV_TAPS * 64 * Cin;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    double s = 0.0;
    for (int k = 0; k < n_slabs; ++k) s += (double)part[(size_t)k * total + j];
    const int ci = j % Cin, co = (j / Cin) % 64, tap = j / (Cin * 64);
    dW[((size_t)co * Cin + ci) * CV_TAPS + tap] = (float)s;
}

template <int T>
static int launch_wgrad(const WgradArgs &a, int Cin, int n_slabs, hipStream_t st)
{
    if (Cin == 64)
        hipLaunchKernelGGL((wgrad64_kernel<T>), dim3(10, n_slabs), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((wgrad2_kernel<T>), dim3(n_slabs), dim3(256), 0, st, a);
    return mx_launch_status();
}

#ifdef WG_DIAG
static unsigned long long *g_wgrad_diag = nullptr;
MX_EXPORT void mx_diag_set_buffer(unsigned long long *p) { g_wgrad_diag = p; }
#endif
// G, amax: (B,64,H/2,352); x: (B,Cin,H,352) block input (pre PReLU / LayerNorm); stats (B,Cin,2);
// slope (Cin,) or NULL for the first block; part: workspace of n_slabs*65*64*Cin floats with
// n_slabs = ceil(B*H / rows_per_slab); dW: (64, Cin, 5, 13) torch layout (overwritten).
MX_EXPORT int mx_conv_block_wgrad(const float *G, const uint8_t *amax, const float *x, const float *stats,
                                  const float *slope, int64_t B, int64_t Cin, int64_t H, int64_t Wv,
                                  int32_t dilation, int64_t rows_per_slab, float *part, float *dW, void *stream)
{
    if (!G || !amax || !x || !stats || !part || !dW || B <= 0 || rows_per_slab <= 0) return MX_ERR_ARG;
    if ((Cin != 64 && Cin != 2) || (Cin == 64 && !slope) || H < 2 || (H & 1) || Wv <= 0 || Wv > CV_PITCH)
        return MX_ERR_UNSUPPORTED;
    const int64_t n_slabs = (B * H + rows_per_slab - 1) / rows_per_slab;
    if (n_slabs > 65535) return MX_ERR_UNSUPPORTED;
#ifdef WG_DIAG
    WgradArgs a{G, amax, x, stats, slope, part, (int)B, (int)H, (int)Wv, (int)rows_per_slab, g_wgrad_diag};
#else
    WgradArgs a{G, amax, x, stats, slope, part, (int)B, (int)H, (int)Wv, (int)rows_per_slab};
#endif
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dilation) {
    case 1: rc = launch_wgrad<1>(a, (int)Cin, (int)n_slabs, st); break;
    case 2: rc = launch_wgrad<2>(a, (int)Cin, (int)n_slabs, st); break;
    case 4: rc = launch_wgrad<4>(a, (int)Cin, (int)n_slabs, st); break;
    case 8: rc = launch_wgrad<8>(a, (int)Cin, (int)n_slabs, st); break;
    case 16: rc = launch_wgrad<16>(a, (int)Cin, (int)n_slabs, st); break;
    default: return MX_ERR_UNSUPPORTED;
    }
    if (rc != MX_OK) return rc;
    const int total = CV_TAPS * 64 * (int)Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, (int)n_slabs, (int)Cin,
                       dW);
    return mx_launch_status();
}
