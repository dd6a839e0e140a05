// flanger.hip -- K2: mono flanger / chorus (reference: mod_extraction/fx.py:72-119).
//
// One workgroup = one clip = NINE wavefronts; the circular delay line (M <= ~34k floats) lives in LDS.
// The reference executes 88 200 dependent python iterations per batch.  Here the clip is walked in chunks of 512 samples
// (FL_V = 8 rows of 64).  For every sample the fp32 index bookkeeping of fx.py:95-103 (write slot, fractional read position,
// prev / next slot) is evaluated with exactly the reference's rounding sequence (no FMA contraction: the build uses
// -ffp-contract=off and explicit __f*_rn).  From the INTEGER slots follows, per sample k of a row, the newest sample
// t[k] = k - (distance back to the write it reads) it depends on; a run of consecutive samples [a, b) has no internal
// read-after-write dependency iff t[k] < a for all k in it, and such a run is processed by its lanes in ONE lock-step
// (all reads, then all writes: the reference's read-before-write order, fx.py:111-115).  Runs are maximal (greedy):
// chorus (delay >= 485 samples) always runs 64 samples per step, a flanger near zero delay degrades gracefully down to
// the reference's one-sample-at-a-time order.
//   FL_V PRODUCER waves (one chunk ahead, one row of the chunk each): input loads, the LFO (resampled in-kernel), the
//            index bookkeeping, the run boundaries of the row (a 64-bit mask), debug outputs -> a two-slot LDS ring of
//            float4 records.
//   CONSUMER wave: only the dependent chain -- per lock-step two ds_read_b32, five dependent fp32 operations, one
//            ds_write_b32 -- and the dry / wet mix + store of the finished chunk.
// One workgroup barrier per chunk (round 4: 8 rows per chunk instead of 4: the barrier and the consumer's per-chunk record
// loads amortise over twice the rows, 0.58 -> 0.62 of the independent floor on the headline draw).  Round 2 ran everything on one wave with one run length per 256-sample chunk:
// 1.40 ms for the slowest of 171 clips x 2 s, ~750 cycles per lock-step of which ~200 are the dependent chain.
//
// Results are bit-identical to the reference for identical mod_sig input.
// Algorithmic HBM traffic: 12 B/sample (x, mod in; y out), 8 B/sample with the 882-point LFO
// resampled in-kernel (util.py:15-29).
#include "common.h"

#ifndef FL_V
#define FL_V 8                 // rows of 64 samples per chunk (4: 0.58 of the independent floor on config 3, 6: 0.60, 8: 0.62 -- the per-chunk barrier and the consumer's record loads amortise over more rows; 12 would need 132 consumer registers)
#endif
#define FL_CHUNK (64 * FL_V)
#define FL_SLOT_FLOATS (FL_CHUNK * 6 + 2 * FL_V)          // FL_CHUNK float4 records, FL_CHUNK 64-bit lane masks (run r of a row in lane r), FL_V run counts
#define FL_RING_FLOATS (2 * FL_SLOT_FLOATS)
#define FL_MAX_M (40960 - FL_RING_FLOATS)                // 160 KB LDS = 40960 floats, minus the ring

#define FL_THREADS (64 * (1 + FL_V))   // consumer wave + one producer wave per row of a chunk
__device__ __forceinline__ unsigned lds_byte_addr(const float *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p;
}

__global__ __launch_bounds__(FL_THREADS) void flanger_kernel(
    const float *__restrict__ x, long long x_stride, const float *__restrict__ mod, int n_mod, float mod_scale,
    const float *__restrict__ lfo_scale, const float *__restrict__ min_delay,
    const float *__restrict__ feedback, const float *__restrict__ depth,
    const float *__restrict__ mix, const float *__restrict__ one_minus_mix,
    const int *__restrict__ max_delay, const int *__restrict__ rows, int N, int lfo_off, int ring_off,
    float *__restrict__ y, long long y_stride, float *__restrict__ mod_up, long long *__restrict__ dbg_prev,
    float *__restrict__ dbg_frac, int probe)
{
    extern __shared__ __attribute__((aligned(16))) float buf[];   // [M delay line | n_mod LFO row (when resampled in-kernel) | ring]
    const int lane = threadIdx.x & 63;
    const bool producer = threadIdx.x >= 64;
    const int pw = (int)(threadIdx.x >> 6) - 1;                // producer wave pw prepares row pw of every chunk
    const int b = rows ? rows[blockIdx.x] : (int)blockIdx.x;
    const int M = max_delay[b];
    const float Mf = (float)M;
    const float ls = lfo_scale[b], md = min_delay[b], fb = feedback[b], dp = depth[b];
    const float mx = mix[b], omm = one_minus_mix[b];
    const float *xb = x + (size_t)b * x_stride;
    const float *mb = mod + (size_t)b * n_mod;
    float *yb = y + (size_t)b * y_stride;
    float *ring = buf + ring_off;                              // slot s at ring + s * FL_SLOT_FLOATS
    constexpr int SLOT = FL_SLOT_FLOATS;

    for (int i = threadIdx.x; i < M; i += FL_THREADS) buf[i] = 0.0f;  // fx.py:92
    const bool resample = (n_mod != N);
    float *lfo = buf + lfo_off;                                // the short LFO row lives in LDS: no gather latency per chunk
    if (resample)
        for (int i = threadIdx.x; i < n_mod; i += FL_THREADS) lfo[i] = mb[i];
    __syncthreads();

    const int n_chunks = (N + FL_CHUNK - 1) / FL_CHUNK;
    // ---- producer state: the next chunk's inputs are prefetched while the current one is prepared
    float xr = 0.25f, mr = 0.5f;
    int w_chunk = 0;                                           // c0 % M, carried instead of a per-sample integer modulo
    if (producer) {
        const int n = pw * 64 + lane;
        xr = n < N && !probe ? xb[n] : 0.25f;
        mr = !resample && n < N && !probe ? mb[n] : 0.5f;
    }
    // records of row pw of chunk c -> ring slot c & 1
    auto build = [&](int c) {
        const int c0 = c * FL_CHUNK, j = pw;
        float4 *rec = reinterpret_cast<float4 *>(ring + (c & 1) * SLOT);
        unsigned long long *run_mask = reinterpret_cast<unsigned long long *>(ring + (c & 1) * SLOT + 4 * FL_CHUNK);
        int *n_runs = reinterpret_cast<int *>(ring + (c & 1) * SLOT + 6 * FL_CHUNK);
        float xn, mn;
        {                                                      // prefetch chunk c + 1
            const int n = c0 + FL_CHUNK + j * 64 + lane;
            xn = n < N && !probe ? xb[n] : 0.25f;
            mn = !resample && n < N && !probe ? mb[n] : 0.5f;
        }
        {
            const int n = c0 + j * 64 + lane;
            const bool valid = n < N;
            float m;
            if (resample) {
                InterpTap t = interp_tap(mod_scale, valid ? n : 0, n_mod);
                m = interp_combine(t, lfo[t.i0], lfo[t.i1]);
                if (mod_up && valid) mod_up[(size_t)b * N + n] = m;
            } else {
                m = mr;
            }
            int w = w_chunk + j * 64 + lane;                   // fx.py:95: n % M without a division
            while (w >= M) w -= M;
            const float d = __fadd_rn(__fmul_rn(ls, m), md);   // fx.py:99
            const float r1 = __fadd_rn(__fsub_rn((float)w, d), Mf);  // fx.py:100
            float r;
            if (r1 >= 0.0f && r1 < Mf) r = r1;                 // fmod is the identity here
            else if (r1 >= Mf && r1 < __fadd_rn(Mf, Mf)) r = __fsub_rn(r1, Mf);  // exact (Sterbenz)
            else r = torch_remainderf(r1, Mf);                 // out-of-contract mod_sig: generic path
            const float fl = floorf(r);
            int prev = (int)fl;                                // fx.py:102
            if (prev < 0) prev = 0;                            // NaN / garbage guard (never hit in contract)
            if (prev >= M) prev = M - 1;
            const int next = prev + 1 == M ? 0 : prev + 1;     // fx.py:103
            const float frac = __fsub_rn(r, fl);               // fx.py:101
            int dp_ = w - prev; if (dp_ <= 0) dp_ += M;        // slot w itself is "M samples ago"
            int dn_ = w - next; if (dn_ <= 0) dn_ += M;
            const int dep = valid ? min(dp_, dn_) : 0x7fffffff;
            if (dbg_prev && valid) dbg_prev[(size_t)b * N + n] = prev;
            if (dbg_frac && valid) dbg_frac[(size_t)b * N + n] = frac;
            // record: x, frac, 1 - frac (fx.py:113, rounded here exactly as there), slots (w | prev << 16; M < 65536)
            rec[j * 64 + lane] = make_float4(xr, frac, __fsub_rn(1.0f, frac), __int_as_float(w | (prev << 16)));
            // maximal dependency-free runs of the row: the run starting at a ends in front of the first k >= a whose
            // newest dependency t[k] = k - dep[k] is inside the run (t[k] >= a).  dep >= 1, so every run is non-empty.
            const int tk = dep > lane ? -1 : lane - dep;
            // Lane r keeps the lane mask of run r (lanes [a, bnd) as a 64-bit exec image): the consumer fetches a step's mask
            // with two v_readlane instead of deriving it.
            unsigned long long mine = 0ull;
            int a = 0, run = 0;
            while (a < 64) {
                const unsigned long long conflict = __ballot(lane >= a && tk >= a);
                const int bnd = conflict ? (int)__builtin_ctzll(conflict) : 64;
                if (lane == run) mine = (~0ull << a) & (~0ull >> (64 - bnd));
                a = bnd;
                ++run;
            }
            run_mask[j * 64 + lane] = mine;
            if (lane == 0) n_runs[j] = run;
        }
        xr = xn;
        mr = mn;
        w_chunk += FL_CHUNK;
        while (w_chunk >= M) w_chunk -= M;
    };

    if (producer) build(0);
    for (int c = 0; c < n_chunks; ++c) {
        __syncthreads();                                       // records of chunk c are complete; the consumer has left slot (c + 1) & 1
        if (producer) {
            if (c + 1 < n_chunks) build(c + 1);
        } else {
            const int c0 = c * FL_CHUNK;
            const float4 *rec = reinterpret_cast<const float4 *>(ring + (c & 1) * SLOT);
            const unsigned long long *run_mask = reinterpret_cast<const unsigned long long *>(ring + (c & 1) * SLOT + 4 * FL_CHUNK);
            const int *n_runs = reinterpret_cast<const int *>(ring + (c & 1) * SLOT + 6 * FL_CHUNK);
            float4 rc[FL_V];
#pragma unroll
            for (int j = 0; j < FL_V; ++j) rc[j] = rec[j * 64 + lane];
            // everything a row needs from LDS and the slot arithmetic, for all four rows, BEFORE the first lock-step loop (the
            // loops are opaque to the compiler and fence memory: whatever is left between two of them is serial time)
            unsigned m_lo_[FL_V], m_hi_[FL_V];
            int n_runs_[FL_V];
            unsigned a_prev_[FL_V], a_next_[FL_V], a_w_[FL_V];
#pragma unroll
            for (int j = 0; j < FL_V; ++j) {
                const int pk = __float_as_int(rc[j].w), w = pk & 0xffff, prev = (pk >> 16) & 0xffff;
                const int next = prev + 1 == M ? 0 : prev + 1;
                const unsigned long long m64 = run_mask[j * 64 + lane];
                m_lo_[j] = (unsigned)m64; m_hi_[j] = (unsigned)(m64 >> 32);
                n_runs_[j] = __builtin_amdgcn_readfirstlane(n_runs[j]);
                a_prev_[j] = lds_byte_addr(buf + prev); a_next_[j] = lds_byte_addr(buf + next); a_w_[j] = lds_byte_addr(buf + w);
            }
            float o[FL_V];
#pragma unroll
            for (int j = 0; j < FL_V; ++j) {
                const float xs = rc[j].x, frac = rc[j].y, omf = rc[j].z;
                const unsigned m_lo = m_lo_[j], m_hi = m_hi_[j];
                const int nr = n_runs_[j];
                const unsigned a_prev = a_prev_[j], a_next = a_next_[j], a_w = a_w_[j];
                // The lock-step loop, written out.  Run r executes under its lane mask, which the producer wave left in lane r of
                // (m_lo, m_hi): two v_readlane into vcc, issued with the step counter behind the two reads of the step before,
                // inside their LDS latency.  The dependent path of a lock-step is read -> 5 fp32 operations -> write ->
                // s_mov exec -> not-taken branch -> read.  The compiler's own versions of this loop (selects onto a private dummy
                // slot, or an exec region per step) put 25-30 instructions there.  A lane is active in exactly one run of its
                // row, so `it` keeps each lane's own interpolated value.
                float it = 0.0f;
                unsigned long long sv;
                int k;
// (the two independent products of fx.py:113 as ONE packed multiplication on fixed register pairs -- v[84:85] = {next, prev},
//  v[86:87] = {frac, 1 - frac}: one instruction less on the dependent path of every lock-step; each product is still its own
//  IEEE multiplication, so the waveform stays bit-identical).  The pair registers are named because AMDGPU inline assembly has
//  no operand modifier for one half of a 64-bit operand (ds_read_b32 must land IN a half of the pair v_pk_mul_f32 consumes); they
//  are declared as clobbers, so the register allocator keeps every operand of this statement and every live value out of
//  v84-v87 whatever FL_V or the register budget is -- the choice of numbers affects nothing but the allocation around the loop.)
#define FL_LOCK_STEP                                                                                                   \
    "ds_read_b32 v85, %[ap]\n"        /* fx.py:111 */                                                                 \
    "ds_read_b32 v84, %[an]\n"        /* fx.py:112 */                                                                 \
    "s_add_i32 %[k], %[k], 1\n"                                                                                        \
    "s_cmp_lt_i32 %[k], %[nr]\n"                                                                                       \
    "v_readlane_b32 vcc_lo, %[mlo], %[k]\n" /* lane nr (mod 64) after the last run: unused */                          \
    "v_readlane_b32 vcc_hi, %[mhi], %[k]\n"                                                                            \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    "v_pk_mul_f32 v[84:85], v[86:87], v[84:85]\n"                                                                \
    "v_add_f32 %[it], v84, v85\n"    /* fx.py:113 */                                                                 \
    "v_mul_f32 v85, %[fb], %[it]\n"                                                                                   \
    "v_add_f32 v85, %[xs], v85\n"                                                                                    \
    "ds_write_b32 %[aw], v85\n"       /* fx.py:114 */                                                                 \
    "s_mov_b64 exec, vcc\n"
                // eight lock-steps per taken branch (a not-taken exit branch costs an issue slot, a taken one refills the
                // instruction buffer on the dependent path)
                asm volatile(
                    "s_mov_b64 %[sv], exec\n"
                    "v_mov_b32 v86, %[frac]\n"
                    "v_mov_b32 v87, %[omf]\n"
                    "s_mov_b32 %[k], 0\n"
                    "v_readlane_b32 vcc_lo, %[mlo], 0\n"
                    "v_readlane_b32 vcc_hi, %[mhi], 0\n"
                    "s_nop 3\n"
                    "s_mov_b64 exec, vcc\n"
                    "Lfl_step_%=:\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc0 Lfl_done_%=\n"
                    FL_LOCK_STEP "s_cbranch_scc1 Lfl_step_%=\n"
                    "Lfl_done_%=:\n"
                    "s_mov_b64 exec, %[sv]\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    : [it] "+v"(it), [sv] "=&s"(sv), [k] "=&s"(k)
                    : [ap] "v"(a_prev), [an] "v"(a_next), [aw] "v"(a_w), [frac] "v"(frac), [omf] "v"(omf), [xs] "v"(xs), [fb] "s"(fb),
                      [mlo] "v"(m_lo), [mhi] "v"(m_hi), [nr] "s"(nr)
                    : "memory", "scc", "vcc", "v84", "v85", "v86", "v87");
                const float oj = __fadd_rn(xs, __fmul_rn(dp, it));                                   // fx.py:115
                o[j] = oj;
            }
#pragma unroll
            for (int j = 0; j < FL_V; ++j) {
                const int n = c0 + j * 64 + lane;
                if (n < N && (!probe || c + 1 == n_chunks)) {  // probe: only the last chunk is stored (keeps the chain live)
                    const float v = __fadd_rn(__fmul_rn(omm, rc[j].x), __fmul_rn(mx, o[j]));         // fx.py:117
                    yb[n] = fminf(fmaxf(v, -1.0f), 1.0f);                                           // fx.py:118
                }
            }
        }
    }
}

// C ABI ---------------------------------------------------------------------------------------
// x: row b at x + b*x_stride (N samples); y likewise with y_stride; mod (B,n_mod) with n_mod == N or any shorter length (resampled in-kernel,
// align_corners=True); per-clip fp32 constants lfo_scale = max_lfo_delay_samples*width,
// min_delay = min_delay_width*max_min_delay_samples, feedback, depth, mix, one_minus_mix;
// max_delay (B,) int32 = delay-line length M per clip (flanger and chorus clips may be mixed in
// one batch); rows: optional list of n_rows clip indices to process (others untouched).
// Optional outputs: mod_up (B,N) resampled LFO; dbg_prev (B,N) int64 / dbg_frac (B,N) for the
// index-parity tests.
static int flanger_fwd_launch(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream, int probe)
{
    if (!x || !mod || !lfo_scale || !min_delay || !feedback || !depth || !mix || !one_minus_mix ||
        !max_delay || !y || B <= 0 || N <= 0 || n_mod <= 0)
        return MX_ERR_ARG;
    if (max_delay_max < 2 || x_stride < N || y_stride < N) return MX_ERR_ARG;
    if (max_delay_max > FL_MAX_M || max_delay_max > 65535 || N >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const int64_t items = rows ? n_rows : B;
    if (items <= 0) return MX_OK;
    static bool attr_set[64] = {};                       // per device: one process may drive several GPUs
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)flanger_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (FL_MAX_M + FL_RING_FLOATS) * sizeof(float));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    // LDS: delay line (max over the batch) + the LFO row when it is resampled in-kernel (n_mod < N) + the record ring
    const int lfo_off = max_delay_max;
    size_t lds_floats = (size_t)max_delay_max + (n_mod != N ? (size_t)n_mod : 0);
    lds_floats = (lds_floats + 3) & ~(size_t)3;                    // the ring holds float4 records
    const int ring_off = (int)lds_floats;
    if (lds_floats > FL_MAX_M) return MX_ERR_UNSUPPORTED;
    const size_t lds = (lds_floats + FL_RING_FLOATS) * sizeof(float);
    hipLaunchKernelGGL(flanger_kernel, dim3((unsigned)items), dim3(FL_THREADS), lds, (hipStream_t)stream, x, (long long)x_stride, mod,
                       (int)n_mod, interp_scale_host(n_mod, N), lfo_scale, min_delay, feedback, depth,
                       mix, one_minus_mix, max_delay, rows, (int)N, lfo_off, ring_off, y, (long long)y_stride, mod_up,
                       (long long *)dbg_prev, dbg_frac, probe);
    return mx_launch_status();
}

MX_EXPORT int mx_flanger_fwd(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream)
{
    return flanger_fwd_launch(x, x_stride, mod, n_mod, lfo_scale, min_delay, feedback, depth, mix, one_minus_mix, max_delay, max_delay_max, rows, n_rows, B, N, y, y_stride, mod_up, dbg_prev, dbg_frac, stream, 0);
}

// Measurement twin (bench.py's serial floor): the SAME launch with no global-memory traffic inside the sample loop -- inputs are constants, only the last chunk is stored.  Results are meaningless; nothing in the product calls it.
MX_EXPORT int mx_flanger_fwd_probe(const float *x, int64_t x_stride, const float *mod, int64_t n_mod, const float *lfo_scale,
                             const float *min_delay, const float *feedback, const float *depth,
                             const float *mix, const float *one_minus_mix, const int32_t *max_delay,
                             int32_t max_delay_max, const int32_t *rows, int64_t n_rows, int64_t B,
                             int64_t N, float *y, int64_t y_stride, float *mod_up, int64_t *dbg_prev, float *dbg_frac,
                             void *stream)
{
    return flanger_fwd_launch(x, x_stride, mod, n_mod, lfo_scale, min_delay, feedback, depth, mix, one_minus_mix, max_delay, max_delay_max, rows, n_rows, B, N, y, y_stride, mod_up, dbg_prev, dbg_frac, stream, 1);
}

// ---- measurement aid: the LDS round trip of one lock-step -------------------------------------------------------------
// One wavefront runs `steps` dependent lock-steps of the flanger's shape on a private LDS array -- two ds_read_b32 of the
// slot written by the previous step, the five fp32 operations of fx.py:113-115, one ds_write_b32 -- with no index
// bookkeeping, no run logic and no global traffic.  bench.py times it with HIP events: (duration / steps) x the lock-steps
// of the slowest clip (counted on the host from the integer slot bookkeeping, tools/flanger_hops.py) is a floor of the
// flanger launch that does NOT come from the flanger kernel itself.  out[0] receives the last value (keeps the chain live).
__global__ __launch_bounds__(64) void lds_roundtrip_kernel(int steps, int stride, float fb, float *__restrict__ out)
{
    __shared__ float line[128];
    const int lane = threadIdx.x;
    line[lane] = 0.5f;
    line[64 + lane] = 0.25f;
    __builtin_amdgcn_wave_barrier();
    float o = 0.0f;
    int slot = lane;
    for (int s = 0; s < steps; ++s) {
        // `stride` is 0 at run time, which the compiler cannot know: it has to issue the reads after the previous step's
        // write (same slot -> a true LDS read-after-write round trip) instead of forwarding the value in a register
        const float pv = line[slot], nv = line[slot + 64];
        const float it = __fadd_rn(__fmul_rn(0.375f, nv), __fmul_rn(0.625f, pv));
        line[slot] = __fadd_rn(0.125f, __fmul_rn(fb, it));
        o = __fadd_rn(0.125f, __fmul_rn(0.5f, it));
        slot = (slot + stride) & 63;
    }
    if (lane == 0) out[0] = o;
}

// steps dependent LDS round trips on one wavefront of one workgroup (see above); out: 1 float
MX_EXPORT int mx_lds_roundtrip_probe(int64_t steps, float *out, void *stream)
{
    if (!out || steps <= 0 || steps >= (1ll << 30)) return MX_ERR_ARG;
    hipLaunchKernelGGL(lds_roundtrip_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)steps, 0, 0.5f, out);
    return mx_launch_status();
}
