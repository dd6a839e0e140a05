// lstm.hip -- K10: the sample-wise LSTM-64 effect model (reference: mod_extraction/models.py:311-339,
// nn.LSTM(2, 64, batch_first) -> Linear(64, 1) -> + x -> tanh) and its truncated BPTT with the L1
// loss fused (lightning.py:355-384: one 1024-sample chunk = one forward, one backward, one
// optimizer step, hidden state carried and detached between chunks).
//
// The recurrence is a chain of T dependent steps per clip, each a 256x64 mat-vec; at the shipped batch
// (128 clips) the chip has more CUs than clips, so a launch lasts T x (latency of one step) and everything
// here is about that latency.  Measured on the part (tools/probe/ubench_valu.hip): a wave issues one vector
// instruction per ~4.8 cycles whatever its neighbours on the SIMD do (up to 4 waves per SIMD), and an
// LDS write -> barrier -> read round trip costs ~250 cycles with 8 waves: a step costs
// (instructions per wave) x 2 ns + ~100 ns.  Hence: one 512-thread workgroup (8 waves) per clip, as few
// instructions per wave and step as possible, exactly ONE barrier per step.
//
//  forward   lane = (unit u, gate pair gp, quarter kq of the 64 h values): 2 gates x 16 k = 16 v_pk_fma_f32
//            (both gates of the pair in one packed FMA, the h value broadcast to both halves by op_sel), four
//            ds_read_b128 of h per lane.  The quarters are summed by two DPP quad_perm adds per gate, every lane
//            applies ONE gate's activation (one exp + one rcp), the 8 lanes of a unit exchange i, f, g, o by DPP
//            (row_half_mirror + quad_perm), c stays in registers, h goes to the next row of a 256-step history in
//            LDS -- rows never collide, hence one barrier per step.  y_t = tanh(fc h_t + b + x_t) is not on the
//            recurrent path: it is evaluated for 256 steps at a time from the history, one step per thread.
//  backward  (serial part) lane = (pair of hidden units, group rg of 16 gate rows): dh_prev[k] = sum_r W[r][k] dg[r]
//            as 16 v_pk_fma_f32 (two units per packed FMA), a reduction over the 16 groups by DPP (the first level hands the
//            partner the unit IT needs: 2 selects + 4 adds), and the element-wise gate derivatives computed by the lanes rg < 8
//            (one (unit, gate) each) from a slab of the forward stash that is staged through LDS 32 steps at a time (coalesced,
//            double buffered; tanh(c) is added to the slab as a seventh plane by one pass per slab).  Round 6: the step is
//            unrolled over a full slab -- every LDS address is a per-slab lane register + an immediate, the gate gradients live
//            in a ring of four buffers (t & 3) -- and the backward step turned out to be bound by the SIMD's vector ISSUE
//            (two recurrence waves per SIMD), not by latency.
//            The weight gradients do not feed the recurrence: dW_hh = sum_t dg_t (x) h_{t-1} accumulates in the SAME
//            kernel as exact-fp32 products (v_mfma_f32_32x32x2_f32 -- which executes on the SIMD's fp32 vector ALUs), on the
//            gate gradients of steps t+1, t+2 that sit in LDS anyway and the h rows of the stash slab.  Round 6: FOUR HELPER
//            WAVES of the workgroup (one per SIMD, 64 gate rows = four tiles each) issue them, so that the in-order recurrence
//            waves do not stall behind two 64-cycle instructions; dW_ih / biases ride along on the A values, the output layer's
//            gradients on helper 0.  (First version: gate gradients written to HBM and a separate GEMM kernel per chunk,
//            0.10-0.14 ms; rounds 2-5: inside the recurrence waves.)  The variant that also leaves the gate gradients
//            (mx_lstm_bwd_dgate) keeps the round-5 shape: rolled loop, products in the recurrence waves.
// One gradient row per clip (state-dict order) is summed over the batch by mx_reduce_rows (deterministic, no atomics).
//
// Activations use v_exp_f32 / v_rcp_f32 (1 ulp each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)),
// tanh(x) = 2 sigmoid(2x) - 1 (absolute error ~1e-7, the recurrent path carries 1e-5 parity, tests/test_gpu_lstm.py).
// Algorithmic HBM traffic: 12 B/sample I/O + 1536 B/sample stash (written, read once).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

#define LS_H 64
#define LS_STASH 384        // floats per time step: gates 256 (i, f, g, o) + c 64 + h 64
#define LS_TB 256           // steps per history block (forward)
#define LS_HP 68            // history row pitch (floats): 17 x 16 B, conflict-free for the per-step row reads
#define LS_FG 16            // forward: steps per unrolled group (LDS offsets of a group's steps are immediates)
#define LS_NPARAM 17473     // 512 + 16384 + 256 + 256 + 64 + 1
#define LS_THREADS 512
#define LS_SLAB 32          // steps per stash slab (backward)
#define LS_PP 72            // plane pitch inside a slab row (64 + 8: the four gate planes land on different banks)
#define LS_ROWP (7 * LS_PP) // slab row pitch (floats): the stash's six planes (i, f, g, o, c, h) + tanh(c), formed once per slab (backward)
#define LS_SLAB_FLOATS ((LS_SLAB + 2) * LS_ROWP + LS_SLAB + 72)   // rows -1 .. 32, dzy, (lfo, x) of 36 steps
#define LS_DGL_FLOATS 1024                                        // backward: ring of four gate-gradient vectors
#define LS_BWD_LDS_FLOATS (2 * LS_SLAB_FLOATS + LS_DGL_FLOATS + LS_THREADS + 768)

typedef float ls_f2 __attribute__((ext_vector_type(2)));
// acc(2) += W[j](2) * v[j] for j = 0..15: 16 packed FMAs in ONE asm block (the compiler pads an s_nop after every
// inline-asm statement that holds vector instructions); v[j] is element j & 1 of the register pair P(j/2),
// broadcast to both halves of the product by op_sel / op_sel_hi
#define LS_PK_LO " op_sel_hi:[1,0,1]\n\t"
#define LS_PK_HI " op_sel:[0,1,0]\n\t"
// acc += sum: chain 0 continues `acc`, chain 1 STARTS with a packed multiply (no zeroed accumulator to set up: one
// v_mov_b64 less per step on the recurrent path; a * b and fma(a, b, 0) are the same number)
#define LS_PK16(acc, W, A, B, C, D)                                                                                   \
    {                                                                                                                \
        ls_f2 acc_b;                                                                                                 \
        LS_PK16_BODY("v_pk_fma_f32 %0, %2, %18, %0" LS_PK_LO, "+v"(acc), acc_b, W, A, B, C, D);                       \
        acc += acc_b;                                                                                                \
    }
// (starting chain 0 with a multiply too, for a zero `acc`, measured SLOWER in the backward kernel: 0.477 -> 0.489 ms)
#define LS_PK16_BODY(FIRST, ACC_OP, accb, W, A, B, C, D)                                                              \
    asm(FIRST "v_pk_mul_f32 %1, %3, %18 op_sel:[0,1]\n\t"                                                             \
        "v_pk_fma_f32 %0, %4, %19, %0" LS_PK_LO "v_pk_fma_f32 %1, %5, %19, %1" LS_PK_HI                              \
        "v_pk_fma_f32 %0, %6, %20, %0" LS_PK_LO "v_pk_fma_f32 %1, %7, %20, %1" LS_PK_HI                              \
        "v_pk_fma_f32 %0, %8, %21, %0" LS_PK_LO "v_pk_fma_f32 %1, %9, %21, %1" LS_PK_HI                              \
        "v_pk_fma_f32 %0, %10, %22, %0" LS_PK_LO "v_pk_fma_f32 %1, %11, %22, %1" LS_PK_HI                            \
        "v_pk_fma_f32 %0, %12, %23, %0" LS_PK_LO "v_pk_fma_f32 %1, %13, %23, %1" LS_PK_HI                            \
        "v_pk_fma_f32 %0, %14, %24, %0" LS_PK_LO "v_pk_fma_f32 %1, %15, %24, %1" LS_PK_HI                            \
        "v_pk_fma_f32 %0, %16, %25, %0" LS_PK_LO "v_pk_fma_f32 %1, %17, %25, %1 op_sel:[0,1,0]"                      \
        : ACC_OP, "=&v"(accb)                                                                                        \
        : "v"(W[0]), "v"(W[1]), "v"(W[2]), "v"(W[3]), "v"(W[4]), "v"(W[5]), "v"(W[6]), "v"(W[7]), "v"(W[8]),          \
          "v"(W[9]), "v"(W[10]), "v"(W[11]), "v"(W[12]), "v"(W[13]), "v"(W[14]), "v"(W[15]),                         \
          "v"((ls_f2){A.x, A.y}), "v"((ls_f2){A.z, A.w}), "v"((ls_f2){B.x, B.y}), "v"((ls_f2){B.z, B.w}),             \
          "v"((ls_f2){C.x, C.y}), "v"((ls_f2){C.z, C.w}), "v"((ls_f2){D.x, D.y}), "v"((ls_f2){D.z, D.w}))
template <int CTRL> __device__ __forceinline__ float ls_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// s * sigmoid(s * x) + (1 - s): sigmoid for s = 1, tanh for s = 2; nsl2e = -s * log2(e), oms = 1 - s
__device__ __forceinline__ float ls_act(float x, float nsl2e, float s, float oms)
{
    const float e = __builtin_amdgcn_exp2f(x * nsl2e);
    return fmaf(__builtin_amdgcn_rcpf(1.0f + e), s, oms);
}
__device__ __forceinline__ float ls_tanh(float x) { return ls_act(x, -2.8853900817779268f, 2.0f, -1.0f); }
#ifndef LS_ABL
#define LS_ABL 0    // probe builds only (tools/probe/lstm_probe.hip): bit 0 no h reads, 1 no activations, 2 no barrier, 3 no h write,
#endif              // 4 no packed FMAs, 5 no stash store
#ifdef LS_DIAG      // tools/probe/lstm_probe.hip: per-phase cycle totals of one forward / backward step (never in the product build)
__device__ unsigned long long ls_diag[8 * 8];
#define LS_STAMP(i)                                                        \
    {                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        dg_acc[i] += t_ - dg_last;                                         \
        dg_last = t_;                                                      \
    }
#define LS_DIAG_INIT unsigned long long dg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_last = __builtin_amdgcn_s_memtime();
#define LS_DIAG_DUMP                                                                             \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                              \
        for (int i_ = 0; i_ < 8; ++i_) ls_diag[(threadIdx.x >> 6) * 8 + i_] = dg_acc[i_];
#else
#define LS_STAMP(i)
#define LS_DIAG_INIT
#define LS_DIAG_DUMP
#endif
// a wave-uniform global pointer pinned to scalar registers: accesses through it take the scalar-base + 32-bit lane offset form
// (address space 1 kept through the integer round trip: a generic pointer would make them flat_* instructions, which also
// count on the LDS counter the step waits on)
typedef __attribute__((address_space(1))) float ls_gfloat;
__device__ __forceinline__ ls_gfloat *ls_uniform_mut(float *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (ls_gfloat *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void ls_barrier()      // LDS-only: outstanding global stores keep flying
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KQ>     // k-split of a gate row over lanes: 4 (512 threads, 16 k per lane) or 2 (256 threads, 32 k per lane)
__global__ __launch_bounds__(128 * KQ) void lstm_fwd_kernel(const float *__restrict__ x, long long xs,
                                                              const float *__restrict__ lfo, long long ls,
                                                              const float *__restrict__ w_ih,
                                                              const float *__restrict__ w_hh,
                                                              const float *__restrict__ b_ih,
                                                              const float *__restrict__ b_hh,
                                                              const float *__restrict__ fc_w,
                                                              const float *__restrict__ fc_b,
                                                              const float *__restrict__ h_in,
                                                              const float *__restrict__ c_in, float *__restrict__ h_out,
                                                              float *__restrict__ c_out, float *__restrict__ y,
                                                              long long ys, float *__restrict__ stash, int T, int probe)
{
    __shared__ __attribute__((aligned(16))) float hist[(LS_TB + 1) * LS_HP];   // row 0 = state entering the block
    __shared__ __attribute__((aligned(16))) float2 xl[LS_TB];                   // (lfo, x) of the block
    __shared__ float dummy[LS_THREADS + LS_FG * LS_HP];                          // sink of the lanes that hold no h (+ a group's row offsets)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int KPL = LS_H / KQ;                  // k values per lane
    const int kq = lane & (KQ - 1), gp = (lane / KQ) & 1, u = wv * (32 / KQ) + lane / (2 * KQ);
    const int q = 2 * gp + (kq & 1);                // the gate this lane activates (lanes kq >= 2 duplicate kq - 2)
    const int ra = (2 * gp) * LS_H + u, rb = (2 * gp + 1) * LS_H + u;
    ls_f2 wp[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) wp[j] = (ls_f2){w_hh[ra * LS_H + KPL * kq + j], w_hh[rb * LS_H + KPL * kq + j]};
    // the input term and both biases enter through the kq = 0 quarter only (input order (lfo, audio): models.py:328)
    const bool k0 = kq == 0;
    const ls_f2 wi0 = k0 ? (ls_f2){w_ih[ra * 2], w_ih[rb * 2]} : (ls_f2){0.f, 0.f};
    const ls_f2 wi1 = k0 ? (ls_f2){w_ih[ra * 2 + 1], w_ih[rb * 2 + 1]} : (ls_f2){0.f, 0.f};
    const ls_f2 bias = k0 ? (ls_f2){b_ih[ra] + b_hh[ra], b_ih[rb] + b_hh[rb]} : (ls_f2){0.f, 0.f};
    const float s = q == 2 ? 2.0f : 1.0f, nsl2e = -s * 1.4426950408889634f, oms = 1.0f - s;
    const bool odd = kq & 1;
    const float fcb = fc_b[0];
    float c = c_in[(size_t)b * LS_H + u];
    if (tid < LS_H) hist[tid] = h_in[(size_t)b * LS_H + tid];
    const float *xb = x + (size_t)b * xs, *lb = lfo + (size_t)b * ls;
    float *yb = y + (size_t)b * ys;
    float *sb = stash ? stash + (size_t)b * T * LS_STASH : nullptr;
    // KQ = 4: quads gp = 0 see (i, f) in their own lanes and (o, g) mirrored: they carry c and h; quads gp = 1 compute
    // junk.  KQ = 2: a quad holds i, f, g, o of one unit: every lane carries c and h.
    const bool valid = KQ == 2 || gp == 0;
    const bool writer = valid && (KQ == 2 ? (lane & 3) == 0 : true);
    float *hw = writer ? hist + LS_HP + u : dummy + tid;         // where this lane's h goes (row 1 = step 0)
    const int hw_step = writer ? LS_HP : 0;
    // stash slot of this lane: its gate activation; KQ = 4: the spare lanes kq = 2, 3 of the valid quads store c and h
    // in the same instruction; KQ = 2: lanes 0 / 1 of the quad store them with a second instruction
    const bool st_c = KQ == 4 ? valid && kq == 2 : (lane & 3) == 0, st_h = KQ == 4 ? valid && kq == 3 : (lane & 3) == 1;
    const int st_off = KQ == 4 ? (st_c ? 256 + u : (st_h ? 320 + u : q * LS_H + u)) : q * LS_H + u;
    const int st_off2 = st_c ? 256 + u : (st_h ? 320 + u : q * LS_H + u);

    for (int t0 = 0; t0 < T; t0 += LS_TB) {
        const int cnt = min(LS_TB, T - t0);
        __syncthreads();                            // the previous block's y pass is done with hist / xl
        if (t0 > 0 && tid < LS_H) hist[tid] = hist[LS_TB * LS_HP + tid];
        if (tid < cnt) xl[tid] = probe ? make_float2(0.5f, 0.25f) : make_float2(lb[t0 + tid], xb[t0 + tid]);
        __syncthreads();
        const float *hr = hist + KPL * kq;
        float *hwp = hw;
        LS_DIAG_INIT
        float *st = sb + (size_t)t0 * LS_STASH + st_off;       // dereferenced only when sb != NULL (wave-uniform test)
#if LS_ABL == 0 && !defined(LS_DIAG)
        // Round 6: blocks whose length is a multiple of LS_FG steps run in unrolled groups -- the row of the h history read and
        // written, the (lfo, x) pair and the stash row of a step are a per-group base + an immediate, the loop counter is a
        // scalar, and the stash store of step t is issued at the head of step t + 1 behind its h reads (it sat between the h
        // write and the barrier: five instructions on every step's critical path).  Same arithmetic, same order.
        if ((cnt % LS_FG) == 0) {
          auto run_groups = [&](auto st_tag) {
            constexpr bool do_st = decltype(st_tag)::value;    // the BPTT stash is written (wave-uniform: two instances of the loop)
            float pend_a = 0.0f, pend_c = 0.0f, pend_h = 0.0f; // the previous step's stash values
            const int hw_group = writer ? LS_FG * LS_HP : 0;
            const float2 *xlg = xl;
            for (int g = 0; g < cnt; g += LS_FG) {
                // stash rows of the group: wave-uniform base in scalar registers + the lane's slot
                ls_gfloat *stg = do_st ? ls_uniform_mut(sb + (size_t)(t0 + g) * LS_STASH) : nullptr;
#pragma unroll
                for (int j = 0; j < LS_FG; ++j) {
                    const float2 in = xlg[j];                  // (first: the input term is then formed while the h reads are in flight)
                    const float4 h0 = *(const float4 *)(hr + j * LS_HP), h1 = *(const float4 *)(hr + j * LS_HP + 4),
                                 h2 = *(const float4 *)(hr + j * LS_HP + 8), h3 = *(const float4 *)(hr + j * LS_HP + 12);
                    float4 h4, h5, h6, h7;
                    if (KQ == 2) {
                        h4 = *(const float4 *)(hr + j * LS_HP + 16); h5 = *(const float4 *)(hr + j * LS_HP + 20);
                        h6 = *(const float4 *)(hr + j * LS_HP + 24); h7 = *(const float4 *)(hr + j * LS_HP + 28);
                    }
                    __builtin_amdgcn_sched_barrier(0);         // (the step's reads first; the deferred store rides in their shadow)
                    if (do_st && (g > 0 || j > 0)) {                   // (the selects live here, not between the barrier and the reads)
                        if (KQ == 4) stg[(j - 1) * LS_STASH + st_off] = st_c ? pend_c : (st_h ? pend_h : pend_a);
                        else {
                            stg[(j - 1) * LS_STASH + st_off] = pend_a;
                            stg[(j - 1) * LS_STASH + st_off2] = st_c ? pend_c : (st_h ? pend_h : pend_a);
                        }
                    }
                    // input term as two packed FMAs (the rolled loop below keeps the unfused 4-instruction form: same value to 1 ulp)
                    ls_f2 acc;
                    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(acc) : "v"(wi0), "v"(in), "v"(bias));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(wi1), "v"(in));
                    LS_PK16(acc, wp, h0, h1, h2, h3);
                    if (KQ == 2) {
                        const ls_f2 *wq = wp + (KQ == 2 ? 16 : 0);
                        LS_PK16(acc, wq, h4, h5, h6, h7);
                    }
                    float pa = acc.x, pb = acc.y;
                    pa += ls_dpp<0xB1>(pa); pb += ls_dpp<0xB1>(pb);     // quad_perm [1,0,3,2]
                    if (KQ == 4) { pa += ls_dpp<0x4E>(pa); pb += ls_dpp<0x4E>(pb); }   // quad_perm [2,3,0,1]: all four quarters
                    const float a = ls_act(odd ? pb : pa, nsl2e, s, oms);
                    float gi, gf, gg, go;
                    if (KQ == 4) {
                        const float m = ls_dpp<0x141>(a);               // row_half_mirror: the other gate pair of the unit
                        gi = ls_dpp<0x00>(a); gf = ls_dpp<0x55>(a); gg = ls_dpp<0x55>(m); go = ls_dpp<0x00>(m);
                    } else {
                        gi = ls_dpp<0x00>(a); gf = ls_dpp<0x55>(a); gg = ls_dpp<0xAA>(a); go = ls_dpp<0xFF>(a);
                    }
                    c = fmaf(gf, c, gi * gg);
                    const float hv = go * ls_tanh(c);
                    hwp[j * LS_HP] = hv;
                    pend_a = a; pend_c = c; pend_h = hv;
                    ls_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                hr += LS_FG * LS_HP;
                hwp += hw_group;
                xlg += LS_FG;
            }
            if (do_st) {                                       // the block's last step
                float *stl = sb + (size_t)(t0 + cnt - 1) * LS_STASH;
                if (KQ == 4) stl[st_off] = st_c ? pend_c : (st_h ? pend_h : pend_a);
                else {
                    stl[st_off] = pend_a;
                    stl[st_off2] = st_c ? pend_c : (st_h ? pend_h : pend_a);
                }
            }
          };
          if (sb && !probe) run_groups(std::true_type{});
          else run_groups(std::false_type{});
        } else
#endif
        for (int tt = 0; tt < cnt; ++tt) {
            float4 h0, h1, h2, h3;
            if (LS_ABL & 1) {
                h0 = h1 = h2 = h3 = make_float4(c, c, c, c);
            } else {
                h0 = *(const float4 *)hr; h1 = *(const float4 *)(hr + 4); h2 = *(const float4 *)(hr + 8);
                h3 = *(const float4 *)(hr + 12);
            }
            float4 h4, h5, h6, h7;
            if (KQ == 2) {
                h4 = *(const float4 *)(hr + 16); h5 = *(const float4 *)(hr + 20); h6 = *(const float4 *)(hr + 24);
                h7 = *(const float4 *)(hr + 28);
            }
            const float2 in = xl[tt];
            LS_STAMP(0)                                         // LDS reads landed
            ls_f2 acc = wi1 * in.y + (wi0 * in.x + bias);
            if (!(LS_ABL & 16)) {
                LS_PK16(acc, wp, h0, h1, h2, h3);
                if (KQ == 2) {
                    const ls_f2 *wq = wp + (KQ == 2 ? 16 : 0);
                    LS_PK16(acc, wq, h4, h5, h6, h7);
                }
            } else {
                acc += (ls_f2){h0.x, h3.w};
            }
            float pa = acc.x, pb = acc.y;
            LS_STAMP(1)                                         // packed FMAs
            pa += ls_dpp<0xB1>(pa); pb += ls_dpp<0xB1>(pb);     // quad_perm [1,0,3,2]
            if (KQ == 4) { pa += ls_dpp<0x4E>(pa); pb += ls_dpp<0x4E>(pb); }   // quad_perm [2,3,0,1]: all four quarters
            const float a = (LS_ABL & 2) ? (odd ? pb : pa) * 0.01f : ls_act(odd ? pb : pa, nsl2e, s, oms);
            float gi, gf, gg, go;
            if (KQ == 4) {
                const float m = ls_dpp<0x141>(a);               // row_half_mirror: the other gate pair of the unit
                gi = ls_dpp<0x00>(a); gf = ls_dpp<0x55>(a); gg = ls_dpp<0x55>(m); go = ls_dpp<0x00>(m);
            } else {
                gi = ls_dpp<0x00>(a); gf = ls_dpp<0x55>(a); gg = ls_dpp<0xAA>(a); go = ls_dpp<0xFF>(a);
            }
            c = fmaf(gf, c, gi * gg);
            const float hv = go * ((LS_ABL & 2) ? c * 0.5f : ls_tanh(c));
            LS_STAMP(2)                                         // reduce, activation, exchange, cell update
            if (!(LS_ABL & 8)) *hwp = hv;
            if (sb && !probe && !(LS_ABL & 32)) {
                if (KQ == 4) {
                    *st = st_c ? c : (st_h ? hv : a);
                } else {
                    *st = a;
                    st[st_off2 - st_off] = st_c ? c : (st_h ? hv : a);
                }
                st += LS_STASH;
            }
            hr += LS_HP;
            hwp += hw_step;
            LS_STAMP(3)                                         // h write landed, stash store issued
            if (!(LS_ABL & 4)) ls_barrier();
            LS_STAMP(4)                                         // barrier
        }
        LS_DIAG_DUMP
        // y_t = tanh(fc . h_t + b + x_t) (models.py:335-337), one step per thread, off the recurrent path
        if (tid < cnt) {
            const float *hr = hist + (tid + 1) * LS_HP;
            float acc = fcb;
#pragma unroll
            for (int k = 0; k < LS_H; k += 4) {
                const float4 h4 = *(const float4 *)(hr + k);
                acc = fmaf(fc_w[k], h4.x, acc);
                acc = fmaf(fc_w[k + 1], h4.y, acc);
                acc = fmaf(fc_w[k + 2], h4.z, acc);
                acc = fmaf(fc_w[k + 3], h4.w, acc);
            }
            if (!probe || t0 + cnt >= T) yb[t0 + tid] = tanhf(acc + xl[tid].y);
        }
        if (t0 + cnt >= T) {
            if (tid < LS_H) h_out[(size_t)b * LS_H + tid] = hist[cnt * LS_HP + tid];
            if (writer && kq == 0) c_out[(size_t)b * LS_H + u] = c;
        }
    }
}

// x (B rows, stride xs) audio, lfo (B rows, stride ls), T samples each; parameters in torch layout:
// w_ih (256,2), w_hh (256,64), b_ih (256), b_hh (256), fc_w (64), fc_b (1); h_in / c_in (B,64): the state
// entering the chunk (read only: the BPTT of the chunk needs it again), h_out / c_out (B,64): the state leaving it
// (may alias h_in / c_in when no backward follows); y (B rows, stride ys); stash (B, T, 384) or NULL when no
// backward follows.
static int lstm_fwd_launch(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                          const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w,
                          const float *fc_b, const float *h_in, const float *c_in, float *h_out, float *c_out, float *y,
                          int64_t y_stride, float *stash, int64_t B, int64_t T, void *stream, int probe)
{
    if (!x || !lfo || !w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b || !h_in || !c_in || !h_out || !c_out || !y ||
        B <= 0 || T <= 0)
        return MX_ERR_ARG;
    if (T >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    // 4 waves (32 k per lane, no idle lanes in the cell update, half the waves at the barrier) for both uses.  Rounds 3-5 ran
    // 8 waves (16 k per lane; the spare lanes carry c and h in the one stash store) when the BPTT stash is written: 0.349 vs
    // 0.343 ms then; with round 6's unrolled groups and the deferred stash store the 4-wave kernel is ahead with the stash
    // too: 0.284 vs 0.305 ms per 128 clips x 1024 steps (0.260 vs 0.315 without).  MODEX_LSTM_KQ = 2 | 4 forces one (experiments).
    static const int kq_env = getenv("MODEX_LSTM_KQ") ? atoi(getenv("MODEX_LSTM_KQ")) : 0;
    const int kq = kq_env == 4 ? 4 : 2;
    if (kq == 2)
        hipLaunchKernelGGL(lstm_fwd_kernel<2>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)x_stride, lfo, (long long)lfo_stride, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, h_in, c_in,
                           h_out, c_out, y, (long long)y_stride, stash, (int)T, probe);
    else
        hipLaunchKernelGGL(lstm_fwd_kernel<4>, dim3((unsigned)B), dim3(512), 0, (hipStream_t)stream, x,
                           (long long)x_stride, lfo, (long long)lfo_stride, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, h_in, c_in,
                           h_out, c_out, y, (long long)y_stride, stash, (int)T, probe);
    return mx_launch_status();
}

MX_EXPORT int mx_lstm_fwd(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                          const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w,
                          const float *fc_b, const float *h_in, const float *c_in, float *h_out, float *c_out, float *y,
                          int64_t y_stride, float *stash, int64_t B, int64_t T, void *stream)
{
    return lstm_fwd_launch(x, x_stride, lfo, lfo_stride, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, h_in, c_in, h_out, c_out, y, y_stride, stash, B, T, stream, 0);
}

// Measurement twin (bench.py's serial floor): the SAME launch with no global-memory traffic inside the sample loop -- inputs are constants, only the last chunk is stored.  Results are meaningless; nothing in the product calls it.
MX_EXPORT int mx_lstm_fwd_probe(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *w_ih,
                          const float *w_hh, const float *b_ih, const float *b_hh, const float *fc_w,
                          const float *fc_b, const float *h_in, const float *c_in, float *h_out, float *c_out, float *y,
                          int64_t y_stride, float *stash, int64_t B, int64_t T, void *stream)
{
    return lstm_fwd_launch(x, x_stride, lfo, lfo_stride, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, h_in, c_in, h_out, c_out, y, y_stride, stash, B, T, stream, 1);
}

// ---- truncated BPTT of one chunk with the L1 loss fused: the serial part --------------------------
// loss = loss_scale * sum_{b,t} |y - wet|  (loss_scale = w_l1 / (B*T) for nn.L1Loss 'mean').
// dgate (B, T, 256): d loss / d pre-activation of every gate row and step, consumed by lstm_wgrad_kernel.
__device__ __forceinline__ int ls_dg_slot(int r)   // LDS slot of gate row r: [16-byte chunk of the 16-run][16-run][4]
{
    return ((r & 15) >> 2) * 64 + (r >> 4) * 4 + (r & 3);
}

// HELP (round 6): the weight-gradient products leave the recurrence waves.  The workgroup grows by LS_HELP_WAVES = 4 waves (one
// per SIMD) that do nothing but the exact-fp32 matrix instructions of dW_hh (+ the dW_ih / bias sums that ride on the same A
// values): helper wave j owns gate rows 64 j .. 64 j + 63 (four 32 x 32 tiles).  They read what the fused version read -- the gate
// gradients of steps t + 1, t + 2 in the rotating LDS buffers and the h rows of the stash slab -- at the same point of the step and
// join the same ONE barrier per step, so nothing about the hand-over changes; the eight recurrence waves no longer stall in-order
// behind two 64-cycle matrix instructions and their five operand reads at the top of every second step.
#define LS_HELP_WAVES 4
#ifndef LSB_ABL
#define LSB_ABL 0    // diagnostic builds only (tools/exp_lstm_bwd.py; wrong results): bit 0 helper waves idle, 1 no prefetch reads of the
#endif               // stash values, 2 no slab staging after the first slab
template <bool DGOUT, bool HELP>    // DGOUT: also write the gate gradients (B, T, 256) -- the input of mx_lstm_dlfo (an UNFROZEN LFO model, lightning.py:258,361)
__global__ __launch_bounds__(LS_THREADS + (HELP ? 64 * LS_HELP_WAVES : 0)) void lstm_bwd_kernel(const float *__restrict__ x, long long xs,
                                                              const float *__restrict__ lfo, long long ls,
                                                              const float *__restrict__ y, long long ys,
                                                              const float *__restrict__ wet, long long ws,
                                                              const float *__restrict__ stash,
                                                              const float *__restrict__ w_hh,
                                                              const float *__restrict__ fc_w,
                                                              const float *__restrict__ h_init,
                                                              const float *__restrict__ c_init, float loss_scale,
                                                              const float *__restrict__ dy, long long dys,
                                                              float *__restrict__ part, float *__restrict__ dgate, int T, int probe)
{
    // d loss / d y of every step: `dy` rows when given (any loss, mx_lstm_bwd), else the fused nn.L1Loss
    // loss_scale * sign(y - wet) (mx_lstm_bwd_l1: one launch and one (B, T) tensor less per optimizer step)
    // slab buffer = [row -1 (only its c plane: c of the step before the slab)] [32 rows of 6 planes] [row 32 (only its h
    // plane: h of the step after the slab)] ; dzy (32) ; lfo (36) ; x (36) of steps t0 .. t0 + 33
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int slab_floats = LS_SLAB_FLOATS;
    float *slab0 = smem, *slab1 = smem + slab_floats;
    float *dgl = smem + 2 * slab_floats;                       // 4 x 256 gate gradients: dg(t) lives in buffer t & 3 (dg(t+1), dg(t+2) are read while dg(t) is written)
    float *dummy = dgl + LS_DGL_FLOATS;                        // sink of the lanes that hold no gate gradient (+ the buffer offset: LS_THREADS + 768 floats)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (HELP && wv >= LS_THREADS / 64) {
        // ---- helper wave: dW_hh = sum_t dg_t (x) h_{t-1}, dW_ih, biases for gate rows 64 hw .. 64 hw + 63
        const int hw = wv - LS_THREADS / 64, c32h = lane & 31, tparh = lane >> 5;
        const int a_slot0 = ls_dg_slot(64 * hw + c32h), a_slot1 = ls_dg_slot(64 * hw + 32 + c32h);
        floatx16 w00, w01, w10, w11;                           // [row tile][column tile]
#pragma unroll
        for (int r = 0; r < 16; ++r) w00[r] = w01[r] = w10[r] = w11[r] = 0.0f;
        float di00 = 0.0f, di01 = 0.0f, di10 = 0.0f, di11 = 0.0f, dbs0 = 0.0f, dbs1 = 0.0f;
        const float *lbh = lfo + (size_t)b * ls, *xbh = x + (size_t)b * xs;
        const int n_slabs_h = (T + LS_SLAB - 1) / LS_SLAB;
        float dfw_h = 0.0f, dfb_h = 0.0f;                      // helper 0: d fc.weight[lane] = sum_t dzy_t h_t[lane], d fc.bias = sum_t dzy_t
        __syncthreads();                                       // the recurrence waves' first slab is in LDS, dgl is zeroed
        __syncthreads();                                       // ... and its tanh(c) plane
        for (int S = n_slabs_h - 1; S >= 0; --S) {
            const int t0 = S * LS_SLAB, cnt = min(LS_SLAB, T - t0);
            const float *sl = (S & 1 ? slab1 : slab0) + LS_ROWP;
            const float *dzlh = (S & 1 ? slab1 : slab0) + (LS_SLAB + 2) * LS_ROWP;
            const float *xll = dzlh + LS_SLAB;
            for (int s_ = cnt - 1; s_ >= 0; --s_) {
                if (!(LSB_ABL & 1) && hw == 0) {               // the output layer's gradients (they were two instructions + one LDS read of every recurrence wave and step)
                    const float dzy_ = dzlh[s_];
                    dfw_h = fmaf(dzy_, sl[s_ * LS_ROWP + 5 * LS_PP + lane], dfw_h);
                    dfb_h += dzy_;
                }
                if (!(LSB_ABL & 1) && ((T - 1 - (t0 + s_)) & 1)) {
                    const float *dgb = dgl + ((s_ + 1 + tparh) & 3) * 256;         // dg(t + 1) / dg(t + 2): buffer t & 3, t0 is a multiple of 4
                    const float av0 = dgb[a_slot0], av1 = dgb[a_slot1];
                    const float *hrow = sl + (s_ + tparh) * LS_ROWP + 5 * LS_PP;
                    const float bv0 = hrow[c32h], bv1 = hrow[32 + c32h];
                    const float li = xll[s_ + 1 + tparh], xi = xll[36 + s_ + 1 + tparh];
#if LSB_ABL & 32     // ablation: the helper's reads and sums, no matrix instructions
                    dbs0 += bv0 + bv1;
#elif LSB_ABL & 64   // ablation (wrong results): the matrix-pipe time of a bf16x3 product instead -- three 32-cycle bf16 instructions per step pair
                    {
                        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                        const bf16x8 a8 = __builtin_bit_cast(bf16x8, make_float4(av0, av1, bv0, bv1));
                        w00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, w00, 0, 0, 0);
                        w01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, w01, 0, 0, 0);
                        w10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, w10, 0, 0, 0);
                    }
#else
                    w00 = mfma32(av0, bv0, w00);
                    w10 = mfma32(av1, bv0, w10);
                    w01 = mfma32(av0, bv1, w01);
                    w11 = mfma32(av1, bv1, w11);
#endif
                    di00 = fmaf(av0, li, di00); di01 = fmaf(av0, xi, di01);
                    di10 = fmaf(av1, li, di10); di11 = fmaf(av1, xi, di11);
                    dbs0 += av0; dbs1 += av1;
                }
                ls_barrier();
            }
            if (S > 0 && cnt == 1) { ls_barrier(); ls_barrier(); }     // (the recurrence waves' one-step slab: store / tanh pass behind the step)
        }
        {   // the steps the pairing left over: dg(0) (x) h_init always, dg(1) (x) h_0 when T is odd
            const float *dgb = dgl + tparh * 256;              // dg(0) in buffer 0, dg(1) in buffer 1
            const bool live = tparh == 0 || (T & 1);
            const float av0 = live ? dgb[a_slot0] : 0.0f, av1 = live ? dgb[a_slot1] : 0.0f;
            const float *hrow = tparh == 0 ? h_init + (size_t)b * LS_H : slab0 + LS_ROWP + 5 * LS_PP;
            const float bv0 = hrow[c32h], bv1 = hrow[32 + c32h];
            const float li = tparh < T ? (probe ? 0.5f : lbh[tparh]) : 0.0f, xi = tparh < T ? (probe ? 0.25f : xbh[tparh]) : 0.0f;
            w00 = mfma32(av0, bv0, w00);
            w10 = mfma32(av1, bv0, w10);
            w01 = mfma32(av0, bv1, w01);
            w11 = mfma32(av1, bv1, w11);
            di00 = fmaf(av0, li, di00); di01 = fmaf(av0, xi, di01);
            di10 = fmaf(av1, li, di10); di11 = fmaf(av1, xi, di11);
            dbs0 += av0; dbs1 += av1;
        }
        float *pbh = part + (size_t)b * LS_NPARAM;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 64 * hw + mfma_row(r, lane);
            pbh[512 + row * LS_H + c32h] = w00[r];
            pbh[512 + row * LS_H + 32 + c32h] = w01[r];
            pbh[512 + (row + 32) * LS_H + c32h] = w10[r];
            pbh[512 + (row + 32) * LS_H + 32 + c32h] = w11[r];
        }
        const float s00 = di00 + __shfl_xor(di00, 32, 64), s01 = di01 + __shfl_xor(di01, 32, 64);
        const float s10 = di10 + __shfl_xor(di10, 32, 64), s11 = di11 + __shfl_xor(di11, 32, 64);
        const float sb0 = dbs0 + __shfl_xor(dbs0, 32, 64), sb1 = dbs1 + __shfl_xor(dbs1, 32, 64);
        if (tparh == 0) {
            const int row = 64 * hw + c32h;
            pbh[row * 2] = s00;
            pbh[row * 2 + 1] = s01;
            pbh[(row + 32) * 2] = s10;
            pbh[(row + 32) * 2 + 1] = s11;
            pbh[512 + 16384 + row] = sb0;
            pbh[512 + 16384 + 256 + row] = sb0;
            pbh[512 + 16384 + row + 32] = sb1;
            pbh[512 + 16384 + 256 + row + 32] = sb1;
        }
        if (hw == 0) {
            pbh[512 + 16384 + 512 + lane] = dfw_h;
            if (lane == 0) pbh[LS_NPARAM - 1] = dfb_h;
        }
        return;
    }
    const int rg = lane & 15, kp = wv * 4 + (lane >> 4);       // 16 gate rows 16 rg .. 16 rg + 15, hidden units 2 kp, 2 kp + 1
    const int e = rg & 7, q = e & 3, k = 2 * kp + (e >> 2);    // the (unit, gate) this lane differentiates (rg >= 8: duplicate)
    const bool second = e >> 2;
    ls_f2 wp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wp[j] = *(const ls_f2 *)(w_hh + (16 * rg + j) * LS_H + 2 * kp);
    const float fcw = fc_w[k];
    // derivative of the own gate: D = alpha + a * (beta - a)   (sigmoid: a (1 - a); tanh: 1 - a^2)
    const float alpha = q == 2 ? 1.0f : 0.0f, beta = q == 2 ? 0.0f : 1.0f;
    // per-lane slab offsets (floats, relative to the row of the step)
    const int off_a = q * LS_PP + k;
    const int off_p = q == 0 ? 2 * LS_PP + k : (q == 1 ? 4 * LS_PP + k - LS_ROWP : (q == 2 ? k : 3 * LS_PP + k));
    const int off_f = 1 * LS_PP + k, off_o = 3 * LS_PP + k, off_c = 6 * LS_PP + k;     // (off_c: the tanh(c) plane)
    const int dg_rd = rg * 4;                                  // + 64 c: the c-th 16-byte chunk of this lane's run of 16 rows
    const bool owner = rg < 8;
    const int dg_wr = ls_dg_slot(q * LS_H + k);
    const float *yb = y + (size_t)b * ys, *wb = dy ? yb : wet + (size_t)b * ws;
    const float *sb = stash + (size_t)b * T * LS_STASH;
    const float *xb = x + (size_t)b * xs, *lb = lfo + (size_t)b * ls;
    // weight gradients (they do not feed the recurrence): dW_hh = sum_t dg_t (x) h_{t-1} accumulates on the otherwise idle
    // matrix pipes, two fp32 MFMAs per wave every second step on the steps (t+1, t+2) whose gate gradients are in LDS;
    // wave wv owns gate rows 32 wv .. 32 wv + 31.  dW_ih / biases ride along on the A fragment, fc on the unit lanes.
    const int c32 = lane & 31, tpar = lane >> 5;
    const int a_slot = ls_dg_slot(32 * wv + c32);
    floatx16 wacc0, wacc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc0[r] = wacc1[r] = 0.0f;
    float dwi0 = 0.0f, dwi1 = 0.0f, dbs = 0.0f, dfw = 0.0f, dfb = 0.0f;
    const int off_h = 5 * LS_PP + k;

    const int n_slabs = (T + LS_SLAB - 1) / LS_SLAB;
    float4 pre[6];
    float pre_c = 0.0f, pre_y = 0.0f, pre_w = 0.0f, pre_xl = 0.0f;
    bool pre_live = false;
    // slab S -> registers (global, coalesced: 32 x 384 contiguous floats), registers -> LDS (plane-padded)
    // FULL (every slab but the clip's last, which is loaded once before the loop): no bounds predicates.  The addresses are a
    // wave-uniform slab base + a 32-bit lane offset (the scalar-base form of the loads): hoisted 64-bit per-lane addresses of the
    // six loads were spilled to scratch memory by the 168-register variant and reloaded in a chain of dependent round trips.
    auto slab_load = [&](int S, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int t0 = S * LS_SLAB, cnt = FULL ? LS_SLAB : min(LS_SLAB, T - t0);
        if (probe) {                                   // serial-floor measurement: no global traffic, constant values
#pragma unroll
            for (int i = 0; i < 6; ++i) pre[i] = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
            pre_c = 0.5f;
            pre_live = true;
            pre_y = 0.0f;
            pre_w = 0.0f;
            pre_xl = 0.25f;
            return;
        }
        const char *gbase = reinterpret_cast<const char *>(sb + (size_t)t0 * LS_STASH);     // wave-uniform
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned e = (unsigned)(i * LS_THREADS + tid) * 4u;                      // element of the 32 x 384 slab
            if (FULL || e < (unsigned)(cnt * LS_STASH)) pre[i] = *reinterpret_cast<const float4 *>(gbase + e * 4u);
            else pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < LS_H) pre_c = t0 > 0 ? sb[(size_t)(t0 - 1) * LS_STASH + 256 + tid] : c_init[(size_t)b * LS_H + tid];
        if (tid >= 64 && tid < 64 + LS_SLAB) {                         // raw y and wet (or dy): d loss / d (pre-tanh output) is formed in
            const int t = t0 + tid - 64;                               // slab_store, 16 steps later -- computing it here put a wait for ALL
            pre_live = FULL || t < T;                                  // outstanding loads (the six slab vectors included) on this step's path
            if (pre_live) {
                pre_y = yb[t];
                pre_w = dy ? dy[(size_t)b * dys + t] : wb[t];
            }
        }
        if (tid >= 160 && tid < 160 + 72) {                            // lfo of steps t0 .. t0 + 35, then x of the same steps
            const int i = tid - 160, t = t0 + (i < 36 ? i : i - 36);
            pre_xl = t < T ? (i < 36 ? lb[t] : xb[t]) : 0.0f;
        }
    };
    auto slab_store = [&](float *dst, const float *above) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = (i * LS_THREADS + tid) * 4, srow = e / LS_STASH, j = e % LS_STASH;
            *(float4 *)(dst + (srow + 1) * LS_ROWP + (j >> 6) * LS_PP + (j & 63)) = pre[i];
        }
        if (tid < LS_H) dst[4 * LS_PP + tid] = pre_c;                  // row -1, c plane
        if (tid >= 64 && tid < 64 + LS_SLAB) {
            float dzy_ = 0.0f;
            if (probe) dzy_ = 1e-3f;
            else if (pre_live) {
                float g = pre_w;                                       // dy given: d loss / d y itself
                if (!dy) {
                    const float e = pre_y - pre_w;
                    g = loss_scale * (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f));
                }
                dzy_ = g * (1.0f - pre_y * pre_y);
            }
            dst[(LS_SLAB + 2) * LS_ROWP + tid - 64] = dzy_;
        }
        // row 32, h plane = h of the step after the slab = row 0 of the slab above, which is the OTHER buffer's (in LDS: no load)
        if (tid >= 96 && tid < 96 + LS_H) dst[(LS_SLAB + 1) * LS_ROWP + 5 * LS_PP + tid - 96] = above ? above[LS_ROWP + 5 * LS_PP + tid - 96] : 0.0f;
        if (tid >= 160 && tid < 160 + 72) dst[(LS_SLAB + 2) * LS_ROWP + LS_SLAB + tid - 160] = pre_xl;
    };

    // tanh(c) of a stored slab, once per (step, unit): 32 x 64 values = four per thread (one 16-byte read, one write), a barrier
    // after slab_store.  Every recurrence wave used to evaluate it in every step for the 8 units it differentiates -- 40 wave
    // instructions per step and workgroup (two of five transcendental) for 64 distinct values, on SIMDs whose vector issue is what
    // bounds the backward step (two recurrence waves + one helper per SIMD; the fp32 matrix instructions execute on the vector ALUs).
    auto slab_tanh = [&](float *dst) {
        const int r = tid >> 4, u4 = (tid & 15) * 4;
        float *row = dst + (r + 1) * LS_ROWP;
        const float4 cv = *(const float4 *)(row + 4 * LS_PP + u4);
        *(float4 *)(row + 6 * LS_PP + u4) = make_float4(ls_tanh(cv.x), ls_tanh(cv.y), ls_tanh(cv.z), ls_tanh(cv.w));
    };
    for (int i = tid; i < LS_DGL_FLOATS; i += LS_THREADS) dgl[i] = 0.0f;        // dh from "step T" is zero
    slab_load(n_slabs - 1, std::false_type{});
    slab_store((n_slabs - 1) & 1 ? slab1 : slab0, nullptr);
    __syncthreads();
    slab_tanh((n_slabs - 1) & 1 ? slab1 : slab0);
    __syncthreads();

    float dc_next = 0.0f;
    float n_a, n_p, n_f, n_o, n_c, n_z, n_h;                          // raw stash values of the next step processed (prefetched one step ahead)
    float *const wr_base = owner ? dgl + dg_wr : dummy + tid;
    const float *const dr_base = dgl + dg_rd;
    // One step.  Called with a compile-time `s` from the unrolled loop over a full slab (round 6): every LDS address of the step
    // is then a per-slab lane register + an immediate -- the ring position of the gate-gradient buffers (t & 3; t0 is a multiple
    // of 32), the slab row, the slab-store point -- and nothing of the step's bookkeeping is vector arithmetic any more.
    auto step = [&](const int s, const int cnt, const int S, const int t0, const float *sl, const float *dzl, const float *xll,
                    const float *pa, const float *pp_, const float *pf, const float *po, const float *pc, const float *ph) {
        const int bc = (s + 1) & 3, bw = s & 3;
        // (0) [fused variant only] every second step: gate gradients of steps t+1, t+2 (x) the states entering them -> dW_hh, dW_ih, db
        if (!HELP && ((T - 1 - (t0 + s)) & 1)) {
            const float av = dgl[((s + 1 + tpar) & 3) * 256 + a_slot];
            const float *hrow = sl + (s + tpar) * LS_ROWP + 5 * LS_PP;      // h_t (tpar 0) / h_{t+1} (tpar 1); row 32 exists
            const float bv0 = hrow[c32], bv1 = hrow[32 + c32];
            const float li = xll[s + 1 + tpar], xi = xll[36 + s + 1 + tpar];
            wacc0 = mfma32(av, bv0, wacc0);
            wacc1 = mfma32(av, bv1, wacc1);
            dwi0 = fmaf(av, li, dwi0);
            dwi1 = fmaf(av, xi, dwi1);
            dbs += av;
        }
        // (1) the 16 gate gradients of step t+1 this lane multiplies
        const float *dr = dr_base + bc * 256;
#if !(LSB_ABL & 8)
        const float4 g0 = *(const float4 *)dr, g1 = *(const float4 *)(dr + 64), g2 = *(const float4 *)(dr + 128),
                     g3 = *(const float4 *)(dr + 192);
#endif
#if LSB_ABL & 8      // ablation: ONE 16-byte read per lane instead of four (the critical reads' share of the LDS pipe)
        const float4 g0 = *(const float4 *)dr, g1 = g0, g2 = g0, g3 = g0;
#endif
        __builtin_amdgcn_sched_barrier(0);                         // (first thing behind the barrier: these four reads head the step's dependent chain)
        // (2) while they arrive: the local derivatives of step t from the values prefetched last step
        const float a = n_a, pp = n_p, f = n_f, o = n_o, dzy = n_z;
        if (!HELP) {
            dfw = fmaf(dzy, n_h, dfw);                             // d fc.weight[k] = sum_t dzy_t h_t[k]   (HELP: helper wave 0)
            dfb += dzy;
        }
        const float tc = n_c;                                      // tanh(c_t), from the slab's seventh plane
        const float kc = o * fmaf(-tc, tc, 1.0f);                  // d h / d c = o (1 - tanh^2 c)
        const float der = fmaf(a, beta - a, alpha);
        const float kq = der * (q == 3 ? tc : pp);                 // i: g i(1-i); f: c_prev f(1-f); g: i (1-g^2); o: tanh(c) o(1-o)
        // (3) prefetch the raw values of step t-1 (row -1 of the slab is never used as a step)
        if (!(LSB_ABL & 2)) {   // unconditional (s = 0 re-reads row 0): the compiler can then count these reads behind the four above
            const int sp = s > 0 ? s - 1 : 0;
            n_a = pa[sp * LS_ROWP]; n_p = pp_[sp * LS_ROWP]; n_f = pf[sp * LS_ROWP]; n_o = po[sp * LS_ROWP]; n_c = pc[sp * LS_ROWP];
            n_z = dzl[sp];
            if (!HELP) n_h = ph[sp * LS_ROWP];
        }
        __builtin_amdgcn_sched_barrier(0);                         // (the prefetch reads stay in FRONT of the FMA chain: sunk to the end of
                                                                   //  the step they sit between the dg write and the barrier, on the critical path)
        // (4) dh_prev[k] = sum_r W[r][k] dg[r] for the unit pair: 16 rows per lane, all-reduce over the 16 row groups
        ls_f2 acc = {0.0f, 0.0f};
        LS_PK16(acc, wp, g0, g1, g2, g3);
        // A lane needs the sum of ONE of the two units (bit 2 of rg says which): the first exchange (row_half_mirror: the partner
        // 7 - l has the other bit 2) hands over the component the partner needs and keeps the own one, the other three levels
        // (lane ^ 1, ^ 2, ^ 8: bit 2 unchanged) then reduce a single value -- 2 selects + 4 DPP adds instead of 8 DPP adds + 1 select.
        float keep = second ? acc.y : acc.x;
        const float send = second ? acc.x : acc.y;
        keep += ls_dpp<0x141>(send);                               // row_half_mirror
        keep += ls_dpp<0xB1>(keep);                                // quad_perm [1,0,3,2]
        keep += ls_dpp<0x4E>(keep);                                // quad_perm [2,3,0,1]
        keep += ls_dpp<0x128>(keep);                               // row_ror:8
        const float dhn = keep;
        // (5) element-wise backward of step t
        const float dh = fmaf(dzy, fcw, dhn);
        const float dc = fmaf(dh, kc, dc_next);
        const float dg = (q == 3 ? dh : dc) * kq;
        dc_next = dc * f;
        wr_base[bw * 256] = dg;
        if (DGOUT && owner) dgate[((size_t)b * T + (t0 + s)) * 256 + q * LS_H + k] = dg;       // row order of weight_ih_l0: gate * 64 + unit
        if (!(LSB_ABL & 4) && S > 0 && cnt > 1 && s == cnt / 2) slab_store((S - 1) & 1 ? slab1 : slab0, S & 1 ? slab1 : slab0);   // the other buffer is idle
        if (!(LSB_ABL & 4) && S > 0 && cnt > 1 && s == cnt / 2 - 1) slab_tanh((S - 1) & 1 ? slab1 : slab0);                      // one barrier later
        ls_barrier();
        __builtin_amdgcn_sched_barrier(0);                         // (nothing of the next step is hoisted in front of the barrier)
    };
    for (int S = n_slabs - 1; S >= 0; --S) {
        const int t0 = S * LS_SLAB, cnt = min(LS_SLAB, T - t0);
        const float *sl = (S & 1 ? slab1 : slab0) + LS_ROWP;           // row 0 of the slab
        const float *dzl = (S & 1 ? slab1 : slab0) + (LS_SLAB + 2) * LS_ROWP;
        const float *xll = dzl + LS_SLAB;                              // lfo[t0 + i] at i, x[t0 + i] at 36 + i
        const float *pa = sl + off_a, *pp_ = sl + off_p, *pf = sl + off_f, *po = sl + off_o, *pc = sl + off_c, *ph = sl + off_h;
        if (!(LSB_ABL & 4) && S > 0) slab_load(S - 1, std::true_type{});
        // raw stash values of the first step processed (the steps prefetch one step ahead)
        {
            const int r0 = (cnt - 1) * LS_ROWP;
            n_a = pa[r0]; n_p = pp_[r0]; n_f = pf[r0]; n_o = po[r0]; n_c = pc[r0]; n_z = dzl[cnt - 1];
            n_h = HELP ? 0.0f : ph[r0];
        }
        if (!DGOUT && cnt == LS_SLAB) {           // (the gate-gradient variant keeps the rolled loop: unrolled, its per-step store addresses spill)
#pragma unroll
            for (int s = LS_SLAB - 1; s >= 0; --s) step(s, LS_SLAB, S, t0, sl, dzl, xll, pa, pp_, pf, po, pc, ph);
        } else {
            for (int s = cnt - 1; s >= 0; --s) step(s, cnt, S, t0, sl, dzl, xll, pa, pp_, pf, po, pc, ph);
        }
        if (S > 0 && cnt == 1) {                   // a one-step slab (only the clip's last): store and tanh pass behind its step
            slab_store((S - 1) & 1 ? slab1 : slab0, S & 1 ? slab1 : slab0);
            ls_barrier();
            slab_tanh((S - 1) & 1 ? slab1 : slab0);
            ls_barrier();
        }
    }
    // the steps the pairing above left over: dg(0) (x) h_init always, dg(1) (x) h_0 when T is odd
    if (!HELP) {
        const float av = tpar == 0 ? dgl[a_slot] : ((T & 1) ? dgl[256 + a_slot] : 0.0f);              // dg(0): buffer 0, dg(1): buffer 1
        const float *hrow = tpar == 0 ? h_init + (size_t)b * LS_H : slab0 + LS_ROWP + 5 * LS_PP;     // slab 0, row 0, h plane
        const float bv0 = hrow[c32], bv1 = hrow[32 + c32];
        const float li = tpar < T ? (probe ? 0.5f : lb[tpar]) : 0.0f, xi = tpar < T ? (probe ? 0.25f : xb[tpar]) : 0.0f;
        wacc0 = mfma32(av, bv0, wacc0);
        wacc1 = mfma32(av, bv1, wacc1);
        dwi0 = fmaf(av, li, dwi0);
        dwi1 = fmaf(av, xi, dwi1);
        dbs += av;
    }
    // one gradient row per clip, state-dict order:
    //   [lstm.weight_ih_l0 (256,2) | lstm.weight_hh_l0 (256,64) | lstm.bias_ih_l0 | lstm.bias_hh_l0 | fc.weight | fc.bias]
    float *pb = part + (size_t)b * LS_NPARAM;
    if (!HELP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wv + mfma_row(r, lane);
            pb[512 + row * LS_H + c32] = wacc0[r];
            pb[512 + row * LS_H + 32 + c32] = wacc1[r];
        }
    }
    if (!HELP) {
        const float s0 = dwi0 + __shfl_xor(dwi0, 32, 64), s1 = dwi1 + __shfl_xor(dwi1, 32, 64);
        const float sbias = dbs + __shfl_xor(dbs, 32, 64);
        if (tpar == 0) {
            const int row = 32 * wv + c32;
            pb[row * 2] = s0;
            pb[row * 2 + 1] = s1;
            pb[512 + 16384 + row] = sbias;
            pb[512 + 16384 + 256 + row] = sbias;
        }
    }
    if (!HELP) {
        if (rg == 0 || rg == 4) pb[512 + 16384 + 512 + k] = dfw;      // lanes e = 0 / 4 hold units 2 kp / 2 kp + 1
        if (tid == 0) pb[LS_NPARAM - 1] = dfb;
    }
}

// part: (B, 17473) gradient rows, one per clip, to be summed with mx_reduce_rows(part, B, 17473, ...).
static int lstm_bwd_l1_launch(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                             int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                             const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                             float loss_scale, const float *dy, int64_t dy_stride, float *part, int64_t B, int64_t T, void *stream, int probe,
                             float *dgate = nullptr)
{
    if (!x || !lfo || !y || (!wet && !dy) || !stash || !w_hh || !fc_w || !h_init || !c_init || !part || B <= 0 || T <= 0)
        return MX_ERR_ARG;
    if (T >= (1ll << 30)) return MX_ERR_UNSUPPORTED;
    const size_t lds = (size_t)LS_BWD_LDS_FLOATS * sizeof(float);
    // MODEX_LSTM_HELPERS=0: the round-5 kernel (weight-gradient matrix instructions inside the recurrence waves); default: on helper waves
    static const bool help = !(getenv("MODEX_LSTM_HELPERS") && atoi(getenv("MODEX_LSTM_HELPERS")) == 0);
    static MxLdsLatch latch[2][2] = {};                       // per device (common.h), per (DGOUT, HELP) instance
#define LS_BWD_LAUNCH(DG, HP)                                                                                                       \
    {                                                                                                                               \
        if (mx_set_dyn_lds(latch[DG][HP], (const void *)lstm_bwd_kernel<DG, HP>, lds) != MX_OK) return MX_ERR_LAUNCH;               \
        hipLaunchKernelGGL((lstm_bwd_kernel<DG, HP>), dim3((unsigned)B), dim3(LS_THREADS + (HP ? 64 * LS_HELP_WAVES : 0)), lds,    \
                           (hipStream_t)stream, x, (long long)x_stride, lfo, (long long)lfo_stride, y, (long long)y_stride, wet,   \
                           (long long)wet_stride, stash, w_hh, fc_w, h_init, c_init, loss_scale, dy, (long long)dy_stride, part,   \
                           DG ? dgate : nullptr, (int)T, probe);                                                                    \
        return mx_launch_status();                                                                                                  \
    }
    if (dgate) LS_BWD_LAUNCH(true, false)     // (168 registers per wave are not enough for this variant: it stays fused)
    if (help) LS_BWD_LAUNCH(false, true) else LS_BWD_LAUNCH(false, false)
#undef LS_BWD_LAUNCH
}

MX_EXPORT int mx_lstm_bwd_l1(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                             int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                             const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                             float loss_scale, float *part, int64_t B, int64_t T, void *stream)
{
    return lstm_bwd_l1_launch(x, x_stride, lfo, lfo_stride, y, y_stride, wet, wet_stride, stash, w_hh, fc_w, h_init, c_init, loss_scale, nullptr, 0, part, B, T, stream, 0);
}

// Measurement twin (bench.py's serial floor): the SAME launch with no global-memory traffic inside the sample loop -- inputs are constants, only the last chunk is stored.  Results are meaningless; nothing in the product calls it.
MX_EXPORT int mx_lstm_bwd_l1_probe(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                             int64_t y_stride, const float *wet, int64_t wet_stride, const float *stash,
                             const float *w_hh, const float *fc_w, const float *h_init, const float *c_init,
                             float loss_scale, float *part, int64_t B, int64_t T, void *stream)
{
    return lstm_bwd_l1_launch(x, x_stride, lfo, lfo_stride, y, y_stride, wet, wet_stride, stash, w_hh, fc_w, h_init, c_init, loss_scale, nullptr, 0, part, B, T, stream, 1);
}


// The same truncated BPTT for ANY loss (lightning.py:380-382 back-propagates whatever calc_and_log_losses returns:
// losses.py:142-160): dy (B rows, stride dy_stride) = d loss / d y of every output sample, evaluated by the caller
// (mx_effect_loss_grad for L1 / MSE / ESR / DC, mx_mrstft_loss for the multi-resolution STFT loss, or their sum).
MX_EXPORT int mx_lstm_bwd(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                          int64_t y_stride, const float *dy, int64_t dy_stride, const float *stash, const float *w_hh,
                          const float *fc_w, const float *h_init, const float *c_init, float *part, int64_t B, int64_t T,
                          void *stream)
{
    if (!dy || dy_stride < T) return MX_ERR_ARG;
    return lstm_bwd_l1_launch(x, x_stride, lfo, lfo_stride, y, y_stride, nullptr, 0, stash, w_hh, fc_w, h_init, c_init, 0.0f,
                              dy, dy_stride, part, B, T, stream, 0);
}

// The same BPTT that ALSO leaves the gate gradients: dgate (B, T, 256), row order of weight_ih_l0 (gate * 64 + unit).  With them
// mx_lstm_dlfo forms d loss / d lfo of every sample -- what an UNFROZEN LFO model needs (lightning.py:258,361: the extractor is
// re-run inside every TBPTT step and receives the effect model's gradient through the LFO it produced).  wet != NULL: nn.L1Loss
// fused (loss_scale as mx_lstm_bwd_l1); else dy (B rows, stride dy_stride) = d loss / d y.
MX_EXPORT int mx_lstm_bwd_dgate(const float *x, int64_t x_stride, const float *lfo, int64_t lfo_stride, const float *y,
                                int64_t y_stride, const float *wet, int64_t wet_stride, const float *dy, int64_t dy_stride,
                                const float *stash, const float *w_hh, const float *fc_w, const float *h_init,
                                const float *c_init, float loss_scale, float *part, float *dgate, int64_t B, int64_t T,
                                void *stream)
{
    if (!dgate || (!wet && !dy) || (wet && dy) || (dy && dy_stride < T)) return MX_ERR_ARG;
    return lstm_bwd_l1_launch(x, x_stride, lfo, lfo_stride, y, y_stride, wet, wet_stride, stash, w_hh, fc_w, h_init, c_init,
                              loss_scale, dy, dy_stride, part, B, T, stream, 0, dgate);
}

// dlfo[b][t] = sum_r weight_ih[r][0] * dgate[b][t][r]   (the LSTM's input is (lfo, audio): models.py:328; column 0 = the LFO)
__global__ __launch_bounds__(256) void lstm_dlfo_kernel(const float *__restrict__ dgate, const float *__restrict__ w_ih,
                                                        long long n_bt, int T, float *__restrict__ dlfo, long long dlfo_stride)
{
    const int lane = threadIdx.x & 63;
    const long long bt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= n_bt) return;
    const float4 g = *reinterpret_cast<const float4 *>(dgate + bt * 256 + 4 * lane);
    const int r = 4 * lane;
    float v = (g.x * w_ih[2 * r] + g.y * w_ih[2 * r + 2]) + (g.z * w_ih[2 * r + 4] + g.w * w_ih[2 * r + 6]);
    v = wave_sum_f32(v);
    if (lane == 0) dlfo[(bt / T) * dlfo_stride + (bt % T)] = v;
}

// dgate (B, T, 256) from mx_lstm_bwd_dgate, w_ih (256, 2) -> dlfo: row b at dlfo + b * dlfo_stride, T values
MX_EXPORT int mx_lstm_dlfo(const float *dgate, const float *w_ih, int64_t B, int64_t T, float *dlfo, int64_t dlfo_stride,
                           void *stream)
{
    if (!dgate || !w_ih || !dlfo || B <= 0 || T <= 0 || dlfo_stride < T) return MX_ERR_ARG;
    if (B * T >= (1ll << 32)) return MX_ERR_UNSUPPORTED;
    const long long n = B * T;
    hipLaunchKernelGGL(lstm_dlfo_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dgate, w_ih, n, (int)T,
                       dlfo, (long long)dlfo_stride);
    return mx_launch_status();
}

// ---- independent latency floor of one recurrent step (bench.py) ---------------------------------------
// The `*_probe` twins above re-run the PRODUCT kernels without global traffic: that floor inherits their schedule.  This
// microbenchmark is built the way the flanger's `mx_lds_roundtrip_probe` is: nothing but the chain of operations ANY schedule of
// the step has to go through on this decomposition (one 512-lane workgroup per clip, the 256 x 64 matrix-vector product split
// 16 k per lane), on one workgroup, with no global memory, no input term, no stash, no output layer, no weight gradients, no
// address bookkeeping beyond a row toggle:
//   kind 0 (forward step, models.py:333):  broadcast read of h_{t-1} from LDS (4 x 16 B per lane) -> 16 packed FMAs (two
//          chains of 8) -> 2 + 2 cross-lane adds -> one gate activation (v_exp_f32, v_rcp_f32) -> exchange of i, f, g, o inside
//          the unit's 8 lanes -> c = f c + i g -> tanh(c) (v_exp_f32, v_rcp_f32) -> h = o tanh(c) -> LDS write -> s_barrier
//   kind 1 (backward step, lightning.py:380): read of the 16 gate gradients of step t + 1 -> 16 packed FMAs -> reduction over
//          the 16 row groups (2 selects + 4 cross-lane adds) -> dh, dc, dg of step t (the local derivatives o (1 - tanh^2 c),
//          a (1 - a) from six LDS values read beside the gradients; tanh c is one of them since round 6) -> LDS write -> s_barrier
//   kind 2 (forward step in the 256-lane decomposition, 32 k per lane: what mx_lstm_fwd launches since round 6): 8 x 16 B of h
//          per lane -> 32 packed FMAs -> 1 + 1 cross-lane adds -> gate activation -> exchange inside the unit's quad -> cell
//          update -> tanh -> LDS write -> s_barrier (four waves)
// `steps` such steps on ONE workgroup; time / steps x T is a floor of a T-step launch that does not come from the product kernels.
__global__ __launch_bounds__(LS_THREADS) void lstm_step_probe_kernel(int steps, int kind, int stride, float *__restrict__ out)
{
    // Round 6: the loops run two steps per iteration so that the double-buffered row is addressed by immediates, as the
    // product kernels address their history / ring rows since this round (the probe must not be slower than what it bounds);
    // `stride` is 0 at run time (the compiler cannot know) and keeps the saved-value reads of the backward step inside the loop.
    __shared__ __attribute__((aligned(16))) float row[2][256 + 16];      // h (64 used) or gate gradients (256) of the previous step
    __shared__ float vals[8 * 64];                                        // stand-in for the saved activations of a step (backward)
    __shared__ float sink[LS_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ls_f2 w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = (ls_f2){0.001f * (float)((tid + 3 * j) % 17) - 0.008f, 0.001f * (float)((tid + 5 * j) % 13) - 0.006f};
    for (int i = tid; i < 2 * (256 + 16); i += blockDim.x) (&row[0][0])[i] = 0.01f * (float)(i % 23) - 0.1f;
    for (int i = tid; i < 8 * 64; i += blockDim.x) vals[i] = 0.3f + 0.001f * (float)(i & 63);
    __syncthreads();
    float acc_out = 0.0f;
    const int n_pairs = (steps + 1) / 2;
    if (kind == 0) {
        const int kq = lane & 3, gp = (lane >> 2) & 1, u = wv * 8 + (lane >> 3);
        const bool odd = kq & 1, writer = gp == 0;
        const int q = 2 * gp + (kq & 1);
        const float sc = q == 2 ? 2.0f : 1.0f, nsl2e = -sc * 1.4426950408889634f, oms = 1.0f - sc;
        float c = 0.1f;
        float *const wr[2] = {writer ? &row[0][u] : &sink[tid], writer ? &row[1][u] : &sink[tid]};
        const float *const rd[2] = {&row[0][16 * kq], &row[1][16 * kq]};
        auto step = [&](const int cur) {
            const float *hr = rd[cur];
            const float4 h0 = *(const float4 *)hr, h1 = *(const float4 *)(hr + 4), h2 = *(const float4 *)(hr + 8), h3 = *(const float4 *)(hr + 12);
            ls_f2 acc = {0.01f, 0.02f};
            LS_PK16(acc, w, h0, h1, h2, h3);
            float pa = acc.x, pb = acc.y;
            pa += ls_dpp<0xB1>(pa); pb += ls_dpp<0xB1>(pb);
            pa += ls_dpp<0x4E>(pa); pb += ls_dpp<0x4E>(pb);
            const float a = ls_act(odd ? pb : pa, nsl2e, sc, oms);
            const float m = ls_dpp<0x141>(a);
            const float gi = ls_dpp<0x00>(a), gf = ls_dpp<0x55>(a), gg = ls_dpp<0x55>(m), go = ls_dpp<0x00>(m);
            c = fmaf(gf, c, gi * gg);
            const float hv = go * ls_tanh(c);
            *wr[cur ^ 1] = hv;
            ls_barrier();
            acc_out = hv;
        };
        for (int t = 0; t < n_pairs; ++t) { step(0); step(1); }
    } else if (kind == 2) {
        // forward step of the 256-lane decomposition (32 k per lane: the kernel the product launches since round 6): a quad holds
        // i, f, g, o of one unit
        const int kq = lane & 1, gp = (lane >> 1) & 1, u = wv * 16 + (lane >> 2);
        const bool odd = kq & 1, writer = (lane & 3) == 0;
        const int q = 2 * gp + kq;
        const float sc = q == 2 ? 2.0f : 1.0f, nsl2e = -sc * 1.4426950408889634f, oms = 1.0f - sc;
        ls_f2 w2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w2[j] = (ls_f2){0.001f * (float)((tid + 7 * j) % 19) - 0.009f, 0.001f * (float)((tid + 11 * j) % 23) - 0.011f};
        float c = 0.1f;
        float *const wr[2] = {writer ? &row[0][u] : &sink[tid], writer ? &row[1][u] : &sink[tid]};
        const float *const rd[2] = {&row[0][32 * kq], &row[1][32 * kq]};
        auto step = [&](const int cur) {
            const float *hr = rd[cur];
            const float4 h0 = *(const float4 *)hr, h1 = *(const float4 *)(hr + 4), h2 = *(const float4 *)(hr + 8), h3 = *(const float4 *)(hr + 12);
            const float4 h4 = *(const float4 *)(hr + 16), h5 = *(const float4 *)(hr + 20), h6 = *(const float4 *)(hr + 24), h7 = *(const float4 *)(hr + 28);
            ls_f2 acc = {0.01f, 0.02f};
            LS_PK16(acc, w, h0, h1, h2, h3);
            LS_PK16(acc, w2, h4, h5, h6, h7);
            float pa = acc.x, pb = acc.y;
            pa += ls_dpp<0xB1>(pa); pb += ls_dpp<0xB1>(pb);
            const float a = ls_act(odd ? pb : pa, nsl2e, sc, oms);
            const float gi = ls_dpp<0x00>(a), gf = ls_dpp<0x55>(a), gg = ls_dpp<0xAA>(a), go = ls_dpp<0xFF>(a);
            c = fmaf(gf, c, gi * gg);
            const float hv = go * ls_tanh(c);
            *wr[cur ^ 1] = hv;
            ls_barrier();
            acc_out = hv;
        };
        for (int t = 0; t < n_pairs; ++t) { step(0); step(1); }
    } else {
        // backward step as the round-6 kernel runs it: tanh(c) is a saved value (formed once per slab, off the chain), the
        // row-group reduction hands over one component in its first level
        const int rg = lane & 15, kp = wv * 4 + (lane >> 4), e = rg & 7, q = e & 3, k = 2 * kp + (e >> 2);
        const bool second = e >> 2, owner = rg < 8;
        const float alpha = q == 2 ? 1.0f : 0.0f, beta = q == 2 ? 0.0f : 1.0f, fcw = 0.01f * (float)k;
        float dc_next = 0.0f;
        float *const wr[2] = {owner ? &row[0][ls_dg_slot(q * LS_H + k)] : &sink[tid], owner ? &row[1][ls_dg_slot(q * LS_H + k)] : &sink[tid]};
        const float *const rd[2] = {&row[0][rg * 4], &row[1][rg * 4]};
        int t = 0;
        auto step = [&](const int cur) {
            const float *dr = rd[cur];
            const float4 g0 = *(const float4 *)dr, g1 = *(const float4 *)(dr + 64), g2 = *(const float4 *)(dr + 128), g3 = *(const float4 *)(dr + 192);
            // the saved values of step t (a, its partner, f, o, tanh c, d loss / d pre-tanh output): six LDS reads, as from a staged stash slab
            const float *v = vals + ((t + stride) & 7) * 64 + (lane & 7);
            const float a = v[0], pp = v[8], f = v[16], o = v[24], tc = v[32], dzy = v[40] * 1e-3f;
            const float kc = o * fmaf(-tc, tc, 1.0f);
            const float der = fmaf(a, beta - a, alpha);
            const float kqv = der * (q == 3 ? tc : pp);
            ls_f2 acc = {0.0f, 0.0f};
            LS_PK16(acc, w, g0, g1, g2, g3);
            float keep = second ? acc.y : acc.x;
            const float send = second ? acc.x : acc.y;
            keep += ls_dpp<0x141>(send);
            keep += ls_dpp<0xB1>(keep);
            keep += ls_dpp<0x4E>(keep);
            keep += ls_dpp<0x128>(keep);
            const float dh = fmaf(dzy, fcw, keep);
            const float dc = fmaf(dh, kc, dc_next);
            const float dg = (q == 3 ? dh : dc) * kqv;
            dc_next = dc * f;
            *wr[cur ^ 1] = dg;
            ls_barrier();
            acc_out = dg;
            ++t;
        };
        for (int p = 0; p < n_pairs; ++p) { step(0); step(1); }
    }
    if (tid == 0) out[0] = acc_out;
}

// kind 0: forward step of the 512-lane decomposition (16 k per lane), 1: backward step (512 lanes), 2: forward step of the 256-lane
// decomposition (32 k per lane; what mx_lstm_fwd launches); `steps` dependent steps on one workgroup; out: 1 float
MX_EXPORT int mx_lstm_step_probe(int32_t kind, int64_t steps, float *out, void *stream)
{
    if (!out || steps <= 0 || steps >= (1ll << 30) || kind < 0 || kind > 2) return MX_ERR_ARG;
    hipLaunchKernelGGL(lstm_step_probe_kernel, dim3(1), dim3(kind == 2 ? 256 : LS_THREADS), 0, (hipStream_t)stream, (int)steps, (int)kind, 0, out);
    return mx_launch_status();
}
