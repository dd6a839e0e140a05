// wave_fft.h -- the register-staged wavefront FFT (512 / 1024 / 2048 points) shared by the MR-STFT loss (mrstft.hip) and the
// log-mel / log-power front end for n_fft != 1024 (melspec.hip).  Reference users: mod_extraction/losses.py:155-156
// (auraloss STFTs), mod_extraction/models.py:170-181,199-208 (torchaudio MelSpectrogram / Spectrogram).
#pragma once
#include "common.h"

#define MR_MAXN 2048

typedef float cf __attribute__((ext_vector_type(2)));          // complex: x = re, y = im (an aligned register pair: the packed-fp32 unit)
// Complex arithmetic on the packed-fp32 instructions, two instructions per multiply, one per add -- written as inline
// assembly because the compiler does not fold "swap the halves and negate ONE of them" into the source modifiers that do
// exactly that (op_sel / neg_lo / neg_hi of v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32): from C++ a complex multiply was
// 6 instructions (v_xor + v_mov to build (-w.im, w.re), a broadcast, v_pk_mul, v_pk_add, v_pk_fma) and a rotation by -+i
// two more, and together with the address arithmetic below the transforms ran ~670 instructions per 1024-point frame
// and lane against ~300 for the butterflies themselves.
//   a * w      = (a.x w.x - a.y w.y,  a.x w.y + a.y w.x):  t = a.yy * (-w.y, w.x);  r = a.xx * w + t
//   a * conj w = (a.x w.x + a.y w.y, -a.x w.y + a.y w.x):  t = a.yy * ( w.y, w.x);  r = a.xx * (w.x, -w.y) + t
// The second product of each component is fused (2 mul + 2 fma).  That costs a property the unfused arithmetic had for
// free: the loss packs the two real signals as x + i y into ONE transform, and the spectra separate EXACTLY for x == y only
// while the arithmetic is symmetric under the index mirror k -> N - k, which maps a butterfly's twiddle w to -i conj(w) and
// thereby SWAPS the two products of a complex multiply -- an FMA rounds one of them and not the other.  The reference gives
// loss == 0 and gradient == 0 exactly for identical signals, so the kernels detect frames whose windowed x and y are
// bit-identical and take Y := X for them (tests/test_gpu_mrstft.py::test_mrstft_identical_signals).
template <bool INV> __device__ __forceinline__ cf cmul_v(cf a, cf w)        // twiddle in vector registers
{
    cf t, r;
    if (!INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    }
    return r;
}
template <bool INV> __device__ __forceinline__ cf cmul_s(cf a, cf w)        // wave-uniform twiddle in scalar registers
{
    cf t, r;
    if (!INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "s"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    }
    return r;
}
// a + (-i) d = (a.x + d.y, a.y - d.x)   /   a + i d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ cf add_mi(cf a, cf d)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ cf add_pi(cf a, cf d)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}

// ---- the FFT: one frame per wavefront (two per wavefront for N = 512), passes fused in registers -----------------
// A frame of N points is held by L lanes, E = N / L complex values per lane.  Radix-4 Stockham passes p = 0.. with
// Ns = 4^p (butterfly j reads src[j + (N/4) c], writes dst[4 (j - k) + k + r Ns], k = j mod Ns) plus one radix-2
// pass for 512 / 2048.  A lane that runs butterflies a + L b of an EVEN pass holds exactly the inputs of E / 4
// butterflies of the next pass, so passes (0,1) and (2,3) run back to back in registers; 2048's radix-2 pass fuses
// with pass 4 the same way.  Three register stages, two exchanges through a wave-private padded LDS buffer, no
// workgroup barrier (first version: the frame spread over 256 threads, one LDS round trip and one __syncthreads per
// pass, twiddles from global memory: 19.2 ms per 256 x 4 s; twiddles in LDS: 14.9 ms).  The index algebra was
// checked against numpy.fft for the three sizes, both directions, before it was written down here.
// Every exchange-buffer slot and twiddle index is (a function of the lane) + (a compile-time constant) -- the forms below
// are checked against the plain algebra (position -> padded slot, butterfly -> twiddle step) by
// tools/probe/check_fft_addr.py -- so each LDS access is one instruction with an immediate offset; the lane parts
// (FftLane) are set up once per kernel:
//   exchange 1 (after passes 0, 1; slot = q + (q >> 4)):   write 17 a + (17 L be + r + 4 r2),   read a + (a >> 4) + const(b, c)
//   exchange 2 (after passes 2, 3; slot = q + 16 (q >> 8)): write 272 (a >> 4) + (a & 15) + (17 L be + 16 r + 64 r2),  read a + const(b, c)
//   twiddles: pass 1: r N / 16 (wave-uniform: nine constants in scalar registers);  pass 2: (a & 15) N / 64 (lane only: three values
//   in registers);  pass 3: (a & 15) N / 256 + r N / 16;  pass 4: a + 64 b (1024), 2 a + 128 (b & 3) (2048);  radix-2: a + const
template <int N> struct WF {
    static constexpr int L = (N == 512) ? 32 : 64;     // lanes per frame
    static constexpr int E = N / L;                    // complex values per lane: 16, 16, 32
    static constexpr int NB = E / 4;                   // radix-4 butterflies per lane and pass: 4, 4, 8
    static constexpr int NBQ = NB / 4;                 // 1, 1, 2
    static constexpr int LEN = N + N / 16;             // padded exchange buffer (complex values)
    static constexpr int WAVES = (N == 2048) ? 2 : 4;  // wavefronts per workgroup (static LDS <= 64 KB)
    static constexpr int FW = 64 / L;                  // frames a wavefront works on at a time
};

template <int N> struct FftLane {
    cf *w1, *w2;                                       // exchange write bases (lane part applied)
    const cf *r1, *r2;                                 // exchange read bases
    cf t2[3];                                          // pass-2 twiddles w, w^2, w^3 of this lane
    const cf *t3[3];                                   // pass-3 twiddle bases: tw_s + m (a & 15) N / 256
    const cf *t4[3];                                   // last radix-4 pass: tw_s + m a (1024), tw_s + 2 m a (2048); unused for 512
    const cf *t5;                                      // radix-2 pass of 512 / 2048: tw_s + a
    cf c1[3][3];                                       // pass-1 twiddles [r - 1][m - 1] = exp(-2 pi i m r / 16): wave-uniform
};
// tw: exp(-2 pi i m / (N * TWS)) table in global memory (TWS = 2048 / N for the MR-STFT's shared 2048-point table, 1 for a table
// of the transform's own length)
template <int N, int TWS = MR_MAXN / N>
__device__ __forceinline__ void fft_lane_setup(FftLane<N> &fl, cf *buf, const cf *tw_s, const float2 *__restrict__ tw, int a)
{
    fl.w1 = buf + 17 * a;
    fl.r1 = buf + a + (a >> 4);
    fl.w2 = buf + 272 * (a >> 4) + (a & 15);
    fl.r2 = buf + a;
#pragma unroll
    for (int m = 1; m <= 3; ++m) {
        const float2 w = tw[(m * (a & 15) * (N / 64)) * TWS];
        fl.t2[m - 1] = {w.x, w.y};
        fl.t3[m - 1] = tw_s + m * (a & 15) * (N / 256);
        fl.t4[m - 1] = tw_s + m * a * (N / 1024);
#pragma unroll
        for (int r = 1; r <= 3; ++r) {
            const float2 c = tw[(m * r * (N / 16)) * TWS];      // uniform address: scalar loads
            fl.c1[r - 1][m - 1] = {c.x, c.y};
        }
    }
    fl.t5 = tw_s + a;
}

// radix-4 butterfly on (already twiddled) v0..v3
template <bool INV> __device__ __forceinline__ void bfly4_core(cf v0, cf v1, cf v2, cf v3, cf (&o)[4])
{
    const cf a0 = v0 + v2, a1 = v0 - v2, a2 = v1 + v3, d = v1 - v3;
    o[0] = a0 + a2;
    o[2] = a0 - a2;
    o[1] = INV ? add_pi(a1, d) : add_mi(a1, d);               // a1 +- (-+ i) d
    o[3] = INV ? add_mi(a1, d) : add_pi(a1, d);
}

// Full transform of the lane-resident frame R (R[b][c] = value at position a + L b + (N/4) c).  Results: Z[i] = bin pos_final<N>(i, a).
template <int N> __device__ __forceinline__ int pos_final(int i, int a)
{
    if (N == 1024) return a + 64 * (i >> 2) + 256 * (i & 3);                      // i = 4 b + r
    if (N == 2048) return a + 64 * ((i >> 2) & 3) + 256 * (i & 3) + 1024 * (i >> 4);  // i = 16 h + 4 b + r
    return a + 32 * ((i >> 2) + 4 * (i & 1)) + 256 * ((i >> 1) & 1);            // 512: i = 4 b + cl + 2 h
}
template <int N, bool INV>
__device__ __forceinline__ void wave_fft(cf (&R)[WF<N>::NB][4], cf (&Z)[WF<N>::E], const FftLane<N> &fl)
{
    constexpr int L = WF<N>::L, NB = WF<N>::NB, NBQ = WF<N>::NBQ;
    cf O[NB][4];
    // ---- passes 0 (no twiddles) and 1 (uniform twiddles), exchange 1
#pragma unroll
    for (int b = 0; b < NB; ++b) bfly4_core<INV>(R[b][0], R[b][1], R[b][2], R[b][3], O[b]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int be = 0; be < NBQ; ++be) {
            cf v1 = O[NBQ + be][r], v2 = O[2 * NBQ + be][r], v3 = O[3 * NBQ + be][r], Pq[4];
            if (r > 0) {
                v1 = cmul_s<INV>(v1, fl.c1[r - 1][0]);
                v2 = cmul_s<INV>(v2, fl.c1[r - 1][1]);
                v3 = cmul_s<INV>(v3, fl.c1[r - 1][2]);
            }
            bfly4_core<INV>(O[be][r], v1, v2, v3, Pq);
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) fl.w1[17 * L * be + r + 4 * r2] = Pq[r2];
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[b][c] = fl.r1[L * b + (N / 4) * c + ((L * b + (N / 4) * c) >> 4)];
    __builtin_amdgcn_wave_barrier();
    // ---- passes 2 (lane twiddles, in registers) and 3, exchange 2
#pragma unroll
    for (int b = 0; b < NB; ++b)
        bfly4_core<INV>(R[b][0], cmul_v<INV>(R[b][1], fl.t2[0]), cmul_v<INV>(R[b][2], fl.t2[1]), cmul_v<INV>(R[b][3], fl.t2[2]), O[b]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const cf w1 = fl.t3[0][r * (N / 16)], w2 = fl.t3[1][2 * r * (N / 16)], w3 = fl.t3[2][3 * r * (N / 16)];
#pragma unroll
        for (int be = 0; be < NBQ; ++be) {
            cf Pq[4];
            bfly4_core<INV>(O[be][r], cmul_v<INV>(O[NBQ + be][r], w1), cmul_v<INV>(O[2 * NBQ + be][r], w2),
                            cmul_v<INV>(O[3 * NBQ + be][r], w3), Pq);
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) fl.w2[17 * L * be + 16 * r + 64 * r2] = Pq[r2];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[b][c] = fl.r2[L * b + (N / 4) * c + 16 * ((L * b + (N / 4) * c) >> 8)];
    __builtin_amdgcn_wave_barrier();
    // ---- the last pass(es)
    if (N == 1024) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {                                      // pass 4: twiddle step a + 64 b
            cf o[4];
            bfly4_core<INV>(R[b][0], cmul_v<INV>(R[b][1], fl.t4[0][64 * b]), cmul_v<INV>(R[b][2], fl.t4[1][2 * 64 * b]),
                            cmul_v<INV>(R[b][3], fl.t4[2][3 * 64 * b]), o);
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[4 * b + r] = o[r];
        }
    } else if (N == 2048) {
#pragma unroll
        for (int b = 0; b < NB; ++b)                                        // pass 4: twiddle step 2 a + 128 (b & 3)
            bfly4_core<INV>(R[b][0], cmul_v<INV>(R[b][1], fl.t4[0][128 * (b & 3)]), cmul_v<INV>(R[b][2], fl.t4[1][2 * 128 * (b & 3)]),
                            cmul_v<INV>(R[b][3], fl.t4[2][3 * 128 * (b & 3)]), O[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                   // radix-2, Ns = 1024: j = a + 64 b + 256 r
                const cf v = cmul_v<INV>(O[(b + 4) % NB][r], fl.t5[64 * b + 256 * r]);
                Z[4 * b + r] = O[b][r] + v;
                Z[16 + 4 * b + r] = O[b][r] - v;
            }
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) {                                // radix-2, Ns = 256: j = a + 32 (b + 4 cl)
                const cf v = cmul_v<INV>(R[b % NB][cl + 2], fl.t5[32 * (b + 4 * cl)]);
                Z[4 * b + cl] = R[b % NB][cl] + v;
                Z[4 * b + cl + 2] = R[b % NB][cl] - v;
            }
    }
}

