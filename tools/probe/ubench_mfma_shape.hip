// ubench_mfma_shape.hip -- does the fp16 MFMA SHAPE change what a power-limited loop delivers?  (VERDICT r03 item 1;
// MI355X_MICROARCH.md "DVFS give-back" item 7 measured 1.12-1.15x for 16x16x32 over 32x32x16 on bf16.)
//
// Bare loops at the wave tile of conv_f16x3_dma_kernel (64 channels x 5.5 column tiles = 176 accumulator registers per
// wave, 4 waves per workgroup, 1 workgroup per CU, 256 workgroups), f16x3 term structure (lo*hi, hi*lo, hi*hi), random
// fp16 operands re-read from LDS with ds_read_b128 at the real kernel's rate (16 reads per 33 MFMAs of 32x32x16 = 32 reads
// per 132 MFMAs of 16x16x32: the same LDS bytes per FLOP).  The sparse pair is v_smfmac_f32_32x32x32_f16 against
// v_smfmac_f32_16x16x64_f16 with one 16-byte read per 32x32 instruction (wgrad_sp_f16.hip's rate).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/_bin/ubench_mfma_shape tools/probe/ubench_mfma_shape.hip
//   gpurun -- ./tools/probe/_bin/ubench_mfma_shape
//
// Prints per mode: ms per launch, TFLOP/s (dense-equivalent for the sparse modes), in-kernel clock from
// s_memtime / s_memrealtime stamps around the loop (median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define LDS_BYTES (64 * 1024)
#define LDS_ALLOC (LDS_BYTES + 16 * 1024 + 2048)

// fragment r of step cnt: one base per step (a single VALU add), r as an immediate offset -- as in the real kernels
__device__ __forceinline__ half8 ldsr(const unsigned char *lds, int cnt, int r, int lane)
{
    const unsigned char *base = lds + ((cnt * 1040) & 0x3FF0) + lane * 16;
    return *reinterpret_cast<const half8 *>(base + r * 1024);
}

// MODE 0: dense 32x32x16, 11 accumulators.  MODE 1: dense 16x16x32, 44 accumulators.
// MODE 2: sparse 32x32x32, 11 accumulators. MODE 3: sparse 16x16x64, 44 accumulators.
// LDSR = 1: fragments re-read from LDS every step; 0: loaded once (registers only).
template <int MODE, int LDSR, int RPM = 4>
__global__ __launch_bounds__(256, 1) void shape_kernel(const _Float16 *__restrict__ src, float *__restrict__ out, int iters,
                                                       unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS_ALLOC / 16; i += 256)
        reinterpret_cast<floatx4 *>(lds)[i] = reinterpret_cast<const floatx4 *>(src)[(i + blockIdx.x * 7) & 8191];
    __syncthreads();
    const unsigned idxw = 0x4E4E4E4Eu ^ (unsigned)(lane * 0x01010101u & 0x44444444u);   // valid 2:4 selectors (pairs differ)

    unsigned long long t0 = 0, r0 = 0;
    if (MODE == 0) {
        floatx16 acc[11];
        for (int i = 0; i < 11; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        half8 FA[2][4], FBH[2][6], FBL[2][6];
        for (int r = 0; r < 4; ++r) FA[0][r] = ldsr(lds, 0, r, lane);
        for (int r = 0; r < 6; ++r) { FBH[0][r] = ldsr(lds, 0, 4 + r, lane); FBL[0][r] = ldsr(lds, 0, 10 + r, lane); }
        for (int r = 0; r < 4; ++r) FA[1][r] = FA[0][r];
        for (int r = 0; r < 6; ++r) { FBH[1][r] = FBH[0][r]; FBL[1][r] = FBL[0][r]; }
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 33; ++i) {
                    const int term = i / 11, u = i - term * 11;
                    const int tile = u < 10 ? (u >> 1) : 5, j = u < 10 ? (u & 1) : 0;
                    const half8 av = FA[f][2 * j + (term == 0 ? 1 : 0)];
                    const half8 bv = term == 1 ? FBL[f][tile] : FBH[f][tile];
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[u], 0, 0, 0);
                }
                if (LDSR) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const half8 v = ldsr(lds, it * 2 + f + 1, r, lane);
                        if (r < 4) FA[f ^ 1][r] = v;
                        else if (r < 10) FBH[f ^ 1][r - 4] = v;
                        else FBL[f ^ 1][r - 10] = v;
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.0f;
        for (int i = 0; i < 11; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[(size_t)blockIdx.x * 256 + tid] = s;
        if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
    } else if (MODE == 1) {
        floatx4 acc[44];
        for (int i = 0; i < 44; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
        // A: 4 channel tiles of 16 x {hi, lo}; B: 12 column tiles of 16 x {hi, lo}
        half8 FA[2][8], FB[2][24];
        for (int r = 0; r < 8; ++r) FA[0][r] = ldsr(lds, 0, r, lane);
        for (int r = 0; r < 24; ++r) FB[0][r] = ldsr(lds, 0, 8 + r, lane);
        for (int r = 0; r < 8; ++r) FA[1][r] = FA[0][r];
        for (int r = 0; r < 24; ++r) FB[1][r] = FB[0][r];
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 132; ++i) {
                    const int term = i / 44, u = i - term * 44;
                    // u < 40: column tile u / 4, channel tile u % 4; u >= 40: column tile 10 + (u - 40) / 2, channel tile (u - 40) % 2
                    const int ct = u < 40 ? (u >> 2) : 10 + ((u - 40) >> 1), ch = u < 40 ? (u & 3) : ((u - 40) & 1);
                    const half8 av = FA[f][ch * 2 + (term == 0 ? 1 : 0)];
                    const half8 bv = FB[f][ct * 2 + (term == 1 ? 1 : 0)];
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[u], 0, 0, 0);
                }
                if (LDSR) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const half8 v = ldsr(lds, it * 2 + f + 1, r, lane);
                        if (r < 8) FA[f ^ 1][r] = v;
                        else FB[f ^ 1][r - 8] = v;
                    }
#pragma unroll
                    for (int g = 0; g < 32; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RPM, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 132 - 32 * RPM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.0f;
        for (int i = 0; i < 44; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[(size_t)blockIdx.x * 256 + tid] = s;
        if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
    } else if (MODE == 2) {
        // sparse 32x32x32: A compressed half8 + index word, B half16 (two 16-byte reads); 11 accumulators x 3 terms
        floatx16 acc[11];
        for (int i = 0; i < 11; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        half8 AH[2][2], AL[2][2];
        half16 BH[2][6], BL[2][6];
        auto ld16 = [&](int cnt, int r) {
            half16 v;
            const half8 a = ldsr(lds, cnt, r, lane), b = ldsr(lds, cnt, r + 1, lane);
            for (int j = 0; j < 8; ++j) { v[j] = a[j]; v[8 + j] = b[j]; }
            return v;
        };
        for (int r = 0; r < 2; ++r) { AH[0][r] = ldsr(lds, 0, r, lane); AL[0][r] = ldsr(lds, 0, 2 + r, lane); }
        for (int r = 0; r < 6; ++r) { BH[0][r] = ld16(0, 4 + 2 * r); BL[0][r] = ld16(0, 16 + 2 * r); }
        for (int r = 0; r < 2; ++r) { AH[1][r] = AH[0][r]; AL[1][r] = AL[0][r]; }
        for (int r = 0; r < 6; ++r) { BH[1][r] = BH[0][r]; BL[1][r] = BL[0][r]; }
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 33; ++i) {
                    const int term = i / 11, u = i - term * 11;
                    const int tile = u < 10 ? (u >> 1) : 5, j = u < 10 ? (u & 1) : 0;
                    const half8 av = term == 0 ? AL[f][j] : AH[f][j];
                    const half16 bv = term == 1 ? BL[f][tile] : BH[f][tile];
                    acc[u] = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, acc[u], (int)idxw, 0, 0);
                }
                if (LDSR) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) { AH[f ^ 1][r] = ldsr(lds, it * 2 + f + 1, r, lane); AL[f ^ 1][r] = ldsr(lds, it * 2 + f + 1, 2 + r, lane); }
#pragma unroll
                    for (int r = 0; r < 6; ++r) { BH[f ^ 1][r] = ld16(it * 2 + f + 1, 4 + 2 * r); BL[f ^ 1][r] = ld16(it * 2 + f + 1, 16 + 2 * r); }
#pragma unroll
                    for (int g = 0; g < 28; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.0f;
        for (int i = 0; i < 11; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[(size_t)blockIdx.x * 256 + tid] = s;
        if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
    } else {
        // sparse 16x16x64: A compressed half8 (16 rows x 32 kept of 64), B half16; 44 accumulators x 3 terms.
        // Same wave tile, same operand bytes per FLOP: 4 channel tiles x {hi, lo} of A, 12 column tiles x {hi, lo} of B
        // cover TWICE the K of the 32x32x32 step, so one step here = two steps there.
        floatx4 acc[44];
        for (int i = 0; i < 44; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
        half8 FA[2][8];
        half16 FB[2][24];
        auto ld16 = [&](int cnt, int r) {
            half16 v;
            const half8 a = ldsr(lds, cnt, r, lane), b = ldsr(lds, cnt, r + 1, lane);
            for (int j = 0; j < 8; ++j) { v[j] = a[j]; v[8 + j] = b[j]; }
            return v;
        };
        for (int r = 0; r < 8; ++r) FA[0][r] = ldsr(lds, 0, r, lane);
        for (int r = 0; r < 24; ++r) FB[0][r] = ld16(0, 8 + 2 * r);
        for (int r = 0; r < 8; ++r) FA[1][r] = FA[0][r];
        for (int r = 0; r < 24; ++r) FB[1][r] = FB[0][r];
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 132; ++i) {
                    const int term = i / 44, u = i - term * 44;
                    const int ct = u < 40 ? (u >> 2) : 10 + ((u - 40) >> 1), ch = u < 40 ? (u & 3) : ((u - 40) & 1);
                    const half8 av = FA[f][ch * 2 + (term == 0 ? 1 : 0)];
                    const half16 bv = FB[f][ct * 2 + (term == 1 ? 1 : 0)];
                    acc[u] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(av, bv, acc[u], (int)idxw, 0, 0);
                }
                if (LDSR) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) FA[f ^ 1][r] = ldsr(lds, it * 2 + f + 1, r, lane);
#pragma unroll
                    for (int r = 0; r < 24; ++r) FB[f ^ 1][r] = ld16(it * 2 + f + 1, 8 + 2 * r);
#pragma unroll
                    for (int g = 0; g < 56; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.0f;
        for (int i = 0; i < 44; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[(size_t)blockIdx.x * 256 + tid] = s;
        if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Mode { const char *name; void (*fn)(const _Float16 *, float *, int, unsigned long long *); double flop_per_iter_wave; };

int main(int argc, char **argv)
{
    const int zeros = argc > 1 && atoi(argv[1]) == 0 ? 1 : 0;      // "0": all-zero operands (ranks by cycles only)
    const int NWG = 256;
    std::vector<_Float16> h(8192 * 8);
    srand(1234);
    for (auto &v : h) v = zeros ? (_Float16)0.0f : (_Float16)((rand() / (float)RAND_MAX) * 2.0f - 1.0f);
    _Float16 *src;
    float *out;
    unsigned long long *stamps;
    CK(hipMalloc(&src, h.size() * 2));
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)NWG * 256 * 4));
    CK(hipMalloc(&stamps, NWG * 16));
    // per loop iteration and wave: two steps; dense-equivalent FLOP
    const double f32 = 2.0 * 33 * 2.0 * 32 * 32 * 16, f16 = 2.0 * 132 * 2.0 * 16 * 16 * 32;
    const double s32 = 2.0 * 33 * 2.0 * 32 * 32 * 32, s16 = 2.0 * 132 * 2.0 * 16 * 16 * 64;
    Mode modes[] = {
        {"dense 32x32x16 regs", shape_kernel<0, 0>, f32}, {"dense 16x16x32 regs", shape_kernel<1, 0>, f16},
        {"dense 32x32x16 lds ", shape_kernel<0, 1>, f32}, {"dense 16x16x32 lds 4:1", shape_kernel<1, 1, 4>, f16},
        {"dense 16x16x32 lds 3:1", shape_kernel<1, 1, 3>, f16}, {"dense 16x16x32 lds 2:1", shape_kernel<1, 1, 2>, f16},
        {"sparse 32x32x32 regs", shape_kernel<2, 0>, s32}, {"sparse 16x16x64 regs", shape_kernel<3, 0>, s16},
        {"sparse 32x32x32 lds ", shape_kernel<2, 1>, s32}, {"sparse 16x16x64 lds ", shape_kernel<3, 1>, s16},
    };
    for (auto &m : modes) CK(hipFuncSetAttribute((const void *)m.fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 6000;     // ~ 2 x 33 x 32 cycles x 6000 = 12.7 M cycles ~ 7 ms
    printf("operands: %s; %d workgroups x 4 waves, %d iterations x 2 steps\n", zeros ? "ZEROS" : "random fp16 in [-1, 1]", NWG, iters);
    for (int round = 0; round < 3; ++round) {
        for (auto &m : modes) {
            // >= 2 s of back-to-back launches before timing (DVFS settles), then 10 timed launches
            hipEvent_t w0, w1;
            CK(hipEventCreate(&w0));
            CK(hipEventCreate(&w1));
            float warm = 0.0f;
            CK(hipEventRecord(w0));
            while (warm < (round == 0 ? 2000.0f : 1000.0f)) {
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(m.fn, dim3(NWG), dim3(256), LDS_ALLOC, 0, src, out, iters, stamps);
                CK(hipEventRecord(w1));
                CK(hipEventSynchronize(w1));
                CK(hipEventElapsedTime(&warm, w0, w1));
            }
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(m.fn, dim3(NWG), dim3(256), LDS_ALLOC, 0, src, out, iters, stamps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 10.0f;
            std::vector<unsigned long long> st(NWG * 2);
            CK(hipMemcpy(st.data(), stamps, NWG * 16, hipMemcpyDeviceToHost));
            std::vector<double> clk(NWG), cyc(NWG);
            for (int i = 0; i < NWG; ++i) { clk[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 0.1; cyc[i] = (double)st[2 * i]; }
            std::sort(clk.begin(), clk.end());
            std::sort(cyc.begin(), cyc.end());
            const double tf = m.flop_per_iter_wave * iters * 4.0 * NWG / (ms * 1e-3) / 1e12;
            printf("round %d  %-22s %8.3f ms  %8.1f TFLOP/s  clock %.3f GHz  loop cycles %.3e\n", round, m.name, ms, tf,
                   clk[NWG / 2], cyc[NWG / 2]);
            fflush(stdout);
        }
    }
    return 0;
}
