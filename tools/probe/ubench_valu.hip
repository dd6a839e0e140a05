// Issue-rate probe for the sample-recurrent kernels: cycles per instruction of v_fmac_f32 (plain), v_fmac_f32_dpp
// (quad_perm), v_pk_fma_f32, v_add_f32_dpp, v_exp_f32/v_rcp_f32, at 1 and 2 waves per SIMD, plus the LDS
// write -> barrier -> read round trip.   hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o _bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE> __global__ void k(float *out, long long *cyc, int iters)
{
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    float a0 = tid * 1e-3f, a1 = a0 + 1.f, w = 1.0001f, h = 0.999f;
    f2 p0 = {a0, a1}, p1 = {a1, a0}, pw = {w, w}, ph = {h, h};
    lds[tid] = a0;
    __syncthreads();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {     // ONE asm block of 8 instructions: the compiler pads s_nop only between blocks
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(h), "v"(w));
            if (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1) : "v"(h), "v"(w));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0]\n v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0]" : "+v"(p0), "+v"(p1) : "v"(pw), "v"(ph));
            if (MODE == 3) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1));
            if (MODE == 4) asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(a0), "+v"(a1));
            if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(h), "v"(w));
            if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(pw), "v"(ph));
            if (MODE == 9) asm volatile("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, 2.0, -1.0\n v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, 2.0, -1.0" : "+v"(a0));   // dependent activation chain
        }
        if (MODE == 7) {            // LDS write -> barrier -> read (b128) round trip, REP/2 per iteration is too many: do 1
            lds[tid] = a0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float4 v = *(const float4 *)&lds[(tid * 4 + 64) & 1020];
            a0 += v.x + v.y + v.z + v.w;
        }
        if (MODE == 8) {            // barrier only
            asm volatile("s_barrier" ::: "memory");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + tid] = a0 + a1 + p0.x + p0.y + p1.x + p1.y;
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[128 + blockIdx.x] = r1 - r0; }
}
template <int MODE> void run(const char *name, int threads, int per_iter)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(128), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, r = 0; for (int i = 0; i < 128; ++i) { s += h[i]; r += h[128 + i]; }
    printf("%-34s threads=%4d  %.2f memtime ticks = %.2f ns per instruction (or per round trip); memtime/memrealtime = %.2f\n", name, threads,
           s / 128 / iters / per_iter, r / 128 / iters / per_iter * 10.0, s / r);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fmac_f32 (2 chains)", threads, REP);
        run<5>("v_fmac_f32 (1 dependent chain)", threads, REP);
        run<1>("v_fmac_f32_dpp quad_perm", threads, REP);
        run<2>("v_pk_fma_f32 op_sel bcast", threads, REP);
        run<6>("v_pk_fma_f32 dependent", threads, REP);
        run<3>("v_add_f32_dpp (dependent)", threads, REP);
        run<4>("v_exp_f32 / v_rcp_f32", threads, REP);
        run<7>("lds write+barrier+read b128", threads, 1);
        run<8>("s_barrier", threads, 1);
        run<9>("exp,add,rcp,fma dependent chain", threads, REP);
    }
    return 0;
}
