// MAC issue cost with DISTINCT operand registers (the LSTM inner product): 16 v_pk_fma_f32 (op_sel broadcast) vs
// 32 v_fmac_f32 vs 32 v_fmac_f32_dpp, one asm block each, 256 / 512 / 1024 threads per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, long long *cyc, int iters)
{
    const int tid = threadIdx.x;
    f2 w[16], acc = {0.f, 0.f}, acc2 = {0.f, 0.f};
    f2 h[8];
    for (int i = 0; i < 16; ++i) w[i] = (f2){1.0f + 1e-3f * (tid + i), 1.0f - 1e-3f * (tid + 2 * i)};
    for (int i = 0; i < 8; ++i) h[i] = (f2){0.5f + 1e-4f * i, 0.25f - 1e-4f * i};
    float a0 = 0.f, a1 = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile("v_pk_fma_f32 %0, %2, %18, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %3, %18, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %4, %19, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %5, %19, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %6, %20, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %7, %20, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %8, %21, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %9, %21, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %10, %22, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %11, %22, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %12, %23, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %13, %23, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %14, %24, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %15, %24, %1 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %0, %16, %25, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %17, %25, %1 op_sel:[0,1,0]"
                         : "+v"(acc), "+v"(acc2)
                         : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
                           "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]), "v"(h[0]), "v"(h[1]), "v"(h[2]),
                           "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]));
        if (MODE == 1 || MODE == 2) {
#define F(i, j) (MODE == 1 ? "v_fmac_f32 %" #i ", %" #j ", %" #j "\n" : "v_fmac_f32_dpp %" #i ", %" #j ", %" #j " quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n")
            // 32 MACs on the 32 scalar halves of w (as weights) x h halves
            float *wf = (float *)w; float *hf = (float *)h;
            if (MODE == 1)
                asm volatile("v_fmac_f32 %0, %2, %18\n v_fmac_f32 %1, %3, %19\n v_fmac_f32 %0, %4, %20\n v_fmac_f32 %1, %5, %21\n"
                             "v_fmac_f32 %0, %6, %22\n v_fmac_f32 %1, %7, %23\n v_fmac_f32 %0, %8, %24\n v_fmac_f32 %1, %9, %25\n"
                             "v_fmac_f32 %0, %10, %18\n v_fmac_f32 %1, %11, %19\n v_fmac_f32 %0, %12, %20\n v_fmac_f32 %1, %13, %21\n"
                             "v_fmac_f32 %0, %14, %22\n v_fmac_f32 %1, %15, %23\n v_fmac_f32 %0, %16, %24\n v_fmac_f32 %1, %17, %25"
                             : "+v"(a0), "+v"(a1)
                             : "v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(wf[3]), "v"(wf[4]), "v"(wf[5]), "v"(wf[6]), "v"(wf[7]), "v"(wf[8]),
                               "v"(wf[9]), "v"(wf[10]), "v"(wf[11]), "v"(wf[12]), "v"(wf[13]), "v"(wf[14]), "v"(wf[15]), "v"(hf[0]),
                               "v"(hf[1]), "v"(hf[2]), "v"(hf[3]), "v"(hf[4]), "v"(hf[5]), "v"(hf[6]), "v"(hf[7]));
            else
                asm volatile("v_fmac_f32_dpp %0, %18, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %19, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %20, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %21, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %22, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %23, %7 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %24, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %25, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %18, %10 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %19, %11 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %20, %12 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %21, %13 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %22, %14 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %23, %15 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f32_dpp %0, %24, %16 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %25, %17 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1)
                             : "v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(wf[3]), "v"(wf[4]), "v"(wf[5]), "v"(wf[6]), "v"(wf[7]), "v"(wf[8]),
                               "v"(wf[9]), "v"(wf[10]), "v"(wf[11]), "v"(wf[12]), "v"(wf[13]), "v"(wf[14]), "v"(wf[15]), "v"(hf[0]),
                               "v"(hf[1]), "v"(hf[2]), "v"(hf[3]), "v"(hf[4]), "v"(hf[5]), "v"(hf[6]), "v"(hf[7]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = acc.x + acc.y + acc2.x + acc2.y + a0 + a1;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int threads, int macs)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(128), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    long long h[128]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 128; ++i) s += h[i];
    printf("%-28s threads=%4d  %.1f cycles per block of 16 instructions = %.2f per MAC-lane\n", name, threads, s / 128 / iters, s / 128 / iters / macs);
}
int main()
{
    for (int threads : {256, 512, 1024}) {
        run<0>("16 x v_pk_fma_f32 (32 MAC)", threads, 32);
        run<1>("16 x v_fmac_f32 (16 MAC)", threads, 16);
        run<2>("16 x v_fmac_f32_dpp (16 MAC)", threads, 16);
    }
    return 0;
}
