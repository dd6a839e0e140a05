// Probe of v_smfmac_f32_32x32x32_f16 operand layouts (no ISA manual in this image): one wave per probe.
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe_kernel(const _Float16* a, const _Float16* b, const int* idx, float* d, int b_per_probe) {
    const int l = threadIdx.x, p = blockIdx.x;
    half8 av; half16 bv;
    for (int j = 0; j < 8; ++j) av[j] = a[((size_t)p * 64 + l) * 8 + j];
    const size_t bo = b_per_probe ? (size_t)p * 64 * 16 : 0;
    for (int j = 0; j < 16; ++j) bv[j] = b[bo + (size_t)l * 16 + j];
    floatx16 c = {};
    c = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, c, idx[p * 64 + l], 0, 0);
    for (int r = 0; r < 16; ++r) d[((size_t)p * 64 + l) * 16 + r] = c[r];
}
extern "C" int probe(const void* a, const void* b, const int* idx, float* d, int n_probes, int b_per_probe) {
    hipLaunchKernelGGL(probe_kernel, dim3(n_probes), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, idx, d, b_per_probe);
    return (int)hipDeviceSynchronize();
}

// ---- v_smfmac_f32_16x16x64_f16: A = half8 (compressed 2:4, 16 rows x 32 of 64 k), B = half16 (64 k x 16 columns), D = floatx4
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void probe16_kernel(const _Float16* a, const _Float16* b, const int* idx, float* d, int b_per_probe) {
    const int l = threadIdx.x, p = blockIdx.x;
    half8 av; half16 bv;
    for (int j = 0; j < 8; ++j) av[j] = a[((size_t)p * 64 + l) * 8 + j];
    const size_t bo = b_per_probe ? (size_t)p * 64 * 16 : 0;
    for (int j = 0; j < 16; ++j) bv[j] = b[bo + (size_t)l * 16 + j];
    floatx4 c = {};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(av, bv, c, idx[p * 64 + l], 0, 0);
    for (int r = 0; r < 4; ++r) d[((size_t)p * 64 + l) * 4 + r] = c[r];
}
extern "C" int probe16(const void* a, const void* b, const int* idx, float* d, int n_probes, int b_per_probe) {
    hipLaunchKernelGGL(probe16_kernel, dim3(n_probes), dim3(64), 0, 0, (const _Float16*)a, (const _Float16*)b, idx, d, b_per_probe);
    return (int)hipDeviceSynchronize();
}
