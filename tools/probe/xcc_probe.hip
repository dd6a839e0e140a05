// Where do the workgroups of a CU-masked stream land?  Every workgroup records XCC_ID and the HW_ID fields (SE, CU).
// Answers whether one 32-bit word of a hipExtStreamCreateWithCUMask mask is one XCD on gfx950 (mod_extraction_amd/streams.py).
#include <hip/hip_runtime.h>
__global__ void where_kernel(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup resident for a while so that the launch spreads over every CU the mask allows
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
extern "C" int where(unsigned* out, int n_wg, int spin, void* stream) {
    hipLaunchKernelGGL(where_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
