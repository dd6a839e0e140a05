// common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes, hard-coded everywhere (no dual paths, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MX_OK 0
#define MX_ERR_ARG (-1)       // bad shape / null pointer / out-of-range parameter
#define MX_ERR_UNSUPPORTED (-2)  // e.g. delay line larger than the LDS budget
#define MX_ERR_LAUNCH (-3)    // hipGetLastError() after launch

#define MX_EXPORT extern "C" __attribute__((visibility("default")))

#define MX_WAVE 64


static inline int mx_launch_status()
{
    return hipGetLastError() == hipSuccess ? MX_OK : MX_ERR_LAUNCH;
}

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a PER-DEVICE property of the loaded
// code object: one process may drive several GPUs, so the "already set" latch is kept per device (a plain static bool would
// leave the second device without the attribute and its launches failing with MX_ERR_LAUNCH).  The latch only caches an
// idempotent driver call; it carries no state a caller could observe.
struct MxLdsLatch { bool set[64]; };
static inline int mx_set_dyn_lds(MxLdsLatch &latch, const void *fn, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool cached = dev >= 0 && dev < 64;
    if (cached && latch.set[dev]) return MX_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return MX_ERR_LAUNCH;
    if (cached) latch.set[dev] = true;
    return MX_OK;
}

// ---- wave-level reductions (64 lanes, DPP/permute based via __shfl_xor) ---------------------
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sum over a 256-thread workgroup (fixed order: lanes by butterfly, then the four waves); red: 4 doubles of LDS
__device__ __forceinline__ double block256_sum_f64(double v, double *red)
{
    v = wave_sum_f64(v);
    __syncthreads();                                            // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// torch.remainder(a, b) for fp32: fmod (exact) + sign fix (aten BinaryOpsKernel.cpp).
__device__ __forceinline__ float torch_remainderf(float a, float b)
{
    float m = fmodf(a, b);
    if ((m != 0.0f) && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

// align_corners=True linear-interpolation source rule of aten (UpSample.h), fp32:
//   real = scale * i;  i0 = min(int(real), n_in-1);  lam1 = clamp(real - i0, 0, 1)
// combined as fma(lam0, x[i0], lam1 * x[i1])  -- the contraction torch's CPU kernel performs.
struct InterpTap { int i0, i1; float lam0, lam1; };
__device__ __forceinline__ InterpTap interp_tap(float scale, int i, int n_in)
{
    InterpTap t;
    float real = __fmul_rn(scale, (float)i);
    int i0 = (int)real;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    float l1 = __fsub_rn(real, (float)i0);
    l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
    t.i0 = i0;
    t.i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    t.lam1 = l1;
    t.lam0 = __fsub_rn(1.0f, l1);
    return t;
}
__device__ __forceinline__ float interp_combine(const InterpTap &t, float x0, float x1)
{
    return __fmaf_rn(t.lam0, x0, __fmul_rn(t.lam1, x1));
}
static inline float interp_scale_host(int64_t n_in, int64_t n_out)
{
    return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
}
