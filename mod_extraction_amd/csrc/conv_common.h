// conv_common.h -- geometry shared by the CNN kernels (K5/K6) of Spectral2DCNN
// (reference: mod_extraction/models.py:183-195).
//
// Activation planes are (B, C, H, CV_PITCH) fp32 with CV_PITCH = 352 floats per row (345 valid
// frames + 7 pad columns) so every row starts on a 128-byte line and 16-byte vector loads are
// aligned; pad columns are treated as zeros by every consumer.
// Convolution geometry is fixed by the model family: 5x13 taps, 64 output channels,
// bin dilation 1, temporal dilation T in {1,2,4,8,16}, "same" zero padding, max-pool (2,1).
#pragma once
#include "common.h"

#define CV_PITCH 352
#define CV_WT 11     // 32-wide MFMA tiles per row
#define CV_KH 5
#define CV_KW 13
#define CV_TAPS 65
#define CV_CO 64

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// D = A(32x2) * B(2x32) + C, exact fp32 (v_mfma_f32_32x32x2_f32):
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__host__ __device__ constexpr int cv_halo(int T) { return ((6 * T + 3) / 4) * 4; }
