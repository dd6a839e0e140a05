// noise.hip -- synthetic dry audio for the benchmark / test batches: uniform noise in [lo, hi) written straight where the effect
// kernels read it (the reference trains on recorded guitar; plain U(-1, 1) clips stand in for it, as in
// the reference's own smoke tests, models.py:344).  No reference counterpart.
//
// Round 5: torch's `uniform_` filled a (B, N + longest phaser lead) staging tensor (180 MB at the headline size, three launches) and
// a device-to-device copy then moved the first N samples of every row into the batch's dry channel -- on the side stream, i.e. on
// CUs the train step's convolutions were using: 0.4 ms of GPU time per step.  This kernel writes every row once, at its own
// length, at its destination (rows without a phaser: the dry channel itself; phaser rows: the staging row the phaser reads its
// lead-in from).  Generator: Philox-4x32-10 (counter = (sample / 4, clip, batch counter), key = seed), the generator behind
// torch's device RNG; values = lo + (hi - lo) * (24-bit uniform).
#include "common.h"

__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0, unsigned k1)
{
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// row i of the launch = clip rows[i] (or i): out[clip * stride + j] for j < (lens ? lens[clip] + len_add : len_add)
__global__ __launch_bounds__(256) void uniform_rows_kernel(float *__restrict__ out, long long stride, const int *__restrict__ rows,
                                                           const int *__restrict__ lens, int len_add, unsigned seed_lo,
                                                           unsigned seed_hi, unsigned counter, float lo, float span)
{
    const int clip = rows ? rows[blockIdx.y] : (int)blockIdx.y;
    const int len = (lens ? lens[clip] : 0) + len_add;
    const int j4 = blockIdx.x * 256 + threadIdx.x;             // group of 4 samples
    if (j4 * 4 >= len) return;
    unsigned c0 = (unsigned)j4, c1 = (unsigned)clip, c2 = counter, c3 = 0x6d6f6478u;
    unsigned k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float s = span * (1.0f / 16777216.0f);
    const float v[4] = {fmaf((float)(c0 >> 8), s, lo), fmaf((float)(c1 >> 8), s, lo), fmaf((float)(c2 >> 8), s, lo),
                        fmaf((float)(c3 >> 8), s, lo)};
    float *dst = out + (size_t)clip * stride + (size_t)j4 * 4;
    if (j4 * 4 + 4 <= len && ((stride & 3) == 0)) {
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j4 * 4 + e < len) dst[e] = v[e];
    }
}

// out: row of clip b at out + b * stride (16-byte aligned base; vector stores when stride % 4 == 0); rows: optional list of n_rows clip
// indices (NULL = clips 0 .. n_rows - 1); lens: optional per-CLIP int32 lengths, len_add is added to them (or is the length when lens
// is NULL); max_len bounds the grid.  seed / counter select the stream: the same (seed, counter, clip, sample) always gives the same value.
MX_EXPORT int mx_uniform_rows(float *out, int64_t stride, const int32_t *rows, int64_t n_rows, const int32_t *lens,
                              int64_t len_add, int64_t max_len, uint64_t seed, uint32_t counter, float lo, float hi, void *stream)
{
    if (!out || n_rows <= 0 || max_len <= 0 || stride < max_len || len_add < 0 || !(hi > lo)) return MX_ERR_ARG;
    if (n_rows > 65535 || max_len >= (1ll << 31)) return MX_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((max_len + 1023) / 1024), (unsigned)n_rows);
    hipLaunchKernelGGL(uniform_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, (long long)stride, rows, lens, (int)len_add,
                       (unsigned)seed, (unsigned)(seed >> 32), counter, lo, hi - lo);
    return mx_launch_status();
}
