// Train-mode BatchNorm statistics out of the PER-WAVE epilogues (ping-pong / dw-reuse kernels): every lane accumulates the sum and the sum
// of squares of its 8 channels over the rows it stores; at the end of the kernel (or when a persistent workgroup moves to another column
// tile) the lanes that share a channel group are added up across the wave and GROUPS lanes issue the double atomics (stat_commit.h's
// replica addressing: ConvDesc::stat_rep).
#pragma once
#include <hip/hip_runtime.h>

struct HnWaveStats {
    float s1[8], s2[8];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    }
    template <typename V4>
    __device__ __forceinline__ void add(const V4& v0, const V4& v1)
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[k] += v0[k]; s2[k] += v0[k] * v0[k];
            s1[4 + k] += v1[k]; s2[4 + k] += v1[k] * v1[k];
        }
    }
    // GROUPS = lanes per row of the epilogue's store pattern (8: 64 columns per wave, 4: 32); lane % GROUPS = the lane's channel group,
    // col0 = first channel of the wave's column slice
    template <int GROUPS>
    __device__ __forceinline__ void flush(double* stat_sum, double* stat_sq, int rep, int cout, int col0, int lane)
    {
#pragma unroll
        for (int off = GROUPS; off < 64; off <<= 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1[k] += __shfl_xor(s1[k], off);
                s2[k] += __shfl_xor(s2[k], off);
            }
        if (lane < GROUPS) {
            const size_t r = rep > 1 ? (size_t)(1 + (blockIdx.x & (unsigned)(rep - 1))) * 2 * (size_t)cout : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                atomicAdd(stat_sum + r + col0 + 8 * lane + k, (double)s1[k]);
                atomicAdd(stat_sq + r + col0 + 8 * lane + k, (double)s2[k]);
            }
        }
        zero();
    }
};
