// Box characterisation for bench.py: what THIS MI355X sustains on the two rooflines the engine is measured against.
// Round 6 measured the same library at 849 panoramas/s (fp32 forward, B = 32) on one box and 748 on another: the matrix-core clock a box
// sustains under load is not a constant of the part, so the bench line carries the box's own measured dense-MFMA rate beside the guide's peak
// (MI355X_MICROARCH.md: 157.3 TF fp32, ~2.5 PF bf16).  Not on any product path.
#include "hn_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// every wave: `iters` rounds of 4 independent 32x32 accumulator chains (one MFMA each per round): the issue pattern of the conv kernels' inner loop
template <bool BF16>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters)
{
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = 1.0f + (float)(threadIdx.x & 7) * 0.125f, b = 0.5f;
    bf16x8 ah, bh;
#pragma unroll
    for (int k = 0; k < 8; ++k) { ah[k] = (__bf16)a; bh[k] = (__bf16)b; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;      // (never true: keeps the chains alive)
}

}  // namespace

// Launches the dense-MFMA rate kernel (dtype 0 = float32 32x32x2, 1 = bf16 32x32x16) on `workgroups` x 4 waves; *flop_out = the FLOP it performs
// (the caller times the launch with events on `stream`).  scratch: workgroups * 256 floats (never written).
extern "C" int hn_probe_mfma(int dtype, int workgroups, int iters, float* scratch, double* flop_out, void* stream)
{
    HN_REQUIRE((dtype == 0 || dtype == 1) && workgroups > 0 && iters > 0 && scratch && flop_out, "hn_probe_mfma: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(mfma_rate_kernel<false>, dim3(workgroups), dim3(256), 0, s, scratch, iters);
    else hipLaunchKernelGGL(mfma_rate_kernel<true>, dim3(workgroups), dim3(256), 0, s, scratch, iters);
    HN_LAUNCH_CHECK();
    const double per_mfma = dtype == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    *flop_out = (double)workgroups * 4.0 * iters * 4.0 * per_mfma;
    return 0;
}
