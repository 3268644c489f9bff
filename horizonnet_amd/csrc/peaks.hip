// Corner-index extraction on the device: sigmoid + periodic maximum filter + peak mask.
// Restates reference inference.py:21-29 (find_N_peaks: scipy maximum_filter(size=r, mode='wrap'),
// where(max == signal), threshold) and the sigmoid of inference.py:80 for a whole batch in one
// launch.  Index work is exact: the mask is a pure comparison of float32 values.
#include "hn_common.h"

namespace {

__global__ __launch_bounds__(256) void find_peaks_kernel(const float* __restrict__ signal, int n, int r, float min_v,
                                                         int apply_sigmoid, uint8_t* __restrict__ mask,
                                                         float* __restrict__ prob)
{
    extern __shared__ float sig[];
    const int b = blockIdx.x;
    const float* s = signal + (size_t)b * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = s[i];
        if (apply_sigmoid) v = 1.0f / (1.0f + expf(-v));
        sig[i] = v;
        if (prob) prob[(size_t)b * n + i] = v;
    }
    __syncthreads();
    const int lo = -(r / 2);       // window [i + lo, i + lo + r - 1], periodic (scipy origin 0)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float m = -INFINITY;
        int j = i + lo;
        j %= n;
        if (j < 0) j += n;
        for (int q = 0; q < r; ++q) {
            m = fmaxf(m, sig[j]);
            j = (j + 1 == n) ? 0 : j + 1;
        }
        const float v = sig[i];
        mask[(size_t)b * n + i] = (m == v && v > min_v) ? 1 : 0;
    }
}

}  // namespace

extern "C" int hn_find_peaks(const float* signal, int B, int n, int r, float min_v, int apply_sigmoid, uint8_t* mask,
                             float* prob, void* stream)
{
    HN_REQUIRE(signal && mask, "find_peaks: null pointer");
    HN_REQUIRE(B >= 0 && n >= 1 && r >= 1 && r <= n && n <= 16384, "find_peaks: bad sizes B=%d n=%d r=%d", B, n, r);
    if (B == 0) return 0;
    hipLaunchKernelGGL(find_peaks_kernel, dim3(B), dim3(256), n * sizeof(float), (hipStream_t)stream, signal, n, r,
                       min_v, apply_sigmoid, mask, prob);
    HN_LAUNCH_CHECK();
    return 0;
}
