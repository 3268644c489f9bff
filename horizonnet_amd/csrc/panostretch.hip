// Pano-Stretch image warp for gfx950: fused coordinate generation (float64) + bilinear gather.
//
// Replaces the three scipy.ndimage.map_coordinates calls + numpy trig at reference
// misc/panostretch.py:91-102 (called from dataset.py:82).  HBM-bound gather: 12,582,912
// algorithmic bytes per 512x1024x3 f32 image.  Coordinates are produced on the fly in float64
// in the reference's exact operation order (the kernel has the DP headroom: ~1 atan + 1 div per
// pixel); per-column terms are computed once per thread and reused over the block's rows,
// tan(v) of the block's rows is shared through LDS.  Sampling restates map_coordinates(order=1,
// mode='wrap'): SciPy's legacy wrap has period len-1, then lerp between floor(c) and
// min(floor(c)+1, len-1), accumulated in double in SciPy's term order, rounded to f32 once.
#include "hn_common.h"

namespace {

constexpr int PS_ROWS = 8;       // rows per workgroup
constexpr int PS_MAXB = 128;     // images per launch (stretch factors travel as kernel arguments)

struct StretchK {
    double kx[PS_MAXB];
    double ky[PS_MAXB];
};

__device__ __forceinline__ double scipy_wrap(double c, double sz)
{
    if (c < 0.0) c += sz * (double)((long long)(-c / sz) + 1);
    else if (c > sz) c -= sz * (double)((long long)(c / sz));
    return c;
}

template <int C>
__global__ __launch_bounds__(256) void pano_stretch_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           StretchK k, int H, int W, int Crt)
{
#pragma clang fp contract(off)
    __shared__ double tan_v[PS_ROWS];
    const int nc = C > 0 ? C : Crt;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * PS_ROWS;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const double PI = 3.141592653589793;
    if (threadIdx.x < PS_ROWS) {
        const int y = y0 + threadIdx.x < H ? y0 + threadIdx.x : H - 1;   // tail rows: computed, never stored
        const double v = (((double)y + 0.5) / (double)H - 0.5) * PI;
        tan_v[threadIdx.x] = tan(v);
    }
    __syncthreads();
    if (x >= W) return;

    const double kx = k.kx[b], ky = k.ky[b];
    // per-column terms: misc/panostretch.py:6-25,92,95
    const double u = (((double)x + 0.5) / (double)W - 0.5) * 2 * PI;
    const double sin_u = sin(u), cos_u = cos(u);
    const double u0 = atan2(sin_u * kx / ky, cos_u);
    const double sin_u0 = sin(u0);
    const double refx = (u0 / (2 * PI) + 0.5) * (double)W - 0.5;
    const double cx = scipy_wrap(refx, (double)(W - 1));
    const double fx = floor(cx);
    int x0 = (int)fx;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);     // only NaN coordinates (odd W: u = 0 -> 0/0, as in the reference) get here
    const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const double wx1 = cx - fx, wx0 = 1.0 - wx1;

    const float* img = src + (size_t)b * H * W * nc;
    float* out = dst + (size_t)b * H * W * nc;

    // Phase 1: coordinates of all rows (independent fp64 chains); phase 2: all gathers in flight; phase 3: stores.
    int ra0[PS_ROWS], ra1[PS_ROWS];
    double wy[PS_ROWS];
#pragma unroll
    for (int r = 0; r < PS_ROWS; ++r) {
        // misc/panostretch.py:93,96
        const double v0 = atan(tan_v[r] * sin_u0 / sin_u * ky);
        const double refy = (v0 / PI + 0.5) * (double)H - 0.5;
        const double cy = scipy_wrap(refy, (double)(H - 1));
        const double fy = floor(cy);
        int yy0 = (int)fy;
        yy0 = yy0 < 0 ? 0 : (yy0 > H - 1 ? H - 1 : yy0);
        const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1;
        ra0[r] = yy0 * W;
        ra1[r] = yy1 * W;
        wy[r] = cy - fy;
    }
    constexpr int NCL = C > 0 ? C : 1;
    if (C > 0) {
        float p00[PS_ROWS][NCL], p01[PS_ROWS][NCL], p10[PS_ROWS][NCL], p11[PS_ROWS][NCL];
#pragma unroll
        for (int r = 0; r < PS_ROWS; ++r) {
#pragma unroll
            for (int c = 0; c < NCL; ++c) {
                p00[r][c] = img[(size_t)(ra0[r] + x0) * NCL + c];
                p01[r][c] = img[(size_t)(ra0[r] + x1) * NCL + c];
                p10[r][c] = img[(size_t)(ra1[r] + x0) * NCL + c];
                p11[r][c] = img[(size_t)(ra1[r] + x1) * NCL + c];
            }
        }
#pragma unroll
        for (int r = 0; r < PS_ROWS; ++r) {
            if (y0 + r >= H) break;
            const double wy1 = wy[r], wy0 = 1.0 - wy1;
            float* o = out + ((size_t)(y0 + r) * W + x) * NCL;
#pragma unroll
            for (int c = 0; c < NCL; ++c) {
                double t = 0.0;                     // SciPy's term order, double accumulate, one rounding to f32
                t += (double)p00[r][c] * wy0 * wx0;
                t += (double)p01[r][c] * wy0 * wx1;
                t += (double)p10[r][c] * wy1 * wx0;
                t += (double)p11[r][c] * wy1 * wx1;
                o[c] = (float)t;
            }
        }
    } else {
        for (int r = 0; r < PS_ROWS && y0 + r < H; ++r) {
            const double wy1 = wy[r], wy0 = 1.0 - wy1;
            const float* q00 = img + (size_t)(ra0[r] + x0) * nc;
            const float* q01 = img + (size_t)(ra0[r] + x1) * nc;
            const float* q10 = img + (size_t)(ra1[r] + x0) * nc;
            const float* q11 = img + (size_t)(ra1[r] + x1) * nc;
            float* o = out + ((size_t)(y0 + r) * W + x) * nc;
            for (int c = 0; c < nc; ++c) {
                double t = 0.0;
                t += (double)q00[c] * wy0 * wx0;
                t += (double)q01[c] * wy0 * wx1;
                t += (double)q10[c] * wy1 * wx0;
                t += (double)q11[c] * wy1 * wx1;
                o[c] = (float)t;
            }
        }
    }
}

}  // namespace

extern "C" int hn_pano_stretch(const float* src, float* dst, const double* kx, const double* ky, int B, int H, int W,
                               int C, void* stream)
{
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    HN_REQUIRE(src && dst && kx && ky, "pano_stretch: null pointer");
    HN_REQUIRE(B >= 0 && H >= 2 && W >= 2 && C >= 1, "pano_stretch: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    hipStream_t s = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += PS_MAXB) {
        const int nb = (B - b0) < PS_MAXB ? (B - b0) : PS_MAXB;
        StretchK k;
        for (int i = 0; i < PS_MAXB; ++i) {
            k.kx[i] = i < nb ? kx[b0 + i] : 1.0;
            k.ky[i] = i < nb ? ky[b0 + i] : 1.0;
            HN_REQUIRE(k.kx[i] > 0.0 && k.ky[i] > 0.0, "pano_stretch: kx, ky must be positive");
        }
        const float* sp = src + (size_t)b0 * H * W * C;
        float* dp = dst + (size_t)b0 * H * W * C;
        dim3 grid((W + 255) / 256, (H + PS_ROWS - 1) / PS_ROWS, nb);
        if (C == 3) hipLaunchKernelGGL(pano_stretch_kernel<3>, grid, dim3(256), 0, s, sp, dp, k, H, W, C);
        else if (C == 1) hipLaunchKernelGGL(pano_stretch_kernel<1>, grid, dim3(256), 0, s, sp, dp, k, H, W, C);
        else if (C == 4) hipLaunchKernelGGL(pano_stretch_kernel<4>, grid, dim3(256), 0, s, sp, dp, k, H, W, C);
        else hipLaunchKernelGGL(pano_stretch_kernel<0>, grid, dim3(256), 0, s, sp, dp, k, H, W, C);
        HN_LAUNCH_CHECK();
    }
    return 0;
}
