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

#include <stdlib.h>

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

// Column / row terms: computed here (device libm), or -- `tab` given -- read from tables the HOST computed with numpy, the
// reference's own arithmetic: tab[b][0][x] = refx, [1][x] = sin(u0), [2][x] = sin(u) (misc/panostretch.py:92,95), tanv[y] =
// tan(v) (:17-24).  With the tables every value except the per-pixel arctangent is bit-identical to the reference's, also
// where refx sits on SciPy's wrap discontinuity (kx == ky: refx(0) = 0 -+ 1e-13 decides between column 0 and column W-1).
template <int C>
__global__ __launch_bounds__(256) void pano_stretch_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           StretchK k, int H, int W, int Crt, const double* __restrict__ tab,
                                                           const double* __restrict__ tanv)
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
        tan_v[threadIdx.x] = tanv ? tanv[y] : tan(v);
    }
    __syncthreads();
    if (x >= W) return;

    const double kx = k.kx[b], ky = k.ky[b];
    // per-column terms: misc/panostretch.py:6-25,92,95
    double sin_u, sin_u0, refx;
    if (tab) {
        const double* tb = tab + (size_t)b * 3 * W;
        refx = tb[x];
        sin_u0 = tb[W + x];
        sin_u = tb[2 * W + x];
    } else {
        const double u = (((double)x + 0.5) / (double)W - 0.5) * 2 * PI;
        sin_u = sin(u);
        const double cos_u = cos(u);
        const double u0 = atan2(sin_u * kx / ky, cos_u);
        sin_u0 = sin(u0);
        refx = (u0 / (2 * PI) + 0.5) * (double)W - 0.5;
    }
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

// ---- symmetric form (power-of-two H and W, 3 channels: the 512 x 1024 RGB panorama of the whole pipeline) ------------------
// The warp is symmetric under x -> W-1-x and y -> H-1-y: with W a power of two u(W-1-x) == -u(x) EXACTLY, sin / atan2 / atan are
// odd, so u0(W-1-x) == -u0(x), and the argument of the per-pixel arctangent, tan v * sin u0 / sin u * ky, is IDENTICAL for x and
// W-1-x and exactly negated for y and H-1-y.  One thread therefore serves a column pair and the block's row pairs: ONE atan,
// ONE chain of divisions per FOUR output pixels, each reproducing pano_stretch_kernel's value bit for bit (tested) -- the
// per-pixel float64 work drops from ~250 to ~100 instructions and the kernel moves from the fp64 issue limit towards the HBM roof.
constexpr int PS_SYM_R = 4;      // row pairs per workgroup

typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));

struct Col {
    int x0, x1;
    double w0, w1;
};

__device__ __forceinline__ Col col_of(double refx, int W)
{
#pragma clang fp contract(off)
    const double cx = scipy_wrap(refx, (double)(W - 1));
    const double fx = floor(cx);
    Col c;
    int x0 = (int)fx;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
    c.x0 = x0;
    c.x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    c.w1 = cx - fx;
    c.w0 = 1.0 - c.w1;
    return c;
}

__device__ __forceinline__ float blend(float p00, float p01, float p10, float p11, double wy0, double wy1, double wx0, double wx1)
{
#pragma clang fp contract(off)
    double t = 0.0;                     // SciPy's term order, double accumulate, one rounding to f32
    t += (double)p00 * wy0 * wx0;
    t += (double)p01 * wy0 * wx1;
    t += (double)p10 * wy1 * wx0;
    t += (double)p11 * wy1 * wx1;
    return (float)t;
}

__global__ __launch_bounds__(256) void pano_stretch_sym3_kernel(const float* __restrict__ src, float* __restrict__ dst, StretchK k,
                                                                int H, int W, const double* __restrict__ tab,
                                                                const double* __restrict__ tanv)
{
#pragma clang fp contract(off)
    __shared__ double tan_v[PS_SYM_R];
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * PS_SYM_R;            // first row of the upper half handled here; its mirror is H-1-y
    const int x = blockIdx.x * 256 + threadIdx.x;    // column of the left half; its mirror is W-1-x
    const double PI = 3.141592653589793;
    if (threadIdx.x < PS_SYM_R) {
        const double v = (((double)(y0 + threadIdx.x) + 0.5) / (double)H - 0.5) * PI;
        tan_v[threadIdx.x] = tanv ? tanv[y0 + threadIdx.x] : tan(v);
    }
    __syncthreads();
    if (x >= W / 2) return;

    const double kx = k.kx[b], ky = k.ky[b];
    const int xb = W - 1 - x;
    double sin_u, sin_u0, refx_a, refx_b;
    if (tab) {          // host (numpy) tables; the launcher has checked that they are mirror-antisymmetric like the device's
        const double* tb = tab + (size_t)b * 3 * W;
        refx_a = tb[x];
        refx_b = tb[xb];
        sin_u0 = tb[W + x];
        sin_u = tb[2 * W + x];
    } else {
        const double u = (((double)x + 0.5) / (double)W - 0.5) * 2 * PI;
        sin_u = sin(u);
        const double cos_u = cos(u);
        const double u0 = atan2(sin_u * kx / ky, cos_u);
        sin_u0 = sin(u0);
        const double q0 = u0 / (2 * PI);
        refx_a = (q0 + 0.5) * (double)W - 0.5;
        refx_b = (-q0 + 0.5) * (double)W - 0.5;               // column W-1-x: u0 -> -u0
    }
    const Col ca = col_of(refx_a, W);
    const Col cb = col_of(refx_b, W);

    const float* img = src + (size_t)b * H * W * 3;
    float* out = dst + (size_t)b * H * W * 3;

    // Phase 1: the PS_SYM_R arctangent chains are independent (instruction-level parallelism hides their dependent latency);
    // phase 2: per row half, all gathers in flight; phase 3: blends + 12-byte stores.
    int ra0[2][PS_SYM_R], ra1[2][PS_SYM_R];
    double wy1[2][PS_SYM_R];
#pragma unroll
    for (int r = 0; r < PS_SYM_R; ++r) {
        const double v0 = atan(tan_v[r] * sin_u0 / sin_u * ky);          // shared by the four pixels (+- for the mirrored rows)
        const double qv = v0 / PI;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double refy = ((h ? -qv : qv) + 0.5) * (double)H - 0.5;
            const double cy = scipy_wrap(refy, (double)(H - 1));
            const double fy = floor(cy);
            int yy0 = (int)fy;
            yy0 = yy0 < 0 ? 0 : (yy0 > H - 1 ? H - 1 : yy0);
            const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1;
            ra0[h][r] = yy0 * W;
            ra1[h][r] = yy1 * W;
            wy1[h][r] = cy - fy;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x3 pa[PS_SYM_R][4], pb[PS_SYM_R][4];       // [row][tap] for column x / column W-1-x
#pragma unroll
        for (int r = 0; r < PS_SYM_R; ++r) {
            pa[r][0] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra0[h][r] + ca.x0) * 3);
            pa[r][1] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra0[h][r] + ca.x1) * 3);
            pa[r][2] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra1[h][r] + ca.x0) * 3);
            pa[r][3] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra1[h][r] + ca.x1) * 3);
            pb[r][0] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra0[h][r] + cb.x0) * 3);
            pb[r][1] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra0[h][r] + cb.x1) * 3);
            pb[r][2] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra1[h][r] + cb.x0) * 3);
            pb[r][3] = *reinterpret_cast<const f32x3*>(img + (size_t)(ra1[h][r] + cb.x1) * 3);
        }
        asm volatile("" ::: "memory");       // every gather of this half is issued before the first store (memory-level parallelism)
#pragma unroll
        for (int r = 0; r < PS_SYM_R; ++r) {
            const int y = h ? H - 1 - (y0 + r) : y0 + r;
            const double w1 = wy1[h][r], w0 = 1.0 - w1;
            f32x3 va, vb;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                va[c] = blend(pa[r][0][c], pa[r][1][c], pa[r][2][c], pa[r][3][c], w0, w1, ca.w0, ca.w1);
                vb[c] = blend(pb[r][0][c], pb[r][1][c], pb[r][2][c], pb[r][3][c], w0, w1, cb.w0, cb.w1);
            }
            // one 12-byte store per pixel (three separate dword stores to two interleaved destinations do not merge)
            *reinterpret_cast<f32x3*>(out + ((size_t)y * W + x) * 3) = va;
            *reinterpret_cast<f32x3*>(out + ((size_t)y * W + xb) * 3) = vb;
        }
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

// env HN_STRETCH_SYM=0 selects the per-pixel kernel everywhere (A/B measurements, the bit-equality test)
static bool stretch_sym_enabled()
{
    const char* e = getenv("HN_STRETCH_SYM");
    return !(e && e[0] == '0');
}

static int pano_stretch_impl(const float* src, float* dst, const double* kx, const double* ky, const double* tab, const double* tanv,
                             int sym_ok, int B, int H, int W, int C, hipStream_t s, const char* who)
{
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    HN_REQUIRE(src && dst && kx && ky, "%s: null pointer", who);
    HN_REQUIRE(B >= 0 && H >= 2 && W >= 2 && C >= 1, "%s: bad shape B=%d H=%d W=%d C=%d", who, B, H, W, C);
    for (int b0 = 0; b0 < B; b0 += PS_MAXB) {
        const int nb = (B - b0) < PS_MAXB ? (B - b0) : PS_MAXB;
        StretchK k;
        for (int i = 0; i < PS_MAXB; ++i) {
            k.kx[i] = i < nb ? kx[b0 + i] : 1.0;
            k.ky[i] = i < nb ? ky[b0 + i] : 1.0;
            HN_REQUIRE(k.kx[i] > 0.0 && k.ky[i] > 0.0, "%s: kx, ky must be positive", who);
        }
        const float* sp = src + (size_t)b0 * H * W * C;
        float* dp = dst + (size_t)b0 * H * W * C;
        const double* tb = tab ? tab + (size_t)b0 * 3 * W : nullptr;
        dim3 grid((W + 255) / 256, (H + PS_ROWS - 1) / PS_ROWS, nb);
        if (C == 3 && pow2(H) && pow2(W) && H >= 2 * PS_SYM_R && sym_ok && stretch_sym_enabled()) {
            hipLaunchKernelGGL(pano_stretch_sym3_kernel, dim3((W / 2 + 255) / 256, H / 2 / PS_SYM_R, nb), dim3(256), 0, s, sp, dp, k, H, W, tb, tanv);
        } else if (C == 3) hipLaunchKernelGGL(pano_stretch_kernel<3>, grid, dim3(256), 0, s, sp, dp, k, H, W, C, tb, tanv);
        else if (C == 1) hipLaunchKernelGGL(pano_stretch_kernel<1>, grid, dim3(256), 0, s, sp, dp, k, H, W, C, tb, tanv);
        else if (C == 4) hipLaunchKernelGGL(pano_stretch_kernel<4>, grid, dim3(256), 0, s, sp, dp, k, H, W, C, tb, tanv);
        else hipLaunchKernelGGL(pano_stretch_kernel<0>, grid, dim3(256), 0, s, sp, dp, k, H, W, C, tb, tanv);
        HN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int hn_pano_stretch(const float* src, float* dst, const double* kx, const double* ky, int B, int H, int W,
                               int C, void* stream)
{
    return pano_stretch_impl(src, dst, kx, ky, nullptr, nullptr, 1, B, H, W, C, (hipStream_t)stream, "pano_stretch");
}

extern "C" int hn_pano_stretch_tables(const float* src, float* dst, const double* kx, const double* ky, const double* col_tables,
                                      const double* tan_v, int tables_mirror_symmetric, int B, int H, int W, int C, void* stream)
{
    HN_REQUIRE(col_tables && tan_v, "pano_stretch_tables: null table");
    return pano_stretch_impl(src, dst, kx, ky, col_tables, tan_v, tables_mirror_symmetric, B, H, W, C, (hipStream_t)stream,
                             "pano_stretch_tables");
}
