// Training-input pipeline for gfx950: one fused gather per batch replaces, per image, reference dataset.py:52
// (uint8 HWC -> float32 / 255), :82 (Pano-Stretch = 3 x scipy map_coordinates), :88-89 (flip), :95-96 (horizontal
// roll), :100-104 (gamma) and :123 (HWC -> CHW).  The dataset lives in HBM as uint8 HWC; the output is the float32
// NCHW batch hn_forward / hn_train_forward consume.
//
// HBM-bound: 1,572,864 B read + 6,291,456 B written = 7,864,320 algorithmic bytes per 512x1024x3 image.
// Index maps compose exactly: out[c][y][x] = stretched[y][xs][c], xs = flip ? W-1-xr : xr, xr = (x - dx) mod W.
// Stretch coordinates are float64 in the reference's operation order (see panostretch.hip); the bilinear blend is
// SciPy's (double accumulate, one rounding to f32).  Gamma: numpy computes float32 ** float32(p) with a SIMD powf
// that is itself only ~1 ulp accurate; here x^p = 2^(p log2 x) is evaluated in float64 to 2^-37 relative error
// (16-entry table reduction + degree-5 polynomials, ~30 fp64 operations instead of ocml pow's ~200) and rounded
// once to f32: correctly rounded except in ~3e-6 of the cases, <= 1 ulp from numpy (tested).
#include "hn_common.h"

#include <stdlib.h>

namespace {

constexpr int AG_ROWS = 8;
constexpr int AG_MAXB = 64;

struct AugParams {
    double kx[AG_MAXB];
    double ky[AG_MAXB];
    float gamma[AG_MAXB];       // already rounded to f32 (numpy: python-float exponent joins a float32 array as float32)
    int index[AG_MAXB];         // which dataset image
    int roll[AG_MAXB];
    unsigned char flip[AG_MAXB];
    unsigned char stretch[AG_MAXB];
    unsigned char use_gamma[AG_MAXB];
};

// tables for pow_01: 1/c_i, log2(c_i) for c_i = 1 + (i + 0.5)/16, and 2^(j/16)
__constant__ double POW_TAB[48] = {
    0x1.f07c1f07c1f08p-1, 0x1.d41d41d41d41dp-1, 0x1.bacf914c1bad0p-1, 0x1.a41a41a41a41ap-1, 0x1.8f9c18f9c18fap-1,
    0x1.7d05f417d05f4p-1, 0x1.6c16c16c16c17p-1, 0x1.5c9882b931057p-1, 0x1.4e5e0a72f0539p-1, 0x1.4141414141414p-1,
    0x1.3521cfb2b78c1p-1, 0x1.29e4129e4129ep-1, 0x1.1f7047dc11f70p-1, 0x1.15b1e5f75270dp-1, 0x1.0c9714fbcda3bp-1,
    0x1.0410410410410p-1,
    0x1.6bad3758efd87p-5, 0x1.08c588cda79e4p-3, 0x1.acf5e2db4ec94p-3, 0x1.24407ab0e073ap-2, 0x1.6e221cd9d0cdep-2,
    0x1.b47ebf73882a1p-2, 0x1.f7a8568cb06cfp-2, 0x1.1bf311e95d00ep-1, 0x1.3abb3faa02167p-1, 0x1.5848226989d34p-1,
    0x1.74b1fd64e0754p-1, 0x1.900e6160002cdp-1, 0x1.aa708f58014d3p-1, 0x1.c3e9ca2e1a055p-1, 0x1.dc899ab3ff56cp-1,
    0x1.f45e08bcf0655p-1,
    0x1.0000000000000p+0, 0x1.0b5586cf9890fp+0, 0x1.172b83c7d517bp+0, 0x1.2387a6e756238p+0, 0x1.306fe0a31b715p+0,
    0x1.3dea64c123422p+0, 0x1.4bfdad5362a27p+0, 0x1.5ab07dd485429p+0, 0x1.6a09e667f3bcdp+0, 0x1.7a11473eb0187p+0,
    0x1.8ace5422aa0dbp+0, 0x1.9c49182a3f090p+0, 0x1.ae89f995ad3adp+0, 0x1.c199bdd85529cp+0, 0x1.d5818dcfba487p+0,
    0x1.ea4afa2a490dap+0};

// x^p for a float32 x in [0, 1] (an image value) and 0.25 <= p <= 4, via float64; tab = POW_TAB staged in LDS.
__device__ __forceinline__ float pow_01(float xf, double p, const double* tab)
{
    if (!(xf > 0.f)) return 0.f;
    const double x = (double)xf;
    const long long bits = __double_as_longlong(x);
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    const int i = (int)((bits >> 48) & 15);
    const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);   // [1, 2)
    const double r = fma(m, tab[i], -1.0);                                                           // |r| <= 1/32
    double q = -0x1.ec709dc3a03fdp-3;                       // log2(1 + r) / r, Taylor to r^5
    q = fma(q, r, 0x1.2776c50ef9bffp-2);
    q = fma(q, r, -0x1.71547652b82fep-2);
    q = fma(q, r, 0x1.ec709dc3a03fdp-2);
    q = fma(q, r, -0x1.71547652b82fep-1);
    q = fma(q, r, 0x1.71547652b82fep+0);
    const double y = p * ((double)e + (tab[16 + i] + r * q));   // p log2 x
    const double n = rint(y * 16.0);
    const double g = fma(n, -0.0625, y);                    // |g| <= 1/32
    double t = 0x1.5d87fe78a6730p-10;                       // 2^g, Taylor to g^5
    t = fma(t, g, 0x1.3b2ab6fba4e77p-7);
    t = fma(t, g, 0x1.c6b08d704a0bfp-5);
    t = fma(t, g, 0x1.ebfbdff82c58ep-3);
    t = fma(t, g, 0x1.62e42fefa39efp-1);
    t = fma(t, g, 1.0);
    const int ni = (int)n;
    return (float)ldexp(tab[32 + (ni & 15)] * t, ni >> 4);
}

__device__ __forceinline__ double wrap_legacy(double c, double sz)
{
    if (c < 0.0) c += sz * (double)((long long)(-c / sz) + 1);
    else if (c > sz) c -= sz * (double)((long long)(c / sz));
    return c;
}

__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ data, float* __restrict__ dst,
                                                      AugParams p, int H, int W)
{
#pragma clang fp contract(off)
    __shared__ double tan_v[AG_ROWS];
    __shared__ double pow_tab[48];
    __shared__ float u8_tab[256];                  // i / 255.0f, correctly rounded once instead of 12 divisions per pixel
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * AG_ROWS;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const double PI = 3.141592653589793;
    if (threadIdx.x < AG_ROWS) {
        const int y = y0 + threadIdx.x < H ? y0 + threadIdx.x : H - 1;
        tan_v[threadIdx.x] = tan((((double)y + 0.5) / (double)H - 0.5) * PI);
    }
    u8_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 48) pow_tab[threadIdx.x - 64] = POW_TAB[threadIdx.x - 64];
    __syncthreads();
    if (x >= W) return;

    int xs = x - p.roll[b];                       // roll normalised to [0, W) by the launcher
    xs = xs < 0 ? xs + W : xs;
    if (p.flip[b]) xs = W - 1 - xs;

    const unsigned char* img = data + (size_t)p.index[b] * H * W * 3;
    float* out = dst + (size_t)b * 3 * H * W;
    const bool gam = p.use_gamma[b] != 0;
    const double ge = (double)p.gamma[b];

    if (!p.stretch[b]) {
        for (int r = 0; r < AG_ROWS && y0 + r < H; ++r) {
            const unsigned char* q = img + ((size_t)(y0 + r) * W + xs) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = u8_tab[q[c]];
                if (gam) v = pow_01(v, ge, pow_tab);
                out[((size_t)c * H + (y0 + r)) * W + x] = v;
            }
        }
        return;
    }

    const double kx = p.kx[b], ky = p.ky[b];
    const double u = (((double)xs + 0.5) / (double)W - 0.5) * 2 * PI;
    const double sin_u = sin(u), cos_u = cos(u);
    const double u0 = atan2(sin_u * kx / ky, cos_u);
    const double sin_u0 = sin(u0);
    const double refx = (u0 / (2 * PI) + 0.5) * (double)W - 0.5;
    const double cx = wrap_legacy(refx, (double)(W - 1));
    const double fx = floor(cx);
    int x0 = (int)fx;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
    const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const double wx1 = cx - fx, wx0 = 1.0 - wx1;

    int ra0[AG_ROWS], ra1[AG_ROWS];
    double wy[AG_ROWS];
#pragma unroll
    for (int r = 0; r < AG_ROWS; ++r) {
        const double v0 = atan(tan_v[r] * sin_u0 / sin_u * ky);
        const double refy = (v0 / PI + 0.5) * (double)H - 0.5;
        const double cy = wrap_legacy(refy, (double)(H - 1));
        const double fy = floor(cy);
        int yy0 = (int)fy;
        yy0 = yy0 < 0 ? 0 : (yy0 > H - 1 ? H - 1 : yy0);
        const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1;
        ra0[r] = yy0 * W;
        ra1[r] = yy1 * W;
        wy[r] = cy - fy;
    }
    float p00[AG_ROWS][3], p01[AG_ROWS][3], p10[AG_ROWS][3], p11[AG_ROWS][3];
#pragma unroll
    for (int r = 0; r < AG_ROWS; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p00[r][c] = u8_tab[img[(size_t)(ra0[r] + x0) * 3 + c]];
            p01[r][c] = u8_tab[img[(size_t)(ra0[r] + x1) * 3 + c]];
            p10[r][c] = u8_tab[img[(size_t)(ra1[r] + x0) * 3 + c]];
            p11[r][c] = u8_tab[img[(size_t)(ra1[r] + x1) * 3 + c]];
        }
    }
#pragma unroll
    for (int r = 0; r < AG_ROWS; ++r) {
        if (y0 + r >= H) break;
        const double wy1 = wy[r], wy0 = 1.0 - wy1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double t = 0.0;
            t += (double)p00[r][c] * wy0 * wx0;
            t += (double)p01[r][c] * wy0 * wx1;
            t += (double)p10[r][c] * wy1 * wx0;
            t += (double)p11[r][c] * wy1 * wx1;
            float v = (float)t;
            if (gam) v = pow_01(v, ge, pow_tab);
            out[((size_t)c * H + (y0 + r)) * W + x] = v;
        }
    }
}

// The 3 bytes of source pixel `pix` with ONE (unaligned) dword load instead of three byte loads: the load window is shifted back
// by one byte when it would end behind the dataset (the very last pixel).
__device__ __forceinline__ unsigned rgb_of(const unsigned char* base, size_t pix, size_t total_bytes)
{
    size_t o = pix * 3;
    unsigned sh = 0;
    if (o + 4 > total_bytes) { o -= 1; sh = 8; }
    unsigned v;
    __builtin_memcpy(&v, base + o, 4);
    return v >> sh;
}

// ---- symmetric form (power-of-two H and W): one arctangent per FOUR output pixels, see pano_stretch_sym3_kernel ------------
// The thread owns the STRETCHED-image column pair (xs, W-1-xs) and the block's row pairs (y, H-1-y); the flip / roll index maps
// are applied on the way out (out column of stretched column xs: undo the flip, then add the roll).  Bit-identical to
// augment_kernel (tested).
constexpr int AG_SYM_R = 4;

struct ColA {
    int x0, x1;
    double w0, w1;
};

__device__ __forceinline__ ColA col_of_a(double refx, int W)
{
#pragma clang fp contract(off)
    const double cx = wrap_legacy(refx, (double)(W - 1));
    const double fx = floor(cx);
    ColA c;
    int x0 = (int)fx;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
    c.x0 = x0;
    c.x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    c.w1 = cx - fx;
    c.w0 = 1.0 - c.w1;
    return c;
}

__global__ __launch_bounds__(256) void augment_sym_kernel(const unsigned char* __restrict__ data, float* __restrict__ dst, AugParams p,
                                                          int H, int W, size_t data_bytes)
{
#pragma clang fp contract(off)
    __shared__ double tan_v[AG_SYM_R];
    __shared__ double pow_tab[48];
    __shared__ float u8_tab[256];
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * AG_SYM_R;
    const int xs = blockIdx.x * 256 + threadIdx.x;       // stretched column of the left half; its mirror is W-1-xs
    const double PI = 3.141592653589793;
    if (threadIdx.x < AG_SYM_R) tan_v[threadIdx.x] = tan((((double)(y0 + threadIdx.x) + 0.5) / (double)H - 0.5) * PI);
    u8_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 48) pow_tab[threadIdx.x - 64] = POW_TAB[threadIdx.x - 64];
    __syncthreads();
    if (xs >= W / 2) return;
    const int xsb = W - 1 - xs;
    // output columns of the two stretched columns: xs = flip ? W-1-(x - roll) : x - roll  =>  x = (flip ? W-1-xs : xs) + roll
    int xoa = (p.flip[b] ? xsb : xs) + p.roll[b];
    int xob = (p.flip[b] ? xs : xsb) + p.roll[b];
    xoa = xoa >= W ? xoa - W : xoa;
    xob = xob >= W ? xob - W : xob;

    float* out = dst + (size_t)b * 3 * H * W;
    const bool gam = p.use_gamma[b] != 0;
    const double ge = (double)p.gamma[b];

    if (!p.stretch[b]) {
#pragma unroll
        for (int r = 0; r < AG_SYM_R; ++r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = h ? H - 1 - (y0 + r) : y0 + r;
                const size_t ip = (size_t)p.index[b] * H * W + (size_t)y * W;
                const unsigned qa = rgb_of(data, ip + xs, data_bytes), qb = rgb_of(data, ip + xsb, data_bytes);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float va = u8_tab[(qa >> (8 * c)) & 0xffu], vb = u8_tab[(qb >> (8 * c)) & 0xffu];
                    if (gam) { va = pow_01(va, ge, pow_tab); vb = pow_01(vb, ge, pow_tab); }
                    out[((size_t)c * H + y) * W + xoa] = va;
                    out[((size_t)c * H + y) * W + xob] = vb;
                }
            }
        }
        return;
    }

    const double kx = p.kx[b], ky = p.ky[b];
    const double u = (((double)xs + 0.5) / (double)W - 0.5) * 2 * PI;
    const double sin_u = sin(u), cos_u = cos(u);
    const double u0 = atan2(sin_u * kx / ky, cos_u);
    const double sin_u0 = sin(u0);
    const double q0 = u0 / (2 * PI);
    const ColA ca = col_of_a((q0 + 0.5) * (double)W - 0.5, W);
    const ColA cb = col_of_a((-q0 + 0.5) * (double)W - 0.5, W);

    // Phase 1: AG_SYM_R independent arctangent chains; phase 2 (per row half): all gathers in flight; phase 3: blend, gamma, stores.
    int ra0[2][AG_SYM_R], ra1[2][AG_SYM_R];
    double wy1[2][AG_SYM_R];
#pragma unroll
    for (int r = 0; r < AG_SYM_R; ++r) {
        const double v0 = atan(tan_v[r] * sin_u0 / sin_u * ky);
        const double qv = v0 / PI;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double refy = ((h ? -qv : qv) + 0.5) * (double)H - 0.5;
            const double cy = wrap_legacy(refy, (double)(H - 1));
            const double fy = floor(cy);
            int yy0 = (int)fy;
            yy0 = yy0 < 0 ? 0 : (yy0 > H - 1 ? H - 1 : yy0);
            const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1;
            ra0[h][r] = yy0 * W;
            ra1[h][r] = yy1 * W;
            wy1[h][r] = cy - fy;
        }
    }
    const size_t ipix = (size_t)p.index[b] * H * W;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned ta[AG_SYM_R][4], tb[AG_SYM_R][4];
#pragma unroll
        for (int r = 0; r < AG_SYM_R; ++r) {
            ta[r][0] = rgb_of(data, ipix + ra0[h][r] + ca.x0, data_bytes);
            ta[r][1] = rgb_of(data, ipix + ra0[h][r] + ca.x1, data_bytes);
            ta[r][2] = rgb_of(data, ipix + ra1[h][r] + ca.x0, data_bytes);
            ta[r][3] = rgb_of(data, ipix + ra1[h][r] + ca.x1, data_bytes);
            tb[r][0] = rgb_of(data, ipix + ra0[h][r] + cb.x0, data_bytes);
            tb[r][1] = rgb_of(data, ipix + ra0[h][r] + cb.x1, data_bytes);
            tb[r][2] = rgb_of(data, ipix + ra1[h][r] + cb.x0, data_bytes);
            tb[r][3] = rgb_of(data, ipix + ra1[h][r] + cb.x1, data_bytes);
        }
        asm volatile("" ::: "memory");       // every gather of this half is issued before the first store
#pragma unroll
        for (int r = 0; r < AG_SYM_R; ++r) {
            const int y = h ? H - 1 - (y0 + r) : y0 + r;
            const double w1 = wy1[h][r], w0 = 1.0 - w1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a0 = u8_tab[(ta[r][0] >> (8 * c)) & 0xffu], a1 = u8_tab[(ta[r][1] >> (8 * c)) & 0xffu];
                const float a2 = u8_tab[(ta[r][2] >> (8 * c)) & 0xffu], a3 = u8_tab[(ta[r][3] >> (8 * c)) & 0xffu];
                const float b0_ = u8_tab[(tb[r][0] >> (8 * c)) & 0xffu], b1_ = u8_tab[(tb[r][1] >> (8 * c)) & 0xffu];
                const float b2_ = u8_tab[(tb[r][2] >> (8 * c)) & 0xffu], b3_ = u8_tab[(tb[r][3] >> (8 * c)) & 0xffu];
                double t1 = 0.0, t2 = 0.0;                   // SciPy's term order, double accumulate, one rounding to f32
                t1 += (double)a0 * w0 * ca.w0;
                t1 += (double)a1 * w0 * ca.w1;
                t1 += (double)a2 * w1 * ca.w0;
                t1 += (double)a3 * w1 * ca.w1;
                t2 += (double)b0_ * w0 * cb.w0;
                t2 += (double)b1_ * w0 * cb.w1;
                t2 += (double)b2_ * w1 * cb.w0;
                t2 += (double)b3_ * w1 * cb.w1;
                float va = (float)t1, vb = (float)t2;
                if (gam) { va = pow_01(va, ge, pow_tab); vb = pow_01(vb, ge, pow_tab); }
                out[((size_t)c * H + y) * W + xoa] = va;
                out[((size_t)c * H + y) * W + xob] = vb;
            }
        }
    }
}

static bool ag_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int hn_augment_batch(const unsigned char* data, int n_images, const int* index, float* dst,
                                const double* kx, const double* ky, const int* flip, const int* roll,
                                const double* gamma, int B, int H, int W, void* stream)
{
    if (B == 0) return 0;
    HN_REQUIRE(data && index && dst, "augment_batch: null pointer");
    HN_REQUIRE(B > 0 && H >= 2 && W >= 2 && n_images > 0, "augment_batch: bad shape B=%d H=%d W=%d N=%d", B, H, W, n_images);
    hipStream_t s = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += AG_MAXB) {
        const int nb = (B - b0) < AG_MAXB ? (B - b0) : AG_MAXB;
        AugParams p = {};
        for (int i = 0; i < nb; ++i) {
            const int j = b0 + i;
            HN_REQUIRE(index[j] >= 0 && index[j] < n_images, "augment_batch: index[%d]=%d outside the dataset (%d images)", j, index[j], n_images);
            p.index[i] = index[j];
            p.kx[i] = kx ? kx[j] : 1.0;
            p.ky[i] = ky ? ky[j] : 1.0;
            HN_REQUIRE(p.kx[i] > 0.0 && p.ky[i] > 0.0, "augment_batch: kx, ky must be positive");
            p.stretch[i] = (kx && ky && !(kx[j] == 1.0 && ky[j] == 1.0)) ? 1 : 0;   // (1, 1) = stretch disabled: exact copy
            p.flip[i] = (flip && flip[j]) ? 1 : 0;
            p.roll[i] = roll ? ((roll[j] % W) + W) % W : 0;
            p.use_gamma[i] = (gamma && gamma[j] != 1.0) ? 1 : 0;
            p.gamma[i] = gamma ? (float)gamma[j] : 1.0f;
            HN_REQUIRE(!p.use_gamma[i] || (p.gamma[i] >= 0.25f && p.gamma[i] <= 4.0f), "augment_batch: gamma %g outside [0.25, 4]", (double)p.gamma[i]);
        }
        const char* env = getenv("HN_STRETCH_SYM");            // "0": the per-pixel kernel everywhere (A/B measurements, the bit-equality test)
        if (ag_pow2(H) && ag_pow2(W) && H >= 2 * AG_SYM_R && !(env && env[0] == '0')) {
            hipLaunchKernelGGL(augment_sym_kernel, dim3((W / 2 + 255) / 256, H / 2 / AG_SYM_R, nb), dim3(256), 0, s, data,
                               dst + (size_t)b0 * 3 * H * W, p, H, W, (size_t)n_images * H * W * 3);
        } else {
            dim3 grid((W + 255) / 256, (H + AG_ROWS - 1) / AG_ROWS, nb);
            hipLaunchKernelGGL(augment_kernel, grid, dim3(256), 0, s, data, dst + (size_t)b0 * 3 * H * W, p, H, W);
        }
        HN_LAUNCH_CHECK();
    }
    return 0;
}
