// HBM-bound helper kernels of the HorizonNet forward (gfx950): layout/normalise prologue,
// max-pool, circular up-sample + flatten, the Linear(1024,12) head, and the weight packers.
// All are coalesced, vectorised where the layout allows, and launched with enough
// workgroups (>> 256) to fill the chip.
#include "hn_common.h"

namespace {

// ---- (x[:, :3] - mean) / std, NCHW -> NHWC4 (4th channel = 0) ------------------------------
// reference model.py:248-252.  One thread per pixel: three coalesced plane reads, one 16-byte store.
__global__ __launch_bounds__(256) void prep_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         long npix_per_img, long total, int C_in)
{
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / npix_per_img;
        const long pix = i - b * npix_per_img;
        const float* src = x + b * C_in * npix_per_img + pix;
        f32x4 v;
        v[0] = (src[0] - mean[0]) / stdv[0];
        v[1] = (src[npix_per_img] - mean[1]) / stdv[1];
        v[2] = (src[2 * npix_per_img] - mean[2]) / stdv[2];
        v[3] = 0.f;
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// ---- 3x3 / stride 2 / pad 1 max-pool, NHWC, ordinary (-inf) padding on BOTH axes -------------
// torchvision ResNet.maxpool as called at reference model.py:76 (wrap_lr_pad skips it, model.py:44).
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int Hi, int Wi, int Ho, int Wo, int C4, long total)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long b = t / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * 2 - 1 + dh;
            if ((unsigned)hi >= (unsigned)Hi) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wo * 2 - 1 + dw;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((b * Hi + hi) * Wi + wi) * (long)C4 + c4) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
    }
}

// ---- circular linear up-sample along W (x f) + (c, h) flatten + concat ----------------------
// reference model.py:151-155 (pad 1 col each side, bilinear align_corners=False to f*(Wq+2), crop f)
// == circular lerp with src = (j + 0.5)/f - 0.5 (SURVEY.md section 4 KAT 2); model.py:175-178 flatten.
// in: [B][hq][Wq][cq];  seq row = t*B + b, column = col0 + c*hq + h.
__global__ __launch_bounds__(256) void upsample_flatten_kernel(const float* __restrict__ in, float* __restrict__ seq,
                                                               int B, int hq, int Wq, int cq, int col0, int f)
{
    const int t = blockIdx.x;            // output column 0..255
    const int b = blockIdx.y;
    // torch: src = scale*(dst+0.5)-0.5 evaluated on the padded grid, dst = t + f, scale = 1/f
    const float src = (1.0f / (float)f) * ((float)(t + f) + 0.5f) - 0.5f;   // >= 0.5, no clamp
    const int i0p = (int)src;            // index in the padded row (pad = 1 column)
    const float w1 = src - (float)i0p;
    const float w0 = 1.0f - w1;
    int i0 = i0p - 1;
    i0 = i0 < 0 ? i0 + Wq : i0;
    int i1 = i0p;                        // (i0p + 1) - 1
    i1 = i1 >= Wq ? i1 - Wq : i1;
    float* dst = seq + ((long)t * B + b) * 1024 + col0;
    const int n = cq * hq;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        // enumerate (h, c) with c fastest so the NHWC reads are coalesced
        const int c = e % cq;
        const int h = e / cq;
        const long base = ((long)b * hq + h) * Wq;
        const float a = in[(base + i0) * cq + c];
        const float bb = in[(base + i1) * cq + c];
        dst[c * hq + h] = w0 * a + w1 * bb;
    }
}

// ---- Linear(1024, 12) + output interleave ---------------------------------------------------
// reference model.py:266-269,278-279: out[b, ch, 4t+s] = lin[t, b, 4ch+s]; cor = ch 0, bon = ch 1,2.
// One wave per (t, b) row: 16 floats per lane, 12 dot products, butterfly reduction.
__global__ __launch_bounds__(256) void linear_head_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ bon,
                                                          float* __restrict__ cor, int T, int B)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)T * B) return;
    const int t = (int)(row / B);
    const int b = (int)(row % B);
    f32x4 xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const f32x4*>(y + row * 1024 + q * 256 + lane * 4);
    float acc[12];
#pragma unroll
    for (int o = 0; o < 12; ++o) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + o * 1024 + q * 256 + lane * 4);
            s += xv[q][0] * wv[0] + xv[q][1] * wv[1] + xv[q][2] * wv[2] + xv[q][3] * wv[3];
        }
        acc[o] = s;
    }
#pragma unroll
    for (int o = 0; o < 12; ++o) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
    }
    if (lane < 12) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < 12; ++o) v = (lane == o) ? acc[o] : v;
        v += bias[lane];
        const int ch = lane >> 2, s = lane & 3;
        const int W = 4 * T;
        if (ch == 0) cor[(long)b * W + 4 * t + s] = v;
        else bon[((long)b * 2 + (ch - 1)) * W + 4 * t + s] = v;
    }
}

// ---- weight packers -------------------------------------------------------------------------
// OIHW -> [Cout][kh][kw][Cin]; the 7x7 stem -> [64][7][8][4] zero padded (8th tap / 4th channel = 0).
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                        int Cout, int Cin, int KH, int KW, int KWp, int Cp)
{
    const long total = (long)Cout * KH * KWp * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        long t = i / Cp;
        const int dw = (int)(t % KWp);
        t /= KWp;
        const int dh = (int)(t % KH);
        const int o = (int)(t / KH);
        float v = 0.f;
        if (c < Cin && dw < KW) v = w[(((long)o * Cin + c) * KH + dh) * KW + dw];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void fold_bn_kernel(const float* gamma, const float* beta, const float* mean,
                                                      const float* var, const float* bias, float* scale,
                                                      float* shift, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const float b = bias ? bias[i] : 0.f;
    if (gamma) {
        const float s = gamma[i] / sqrtf(var[i] + 1e-5f);
        scale[i] = s;
        shift[i] = (b - mean[i]) * s + beta[i];
    } else {
        scale[i] = 1.f;
        shift[i] = b;
    }
}

__global__ __launch_bounds__(256) void add_vec_kernel(const float* a, const float* b, float* out, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = a[i] + (b ? b[i] : 0.f);
}

inline unsigned grid_for(long total, int per_block = 256, long cap = 256L * 16)
{
    long g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

int hn_launch_prep_nhwc4(const float* x, float* out, int B, int C_in, int H, int W, hipStream_t s)
{
    const long npix = (long)H * W, total = npix * B;
    hipLaunchKernelGGL(prep_nhwc4_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, out, npix, total, C_in);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_maxpool(const float* in, float* out, int B, int Hi, int Wi, int C, hipStream_t s)
{
    HN_REQUIRE(C % 4 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "maxpool: C%%4, even H/W required");
    const int Ho = Hi / 2, Wo = Wi / 2, C4 = C / 4;
    const long total = (long)B * Ho * Wo * C4;
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, Hi, Wi, Ho, Wo, C4, total);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_upsample_flatten(const float* in, float* seq, int B, int hq, int Wq, int cq, int col0, hipStream_t s)
{
    HN_REQUIRE(Wq > 0 && 256 % Wq == 0, "upsample: 256 %% Wq(%d) != 0", Wq);
    HN_REQUIRE(col0 + cq * hq <= 1024, "upsample: column range exceeds 1024");
    hipLaunchKernelGGL(upsample_flatten_kernel, dim3(256, B), dim3(256), 0, s, in, seq, B, hq, Wq, cq, col0, 256 / Wq);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_linear_head(const float* y, const float* w, const float* bias, float* bon, float* cor, int T, int B,
                          hipStream_t s)
{
    const long rows = (long)T * B;
    hipLaunchKernelGGL(linear_head_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, y, w, bias, bon, cor, T, B);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_pack_conv(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s)
{
    const bool stem = (KH == 7);
    const int KWp = stem ? 8 : KW, Cp = stem ? 4 : Cin;
    const long total = (long)Cout * KH * KWp * Cp;
    hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, out, Cout, Cin, KH, KW, KWp, Cp);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                      float* scale, float* shift, int C, hipStream_t s)
{
    hipLaunchKernelGGL(fold_bn_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, mean, var, bias, scale, shift, C);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_add_vec(const float* a, const float* b, float* out, long n, hipStream_t s)
{
    hipLaunchKernelGGL(add_vec_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, out, n);
    HN_LAUNCH_CHECK();
    return 0;
}
