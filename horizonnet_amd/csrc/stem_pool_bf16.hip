// Fused bf16 stem for the 512 x 1024 panorama: 7x7 / 2 convolution (circular in W, zero rows in H) + folded BatchNorm + ReLU +
// 3x3 / 2 max-pool in ONE kernel (reference model.py:73-81 / torchvision resnet conv1, bn1, relu, maxpool; the LR padding of
// model.py:28-61 is the wrap of the column index).
//
// Why: as an implicit GEMM the stem expands every input pixel 12x on its way into LDS (64 KiB of LDS-DMA per 128 output pixels
// for 5 KiB of image) and writes 537 MB of activations that the pool kernel reads straight back (0.39 + 0.20 ms of a 7.6 ms
// forward, neither at a roofline).  Here a workgroup walks a band of convolution rows of one image at full width:
//   * input rows sit ONCE in an 8-row LDS ring as normalised NHWC4 bf16 (1024 px x 4 ch = 8 KiB): every thread reads two pixels
//     of the three float32 planes a whole convolution row ahead, holds them in registers, and writes (x - mean) / std as 16 bytes
//     (the separate normalise / re-layout pass and its 134 MB round trip are gone; rows beyond the image are zeros);
//     the MFMA B fragments are read straight out of the ring -- 8 consecutive k = two neighbouring pixels = 16 aligned bytes at
//     column (2*px - 4 + 2*g) mod 1024 -- no im2col copy exists anywhere;
//   * the weights (64 x 224 of the packed [64][256] stem matrix, k = (filter row, 8-pixel window, 4 channels), the engine's
//     existing packing) are the A operand, so a lane holds ONE pixel and 32 channels of the 32x32 result: scale/shift/ReLU,
//     the bf16 rounding and the running maximum over the three rows of a pool window stay in registers (post-ReLU values are
//     >= 0, so their bf16 bit patterns order like unsigned integers and 0 stands in for the -inf padding);
//   * every second row the row maxima go through a 37 KiB LDS strip (half a row at a time) for the 3-wide horizontal maximum
//     and leave as whole 128-byte pooled pixels.
// The k order and the 16-wide MFMA steps are those of the implicit-GEMM stem, so the result is bit-identical to
// hn_launch_prep_nhwc4_bf16 + hn_launch_conv_bf16(stem) + hn_launch_maxpool_bf16 (tests/test_gpu_bf16.py).  Algorithmic HBM bytes
// at B = 32: 403 MB in (float32 planes), 134 MB out (was 403 + 134, 134 + 537, 537 + 134).
#include "hn_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int IMG_H = 512, IMG_W = 1024;
constexpr int POOL_H = 128, POOL_W = 256;
constexpr int ROW_BYTES = IMG_W * 8;                 // one NHWC4 bf16 input row
constexpr int RING_ROWS = 8;
constexpr int W_STRIDE = 464;                        // bytes per output channel in LDS: 224 k x 2 B + 16 (odd multiple of 16: conflict-free b128 rows)
constexpr int H_STRIDE = 144;                        // bytes per pixel of the pooling strip: 64 ch x 2 B + 16
constexpr int RING_OFF = 0;
constexpr int W_OFF = RING_OFF + RING_ROWS * ROW_BYTES;            // 65536
constexpr int SC_OFF = W_OFF + 64 * W_STRIDE;                      // + 29696
constexpr int H_OFF = SC_OFF + 512;                                // scale[64], shift[64] float32
constexpr int LDS_BYTES = H_OFF + 257 * H_STRIDE;                  // + 37008 = 132752
// optional tail: layer1.0.conv1 (1x1, 64 -> 64, BN, ReLU) on every pooled half row while it is still in LDS
constexpr int W1_STRIDE = 144;                                     // 64 k x 2 B + 16
constexpr int W1_OFF = LDS_BYTES;                                  // [64][W1_STRIDE]
constexpr int SC1_OFF = W1_OFF + 64 * W1_STRIDE;                   // scale[64], shift[64] float32
constexpr int P_OFF = SC1_OFF + 512;                               // pooled half row [128][H_STRIDE] bf16: the 1x1 conv's B operand
constexpr int LDS_BYTES_C1 = P_OFF + 128 * H_STRIDE;               // 160912
static_assert(LDS_BYTES_C1 <= 160 * 1024, "LDS");

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32x4 pk_max4(u32x4 a, u32x4 b)
{
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(u16x8, a), __builtin_bit_cast(u16x8, b)));
}
// grid = B * (128 / pseg) workgroups of 512 threads; workgroup -> image b, pooled rows [py0, py0 + pseg)
// TRAIN: the training step's stem conv -- convolution rows [2 py0, 2 (py0 + pseg)) written as z (bf16, no affine / ReLU / pooling: the
// BatchNorm of a training step needs the batch statistics first) into y [B][256][512][64], with the per-channel sum / sum of squares
// of the float32 accumulators added to stat_sum / stat_sq (the implicit-GEMM stem's STATS epilogue, 1.5 ms at B = 64).
template <bool TRAIN>
__global__ __launch_bounds__(512) void stem_pool_bf16_kernel(const float* __restrict__ x, int C_in, const u16* __restrict__ wpk,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             u16* __restrict__ y, int pseg, const u16* __restrict__ w1,
                                                             const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                             u16* __restrict__ t1, double* __restrict__ stat_sum, double* __restrict__ stat_sq)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 31;
    const int half = lane >> 5;
    const int segs = POOL_H / pseg;
    const int b = blockIdx.x / segs;
    const int py0 = (blockIdx.x % segs) * pseg;

    const float* ximg = x + (size_t)b * C_in * IMG_H * IMG_W + 2 * tid;     // this thread's two pixels of a row, channel 0

    // ---- one-time: weights (224 of every 256 packed k), scale / shift ----
    for (int i = tid; i < 64 * 28; i += 512) {              // 28 sixteen-byte pieces per output channel
        const int o = i / 28, pc = i - o * 28;
        *reinterpret_cast<u32x4*>(smem + W_OFF + o * W_STRIDE + pc * 16) = *reinterpret_cast<const u32x4*>(wpk + o * 256 + pc * 8);
    }
    if (tid < 64) {
        reinterpret_cast<float*>(smem + SC_OFF)[tid] = TRAIN ? 0.f : scale[tid];       // TRAIN: the statistics' reduction scratch
        reinterpret_cast<float*>(smem + SC_OFF)[64 + tid] = TRAIN ? 0.f : shift[tid];
    }
    const bool fuse_c1 = !TRAIN && w1 != nullptr;           // (launch-uniform; the launch then carries LDS_BYTES_C1)
    if (fuse_c1) {
        {                                                   // 64 rows x 8 sixteen-byte pieces = 512 threads
            const int o = tid >> 3, pc = tid & 7;
            *reinterpret_cast<u32x4*>(smem + W1_OFF + o * W1_STRIDE + pc * 16) = *reinterpret_cast<const u32x4*>(w1 + o * 64 + pc * 8);
        }
        if (tid < 64) {
            reinterpret_cast<float*>(smem + SC1_OFF)[tid] = scale1[tid];
            reinterpret_cast<float*>(smem + SC1_OFF)[64 + tid] = shift1[tid];
        }
    }

    // input row `row`: thread t holds pixels 2t, 2t + 1 of the three planes (load_row), later writes them normalised as NHWC4
    // bf16 into ring slot (row & 7) (store_row; reference model.py:248-252: (x[:, :3] - mean) / std); rows outside the image = 0
    auto load_row = [&](int row, float2 (&v)[3]) {
        if ((unsigned)row < (unsigned)IMG_H) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const float2*>(ximg + ((size_t)c * IMG_H + row) * IMG_W);
        }
    };
    auto store_row = [&](int row, const float2 (&v)[3]) {
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        u32x4 o = {0u, 0u, 0u, 0u};
        if ((unsigned)row < (unsigned)IMG_H)
            o = u32x4{pack_bf16((v[0].x - mean[0]) / stdv[0], (v[1].x - mean[1]) / stdv[1]), pack_bf16((v[2].x - mean[2]) / stdv[2], 0.f),
                      pack_bf16((v[0].y - mean[0]) / stdv[0], (v[1].y - mean[1]) / stdv[1]), pack_bf16((v[2].y - mean[2]) / stdv[2], 0.f)};
        *reinterpret_cast<u32x4*>(smem + RING_OFF + ((row + 8) & 7) * ROW_BYTES + tid * 16) = o;
    };

    // this wave's pixels: conv columns 64 * wave + 32 * t + fr, t = 0, 1
    unsigned xoff[2][2];                                    // [t][k-step parity]: byte offset of the lane's 16-byte window piece in a ring row
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int px = 64 * wave + 32 * t + fr;
            xoff[t][sub] = (unsigned)(((2 * px - 4 + 2 * (2 * sub + half)) & (IMG_W - 1)) * 8);
        }
    const char* wbase = smem + W_OFF + fr * W_STRIDE + half * 16;      // + j * 32 * W_STRIDE + s * 32

    int r = TRAIN ? 2 * py0 : 2 * py0 - 1;                  // first convolution row of the band (row -1 does not exist: its maxima are 0)
    const int r_last = 2 * (py0 + pseg) - 1;
    const bool seed_row = !TRAIN && r >= 0;                 // the band's first row only seeds the running maximum
    if (r < 0) r = 0;
    float ssum[2][16], ssq[2][16];                          // TRAIN: this lane's pixels, channel 32 j + 8 (q >> 2) + 4 half + (q & 3)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) { ssum[j][q] = 0.f; ssq[j][q] = 0.f; }
    float2 st[2][3] = {};                                   // the two rows in flight (registers are the ring's ninth and tenth slot)
    for (int row = 2 * r - 3; row <= 2 * r + 2; ++row) {
        load_row(row, st[0]);
        store_row(row, st[0]);
    }
    load_row(2 * r + 3, st[0]);
    load_row(2 * r + 4, st[1]);

    unsigned vm[2][2][8];                                   // running row maximum, packed bf16 pairs: [t][j][g * 2 + pair]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) vm[t][j][q] = 0u;

    const int r_first = r;
    for (; r <= r_last; ++r) {
        __syncthreads();                                    // rows 2r-3 .. 2r+2 are in the ring; the slots of rows 2r-5, 2r-4 are free
        store_row(2 * r + 3, st[0]);                        // requested one convolution row ago
        store_row(2 * r + 4, st[1]);
        load_row(2 * r + 5, st[0]);                         // needed one convolution row from now
        load_row(2 * r + 6, st[1]);

        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[t][j][q] = 0.f;

        // k-step s: filter row s >> 1, window pixels 4 * (s & 1) .. + 3.  The fragments of step s + 1 are requested before the
        // MFMAs of step s (two register sets); the barrier that publishes row 2r+3 (read from step 12 on) sits in the middle.
        u32x4 xb[2][2], wa[2][2];
        auto ldfrag = [&](int s, int buf) {
            const char* rowp = smem + RING_OFF + ((2 * r - 3 + (s >> 1) + 8) & 7) * ROW_BYTES;
#pragma unroll
            for (int t = 0; t < 2; ++t) xb[buf][t] = *reinterpret_cast<const u32x4*>(rowp + xoff[t][s & 1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) wa[buf][j] = *reinterpret_cast<const u32x4*>(wbase + j * 32 * W_STRIDE + s * 32);
        };
        auto mma = [&](int buf) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[buf][j]), __builtin_bit_cast(bf16x8, xb[buf][t]),
                                                                        acc[t][j], 0, 0, 0);
        };
        ldfrag(0, 0);
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            if (s == 6) __syncthreads();                    // rows 2r+3, 2r+4 are written by every thread
            if (s + 1 < 14) ldfrag(s + 1, (s + 1) & 1);
            mma(s & 1);
        }

        if (TRAIN) {
            // statistics of the float32 accumulators; z row r as bf16 through the strip, half a row (256 pixels) at a time
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        ssum[j][q] += acc[t][j][q];
                        ssq[j][q] += acc[t][j][q] * acc[t][j][q];
                    }
            u16* zrow = y + ((size_t)(b * 256 + r) * 512) * 64;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if ((wave >> 2) == hf) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        char* hp = smem + H_OFF + (64 * (wave & 3) + 32 * t + fr) * H_STRIDE + 8 * half;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const u32x2 v = {pack_bf16(acc[t][j][4 * g], acc[t][j][4 * g + 1]), pack_bf16(acc[t][j][4 * g + 2], acc[t][j][4 * g + 3])};
                                *reinterpret_cast<u32x2*>(hp + (32 * j + 8 * g) * 2) = v;
                            }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int id = tid + 512 * it;
                    const int q = id >> 3, cg = id & 7;
                    *reinterpret_cast<u32x4*>(zrow + (size_t)(256 * hf + q) * 64 + cg * 8) =
                        *reinterpret_cast<const u32x4*>(smem + H_OFF + q * H_STRIDE + cg * 16);
                }
                __syncthreads();
            }
            continue;
        }
        // ---- scale / shift / ReLU / bf16, running maximum; lane = pixel, register q -> channel 32 j + 8 (q >> 2) + 4 half + (q & 3) ----
        const bool emit = (r & 1) && (r > r_first || !seed_row);
        unsigned outp[2][2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(smem + SC_OFF + (32 * j + 8 * g + 4 * half) * 4);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(smem + SC_OFF + 256 + (32 * j + 8 * g + 4 * half) * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 v = {acc[t][j][4 * g], acc[t][j][4 * g + 1], acc[t][j][4 * g + 2], acc[t][j][4 * g + 3]};
                    v = v * sc + sf;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    const unsigned p0 = pack_bf16(v[0], v[1]), p1 = pack_bf16(v[2], v[3]);
                    const unsigned m0 = pk_max(vm[t][j][2 * g], p0), m1 = pk_max(vm[t][j][2 * g + 1], p1);
                    outp[t][j][2 * g] = m0;
                    outp[t][j][2 * g + 1] = m1;
                    // an odd row closes a window and opens the next one; an even row just joins the running maximum
                    vm[t][j][2 * g] = (r & 1) ? p0 : m0;
                    vm[t][j][2 * g + 1] = (r & 1) ? p1 : m1;
                }
            }
        if (!emit) continue;

        // ---- horizontal 3-maximum + store of pooled row (r - 1) / 2, half a row at a time through the LDS strip ----
        const int py = (r - 1) >> 1;
        u16* yrow = y + ((size_t)(b * POOL_H + py) * POOL_W) * 64;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if ((wave >> 2) == hf) {                        // the four waves that own this half write their pixels to slots 1 .. 256
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    char* hp = smem + H_OFF + (64 * (wave & 3) + 32 * t + fr + 1) * H_STRIDE + 8 * half;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x2 v = {outp[t][j][2 * g], outp[t][j][2 * g + 1]};
                            *reinterpret_cast<u32x2*>(hp + (32 * j + 8 * g) * 2) = v;
                        }
                }
            }
            if (hf == 0) {                                  // slot 0 = the pixel left of the half: nothing (0) / pixel 255 (wave 3's last lanes)
                if (wave == 0 && lane < 8) *reinterpret_cast<u32x4*>(smem + H_OFF + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            } else if (wave == 3 && fr == 31) {
                char* hp = smem + H_OFF + 8 * half;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x2 v = {outp[1][j][2 * g], outp[1][j][2 * g + 1]};
                        *reinterpret_cast<u32x2*>(hp + (32 * j + 8 * g) * 2) = v;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int id = tid + 512 * it;
                const int q = id >> 3, cg = id & 7;          // pooled pixel of this half, group of 8 channels
                const char* hp = smem + H_OFF + (2 * q) * H_STRIDE + cg * 16;
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(hp);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(hp + H_STRIDE);
                const u32x4 a2 = *reinterpret_cast<const u32x4*>(hp + 2 * H_STRIDE);
                const u32x4 pv = pk_max4(pk_max4(a0, a1), a2);
                *reinterpret_cast<u32x4*>(yrow + (size_t)(128 * hf + q) * 64 + cg * 8) = pv;
                if (fuse_c1) *reinterpret_cast<u32x4*>(smem + P_OFF + q * H_STRIDE + cg * 16) = pv;
            }
            __syncthreads();
            if (fuse_c1) {
                // layer1.0.conv1 on the 128 pooled pixels: wave = (32 pixels, 32 output channels), K = 64 = four MFMA steps in the
                // separate kernel's order; weights A, pixels B, so the BN / ReLU / rounding again happen on (pixel, 32 channels)
                // lanes; the result leaves through the (now free) pooling strip as 128-byte pixels
                const int pt = wave & 3, ct = wave >> 2;
                f32x16 a1;
#pragma unroll
                for (int q = 0; q < 16; ++q) a1[q] = 0.f;
                const char* wp = smem + W1_OFF + (ct * 32 + fr) * W1_STRIDE + half * 16;
                const char* xp = smem + P_OFF + (pt * 32 + fr) * H_STRIDE + half * 16;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const u32x4 wa = *reinterpret_cast<const u32x4*>(wp + kk * 32);
                    const u32x4 xb = *reinterpret_cast<const u32x4*>(xp + kk * 32);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, xb), a1, 0, 0, 0);
                }
                char* tp = smem + H_OFF + (pt * 32 + fr) * H_STRIDE + (ct * 32) * 2 + 8 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(smem + SC1_OFF + (32 * ct + 8 * g + 4 * half) * 4);
                    const f32x4 sf = *reinterpret_cast<const f32x4*>(smem + SC1_OFF + 256 + (32 * ct + 8 * g + 4 * half) * 4);
                    f32x4 v = {a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]};
                    v = v * sc + sf;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    const u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(tp + 16 * g) = o;
                }
                __syncthreads();
                u16* trow = t1 + ((size_t)(b * POOL_H + py) * POOL_W) * 64;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int id = tid + 512 * it;
                    const int q = id >> 3, cg = id & 7;
                    *reinterpret_cast<u32x4*>(trow + (size_t)(128 * hf + q) * 64 + cg * 8) =
                        *reinterpret_cast<const u32x4*>(smem + H_OFF + q * H_STRIDE + cg * 16);
                }
                __syncthreads();
            }
        }
    }
    if (TRAIN) {
        // per channel: the 32 pixel lanes of a half wave (xor butterfly), then the eight waves' partials through LDS in WAVE ORDER (round 6: they
        // used to meet in float LDS atomics, i.e. in arrival order -- the one float-order dependence of the training forward: 1 % of B = 3
        // forwards took a different bf16 rounding path downstream, tools/soak_determinism.py), then one float64 atomic per channel and workgroup
        // (a double sum of float partials of similar magnitude is exact, so its order does not matter)
        float* part = reinterpret_cast<float*>(smem + H_OFF);             // [8 waves][128]: sums, sums of squares (the pooling strip is free now)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float a = ssum[j][q], c2 = ssq[j][q];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    a += __shfl_xor(a, m, 64);
                    c2 += __shfl_xor(c2, m, 64);
                }
                if (fr == 0) {
                    const int ch = 32 * j + 8 * (q >> 2) + 4 * half + (q & 3);
                    part[wave * 128 + ch] = a;
                    part[wave * 128 + 64 + ch] = c2;
                }
            }
        __syncthreads();
        if (tid < 128) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += (double)part[w * 128 + tid];
            atomicAdd((tid < 64 ? stat_sum : stat_sq) + (tid & 63), t);
        }
    }
}

}  // namespace

// x: [B][C_in >= 3][512][1024] float32 (the first three planes are used), wpk: the packed stem matrix [64][256] bf16
// (hn_launch_pack_conv_bf16, stem), scale / shift: folded BatchNorm [64], y: [B][128][256][64] bf16.
// w1 (optional): layer1.0.conv1's packed [64][64] bf16 weights with its folded BatchNorm -> t1 [B][128][256][64] bf16 is written too.
int hn_launch_stem_pool_bf16(const float* x, int C_in, const void* wpk, const float* scale, const float* shift, void* y, int B, hipStream_t s,
                             const void* w1, const float* scale1, const float* shift1, void* t1)
{
    HN_REQUIRE(B >= 1 && C_in >= 3, "stem_pool bf16: empty batch / fewer than 3 input planes");
    HN_REQUIRE(!w1 || (scale1 && shift1 && t1), "stem_pool bf16: the fused 1x1 conv needs its scale / shift / output");
    static bool attr_done[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_C1));
        attr_done[dev] = true;
    }
    // pooled rows per workgroup: as many as still give ~256 workgroups (a band recomputes one convolution row in 2 * pseg + 1)
    int pseg = 16;
    while (pseg > 1 && (long)B * (POOL_H / pseg) < 256) pseg >>= 1;
    hipLaunchKernelGGL(stem_pool_bf16_kernel<false>, dim3((unsigned)(B * (POOL_H / pseg))), dim3(512), w1 ? LDS_BYTES_C1 : LDS_BYTES, s, x, C_in,
                       reinterpret_cast<const u16*>(wpk), scale, shift, reinterpret_cast<u16*>(y), pseg, reinterpret_cast<const u16*>(w1), scale1,
                       shift1, reinterpret_cast<u16*>(t1), (double*)nullptr, (double*)nullptr);
    HN_LAUNCH_CHECK();
    return 0;
}

// The training step's stem conv (train_precision bf16): z [B][256][512][64] bf16 = conv7x7/2 of the normalised input (no affine), with
// the per-channel sum / sum of squares of the float32 accumulators ADDED to stat_sum / stat_sq [64] (zeroed by the caller).
int hn_launch_stem_conv_train_bf16(const float* x, int C_in, const void* wpk, void* z, double* stat_sum, double* stat_sq, int B, hipStream_t s)
{
    HN_REQUIRE(B >= 1 && C_in >= 3 && stat_sum && stat_sq, "stem conv (train) bf16: bad arguments");
    static bool attr_done[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done[dev] = true;
    }
    int pseg = 16;
    while (pseg > 1 && (long)B * (POOL_H / pseg) < 256) pseg >>= 1;
    hipLaunchKernelGGL(stem_pool_bf16_kernel<true>, dim3((unsigned)(B * (POOL_H / pseg))), dim3(512), LDS_BYTES, s, x, C_in,
                       reinterpret_cast<const u16*>(wpk), (const float*)nullptr, (const float*)nullptr, reinterpret_cast<u16*>(z), pseg,
                       (const u16*)nullptr, (const float*)nullptr, (const float*)nullptr, (u16*)nullptr, stat_sum, stat_sq);
    HN_LAUNCH_CHECK();
    return 0;
}
