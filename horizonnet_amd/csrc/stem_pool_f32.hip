// Fused float32 stem for the 512 x 1024 panorama: input normalisation + 7x7 / 2 convolution (circular in W, zero rows in H) + folded
// BatchNorm + ReLU + 3x3 / 2 max-pool in ONE kernel (reference model.py:248-252,73-76 / torchvision resnet conv1, bn1, relu, maxpool; the LR
// padding of model.py:28-61 is the wrap of the column index).  The float32 sibling of stem_pool_bf16.hip.
//
// Why: as three launches (prep_nhwc4 + the implicit-GEMM stem + maxpool) the float32 stem moved 6.2 GB per batch of 32 -- the 2.1 GB stem
// activation is written by the conv only for the pool to read it back -- and ran 1.5 ms for 0.5 ms of matrix work (57 TF/s, the slowest
// launch group of the float32 forward).  Here a workgroup (8 waves) walks a band of convolution rows of one image at full width:
//   * input rows sit ONCE in an 8-row LDS ring as normalised float32 PLANES ([row][channel][1024]: 12 KiB per row; rows beyond the image are
//     zeros): every thread reads two pixels of the three planes a whole convolution row ahead and writes (x - mean) / std (prep_nhwc4's
//     expression) two rows later; no NHWC4 copy, no im2col;
//   * v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (64 output channels = two row tiles) and 64 pixels per wave as B: the two k of
//     an instruction are two neighbouring filter taps of one (filter row, channel) -- lanes 0..31 tap 2g, lanes 32..63 tap 2g + 1 -- so the B
//     fragment is ONE ds_read_b32 at column (2 px - 3 + 2g + half) mod 1024 of the channel's plane (lane stride 2 floats: every bank serves
//     exactly two lanes, the rate of any 64-lane read) and the A fragment one ds_read_b32 of the [64][7][3][8] weight image (row stride 169
//     floats: conflict-free).  K = 7 rows x 3 channels x 8 taps (the 8th tap's weights are zero): 21 steps x 16 MFMAs per convolution row;
//   * a lane ends up with ONE pixel and 32 channels: scale / shift / ReLU and the running maximum over the three rows of a pool window stay in
//     registers (post-ReLU values are >= 0, so 0 stands in for the -inf padding of the pool);
//   * every second row the 3-wide horizontal maximum is taken with two lane shuffles (the pixel left of a wave's first one comes from the
//     neighbouring wave through 2 KiB of LDS) and the even lanes store their pooled pixel: 32 channels as eight 16-byte pieces.
// Algorithmic HBM bytes at B = 32: 403 MB in, 537 MB out (was 403 + 537, 537 + 2147, 2147 + 537).  The k order differs from the
// implicit-GEMM stem's (taps inside channels instead of channels inside taps), so results agree to float32 rounding, not bit for bit.
#include "hn_common.h"

namespace {

constexpr int IMG_H = 512, IMG_W = 1024;
constexpr int POOL_H = 128, POOL_W = 256;
constexpr int ROW_FLOATS = 3 * IMG_W;                // one ring row: three planes
constexpr int RING_ROWS = 8;
constexpr int KPC = 7 * 3 * 8;                       // 168 k per output channel
constexpr int W_STRIDE = KPC + 1;                    // 169: odd -> the 32 rows of an A fragment hit 32 different banks
constexpr int RING_OFF = 0;                          // float offsets
constexpr int W_OFF = RING_OFF + RING_ROWS * ROW_FLOATS;           // 24576
constexpr int SC_OFF = W_OFF + 64 * W_STRIDE;                      // + 10816
constexpr int EDGE_OFF = SC_OFF + 128;                             // [8 waves][64 channels]: each wave's last pixel, for its right neighbour
constexpr int LDS_FLOATS = EDGE_OFF + 8 * 64;
constexpr int LDS_BYTES = LDS_FLOATS * 4;                          // 144,128
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

// grid = B * (128 / pseg) workgroups of 512 threads; workgroup -> image b, pooled rows [py0, py0 + pseg)
__global__ __launch_bounds__(512) void stem_pool_f32_kernel(const float* __restrict__ x, int C_in, const float* __restrict__ wpk,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ y, int pseg)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 31;
    const int half = lane >> 5;
    const int segs = POOL_H / pseg;
    const int b = blockIdx.x / segs;
    const int py0 = (blockIdx.x % segs) * pseg;

    const float* ximg = x + (size_t)b * C_in * IMG_H * IMG_W + 2 * tid;     // this thread's two pixels of a row, channel 0

    // ---- one-time: weights [o][dh][c][tap] out of the engine's packed [64][7][8 taps][4 ch] stem matrix, scale / shift ----
    for (int i = tid; i < 64 * KPC; i += 512) {
        const int o = i / KPC, k = i - o * KPC;
        const int dh = k / 24, rem = k - dh * 24;
        const int c = rem >> 3, tap = rem & 7;
        smem[W_OFF + o * W_STRIDE + k] = tap < 7 ? wpk[o * 224 + dh * 32 + tap * 4 + c] : 0.f;
    }
    if (tid < 64) {
        smem[SC_OFF + tid] = scale[tid];
        smem[SC_OFF + 64 + tid] = shift[tid];
    }

    // input row `row`: thread t holds pixels 2t, 2t + 1 of the three planes (load_row), later writes them normalised into ring slot
    // (row & 7) (store_row; reference model.py:248-252: (x[:, :3] - mean) / std); rows outside the image = 0
    auto load_row = [&](int row, float2 (&v)[3]) {
        if ((unsigned)row < (unsigned)IMG_H) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const float2*>(ximg + ((size_t)c * IMG_H + row) * IMG_W);
        }
    };
    auto store_row = [&](int row, const float2 (&v)[3]) {
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        float* dst = smem + RING_OFF + ((row + 8) & 7) * ROW_FLOATS + 2 * tid;
        const bool in = (unsigned)row < (unsigned)IMG_H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float2 o = {0.f, 0.f};
            if (in) o = float2{(v[c].x - mean[c]) / stdv[c], (v[c].y - mean[c]) / stdv[c]};
            *reinterpret_cast<float2*>(dst + c * IMG_W) = o;
        }
    };

    // this wave's pixels: conv columns 64 * wave + 32 * t + fr, t = 0, 1; xoff[t][g]: column of tap 2g + half inside a plane
    int xoff[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int px = 64 * wave + 32 * t + fr;
            xoff[t][g] = (2 * px - 3 + 2 * g + half) & (IMG_W - 1);
        }
    const float* wbase = smem + W_OFF + fr * W_STRIDE + half;          // + j * 32 * W_STRIDE + (dh * 3 + c) * 8 + 2 g

    int r = 2 * py0 - 1;                                    // first convolution row of the band (row -1 does not exist: its maxima are 0)
    const int r_last = 2 * (py0 + pseg) - 1;
    const bool seed_row = r >= 0;                           // the band's first row only seeds the running maximum
    if (r < 0) r = 0;
    float2 st[2][3] = {};                                   // the two rows in flight (registers are the ring's ninth and tenth slot)
    for (int row = 2 * r - 3; row <= 2 * r + 2; ++row) {
        load_row(row, st[0]);
        store_row(row, st[0]);
    }
    load_row(2 * r + 3, st[0]);
    load_row(2 * r + 4, st[1]);

    float vm[2][2][16];                                     // running row maximum: [t][j][q], channel 32 j + 8 (q >> 2) + 4 half + (q & 3)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) vm[t][j][q] = 0.f;

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

        // step s = (dh, c): filter row dh = s / 3, channel c = s % 3; inside a step the four tap pairs g (taps 2g, 2g + 1) x 2 pixel tiles x 2
        // channel tiles = 16 MFMAs.  The 16 fragment words of step s + 1 are requested before the MFMAs of step s (two register sets; the
        // scheduling barriers keep the compiler from sinking the reads back to their first use); the barrier that publishes row 2r+3
        // (read from step 18 on) sits in the middle of the loop.
        float xb[2][2][4], wa[2][2][4];
        auto ldfrag = [&](int s, int buf) {
            const int dh = s / 3, c = s - 3 * dh;
            const float* rowp = smem + RING_OFF + ((2 * r - 3 + dh + 8) & 7) * ROW_FLOATS + c * IMG_W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 0; t < 2; ++t) xb[buf][t][g] = rowp[xoff[t][g]];
#pragma unroll
                for (int j = 0; j < 2; ++j) wa[buf][j][g] = wbase[j * 32 * W_STRIDE + s * 8 + 2 * g];
            }
        };
        auto mma = [&](int buf) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[buf][j][g], xb[buf][t][g], acc[t][j], 0, 0, 0);
        };
        ldfrag(0, 0);
#pragma unroll
        for (int s = 0; s < 21; ++s) {
            if (s == 12) __syncthreads();                   // rows 2r+3, 2r+4 are written by every thread
            if (s + 1 < 21) ldfrag(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(s & 1);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- scale / shift / ReLU, running maximum; lane = pixel, register q -> channel 32 j + 8 (q >> 2) + 4 half + (q & 3) ----
        const bool odd = (r & 1) != 0;
        const bool emit = odd && (r > r_first || !seed_row);
        float outv[2][2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(smem + SC_OFF + 32 * j + 8 * g + 4 * half);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(smem + SC_OFF + 64 + 32 * j + 8 * g + 4 * half);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float cur = fmaxf(acc[t][j][4 * g + k] * sc[k] + sf[k], 0.f);
                        const float m = fmaxf(vm[t][j][4 * g + k], cur);
                        outv[t][j][4 * g + k] = m;
                        // an odd row closes a window and opens the next one; an even row just joins the running maximum
                        vm[t][j][4 * g + k] = odd ? cur : m;
                    }
            }
        if (!emit) continue;

        // ---- horizontal 3-maximum (pooled column q = conv columns 2q - 1, 2q, 2q + 1) + store of pooled row (r - 1) / 2 ----
        // the pixel left of a wave's first pixel belongs to the neighbouring wave: its last lane's 64 channels go through LDS
        if (fr == 31) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) smem[EDGE_OFF + wave * 64 + 32 * j + 8 * (q >> 2) + 4 * half + (q & 3)] = outv[1][j][q];
        }
        __syncthreads();
        const int py = (r - 1) >> 1;
        float* yrow = y + ((size_t)(b * POOL_H + py) * POOL_W) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 pv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = 4 * g + k;
                        const float v = outv[t][j][q];
                        float left = __shfl_up(v, 1, 64);
                        const float right = __shfl_down(v, 1, 64);
                        if (t == 1) {
                            const float prev = __shfl(outv[0][j][q], (lane & 32) | 31, 64);       // tile 0's last pixel, same half
                            left = fr == 0 ? prev : left;
                        } else if (fr == 0) {
                            left = wave > 0 ? smem[EDGE_OFF + (wave - 1) * 64 + 32 * j + 8 * g + 4 * half + k] : 0.f;
                        }
                        pv[k] = fmaxf(fmaxf(left, v), right);
                    }
                    if ((fr & 1) == 0) {
                        const int pp = 32 * wave + 16 * t + (fr >> 1);
                        *reinterpret_cast<f32x4*>(yrow + (size_t)pp * 64 + 32 * j + 8 * g + 4 * half) = pv;
                    }
                }
    }
}

}  // namespace

// x: [B][C_in >= 3][512][1024] float32 (the first three planes are used), wpk: the engine's packed float32 stem matrix [64][7][8][4]
// (hn_launch_pack_conv, stem), scale / shift: folded BatchNorm [64], y: the pooled activation [B][128][256][64] float32.
int hn_launch_stem_pool_f32(const float* x, int C_in, const float* wpk, const float* scale, const float* shift, float* y, int B, hipStream_t s)
{
    HN_REQUIRE(B >= 1 && C_in >= 3, "stem_pool f32: empty batch / fewer than 3 input planes");
    static bool attr_done[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pool_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done[dev] = true;
    }
    // pooled rows per workgroup: as many as still give ~256 workgroups (a band recomputes one convolution row in 2 * pseg + 1)
    int pseg = 16;
    while (pseg > 1 && (long)B * (POOL_H / pseg) < 256) pseg >>= 1;
    hipLaunchKernelGGL(stem_pool_f32_kernel, dim3((unsigned)(B * (POOL_H / pseg))), dim3(512), LDS_BYTES, s, x, C_in, wpk, scale, shift, y, pseg);
    HN_LAUNCH_CHECK();
    return 0;
}
