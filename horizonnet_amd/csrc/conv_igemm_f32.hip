// Implicit-GEMM convolution for gfx950, float32 on the matrix cores.
//
//   y[m][n] = act( scale[n] * sum_k A[m][k] * W[n][k] + shift[n] (+ res[m][n]) )
//   m = (b, ho, wo) output pixel, n = output channel, k = (dh, dw, c) filter tap x input channel.
//
// Replaces the MKLDNN/cuDNN convolution + BatchNorm + ReLU delegates behind the 53 backbone
// convs (torchvision Bottleneck, reference model.py:73-81) and the 16 height-compression convs
// (reference model.py:123-135).  Padding follows reference model.py:27-55: zeros above/below,
// CIRCULAR left/right -- here it is index arithmetic in the tile loader, never a padded copy.
//
// Design (MI355X): NHWC activations so a filter tap's 32 input channels are one 128-byte
// line; 256-thread workgroups (one wave per SIMD), BK = 32; tiles staged through registers
// into padded LDS rows (stride 36 floats -> conflict-free ds_read_b128 fragment reads);
// v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain, 155 TF roof).  Each lane fetches 4 consecutive
// k with ONE ds_read_b128 and feeds 4 MFMAs: lanes 0-31 carry k = 8j..8j+3 and lanes 32-63
// k = 8j+4..8j+7, so MFMA q consumes k = 8j+q and 8j+4+q (any fixed k order is a valid sum).
#include "hn_common.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // floats; 144-byte rows keep 16-byte alignment

struct ConvArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int Hi, Wi, Cin, Ho, Wo, Cout;
    int KW, sh, sw, ph, pw;
    int M, K, nk, relu, ldy;
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p)
{
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int AP = BM / 32;      // A rows staged per thread
    constexpr int BP = BN / 32;      // W rows staged per thread
    constexpr int A_TILE = BM * LDS_STRIDE;
    constexpr int STAGE = (BM + BN) * LDS_STRIDE;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3;       // 0..31
    const int lcol = tid & 7;        // 16-byte column inside the 128-byte row

    const int NT = p.Cout / BN;
    const int nt = blockIdx.x % NT;
    const int mt = blockIdx.x / NT;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    // ---- per-thread A-row descriptors (output pixel -> input window origin) ----
    size_t a_base[AP];
    int a_hi0[AP], a_wi0[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = m0 + lrow + 32 * q;
        if (m < p.M) {
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int b = t / p.Ho;
            a_base[q] = (size_t)b * p.Hi * p.Wi;
            a_hi0[q] = ho * p.sh - p.ph;
            a_wi0[q] = wo * p.sw - p.pw;
        } else {
            a_base[q] = 0;
            a_hi0[q] = -(1 << 28);   // always out of range -> zero rows
            a_wi0[q] = 0;
        }
    }
    const float* wrow[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) wrow[q] = p.w + (size_t)(n0 + lrow + 32 * q) * p.K + lcol * 4;

    f32x4 ra[AP], rb[BP];
    int dh = 0, dw = 0, c0 = 0;      // filter-tap cursor of the chunk being fetched

    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int hi = a_hi0[q] + dh;
            int wi = a_wi0[q] + (STEM ? lcol : dw);
            wi = wi < 0 ? wi + p.Wi : wi;
            wi = wi >= p.Wi ? wi - p.Wi : wi;
            // branch-free: clamp the row, load, then zero the H-padding rows
            const bool ok = (unsigned)hi < (unsigned)p.Hi;
            const int hic = ok ? hi : 0;
            const size_t pix = a_base[q] + (size_t)hic * p.Wi + wi;
            const float* src = STEM ? p.x + pix * 4 : p.x + pix * p.Cin + c0 + lcol * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v = ok ? v : z;
            ra[q] = v;
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) rb[q] = *reinterpret_cast<const f32x4*>(wrow[q] + (size_t)kc * BK);
        // advance the tap cursor to the next chunk
        if (STEM) {
            dh += 1;
        } else {
            c0 += BK;
            if (c0 == p.Cin) {
                c0 = 0;
                if (++dw == p.KW) { dw = 0; ++dh; }
            }
        }
    };
    auto stage = [&](int buf) {
        float* a_s = smem + buf * STAGE;
        float* b_s = a_s + A_TILE;
#pragma unroll
        for (int q = 0; q < AP; ++q)
            *reinterpret_cast<f32x4*>(a_s + (lrow + 32 * q) * LDS_STRIDE + lcol * 4) = ra[q];
#pragma unroll
        for (int q = 0; q < BP; ++q)
            *reinterpret_cast<f32x4*>(b_s + (lrow + 32 * q) * LDS_STRIDE + lcol * 4) = rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31;
    const int fk = (lane >> 5) * 4;

    fetch(0);
    stage(0);
    __syncthreads();

    for (int kc = 0; kc < p.nk; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < p.nk;
        if (more) fetch(kc + 1);

        const float* a_s = smem + buf * STAGE + (wm * WM + fr) * LDS_STRIDE + fk;
        const float* b_s = smem + buf * STAGE + A_TILE + (wn * WN + fr) * LDS_STRIDE + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(a_s + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(b_s + j * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
        }

        if (more) stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: folded BN / bias, residual, ReLU; lanes 0-31 write one 128-byte row segment ----
    const int half = lane >> 5;
    auto epilogue = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + fr;
            const float sc = p.scale[n];
            const float sf = p.shift[n];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow0 = m0 + wm * WM + i * 32 + 4 * half;
                float rv[16];
                if (HAS_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                        const int mc = m < p.M ? m : p.M - 1;          // clamped: loads stay unconditional
                        rv[r] = p.res[(size_t)mc * p.Cout + n];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] * sc + sf;
                    if (HAS_RES) v += rv[r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (m < p.M) p.y[(size_t)m * p.ldy + n] = v;
                }
            }
        }
    };
    if (p.res) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM>
int launch_cfg(const ConvArgs& a, hipStream_t s)
{
    const size_t lds = 2 * (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
    auto kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, STEM>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int MT = hn_cdiv(a.M, BM);
    const int NT = a.Cout / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(MT * NT)), dim3(256), lds, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int hn_launch_conv(const ConvDesc& d, hipStream_t s)
{
    ConvArgs a;
    a.x = d.x; a.w = d.w; a.scale = d.scale; a.shift = d.shift; a.res = d.res; a.y = d.y;
    a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.KW = d.KW; a.sh = d.sh; a.sw = d.sw; a.ph = d.ph; a.pw = d.pw;
    a.M = d.B * d.Ho * d.Wo;
    a.relu = d.relu;
    a.ldy = d.ldy ? d.ldy : d.Cout;
    if (d.stem) {
        HN_REQUIRE(d.KH == 7 && d.KW == 7 && d.Cout == 64 && d.Cin == 4, "stem conv: expects 7x7, Cin(padded)=4, Cout=64");
        a.K = 7 * BK;
        a.nk = 7;
        return launch_cfg<128, 64, 2, 2, true>(a, s);
    }
    HN_REQUIRE(d.Cin % BK == 0, "conv: Cin=%d must be a multiple of %d", d.Cin, BK);
    HN_REQUIRE(d.Cout % 32 == 0, "conv: Cout=%d must be a multiple of 32", d.Cout);
    a.K = d.KH * d.KW * d.Cin;
    a.nk = a.K / BK;
    const long M = a.M;
    if (d.Cout % 128 == 0) {
        const long blocks128 = (long)hn_cdiv(M, 128) * (d.Cout / 128);
        if (blocks128 >= 512) return launch_cfg<128, 128, 2, 2, false>(a, s);
        return launch_cfg<64, 128, 2, 2, false>(a, s);
    }
    if (d.Cout % 64 == 0) {
        const long blocks128 = (long)hn_cdiv(M, 128) * (d.Cout / 64);
        if (blocks128 >= 512) return launch_cfg<128, 64, 2, 2, false>(a, s);
        return launch_cfg<64, 64, 2, 2, false>(a, s);
    }
    return launch_cfg<128, 32, 4, 1, false>(a, s);
}
