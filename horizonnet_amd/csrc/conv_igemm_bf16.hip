// Implicit-GEMM convolution for gfx950, bf16 operands on the matrix cores, f32 accumulation.
//
// Same algorithm, data flow and boundary semantics as conv_igemm_f32.hip (see the header there):
//   y[m][n] = act( scale[n] * sum_k A[m][k] * W[n][k] + shift[n] (+ res[m][n]) )
// with NHWC bf16 activations, packed bf16 weights [Cout][Cin / 64][kh][kw][64] (the filter taps are the INNER loop of a
// 64-channel chunk, see fetch()), f32 scale/shift (folded BN), bf16 (or f32) output.  It serves the bf16 inference mode that BASELINE configs 3-5 name (the reference trains
// and infers under autocast, train.py:51,273).  What changes against the f32 kernel:
//   * BK = 64 elements, so a tile row is again ONE 128-byte line = 8 chunks of 16 bytes; the LDS-DMA
//     staging, the source-side slot swizzle c ^ ((r >> 1) & 7) and the buffer-load addressing are
//     byte-for-byte the f32 kernel's;
//   * v_mfma_f32_32x32x16_bf16: one 16-byte fragment (8 consecutive k) per lane per operand feeds ONE
//     MFMA (lanes 0-31 carry k = 16s..16s+7, lanes 32-63 k = 16s+8..16s+15);
//   * the epilogue converts with v_cvt_pk_bf16_f32 (round to nearest even) and stores 16 bytes per lane.
// The 7x7 stem reads an NHWC4 bf16 image (3 real channels + 0; 8 bytes per pixel).  One 128-byte K chunk holds TWO filter
// rows: 2 x (8 pixels x 4 channels), the 8-pixel window starting at the even pixel 2*wo - 4 (tap 0 and channel 3 carry
// zero weights) so every 16-byte DMA piece is an aligned pixel pair; 7 filter rows -> 4 chunks, K = 256 (the NHWC8 /
// one-row-per-chunk form of round 1 spent K = 448 on the same 147 real products).
#include "hn_common.h"
#include "conv_bf16_args.h"
#include "stat_commit.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

// OUT_F32: the output AND the residual are float32 (train-mode z / dX (+ identity gradient), LSTM gate pre-activations).
// TR: data-gradient mode (x = dY bf16 on the (Hi,Wi) grid, y = dX on the (Ho,Wo) grid, per-class tap lists).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM, bool OUT_F32, bool TR>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(ConvArgsH p)
{
    static_assert(!(STEM && TR), "the stem has no data gradient");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int AP = BM / 32, BP = BN / 32;
    constexpr int A_BYTES = BM * ROWB;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3;
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);

    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int NT = p.Cout / BN;
    const int nt = bid % NT;
    const int mt = bid / NT;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const int hw_out = TR ? p.cHo * p.cWo : p.Ho * p.Wo;
    const int b_first = m0 / hw_out;
    const size_t img_elems = (size_t)p.Hi * p.Wi * (STEM ? 4 : p.xstride);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.x + (size_t)b_first * img_elems), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w + (size_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);

    int a_pix0[AP], a_hi0[AP], a_wi0[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = m0 + lrow + 32 * q;
        if (m < p.M) {
            const int gw = TR ? p.cWo : p.Wo, gh = TR ? p.cHo : p.Ho;
            const int wo = m % gw;
            const int t = m / gw;
            const int ho = t % gh;
            const int b = t / gh;
            a_pix0[q] = (b - b_first) * p.Hi * p.Wi;
            a_hi0[q] = TR ? p.ca + p.sh * ho + p.ph : ho * p.sh - p.ph;
            a_wi0[q] = TR ? p.cb + p.sw * wo + p.pw : wo * p.sw - p.pw;
        } else {
            a_pix0[q] = -1;
            a_hi0[q] = 0;
            a_wi0[q] = 0;
        }
    }
    unsigned w_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) w_off[q] = (unsigned)((lrow + 32 * q) * p.K + lcol * 8) * 2u;

    unsigned a_off[AP];
    auto tap_offsets = [&](int dh, int dw) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            int hi, wi;
            bool ok = a_pix0[q] >= 0;
            if (TR) {                                            // dh / dw index the class's tap lists
                const int th = a_hi0[q] - p.tdh[dh];
                int tw = a_wi0[q] - p.tdw[dw];
                tw = tw < 0 ? tw + p.Wo : tw;                     // circular on the dX grid (width Wo)
                tw = tw >= p.Wo ? tw - p.Wo : tw;
                hi = th >> p.sh_log2;
                wi = tw >> p.sw_log2;
                ok = ok && th >= 0 && hi < p.Hi;
            } else {
                if (STEM) {      // dh = chunk index: piece lcol -> filter row 2*dh + (lcol >> 2), pixel pair (lcol & 3) of the window
                    const int frow = 2 * dh + (lcol >> 2);
                    hi = a_hi0[q] + frow;
                    wi = a_wi0[q] - 1 + 2 * (lcol & 3);             // a_wi0 = 2*wo - 3: the window starts at the even pixel 2*wo - 4
                    ok = ok && frow < 7;
                } else {
                    hi = a_hi0[q] + dh;
                    wi = a_wi0[q] + dw;
                }
                wi = wi < 0 ? wi + p.Wi : wi;
                wi = wi >= p.Wi ? wi - p.Wi : wi;
                ok = ok && ((unsigned)hi < (unsigned)p.Hi);
            }
            const unsigned pix = (unsigned)(a_pix0[q] + hi * p.Wi + wi);
            const unsigned off = STEM ? pix * 8u : (pix * (unsigned)p.xstride + (unsigned)lcol * 8u) * 2u;
            a_off[q] = ok ? off : OOB;
        }
    };

    // K range of this workgroup: everything, or slice blockIdx.y of a split-K launch (deep-K convs with few output tiles:
    // the tails of the height-compression chains).  Slices are contiguous chunk ranges in the usual k order.
    int kb = 0, ke = p.nk;
    int dh = 0, dw = 0, c0 = 0;
    if (!STEM && !TR && p.ksplit > 1) {
        const int sl = blockIdx.y;
        kb = (int)((long)p.nk * sl / p.ksplit);
        ke = (int)((long)p.nk * (sl + 1) / p.ksplit);
        const int ntap = p.KH * p.KW;                    // chunk index = channel chunk * taps + tap
        const int cc = kb / ntap;
        const int tap = kb - cc * ntap;
        c0 = cc * BKE;
        dh = tap / p.KW;
        dw = tap - dh * p.KW;
    }
    tap_offsets(dh, dw);

    auto fetch = [&](int kc) {       // chunk kc -> LDS stage (kc & 1)
        char* a_s = smem + (kc & 1) * STAGE_BYTES;
        char* b_s = a_s + A_BYTES;
#pragma unroll
        for (int q = 0; q < AP; ++q) dma16(rsrc_a, a_s + (q * 4 + wave) * 1024, a_off[q], (unsigned)c0 * 2u);
#pragma unroll
        for (int q = 0; q < BP; ++q) dma16(rsrc_w, b_s + (q * 4 + wave) * 1024, w_off[q], (unsigned)kc * (unsigned)ROWB);
        if (STEM) {
            dh += 1;
            tap_offsets(dh, 0);
        } else if (TR) {                 // data gradient: taps outer, channels inner (its own per-class packing)
            c0 += BKE;
            if (c0 == p.Cin) {
                c0 = 0;
                if (++dw == p.ntdw) { dw = 0; ++dh; }
                if (dh < p.ntdh) tap_offsets(dh, dw);
            }
        } else {
            // forward: the filter taps are the INNER loop of a 64-channel chunk -- the chunk of tap (dh, dw + 1) is the chunk of
            // (dh, dw) moved by one pixel, so 127 of its 128 lines were fetched one iteration ago (L1 / L2 hits) instead of
            // Cin / 64 iterations ago; rocprofv3 FETCH_SIZE of the 3x3 convs was 3-5x their input with taps outer
            if (++dw == p.KW) {
                dw = 0;
                if (++dh == p.KH) { dh = 0; c0 += BKE; }
            }
            if (p.KH * p.KW > 1) tap_offsets(dh, dw);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;

    [[maybe_unused]] const size_t tr_wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    HN_TR_STAMP(tr_wg, 0);
    HN_TR_HWID(tr_wg);
    if (!TR || p.nk > 0) {      // a parity class no tap reaches (1x1 stride 2) has K = 0: dX = add there
        fetch(kb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    HN_TR_STAMP(tr_wg, 1);

    {
        u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        auto ldfrag = [&](u32x4 (&fa)[TM], u32x4 (&fb)[TN], int buf, int ks) {
            const int koff = ((2 * ks + half) ^ fswz) * 16;
            const char* a_s = smem + buf * STAGE_BYTES + (wm * WM + fr) * ROWB + koff;
            const char* b_s = smem + buf * STAGE_BYTES + A_BYTES + (wn * WN + fr) * ROWB + koff;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(a_s + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b_s + j * 32 * ROWB);
        };
        auto mma = [&](const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                        acc[i][j], 0, 0, 0);
        };
        if (!TR || p.nk > 0) ldfrag(fa0, fb0, kb & 1, 0);
        for (int kc = kb; kc < ke; ++kc) {
            const int buf = kc & 1;
            const bool more = kc + 1 < ke;
            if (more) fetch(kc + 1);
            ldfrag(fa1, fb1, buf, 1);
            mma(fa0, fb0);
            ldfrag(fa0, fb0, buf, 2);
            mma(fa1, fb1);
            ldfrag(fa1, fb1, buf, 3);
            mma(fa0, fb0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of chunk kc+1 has landed
            __syncthreads();
            if (more) ldfrag(fa0, fb0, buf ^ 1, 0);
            mma(fa1, fb1);
        }
        __syncthreads();
    }
    HN_TR_STAMP(tr_wg, 2);

    // ---- epilogue: f32 accumulators -> LDS -> 8 channels per lane: scale/shift, residual, ReLU, bf16 (or f32) rows ----
    constexpr int CS = BN + 4;
    constexpr int EROWS = WM;
    static_assert(EROWS * CS * 4 <= 2 * STAGE_BYTES, "C tile must fit in the staging LDS");
    constexpr int TPR = BN / 8;
    constexpr int RPP = 256 / TPR;
    constexpr int NPS = EROWS >= RPP ? EROWS / RPP : 1;     // BN = 32: one sweep, only the first EROWS row-lanes work
    float* cs = reinterpret_cast<float*>(smem);
    const int ccol = (tid % TPR) * 8;
    const int crow = tid / TPR;
    // output pixel of GEMM row m: dense, except in data-gradient mode where the rows enumerate one parity class
    auto out_pix = [&](int m) -> size_t {
        if (!TR) return (size_t)m;
        const int wo = m % p.cWo;
        const int t = m / p.cWo;
        const int ho = t % p.cHo;
        const int b = t / p.cHo;
        return ((size_t)b * p.Ho + (p.ca + p.sh * ho)) * p.Wo + (p.cb + p.sw * wo);
    };
    auto epilogue = [&](auto has_res) {
        constexpr bool HAS_RES = decltype(has_res)::value;
        u32x4 rres[HAS_RES ? WAVES_M * NPS : 1];            // bf16 residual: 8 channels
        f32x4 rres_lo[HAS_RES && OUT_F32 ? WAVES_M * NPS : 1], rres_hi[HAS_RES && OUT_F32 ? WAVES_M * NPS : 1];   // f32 residual
        if (HAS_RES) {
#pragma unroll
            for (int ps = 0; ps < WAVES_M * NPS; ++ps) {
                const int m = m0 + (ps / NPS) * EROWS + crow + (ps % NPS) * RPP;
                const int mc = m < p.M ? m : p.M - 1;
                if (OUT_F32) {
                    const float* rp = reinterpret_cast<const float*>(p.res) + out_pix(mc) * p.Cout + n0 + ccol;
                    rres_lo[ps] = *reinterpret_cast<const f32x4*>(rp);
                    rres_hi[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
                } else {
                    rres[ps] = *reinterpret_cast<const u32x4*>(p.res + out_pix(mc) * p.Cout + n0 + ccol);
                }
            }
        }
        f32x4 st1a = {0.f, 0.f, 0.f, 0.f}, st1b = st1a, st2a = st1a, st2b = st1a;     // BN statistics of this thread's 8 columns
        // bn_z: the reduce pass of the unit whose gradient these rows are (zhat = z * za + zb, as bn_bwd_reduce_h8_kernel)
        f32x4 za0 = st1a, za1 = st1a, zb0 = st1a, zb1 = st1a;
        const bool bn_red = !OUT_F32 && !TR && p.bn_z != nullptr;
        if (bn_red) {
            za0 = *reinterpret_cast<const f32x4*>(p.bn_invstd + n0 + ccol);
            za1 = *reinterpret_cast<const f32x4*>(p.bn_invstd + n0 + ccol + 4);
            zb0 = -*reinterpret_cast<const f32x4*>(p.bn_mean + n0 + ccol) * za0;
            zb1 = -*reinterpret_cast<const f32x4*>(p.bn_mean + n0 + ccol + 4) * za1;
        }
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(p.scale + n0 + ccol);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(p.scale + n0 + ccol + 4);
        const f32x4 sf0 = *reinterpret_cast<const f32x4*>(p.shift + n0 + ccol);
        const f32x4 sf1 = *reinterpret_cast<const f32x4*>(p.shift + n0 + ccol + 4);
#pragma unroll
        for (int h = 0; h < WAVES_M; ++h) {
            if (h > 0) __syncthreads();
            if (wm == h) {
                float* c_w = cs + (4 * half) * CS + wn * WN + fr;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            c_w[(i * 32 + (r & 3) + 8 * (r >> 2)) * CS + j * 32] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = crow + ps * RPP;
                if (row >= EROWS) continue;
                const int m = m0 + h * EROWS + row;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + row * CS + ccol);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + row * CS + ccol + 4);
                if (p.ksplit <= 1) {          // (split-K partial tiles stay raw: the reduce kernel applies scale / shift once)
                    v0 = v0 * sc0 + sf0;
                    v1 = v1 * sc1 + sf1;
                }
                if (HAS_RES) {
                    if (OUT_F32) {
                        v0 += rres_lo[h * NPS + ps];
                        v1 += rres_hi[h * NPS + ps];
                    } else {
                        const u32x4 rr = rres[h * NPS + ps];
                        v0[0] += bf16_lo(rr[0]); v0[1] += bf16_hi(rr[0]); v0[2] += bf16_lo(rr[1]); v0[3] += bf16_hi(rr[1]);
                        v1[0] += bf16_lo(rr[2]); v1[1] += bf16_hi(rr[2]); v1[2] += bf16_lo(rr[3]); v1[3] += bf16_hi(rr[3]);
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
                }
                if (m < p.M) {
                    if (p.stat_sum) {            // wave-uniform: inference launches skip the statistics arithmetic
                        st1a += v0; st1b += v1;
                        st2a += v0 * v0; st2b += v1 * v1;
                    }
                    if (OUT_F32) {
                        float* yo = reinterpret_cast<float*>(p.y) + (p.ksplit > 1 ? (size_t)blockIdx.y * p.M * p.ldy : (size_t)0) +
                                    out_pix(m) * p.ldy + n0 + ccol;
                        *reinterpret_cast<f32x4*>(yo) = v0;
                        *reinterpret_cast<f32x4*>(yo + 4) = v1;
                    } else {
                        u32x4 o;
                        o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                        o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.y) + out_pix(m) * p.ldy + n0 + ccol) = o;
                        if (bn_red) {           // (wave-uniform) g = stored (rounded) dy where the unit's ReLU passed; sums of g and g * zhat
                            const size_t e = (size_t)m * p.Cout + n0 + ccol;
                            const u32x4 zv = *reinterpret_cast<const u32x4*>(p.bn_z + e);
                            const unsigned mm = *reinterpret_cast<const unsigned short*>(p.bn_mask + (e >> 2));
                            const unsigned mk8 = (mm & 0xfu) | ((mm >> 4) & 0xf0u);
                            f32x4 g0, g1, z0, z1;
                            g0[0] = bf16_lo(o[0]); g0[1] = bf16_hi(o[0]); g0[2] = bf16_lo(o[1]); g0[3] = bf16_hi(o[1]);
                            g1[0] = bf16_lo(o[2]); g1[1] = bf16_hi(o[2]); g1[2] = bf16_lo(o[3]); g1[3] = bf16_hi(o[3]);
                            z0[0] = bf16_lo(zv[0]); z0[1] = bf16_hi(zv[0]); z0[2] = bf16_lo(zv[1]); z0[3] = bf16_hi(zv[1]);
                            z1[0] = bf16_lo(zv[2]); z1[1] = bf16_hi(zv[2]); z1[2] = bf16_lo(zv[3]); z1[3] = bf16_hi(zv[3]);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                g0[k] = ((mk8 >> k) & 1u) ? g0[k] : 0.f;
                                g1[k] = ((mk8 >> (4 + k)) & 1u) ? g1[k] : 0.f;
                            }
                            st1a += g0; st1b += g1;
                            st2a += g0 * (z0 * za0 + zb0); st2b += g1 * (z1 * za1 + zb1);
                        }
                        if (p.mask_out) {        // (wave-uniform) the training forward's fused BatchNorm + ReLU: the adjoint's bit mask, v > 0 as affine_act_kernel has it
                            unsigned mk = 0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) mk |= (v0[k] > 0.f ? 1u : 0u) << k | (v1[k] > 0.f ? 1u : 0u) << (8 + k);
                            *reinterpret_cast<unsigned short*>(p.mask_out + ((out_pix(m) * p.ldy + n0 + ccol) >> 2)) = (unsigned short)mk;
                        }
                    }
                }
            }
            HN_TR_STAMP(tr_wg, 3 + h);
        }
        if (p.stat_sum || bn_red) {        // see conv_igemm_f32.hip: per-channel sums of the stored tile -> one f64 atomic per channel
            __syncthreads();
            float* red = cs;                         // [2][RPP][BN]
            if (crow < RPP) {
                *reinterpret_cast<f32x4*>(red + crow * BN + ccol) = st1a;
                *reinterpret_cast<f32x4*>(red + crow * BN + ccol + 4) = st1b;
                *reinterpret_cast<f32x4*>(red + (RPP + crow) * BN + ccol) = st2a;
                *reinterpret_cast<f32x4*>(red + (RPP + crow) * BN + ccol + 4) = st2b;
            }
            __syncthreads();
            float a1 = 0.f, a2 = 0.f;
            if (tid < BN) {
#pragma unroll
                for (int r = 0; r < RPP; ++r) {
                    a1 += red[r * BN + tid];
                    a2 += red[(RPP + r) * BN + tid];
                }
            }
            if (bn_red) {            // per-tile partials, plain stores: [tile row mt][S1 | S2][Cout]
                if (tid < BN) {
                    p.bn_slab[((size_t)mt * 2) * p.Cout + n0 + tid] = a1;
                    p.bn_slab[((size_t)mt * 2 + 1) * p.Cout + n0 + tid] = a2;
                }
            } else
            hn_stat_commit(p.stat_sum, p.stat_sq, p.stat_rep, p.Cout, n0, BN, tid, a1, a2);
        }
    };
    if (p.res) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

// ---- bottleneck block 0 tail in ONE launch: y = relu(bn3(conv3(t2)) + bn_d(downsample(x)))  (model.py:78-81) -------------
// Both convolutions are 1x1 GEMMs into the same output tile.  Run separately the downsample branch writes its
// [M][Cout] result to HBM and conv3 reads it back as the residual -- for layer1 that is 2 x 537 MB of the block's
// 2.4 GB at B = 32, all of it on the HBM roof.  Here the tile keeps TWO accumulator sets: K chunks 0..nk1-1 come from t2
// (dense rows) into acc1, chunks nk1.. from x (strided pixels of the block input) into acc2; the epilogue applies each
// branch's own folded-BN scale/shift, rounds the downsample branch to bf16 exactly where the two-launch form stores it,
// adds, ReLUs, stores.  Bit-identical to conv3(res = downsample(x)) by construction (same k order, same rounding points).
struct DualArgsH {
    const u16* a1;      // t2 [M][K1]
    const u16* w1;      // conv3 weights [Cout][K1]
    const float* scale1;
    const float* shift1;
    const u16* a2;      // block input x [B][Hi2][Wi2][K2]
    const u16* w2;      // downsample weights [Cout][K2]
    const float* scale2;
    const float* shift2;
    u16* y;             // [M][Cout]
    int M, Cout, K1, K2, nk1, nk2;
    int Ho, Wo, Hi2, Wi2, s2;       // output grid; input grid and stride of the downsample conv
    int xcd_swizzle;
};

__global__ __launch_bounds__(256, 2) void conv1x1_dual_bf16_kernel(DualArgsH p)
{
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, TM = 2, TN = 2, AP = 4, BP = 4;
    constexpr int A_BYTES = BM * ROWB;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 3;
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);

    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int NT = p.Cout / BN;
    const int nt = bid % NT;
    const int mt = bid / NT;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const int hw_out = p.Ho * p.Wo;
    const int b_first = m0 / hw_out;
    const __amdgpu_buffer_rsrc_t rsrc_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.a1 + (size_t)m0 * p.K1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u16*>(p.a2 + (size_t)b_first * p.Hi2 * p.Wi2 * p.K2), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w1 + (size_t)n0 * p.K1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w2 + (size_t)n0 * p.K2), 0, 0x7fffffff, 0x00020000);

    unsigned a1_off[AP], a2_off[AP], w1_off[BP], w2_off[BP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = m0 + lrow + 32 * q;
        if (m < p.M) {
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int b = t / p.Ho;
            a1_off[q] = (unsigned)((lrow + 32 * q) * p.K1 + lcol * 8) * 2u;
            const unsigned pix = (unsigned)((b - b_first) * p.Hi2 * p.Wi2 + ho * p.s2 * p.Wi2 + wo * p.s2);
            a2_off[q] = (pix * (unsigned)p.K2 + (unsigned)lcol * 8u) * 2u;
        } else {
            a1_off[q] = OOB;
            a2_off[q] = OOB;
        }
    }
#pragma unroll
    for (int q = 0; q < BP; ++q) {
        w1_off[q] = (unsigned)((lrow + 32 * q) * p.K1 + lcol * 8) * 2u;
        w2_off[q] = (unsigned)((lrow + 32 * q) * p.K2 + lcol * 8) * 2u;
    }

    const int nk = p.nk1 + p.nk2;
    // chunk kc -> LDS stage (kc & 1).  Two loaders with their own descriptors / offsets (a runtime choice between the two
    // inside one loader makes hipcc spill both sets to scratch and select through memory)
    auto fetch1 = [&](int kc) __attribute__((always_inline)) {
        char* a_s = smem + (kc & 1) * STAGE_BYTES;
        char* b_s = a_s + A_BYTES;
        const unsigned so = (unsigned)kc * (unsigned)ROWB;
#pragma unroll
        for (int q = 0; q < AP; ++q) dma16(rsrc_a1, a_s + (q * 4 + wave) * 1024, a1_off[q], so);
#pragma unroll
        for (int q = 0; q < BP; ++q) dma16(rsrc_w1, b_s + (q * 4 + wave) * 1024, w1_off[q], so);
    };
    auto fetch2 = [&](int kc) __attribute__((always_inline)) {      // kc counts over both branches; its chunk of x / W_d is kc - nk1
        char* a_s = smem + (kc & 1) * STAGE_BYTES;
        char* b_s = a_s + A_BYTES;
        const unsigned so = (unsigned)(kc - p.nk1) * (unsigned)ROWB;
#pragma unroll
        for (int q = 0; q < AP; ++q) dma16(rsrc_a2, a_s + (q * 4 + wave) * 1024, a2_off[q], so);
#pragma unroll
        for (int q = 0; q < BP; ++q) dma16(rsrc_w2, b_s + (q * 4 + wave) * 1024, w2_off[q], so);
    };

    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;

    fetch1(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        auto ldfrag = [&](u32x4 (&fa)[TM], u32x4 (&fb)[TN], int buf, int ks) __attribute__((always_inline)) {
            const int koff = ((2 * ks + half) ^ fswz) * 16;
            const char* a_s = smem + buf * STAGE_BYTES + (wm * WM + fr) * ROWB + koff;
            const char* b_s = smem + buf * STAGE_BYTES + A_BYTES + (wn * WN + fr) * ROWB + koff;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(a_s + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b_s + j * 32 * ROWB);
        };
        auto mma = [&](f32x16 (&acc)[TM][TN], const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                        acc[i][j], 0, 0, 0);
        };
        ldfrag(fa0, fb0, 0, 0);
        // one chunk sequence, two accumulator sets; every (accumulator, prefetch source) pair is its own instantiation of the
        // loop body: static register indices, static descriptors
        auto body = [&](auto& acc, int kc, auto prefetch) __attribute__((always_inline)) {
            const int buf = kc & 1;
            prefetch();
            ldfrag(fa1, fb1, buf, 1);
            mma(acc, fa0, fb0);
            ldfrag(fa0, fb0, buf, 2);
            mma(acc, fa1, fb1);
            ldfrag(fa1, fb1, buf, 3);
            mma(acc, fa0, fb0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kc + 1 < nk) ldfrag(fa0, fb0, buf ^ 1, 0);
            mma(acc, fa1, fb1);
        };
        for (int kc = 0; kc + 1 < p.nk1; ++kc) body(acc1, kc, [&]() __attribute__((always_inline)) { fetch1(kc + 1); });
        body(acc1, p.nk1 - 1, [&]() __attribute__((always_inline)) { fetch2(p.nk1); });          // nk2 >= 1
        for (int kc = p.nk1; kc + 1 < nk; ++kc) body(acc2, kc, [&]() __attribute__((always_inline)) { fetch2(kc + 1); });
        body(acc2, nk - 1, [&]() __attribute__((always_inline)) {});
        __syncthreads();
    }

    // ---- epilogue: per wave row h, acc1 and acc2 pass through the LDS C tile one after the other ----
    constexpr int CS = BN + 4;
    constexpr int EROWS = WM;
    constexpr int TPR = BN / 8;          // 16
    constexpr int RPP = 256 / TPR;       // 16 rows per pass
    constexpr int NPS = EROWS / RPP;     // 4
    float* cs = reinterpret_cast<float*>(smem);
    const int ccol = (tid % TPR) * 8;
    const int crow = tid / TPR;
    const f32x4 s1a = *reinterpret_cast<const f32x4*>(p.scale1 + n0 + ccol), s1b = *reinterpret_cast<const f32x4*>(p.scale1 + n0 + ccol + 4);
    const f32x4 t1a = *reinterpret_cast<const f32x4*>(p.shift1 + n0 + ccol), t1b = *reinterpret_cast<const f32x4*>(p.shift1 + n0 + ccol + 4);
    const f32x4 s2a = *reinterpret_cast<const f32x4*>(p.scale2 + n0 + ccol), s2b = *reinterpret_cast<const f32x4*>(p.scale2 + n0 + ccol + 4);
    const f32x4 t2a = *reinterpret_cast<const f32x4*>(p.shift2 + n0 + ccol), t2b = *reinterpret_cast<const f32x4*>(p.shift2 + n0 + ccol + 4);
    auto to_lds = [&](const f32x16 (&acc)[TM][TN]) __attribute__((always_inline)) {
        float* c_w = cs + (4 * half) * CS + wn * WN + fr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) c_w[(i * 32 + (r & 3) + 8 * (r >> 2)) * CS + j * 32] = acc[i][j][r];
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 d0[NPS], d1[NPS];
        if (h > 0) __syncthreads();
        if (wm == h) to_lds(acc2);
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {        // downsample branch: bn_d, then the bf16 rounding of its stored form
            const float* src = cs + (crow + ps * RPP) * CS + ccol;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * s2a + t2a;
            f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4) * s2b + t2b;
            const unsigned q0 = pack_bf16(v0[0], v0[1]), q1 = pack_bf16(v0[2], v0[3]), q2 = pack_bf16(v1[0], v1[1]), q3 = pack_bf16(v1[2], v1[3]);
            d0[ps] = f32x4{bf16_lo(q0), bf16_hi(q0), bf16_lo(q1), bf16_hi(q1)};
            d1[ps] = f32x4{bf16_lo(q2), bf16_hi(q2), bf16_lo(q3), bf16_hi(q3)};
        }
        __syncthreads();
        if (wm == h) to_lds(acc1);
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int row = crow + ps * RPP;
            const int m = m0 + h * EROWS + row;
            const float* src = cs + row * CS + ccol;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * s1a + t1a + d0[ps];
            f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4) * s1b + t1b + d1[ps];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
            if (m < p.M) {
                u32x4 o;
                o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                *reinterpret_cast<u32x4*>(p.y + (size_t)m * p.Cout + n0 + ccol) = o;
            }
        }
    }
}

// ---- bottleneck tail CHAINED into the next block's conv1 (layer1; reference model.py:78-81 + torchvision Bottleneck) ----------
//   out = relu(bn3(conv3(t2)) + x)                (1x1, 64 -> 256, the block's output: kept, it is the next residual)
//   t1' = relu(bn1'(conv1'(out)))                 (1x1, 256 -> 64, the NEXT block's first conv)
// As two launches the 537 MB `out` (B = 32) is written and immediately read back; on layer1 both launches sit on the HBM roof.
// Here a workgroup owns 64 pixels: phase 1 computes their 256 output channels (one 64-deep K chunk), the epilogue stores the
// bf16 rows to HBM AND into LDS as the A operand of phase 2 (4 K chunks x 64 rows x 128 B, the tile loader's slot swizzle),
// phase 2 multiplies them with W1' (prefetched by LDS-DMA at kernel start) and stores t1'.  80 KB of LDS, 2 workgroups per CU.
// Same k order and rounding points as the two-launch form: bit-identical (tested).
struct ChainArgsH {
    const u16* a1;       // t2 [M][64]
    const u16* w3;       // [256][64]
    const float* scale3;
    const float* shift3;
    const u16* res;      // block input x [M][256]                      (plain form)
    const u16* ad;       // block input [M][64] of the downsample branch (DUAL form: block 0, stride 1)
    const u16* wd;       // downsample weights [256][64]
    const float* scale_d;
    const float* shift_d;
    u16* y;              // out [M][256]
    const u16* w1n;      // next conv1 [64][256]
    const float* scale1n;
    const float* shift1n;
    u16* y2;             // t1' [M][64 * n2_passes]
    int M;
    int xcd_swizzle;
    int n2_passes;       // 1: conv1' has 64 output channels (layer1 blocks); 2: 128 (layer2.0.conv1), two passes over the same A operand
};

constexpr int CH_BM = 64, CH_N1 = 256, CH_N2 = 64;
constexpr int CH_CS = 0;                         // f32 C staging: 16 rows x 256 floats (epilogue 1) / 64 rows x 64 floats (epilogue 2)
constexpr int CH_A2 = 16384;                     // out tile as A operand: 4 chunks x 64 rows x 128 B
constexpr int CH_W1N = CH_A2 + 32768;            // W1': 4 chunks x 64 rows x 128 B
constexpr int CH_LDS = CH_W1N + 32768;           // 81920
constexpr int CH_A1 = 0;                         // phase-1 operands overlay the staging / A2 areas (dead before those are written)
constexpr int CH_W3 = 8192;
constexpr int CH_AD = 40960;                     // DUAL: downsample operands (the W1' area is loaded after phase 1 instead of up front)
constexpr int CH_WD = 49152;

// DUAL: block 0 of layer1 -- out = relu(bn3(conv3(t2)) + bf16(bn_d(downsample(x)))) like conv1x1_dual_bf16_kernel (two accumulator
// sets, the downsample branch rounded to bf16 where the two-launch form stores it), then the chained conv1'.
template <bool DUAL>
__global__ __launch_bounds__(256, 2) void conv1x1_chain_bf16_kernel(ChainArgsH p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = tid >> 3;
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);

    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = bid * CH_BM;

    const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.a1 + (size_t)m0 * 64), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w3), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1n = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w1n), 0, 0x7fffffff, 0x00020000);
    auto load_w1n = [&](int pass) __attribute__((always_inline)) {   // W1' rows 64 pass .. +64: chunk c = their input channels 64c .. 64c+63
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                dma16(rsrc_w1n, smem + CH_W1N + c * 8192 + (q * 4 + wave) * 1024, (unsigned)((64 * pass + lrow + 32 * q) * 256 + lcol * 8) * 2u,
                      (unsigned)c * ROWB);
    };

    // ---- every load of the tile is issued up front: t2 rows + W3 (phase 1), then W1' (phase 2) and the residual rows (plain) or
    //      the downsample operands (DUAL; W1' follows after phase 1, into the area they free) ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {                      // t2: 64 rows x 128 B
        const int row = lrow + 32 * q;
        const unsigned off = (m0 + row < p.M) ? (unsigned)(row * 64 + lcol * 8) * 2u : OOB;
        dma16(rsrc_a1, smem + CH_A1 + (q * 4 + wave) * 1024, off, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)                        // W3: 256 rows x 128 B
        dma16(rsrc_w3, smem + CH_W3 + (q * 4 + wave) * 1024, (unsigned)((lrow + 32 * q) * 64 + lcol * 8) * 2u, 0);
    const int ccol = (tid & 31) * 8;                   // epilogue 1: 32 threads per row (8 channels each), 8 rows per step
    const int crow = tid >> 5;
    u32x4 rres[DUAL ? 1 : 8];
    if (DUAL) {
        const __amdgpu_buffer_rsrc_t rsrc_ad = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.ad + (size_t)m0 * 64), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wd), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = lrow + 32 * q;
            const unsigned off = (m0 + row < p.M) ? (unsigned)(row * 64 + lcol * 8) * 2u : OOB;
            dma16(rsrc_ad, smem + CH_AD + (q * 4 + wave) * 1024, off, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            dma16(rsrc_wd, smem + CH_WD + (q * 4 + wave) * 1024, (unsigned)((lrow + 32 * q) * 64 + lcol * 8) * 2u, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_w1n(0);
        asm volatile("" ::: "memory");                 // (pins the issue order the counted wait below relies on: DMA pieces, then residual loads)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + it * 8 + crow;
            const int mc = m < p.M ? m : p.M - 1;
            rres[it] = *reinterpret_cast<const u32x4*>(p.res + (size_t)mc * CH_N1 + ccol);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // the 18 LDS-DMA pieces (issued first) have landed; the 8 residual loads may be in flight
    }
    __syncthreads();

    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;

    // ---- phase 1: 64 x 256 x 64 (DUAL: twice, two accumulator sets); wave (wm, wn) owns rows 32 wm .. +32, columns 128 wn .. +128 ----
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4], accd[4];                            // (accd is dead code in the plain form)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
            accd[j][r] = 0.f;
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int koff = ((2 * ks + half) ^ fswz) * 16;
        const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + CH_A1 + (wm * 32 + fr) * ROWB + koff);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 fb = *reinterpret_cast<const u32x4*>(smem + CH_W3 + (wn * 128 + j * 32 + fr) * ROWB + koff);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[j], 0, 0, 0);
        }
    }
    if (DUAL) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int koff = ((2 * ks + half) ^ fswz) * 16;
            const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + CH_AD + (wm * 32 + fr) * ROWB + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 fb = *reinterpret_cast<const u32x4*>(smem + CH_WD + (wn * 128 + j * 32 + fr) * ROWB + koff);
                accd[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), accd[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // every wave has read the phase-1 operands: their LDS is free
    if (DUAL) load_w1n(0);                             // lands during epilogue 1

    // ---- epilogue 1: four passes of 16 rows through the f32 staging tile; rows -> HBM (out) and -> LDS (phase-2 A operand) ----
    float* cs = reinterpret_cast<float*>(smem + CH_CS);
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(p.scale3 + ccol), sc1 = *reinterpret_cast<const f32x4*>(p.scale3 + ccol + 4);
    const f32x4 sf0 = *reinterpret_cast<const f32x4*>(p.shift3 + ccol), sf1 = *reinterpret_cast<const f32x4*>(p.shift3 + ccol + 4);
    f32x4 sd0 = sc0, sd1 = sc1, td0 = sf0, td1 = sf1;
    if (DUAL) {
        sd0 = *reinterpret_cast<const f32x4*>(p.scale_d + ccol); sd1 = *reinterpret_cast<const f32x4*>(p.scale_d + ccol + 4);
        td0 = *reinterpret_cast<const f32x4*>(p.shift_d + ccol); td1 = *reinterpret_cast<const f32x4*>(p.shift_d + ccol + 4);
    }
    auto to_stage = [&](const f32x16 (&a4)[4], int ps) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 8 * (ps & 1) + rr;                         // accumulator registers of rows 16 (ps & 1) .. +16 of the wave's 32
                const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;  // row inside the pass, 0..15
                const int col = (wn * 128 + j * 32 + fr) ^ (((lr >> 2) & 1) << 5);   // rows 4 apart: other half of the bank space
                cs[lr * CH_N1 + col] = a4[j][r];
            }
    };
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        f32x4 d0[2], d1[2];
        if constexpr (DUAL) {                          // downsample branch first: bn_d, then the bf16 rounding of its stored form
            if (wm == (ps >> 1)) to_stage(accd, ps);
            __syncthreads();
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int lr = crow + 8 * st;
                const float* src = cs + lr * CH_N1 + (ccol ^ (((lr >> 2) & 1) << 5));
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * sd0 + td0;
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4) * sd1 + td1;
                const unsigned q0 = pack_bf16(v0[0], v0[1]), q1 = pack_bf16(v0[2], v0[3]), q2 = pack_bf16(v1[0], v1[1]), q3 = pack_bf16(v1[2], v1[3]);
                d0[st] = f32x4{bf16_lo(q0), bf16_hi(q0), bf16_lo(q1), bf16_hi(q1)};
                d1[st] = f32x4{bf16_lo(q2), bf16_hi(q2), bf16_lo(q3), bf16_hi(q3)};
            }
            __syncthreads();
        }
        if (wm == (ps >> 1)) to_stage(acc, ps);
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int lr = crow + 8 * st;
            const int R = ps * 16 + lr;                                       // row inside the 64-row tile
            const float* src = cs + lr * CH_N1 + (ccol ^ (((lr >> 2) & 1) << 5));
            f32x4 v0, v1;
            if (DUAL) {
                v0 = *reinterpret_cast<const f32x4*>(src) * sc0 + sf0 + d0[st];
                v1 = *reinterpret_cast<const f32x4*>(src + 4) * sc1 + sf1 + d1[st];
            } else {
                v0 = *reinterpret_cast<const f32x4*>(src);
                v1 = *reinterpret_cast<const f32x4*>(src + 4);
                v0 = v0 * sc0 + sf0;
                v1 = v1 * sc1 + sf1;
                const u32x4 rr = rres[ps * 2 + st];
                v0[0] += bf16_lo(rr[0]); v0[1] += bf16_hi(rr[0]); v0[2] += bf16_lo(rr[1]); v0[3] += bf16_hi(rr[1]);
                v1[0] += bf16_lo(rr[2]); v1[1] += bf16_hi(rr[2]); v1[2] += bf16_lo(rr[3]); v1[3] += bf16_hi(rr[3]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
            u32x4 o;
            o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
            o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
            if (m0 + R < p.M) *reinterpret_cast<u32x4*>(p.y + (size_t)(m0 + R) * CH_N1 + ccol) = o;
            // phase-2 A operand: K chunk ccol / 64, 16-byte slot (ccol % 64) / 8, swizzled like the tile loader's rows
            *reinterpret_cast<u32x4*>(smem + CH_A2 + (ccol >> 6) * 8192 + R * ROWB + ((((ccol & 63) >> 3) ^ ((R >> 1) & 7)) << 4)) = o;
        }
        __syncthreads();
    }

    // ---- phase 2: 64 x 64 x 256 from LDS per pass; wave (wm, wn) owns rows 32 wm .. +32, columns 32 wn .. +32 of the pass ----
    const int ldy2 = CH_N2 * p.n2_passes;
    for (int pass = 0; pass < p.n2_passes; ++pass) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this pass's W1' rows have landed
        __syncthreads();
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int koff = ((2 * ks + half) ^ fswz) * 16;
                const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + CH_A2 + c * 8192 + (wm * 32 + fr) * ROWB + koff);
                const u32x4 fb = *reinterpret_cast<const u32x4*>(smem + CH_W1N + c * 8192 + (wn * 32 + fr) * ROWB + koff);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc2, 0, 0, 0);
            }
        __syncthreads();                               // every wave has read this pass's W1' (and the staging tile of the previous pass)
        if (pass + 1 < p.n2_passes) load_w1n(pass + 1);   // lands during this pass's epilogue
        // ---- epilogue 2: 64 x 64 through the staging tile ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            cs[row * CH_N2 + ((wn * 32 + fr) ^ (((row >> 2) & 1) << 5))] = acc2[r];
        }
        __syncthreads();
        {
            const int c2 = (tid & 7) * 8;
            const int cg = CH_N2 * pass + c2;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale1n + cg), s1 = *reinterpret_cast<const f32x4*>(p.scale1n + cg + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.shift1n + cg), t1 = *reinterpret_cast<const f32x4*>(p.shift1n + cg + 4);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int row = (tid >> 3) + 32 * st;
                const float* src = cs + row * CH_N2 + (c2 ^ (((row >> 2) & 1) << 5));
                f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * s0 + t0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4) * s1 + t1;
#pragma unroll
                for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
                u32x4 o;
                o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                if (m0 + row < p.M) *reinterpret_cast<u32x4*>(p.y2 + (size_t)(m0 + row) * ldy2 + cg) = o;
            }
        }
    }
}

// ---- 8-wave large-tile variant (inference forward; no stem / data-gradient / statistics modes) ---------------------
// 512 threads = 2 waves per SIMD from ONE workgroup, tiles 256x256 (wave tile 128x64), 256x128 and 128x256 (64x64):
// the 128x128 / 4-wave kernel above moves 0.5 KiB from L2 into LDS per MFMA -- 64 B/clk/CU at the full matrix rate, more
// than the ~56 B/clk/CU the L2 delivers -- and reads 1 KiB of LDS per MFMA; the 256x256 tile halves the first (0.25 KiB)
// and cuts the second to 0.75 KiB, with one workgroup per CU (128 KiB of LDS for the two stages) whose second wave per
// SIMD covers the other's barrier / fragment-read gaps.  Loader, swizzle, OOB handling and k order are the 4-wave
// kernel's (same results bit for bit: the accumulation order over k does not depend on the tile shape).
// STATS: train-mode BatchNorm statistics of the stored rows (per-channel sum / sum of squares -> one f64 atomic per channel and
// workgroup), as in the 4-wave kernel's epilogue; a separate instantiation so that the inference kernel's registers stay as they are.
// PERSIST: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ... of the (XCD-swizzled) tile list.  What it buys is
// measured with tools/conv_trace.py: a 256x256 workgroup owns its CU alone (128 KiB of LDS), so between two tiles the CU sat
// idle for the dispatch of the next workgroup (1.6-1.7 us) plus the first chunk's flight (1.2-1.9 us) -- 15-20 % of a 1x1
// expansion's tile.  Here the next tile's first chunk is requested right after the k loop, into stage 0, and lands while the
// epilogue runs out of a slab at the TOP of the 160 KiB (over stage 1 and the 32 KiB above it).  Same arithmetic, same bits.
// TR: data-gradient mode of the strided convs (one stride-parity class of dX pixels per launch, per-class tap lists and packing,
// see the 4-wave kernel); bf16 gradients only.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool OUT_F32, bool STATS = false, bool PERSIST = false, bool TR = false>
__global__ __launch_bounds__(512) void conv_igemm_bf16_w8_kernel(ConvArgsH p)
{
    static_assert(WAVES_M * WAVES_N == 8, "8 waves per workgroup");
    static_assert(!(PERSIST && STATS), "the persistent form is an inference kernel");
    static_assert(!(TR && (OUT_F32 || STATS)), "data-gradient mode: bf16 gradients, no statistics");
    constexpr int NW = 8;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int AP = BM / 64, BP = BN / 64;              // loader passes: 64 rows (512 threads x 16 bytes) each
    constexpr int A_BYTES = BM * ROWB;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int CS = BN + 4;
    constexpr int SLAB = 32 * CS;                           // floats
    static_assert(WAVES_M * SLAB * 4 <= 2 * STAGE_BYTES, "C slabs must fit in the staging LDS");
    constexpr int LDS_P = 160 * 1024;                       // persistent launches take the whole LDS
    constexpr int SLAB_OFF = PERSIST ? LDS_P - WAVES_M * SLAB * 4 : 0;     // bytes
    static_assert(!PERSIST || (SLAB_OFF >= STAGE_BYTES && SLAB_OFF % 16 == 0), "epilogue slabs must leave stage 0 alone");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3;                              // 0..63
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);

    const int NT = p.Cout / BN;
    const int hw_out = TR ? p.cHo * p.cWo : p.Ho * p.Wo;
    const size_t img_elems = (size_t)p.Hi * p.Wi * p.xstride;
    const int total = PERSIST ? ((p.M + BM - 1) / BM) * NT : (int)gridDim.x;

    // ---- per-tile state (re-made by setup() for every tile of a persistent workgroup) ----
    int m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    int a_pix0[AP], a_hi0[AP], a_wi0[AP];
    unsigned a_off[AP];
    int dh = 0, dw = 0, c0 = 0;
    unsigned w_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) w_off[q] = (unsigned)((lrow + 64 * q) * p.K + lcol * 8) * 2u;

    auto tap_offsets = [&](int th, int tw) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            int hi, wi;
            bool ok = a_pix0[q] >= 0;
            if (TR) {                                            // th / tw index the class's tap lists
                const int t_h = a_hi0[q] - p.tdh[th];
                int t_w = a_wi0[q] - p.tdw[tw];
                t_w = t_w < 0 ? t_w + p.Wo : t_w;                 // circular on the dX grid (width Wo)
                t_w = t_w >= p.Wo ? t_w - p.Wo : t_w;
                hi = t_h >> p.sh_log2;
                wi = t_w >> p.sw_log2;
                ok = ok && t_h >= 0 && hi < p.Hi;
            } else {
                hi = a_hi0[q] + th;
                wi = a_wi0[q] + tw;
                wi = wi < 0 ? wi + p.Wi : wi;
                wi = wi >= p.Wi ? wi - p.Wi : wi;
                ok = ok && ((unsigned)hi < (unsigned)p.Hi);
            }
            const unsigned pix = (unsigned)(a_pix0[q] + hi * p.Wi + wi);
            a_off[q] = ok ? (pix * (unsigned)p.xstride + (unsigned)lcol * 8u) * 2u : OOB;
        }
    };
    auto setup = [&](int vb) {       // tile vb of the launch order -> (m0, n0), buffer resources, row coordinates, first tap
        int bid = vb;
        if (p.xcd_swizzle) {
            const int q = total >> 3, r = total & 7;
            const int xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int nt = bid % NT;
        const int mt = bid / NT;
        m0 = mt * BM;
        n0 = nt * BN;
        const int b_first = m0 / hw_out;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.x + (size_t)b_first * img_elems), 0, 0x7fffffff, 0x00020000);
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.w + (size_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int m = m0 + lrow + 64 * q;
            if (m < p.M) {
                const int gw = TR ? p.cWo : p.Wo, gh = TR ? p.cHo : p.Ho;
                const int wo = m % gw;
                const int t = m / gw;
                const int ho = t % gh;
                const int b = t / gh;
                a_pix0[q] = (b - b_first) * p.Hi * p.Wi;
                a_hi0[q] = TR ? p.ca + p.sh * ho + p.ph : ho * p.sh - p.ph;
                a_wi0[q] = TR ? p.cb + p.sw * wo + p.pw : wo * p.sw - p.pw;
            } else {
                a_pix0[q] = -1;
                a_hi0[q] = 0;
                a_wi0[q] = 0;
            }
        }
        dh = 0; dw = 0; c0 = 0;
        tap_offsets(0, 0);
    };
    // output pixel of GEMM row m: dense, except in data-gradient mode where the rows enumerate one parity class
    auto out_pix = [&](int m) -> size_t {
        if (!TR) return (size_t)m;
        const int wo = m % p.cWo;
        const int t = m / p.cWo;
        const int ho = t % p.cHo;
        const int b = t / p.cHo;
        return ((size_t)b * p.Ho + (p.ca + p.sh * ho)) * p.Wo + (p.cb + p.sw * wo);
    };

    auto fetch = [&](int kc) {       // chunk kc -> LDS stage (kc & 1)
        char* a_s = smem + (kc & 1) * STAGE_BYTES;
        char* b_s = a_s + A_BYTES;
#pragma unroll
        for (int q = 0; q < AP; ++q) dma16(rsrc_a, a_s + (q * NW + wave) * 1024, a_off[q], (unsigned)c0 * 2u);
#pragma unroll
        for (int q = 0; q < BP; ++q) dma16(rsrc_w, b_s + (q * NW + wave) * 1024, w_off[q], (unsigned)kc * (unsigned)ROWB);
        if (TR) {                        // data gradient: taps outer, channels inner (its own per-class packing)
            c0 += BKE;
            if (c0 == p.Cin) {
                c0 = 0;
                if (++dw == p.ntdw) { dw = 0; ++dh; }
                if (dh < p.ntdh) tap_offsets(dh, dw);
            }
        } else {
            if (++dw == p.KW) {          // taps inner, 64-channel chunks outer (see the 4-wave kernel's fetch)
                dw = 0;
                if (++dh == p.KH) { dh = 0; c0 += BKE; }
            }
            if (p.KH * p.KW > 1) tap_offsets(dh, dw);
        }
    };

    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;

    // epilogue geometry: per 32-row MFMA band i, every wave row writes its band into its own LDS slab ([32][BN] floats);
    // the 512 threads then sweep the WAVES_M slabs as whole rows: scale/shift, residual, ReLU, 16-byte stores
    constexpr int TPR = BN / 8;                             // threads per output row (8 channels each)
    constexpr int RPP = 512 / TPR;                          // rows per sweep pass
    constexpr int NPASS = WAVES_M * 32 / RPP;
    static_assert(NPASS >= 1 && (WAVES_M * 32) % RPP == 0, "sweep geometry");
    float* cs = reinterpret_cast<float*>(smem + SLAB_OFF);
    const int ccol = (tid % TPR) * 8;
    const int crow = tid / TPR;
    const bool has_res = p.res != nullptr;
    f32x4 st1a = {0.f, 0.f, 0.f, 0.f}, st1b = st1a, st2a = st1a, st2b = st1a;     // STATS: sums of this thread's 8 columns

    int vb = blockIdx.x;
    [[maybe_unused]] int tr_wg = vb;
    HN_TR_STAMP(tr_wg, 0);
    HN_TR_HWID(tr_wg);
    setup(vb);
    fetch(0);

    while (true) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // chunk 0 of this tile is in stage 0 (and the previous tile's slab reads are over)
        HN_TR_STAMP(tr_wg, 1);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
            auto ldfrag = [&](u32x4 (&fa)[TM], u32x4 (&fb)[TN], int buf, int ks) {
                const int koff = ((2 * ks + half) ^ fswz) * 16;
                const char* a_s = smem + buf * STAGE_BYTES + (wm * WM + fr) * ROWB + koff;
                const char* b_s = smem + buf * STAGE_BYTES + A_BYTES + (wn * WN + fr) * ROWB + koff;
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(a_s + i * 32 * ROWB);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b_s + j * 32 * ROWB);
            };
            auto mma = [&](const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                            acc[i][j], 0, 0, 0);
            };
            ldfrag(fa0, fb0, 0, 0);
            for (int kc = 0; kc < p.nk; ++kc) {
                const int buf = kc & 1;
                const bool more = kc + 1 < p.nk;
                if (more) fetch(kc + 1);
                ldfrag(fa1, fb1, buf, 1);
                mma(fa0, fb0);
                ldfrag(fa0, fb0, buf, 2);
                mma(fa1, fb1);
                ldfrag(fa1, fb1, buf, 3);
                mma(fa0, fb0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of chunk kc+1 has landed
                __syncthreads();
                if (more) ldfrag(fa0, fb0, buf ^ 1, 0);
                mma(fa1, fb1);
            }
            __syncthreads();
        }
        HN_TR_STAMP(tr_wg, 2);

        // the tile being written; a persistent workgroup now points its loader at the next tile and requests its first chunk
        const int em0 = m0, en0 = n0;
        [[maybe_unused]] const int tr_done = tr_wg;
        bool has_next = false;
        if (PERSIST) {
            vb += (int)gridDim.x;
            has_next = vb < total;
            if (has_next) {
                tr_wg = vb;
                HN_TR_STAMP(tr_wg, 0);
                HN_TR_HWID(tr_wg);
                setup(vb);
                fetch(0);
            }
        }

        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(p.scale + en0 + ccol);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(p.scale + en0 + ccol + 4);
        const f32x4 sf0 = *reinterpret_cast<const f32x4*>(p.shift + en0 + ccol);
        const f32x4 sf1 = *reinterpret_cast<const f32x4*>(p.shift + en0 + ccol + 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            u32x4 rres[NPASS];
            if (has_res) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int sr = crow + ps * RPP;             // 0 .. WAVES_M*32-1: slab sr / 32, row sr % 32
                    const int m = em0 + (sr >> 5) * WM + i * 32 + (sr & 31);
                    const int mc = m < p.M ? m : p.M - 1;
                    rres[ps] = *reinterpret_cast<const u32x4*>(p.res + out_pix(mc) * p.Cout + en0 + ccol);
                }
            }
            if (i > 0) __syncthreads();
            {
                float* c_w = cs + wm * SLAB + (4 * half) * CS + wn * WN + fr;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) c_w[((r & 3) + 8 * (r >> 2)) * CS + j * 32] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int sr = crow + ps * RPP;
                const int m = em0 + (sr >> 5) * WM + i * 32 + (sr & 31);
                const float* src = cs + (sr >> 5) * SLAB + (sr & 31) * CS + ccol;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
                v0 = v0 * sc0 + sf0;
                v1 = v1 * sc1 + sf1;
                if (has_res) {
                    const u32x4 rr = rres[ps];
                    v0[0] += bf16_lo(rr[0]); v0[1] += bf16_hi(rr[0]); v0[2] += bf16_lo(rr[1]); v0[3] += bf16_hi(rr[1]);
                    v1[0] += bf16_lo(rr[2]); v1[1] += bf16_hi(rr[2]); v1[2] += bf16_lo(rr[3]); v1[3] += bf16_hi(rr[3]);
                }
                if (p.relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
                }
                if (m < p.M) {
                    if (STATS) {
                        st1a += v0; st1b += v1;
                        st2a += v0 * v0; st2b += v1 * v1;
                    }
                    if (OUT_F32) {
                        float* yo = reinterpret_cast<float*>(p.y) + out_pix(m) * p.ldy + en0 + ccol;
                        *reinterpret_cast<f32x4*>(yo) = v0;
                        *reinterpret_cast<f32x4*>(yo + 4) = v1;
                    } else {
                        u32x4 o;
                        o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                        o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.y) + out_pix(m) * p.ldy + en0 + ccol) = o;
                    }
                }
            }
            HN_TR_STAMP(tr_done, 3 + i);
        }
        if (!PERSIST || !has_next) break;
    }
    if (STATS) {             // per-channel sums of the stored tile: threads of one column group -> LDS -> one f64 atomic per channel
        static_assert(2 * RPP * BN * 4 <= 2 * STAGE_BYTES, "statistics scratch must fit in the staging LDS");
        __syncthreads();
        float* red = cs;                         // [2][RPP][BN]
        *reinterpret_cast<f32x4*>(red + crow * BN + ccol) = st1a;
        *reinterpret_cast<f32x4*>(red + crow * BN + ccol + 4) = st1b;
        *reinterpret_cast<f32x4*>(red + (RPP + crow) * BN + ccol) = st2a;
        *reinterpret_cast<f32x4*>(red + (RPP + crow) * BN + ccol + 4) = st2b;
        __syncthreads();
        float a1 = 0.f, a2 = 0.f;
        if (tid < BN) {
#pragma unroll
            for (int r = 0; r < RPP; ++r) {
                a1 += red[r * BN + tid];
                a2 += red[(RPP + r) * BN + tid];
            }
        }
        hn_stat_commit(p.stat_sum, p.stat_sq, p.stat_rep, p.Cout, n0, BN, tid, a1, a2);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool OUT_F32, bool STATS = false, bool PERSIST = false, bool TR = false>
int launch_cfg_w8(const ConvArgsH& a, hipStream_t s)
{
    const size_t lds = PERSIST ? (size_t)160 * 1024 : 2 * (size_t)(BM + BN) * ROWB;
    auto kern = conv_igemm_bf16_w8_kernel<BM, BN, WAVES_M, WAVES_N, OUT_F32, STATS, PERSIST, TR>;
    static bool attr_done[64] = {};   // per instantiation, per device
    static int n_cu[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HN_HIP(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_done[dev] = true;
    }
    const int MT = hn_cdiv(a.M, BM);
    const int NT = a.Cout / BN;
    const int tiles = MT * NT;
    int grid = tiles;
    if (PERSIST) {
        const int cus = dev < 64 && n_cu[dev] > 0 ? n_cu[dev] : 256;
        grid = tiles < cus ? tiles : cus;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

// Tile choice for the inference forward: a large tile only where it still fills the 256 CUs and K is deep enough to
// amortise its prologue / epilogue.  HN_BF16_W8 = 0 disables the 8-wave kernels, 1 = 256x256, 2 = 256x128, 3 = 128x256,
// 4 / 5 = the ping-pong persistent 256x256 kernel without / with s_setprio, 6 / 7 = the dw-reuse 3x3 kernel with 256x256 / 512x128 tiles
// force one (A/B runs, tests).
template <bool OUT_F32>
int dispatch_w8(const ConvArgsH& a, int Cout, hipStream_t s, bool* taken)
{
    const char* env = getenv("HN_BF16_W8");              // read per call: tools/conv_sweep.py flips it between launches
    const int force = env ? atoi(env) : -1;
    *taken = true;
    if (force == 0) { *taken = false; return 0; }
    const long M = a.M;
    if (a.stat_sum != nullptr) {
        // train-mode forward (batch statistics in the epilogue): the 256x256 tile under the same >= 224-workgroup rule, bf16 z only
        static const char* tenv = getenv("HN_BF16_W8_TRAIN");
        const long t = Cout % 256 == 0 ? (long)hn_cdiv(M, 256) * (Cout / 256) : 0;
        // 64 output channels, 3x3, stride 1 along W (layer1's conv2): the dw-reuse kernel's statistics variant (conv3x3_dwr64_bf16.hip)
        static const char* d64t = getenv("HN_BF16_DWR64");
        if (!OUT_F32 && !(d64t && atoi(d64t) == 0) && a.res == nullptr && a.relu == 0 && Cout == 64 && (long)hn_cdiv(M, 512) >= 224 && hn_conv_bf16_dwr_ok(a, 2))
            return hn_launch_conv_bf16_dwr(a, 0, 2, s);
        if (!OUT_F32 && !(tenv && atoi(tenv) == 0) && a.res == nullptr && t >= 224) return launch_cfg_w8<256, 256, 2, 4, false, true>(a, s);
        *taken = false;
        return 0;
    }
    const long t256 = Cout % 256 == 0 ? (long)hn_cdiv(M, 256) * (Cout / 256) : 0;
    // tiles a persistent 8-wave kernel needs before it is preferred to the 4-wave kernels.  224 = "fills the chip on its own" (the rule the
    // isolated sweep gives); HN_BF16_MIN_TILES lowers it for A/B runs of the PIPELINED forward, where a kernel that keeps 128 CUs busy at
    // a higher per-CU rate leaves the other 128 to the other stream
    const char* mte = getenv("HN_BF16_MIN_TILES");
    const long min_tiles = mte ? atol(mte) : 224;
    const long t128x256 = Cout % 256 == 0 ? (long)hn_cdiv(M, 128) * (Cout / 256) : 0;
    const long t256x128 = Cout % 128 == 0 ? (long)hn_cdiv(M, 256) * (Cout / 128) : 0;
    if ((force == 4 || force == 5) && t256 && a.nk >= 2) return hn_launch_conv_bf16_pp(a, OUT_F32 ? 1 : 0, force - 4, s);
    if ((force == 6 || force == 7) && hn_conv_bf16_dwr_ok(a, force - 6)) return hn_launch_conv_bf16_dwr(a, OUT_F32 ? 1 : 0, force - 6, s);
    if (force == 8 && !OUT_F32 && hn_conv_bf16_dwr_ok(a, 2)) return hn_launch_conv_bf16_dwr(a, 0, 2, s);
    if (force == 1 && t256) return launch_cfg_w8<256, 256, 2, 4, OUT_F32>(a, s);
    if (force == 2 && t256x128) return launch_cfg_w8<256, 128, 4, 2, OUT_F32>(a, s);
    if (force == 3 && t128x256) return launch_cfg_w8<128, 256, 2, 4, OUT_F32>(a, s);
    // measured on every conv shape of the B = 32 forward (tools/conv_sweep.py, profiles/r2_conv_tile_sweep.txt): the
    // 256x256 tile wins by 10-25 % wherever it yields >= 224 workgroups (1.0-1.16 PF on the deep-K convs vs 0.8-0.96),
    // also on the HBM-bound 1x1 convs; with 128 workgroups (layer4 3x3, ghc3.0) half the CUs idle and it loses; the
    // 256x128 / 128x256 shapes (64x64 wave tiles) never beat the 4-wave 128x128 kernel and are kept for A/B runs only
    // 3x3 convs with stride 1 along W: the activations of a filter row loaded once (conv3x3_dwr_bf16.hip); HN_BF16_DWR=0 for A/B runs
    const char* dwe = getenv("HN_BF16_DWR");
    const bool dwr = !(dwe && atoi(dwe) == 0);
    if (force < 0 && dwr && t256 >= min_tiles && hn_conv_bf16_dwr_ok(a, 0)) return hn_launch_conv_bf16_dwr(a, OUT_F32 ? 1 : 0, 0, s);
    if (force < 0 && dwr && t256 < min_tiles && (long)hn_cdiv(M, 512) * (Cout / 128) >= min_tiles && hn_conv_bf16_dwr_ok(a, 1))
        return hn_launch_conv_bf16_dwr(a, OUT_F32 ? 1 : 0, 1, s);
    // 64 output channels (layer1's conv2, the 64-channel height-compression convs): 512 x 64 tiles, same rule
    const char* d64 = getenv("HN_BF16_DWR64");        // 0 for A/B runs
    if (force < 0 && dwr && !(d64 && atoi(d64) == 0) && !OUT_F32 && Cout == 64 && (long)hn_cdiv(M, 512) >= 224 && hn_conv_bf16_dwr_ok(a, 2))
        return hn_launch_conv_bf16_dwr(a, 0, 2, s);
    if (force < 0 && t256 >= min_tiles) {
        // ping-pong persistent kernel (conv_igemm_bf16_pp.hip); HN_BF16_PP=0 for A/B runs, 2 = with s_setprio
        const char* ppe = getenv("HN_BF16_PP");
        const int pp = ppe ? atoi(ppe) : 1;
        if (pp > 0 && a.nk >= 2) return hn_launch_conv_bf16_pp(a, OUT_F32 ? 1 : 0, pp - 1, s);
        // more than one tile per CU: the persistent form (next tile's first chunk in flight under the epilogue, no workgroup
        // dispatch between tiles); HN_W8_PERSIST=0 for A/B runs
        const char* pe = getenv("HN_W8_PERSIST");
        if (t256 > 256 && !(pe && atoi(pe) == 0)) return launch_cfg_w8<256, 256, 2, 4, OUT_F32, false, true>(a, s);
        return launch_cfg_w8<256, 256, 2, 4, OUT_F32>(a, s);
    }
    *taken = false;
    return 0;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STEM, bool OUT_F32, bool TR = false>
int launch_cfg_h(const ConvArgsH& a, hipStream_t s)
{
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    auto kern = conv_igemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STEM, OUT_F32, TR>;
    static bool attr_done[64] = {};   // per instantiation, per device
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[dev] = true;
    }
    const int MT = hn_cdiv(a.M, BM);
    const int NT = a.Cout / BN;
    // a one-chunk GEMM (1x1 conv with 64 input channels: layer1's conv3 / downsample, HBM-bound) never touches the second
    // LDS stage: launch with one stage (+ the epilogue's C tile) so that 4 workgroups instead of 2 share a CU and their
    // load / compute / store phases overlap
    constexpr size_t c_tile = (size_t)(BM / WAVES_M) * (BN + 4) * 4;
    constexpr size_t one_stage = (size_t)(BM + BN) * ROWB;
    const size_t lds_launch = (a.nk <= 1 && !a.stat_sum) ? (c_tile > one_stage ? c_tile : one_stage) : lds;
    hipLaunchKernelGGL(kern, dim3((unsigned)(MT * NT), (unsigned)(a.ksplit > 1 ? a.ksplit : 1)), dim3(256), lds_launch, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

template <bool OUT_F32, bool TR = false>
int dispatch(const ConvArgsH& a, int Cout, hipStream_t s)
{
    const long M = a.M;
    if constexpr (TR && !OUT_F32) {
        // data gradient of a strided conv with bf16 gradients: the 256x256 8-wave kernel under the forward dispatcher's rule
        // (the height-compression convs' data gradients are 0.3-0.6 TFLOP each); HN_DGRAD_W8=0 for A/B runs
        const char* de = getenv("HN_DGRAD_W8");                 // read per call: the stage test flips it between launches
        const long t256 = Cout % 256 == 0 ? (long)hn_cdiv(M, 256) * (Cout / 256) : 0;
        if (!(de && atoi(de) == 0) && a.nk > 0 && t256 >= 224) {
            if (t256 > 256) return launch_cfg_w8<256, 256, 2, 4, false, false, true, true>(a, s);
            return launch_cfg_w8<256, 256, 2, 4, false, false, false, true>(a, s);
        }
    }
    if (Cout % 128 == 0) {
        const long blocks128 = (long)hn_cdiv(M, 128) * (Cout / 128);
        if (blocks128 >= 512) return launch_cfg_h<128, 128, 2, 2, false, OUT_F32, TR>(a, s);
        return launch_cfg_h<64, 128, 2, 2, false, OUT_F32, TR>(a, s);
    }
    if (Cout % 64 == 0) {
        const long blocks128 = (long)hn_cdiv(M, 128) * (Cout / 64);
        if (blocks128 >= 512) return launch_cfg_h<128, 64, 2, 2, false, OUT_F32, TR>(a, s);
        return launch_cfg_h<64, 64, 2, 2, false, OUT_F32, TR>(a, s);
    }
    return launch_cfg_h<128, 32, 4, 1, false, OUT_F32, TR>(a, s);
}

// ---- bf16 helper kernels -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, u16* __restrict__ out, long n8)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + i * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 8 + 4);
        u32x4 o;
        o[0] = pack_bf16(a[0], a[1]); o[1] = pack_bf16(a[2], a[3]); o[2] = pack_bf16(b[0], b[1]); o[3] = pack_bf16(b[2], b[3]);
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

// (x[:, :3] - mean) / std, NCHW f32 -> NHWC4 bf16 (channel 3 = 0; 8 bytes per pixel)   (reference model.py:248-252)
__global__ __launch_bounds__(256) void prep_nhwc4_bf16_kernel(const float* __restrict__ x, u16* __restrict__ out, long npix, long total,
                                                              int C_in)
{
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    // 2 pixels (16 bytes out) per thread
    for (long i2 = (long)blockIdx.x * blockDim.x + threadIdx.x; i2 * 2 < total; i2 += (long)gridDim.x * blockDim.x) {
        const long i = i2 * 2;
        const long b = i / npix;
        const long pix = i - b * npix;                       // npix is even: the pair never straddles two images
        const float* src = x + b * C_in * npix + pix;
        const float2 c0 = *reinterpret_cast<const float2*>(src);
        const float2 c1 = *reinterpret_cast<const float2*>(src + npix);
        const float2 c2 = *reinterpret_cast<const float2*>(src + 2 * npix);
        u32x4 o = {pack_bf16((c0.x - mean[0]) / stdv[0], (c1.x - mean[1]) / stdv[1]), pack_bf16((c2.x - mean[2]) / stdv[2], 0.f),
                   pack_bf16((c0.y - mean[0]) / stdv[0], (c1.y - mean[1]) / stdv[1]), pack_bf16((c2.y - mean[2]) / stdv[2], 0.f)};
        *reinterpret_cast<u32x4*>(out + i * 4) = o;
    }
}

// 3x3 / 2 / pad 1 max-pool, NHWC bf16, 8 channels per thread (ordinary -inf padding)
__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const u16* __restrict__ in, u16* __restrict__ out, int Hi, int Wi, int Ho, int Wo,
                                                           int C8, long total)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long b = t / Ho;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * 2 - 1 + dh;
            if ((unsigned)hi >= (unsigned)Hi) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wo * 2 - 1 + dw;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(in + (((b * Hi + hi) * Wi + wi) * (long)C8 + c8) * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    m[2 * k] = fmaxf(m[2 * k], bf16_lo(v[k]));
                    m[2 * k + 1] = fmaxf(m[2 * k + 1], bf16_hi(v[k]));
                }
            }
        }
        u32x4 o = {pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3]), pack_bf16(m[4], m[5]), pack_bf16(m[6], m[7])};
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

// circular linear up-sample along W + (c, h) flatten, bf16 in -> bf16 sequence matrix (see elementwise.hip)
__global__ __launch_bounds__(256) void upsample_flatten_bf16_kernel(const u16* __restrict__ in, u16* __restrict__ seq, int B, int hq, int Wq,
                                                                    int cq, int col0, int f)
{
    const int t = blockIdx.x;
    const int b = blockIdx.y;
    const float src = (1.0f / (float)f) * ((float)(t + f) + 0.5f) - 0.5f;
    const int i0p = (int)src;
    const float w1 = src - (float)i0p;
    const float w0 = 1.0f - w1;
    int i0 = i0p - 1;
    i0 = i0 < 0 ? i0 + Wq : i0;
    int i1 = i0p;
    i1 = i1 >= Wq ? i1 - Wq : i1;
    u16* dst = seq + ((long)t * B + b) * 1024 + col0;
    const int n = cq * hq;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int c = e % cq;
        const int h = e / cq;
        const long base = ((long)b * hq + h) * Wq;
        const float a = __builtin_bit_cast(float, (unsigned)in[(base + i0) * cq + c] << 16);
        const float bb = __builtin_bit_cast(float, (unsigned)in[(base + i1) * cq + c] << 16);
        dst[c * hq + h] = (u16)(pack_bf16(w0 * a + w1 * bb, 0.f) & 0xffffu);
    }
}

// OIHW f32 -> packed bf16 [Cout][Cp / 64][kh][KWp][64] (Cp a multiple of 64)
__global__ __launch_bounds__(256) void pack_conv_bf16_kernel(const float* __restrict__ w, u16* __restrict__ out, int Cout, int Cin, int KH,
                                                             int KW, int KWp, int Cp)
{
    const long total = (long)Cout * KH * KWp * Cp;
    const int ntap = KH * KWp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // k = (64-channel chunk, tap, channel in chunk): the forward kernels' chunk order (taps inner)
        const int e = (int)(i % BKE);
        long t = i / BKE;
        const int tap = (int)(t % ntap);
        t /= ntap;
        const int cc = (int)(t % (Cp / BKE));
        const int o = (int)(t / (Cp / BKE));
        const int dh = tap / KWp, dw = tap - dh * KWp;
        const int c = cc * BKE + e;
        float v = 0.f;
        if (c < Cin && dw < KW) v = w[(((long)o * Cin + c) * KH + dh) * KW + dw];
        out[i] = (u16)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// 7x7 stem: OIHW f32 [64][3][7][7] -> packed bf16 [64][4 chunks][2 filter rows][8 taps][4 channels] (tap t <-> dw = t - 1)
__global__ __launch_bounds__(256) void pack_stem_bf16_kernel(const float* __restrict__ w, u16* __restrict__ out, int Cout)
{
    const long total = (long)Cout * 256;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 3);
        const int t = (int)((i >> 2) & 7);
        const int dh = (int)((i >> 5) & 7);                  // chunk * 2 + row-in-chunk
        const int o = (int)(i >> 8);
        float v = 0.f;
        if (ch < 3 && t >= 1 && dh < 7) v = w[(((long)o * 3 + ch) * 7 + dh) * 7 + (t - 1)];
        out[i] = (u16)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// OIHW f32 -> [Cin][Cout / 64][KH][KW][64] bf16 with flipped taps: the weights of a stride-1 conv's data gradient written as a
// forward conv (channel roles swapped) in the forward kernels' K order (see hn_launch_conv_dgrad_bf16)
__global__ __launch_bounds__(256) void pack_dgrad_fwd_bf16_kernel(const float* __restrict__ w, u16* __restrict__ out, int Cout, int Cin,
                                                                  int KH, int KW)
{
    const long total = (long)Cin * KH * KW * Cout;
    const int ntap = KH * KW, nch = Cout / BKE;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % BKE);
        long t = i / BKE;
        const int tap = (int)(t % ntap);
        t /= ntap;
        const int cc = (int)(t % nch);
        const int c = (int)(t / nch);
        const int dh = tap / KW, dw = tap - dh * KW;
        const float v = w[(((long)(cc * BKE + e) * Cin + c) * KH + (KH - 1 - dh)) * KW + (KW - 1 - dw)];
        out[i] = (u16)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// OIHW f32 -> per-class data-gradient packing [Cin][ndh][ndw][Cout] in bf16 (only the taps that reach the class)
__global__ __launch_bounds__(256) void pack_dgrad_class_bf16_kernel(const float* __restrict__ w, u16* __restrict__ out, int Cout, int Cin,
                                                                    int KH, int KW, int dh0, int dh1, int dh2, int ndh, int dw0, int dw1,
                                                                    int dw2, int ndw)
{
    const long total = (long)Cin * ndh * ndw * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % Cout);
        long t = i / Cout;
        const int iw = (int)(t % ndw);
        t /= ndw;
        const int ih = (int)(t % ndh);
        const int c = (int)(t / ndh);
        const int dh = ih == 0 ? dh0 : (ih == 1 ? dh1 : dh2), dw = iw == 0 ? dw0 : (iw == 1 ? dw1 : dw2);
        out[i] = (u16)(pack_bf16(w[(((long)o * Cin + c) * KH + dh) * KW + dw], 0.f) & 0xffffu);
    }
}

// split-K tail: out[m][n] = act(scale[n] * sum_s partial[s][m][n] + shift[n]) -> bf16, slices summed in index order
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ part, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, u16* __restrict__ out, long MN8, int N8,
                                                                 long slice_elems, int S, int relu)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < MN8; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N8) * 8;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int sl = 0; sl < S; ++sl) {
            const float* q = part + sl * slice_elems + i * 8;
            a0 += *reinterpret_cast<const f32x4*>(q);
            a1 += *reinterpret_cast<const f32x4*>(q + 4);
        }
        a0 = a0 * *reinterpret_cast<const f32x4*>(scale + n) + *reinterpret_cast<const f32x4*>(shift + n);
        a1 = a1 * *reinterpret_cast<const f32x4*>(scale + n + 4) + *reinterpret_cast<const f32x4*>(shift + n + 4);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] = fmaxf(a0[k], 0.f); a1[k] = fmaxf(a1[k], 0.f); }
        }
        u32x4 o = {pack_bf16(a0[0], a0[1]), pack_bf16(a0[2], a0[3]), pack_bf16(a1[0], a1[1]), pack_bf16(a1[2], a1[3])};
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

// the 4-wave kernel a conv with the fused BatchNorm reduce runs on (its epilogue holds the sums): 128-row tiles, 128 or 64 columns
inline int dispatch_bn(const ConvArgsH& a, int Cout, hipStream_t s)
{
    if (Cout % 128 == 0) return launch_cfg_h<128, 128, 2, 2, false, false, false>(a, s);
    return launch_cfg_h<128, 64, 2, 2, false, false, false>(a, s);
}

// out[i] += sum over this block's share of the tiles of slab[t][i] (double; out zeroed by the caller): blockIdx.y = tile range, 16 elements x 16
// tile lanes per workgroup, eight loads in flight per lane
__global__ __launch_bounds__(256) void slab_colsum_kernel(const float* __restrict__ slab, int tiles, int n, double* __restrict__ out)
{
    __shared__ double red[256];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int per = (tiles + gridDim.y - 1) / gridDim.y;
    const int t0 = blockIdx.y * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    for (int b = blockIdx.x; b * 16 < n; b += gridDim.x) {
        const int i = b * 16 + il;
        double t = 0.0;
        if (i < n) {
            int q = t0 + sl;
            for (; q + 7 * 16 < t1; q += 8 * 16) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(q + 16 * u) * n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += (double)v[u];
            }
            for (; q < t1; q += 16) t += (double)slab[(size_t)q * n + i];
        }
        red[threadIdx.x] = t;
        __syncthreads();
        if (sl == 0 && i < n) {
            for (int u = 1; u < 16; ++u) t += red[u * 16 + il];
            atomicAdd(out + i, t);
        }
        __syncthreads();
    }
}

inline unsigned grid_for(long total, long cap = 256L * 16)
{
    long g = (total + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

#ifdef HN_CONV_TRACE
static unsigned long long* g_conv_trace = nullptr;
extern "C" int hn_debug_conv_trace(void* buf) { g_conv_trace = static_cast<unsigned long long*>(buf); return 0; }
#endif

int hn_conv_bf16_bn_tile_rows(int Cout, long M) { (void)Cout; (void)M; return 128; }

int hn_launch_slab_colsum(const float* slab, int tiles, int n, double* out, hipStream_t s)
{
    int g = (n + 15) / 16;
    if (g > 4096) g = 4096;
    int gy = tiles / 128;                  // >= 128 tiles (8 per lane) per block
    gy = gy < 1 ? 1 : (gy > 128 ? 128 : gy);
    hipLaunchKernelGGL(slab_colsum_kernel, dim3((unsigned)g, (unsigned)gy), dim3(256), 0, s, slab, tiles, n, out);
    HN_LAUNCH_CHECK();
    return 0;
}

// d.x / d.w / d.res / d.y are reinterpreted as bf16 buffers (d.y as f32 when out_f32)
int hn_launch_conv_bf16(const ConvDesc& d, int out_f32, hipStream_t s)
{
    static const char* env = getenv("HN_XCD_SWIZZLE");
    ConvArgsH a;
    a.xcd_swizzle = env ? atoi(env) : 1;
    a.x = reinterpret_cast<const u16*>(d.x); a.w = reinterpret_cast<const u16*>(d.w); a.scale = d.scale; a.shift = d.shift;
    a.res = reinterpret_cast<const u16*>(d.res); a.y = d.y;
    a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout;
    a.KH = d.KH; a.KW = d.KW; a.sh = d.sh; a.sw = d.sw; a.ph = d.ph; a.pw = d.pw;
    a.xstride = d.xstride ? d.xstride : d.Cin;
    a.M = d.B * d.Ho * d.Wo;
    a.relu = d.relu;
    a.ldy = d.ldy ? d.ldy : d.Cout;
    a.stat_sum = d.stat_sum; a.stat_sq = d.stat_sq; a.stat_rep = d.stat_rep > 1 ? d.stat_rep : 1;
    a.ksplit = 1;
    a.sh_log2 = a.sw_log2 = a.ca = a.cb = a.cHo = a.cWo = a.ntdh = a.ntdw = 0;
    for (int i = 0; i < 3; ++i) a.tdh[i] = a.tdw[i] = 0;
    a.mask_out = d.mask_out;
    a.bn_z = reinterpret_cast<const u16*>(d.bn_z); a.bn_mask = d.bn_mask; a.bn_mean = d.bn_mean; a.bn_invstd = d.bn_invstd; a.bn_slab = d.bn_slab;
    HN_REQUIRE(!d.bn_z || (!out_f32 && d.bn_mask && d.bn_mean && d.bn_invstd && d.bn_slab && !d.stat_sum && !d.stem && !d.mask_out && d.KH == 1 && d.KW == 1 &&
                           d.sh == 1 && d.sw == 1 && a.ldy == d.Cout && d.Cout % 64 == 0),
               "conv bf16: the fused BatchNorm reduce goes with a dense 1x1 bf16 conv");
    HN_REQUIRE(!d.mask_out || (!out_f32 && d.relu && !d.stem && !d.stat_sum && d.Cout % 8 == 0), "conv bf16: mask_out goes with a bf16 ReLU output");
#ifdef HN_CONV_TRACE
    a.trace = g_conv_trace;
#endif
    HN_REQUIRE(!d.res || !out_f32, "conv bf16: a residual with float32 output is only used by the data-gradient launcher");
    {
        const double span = 256.0 / ((double)d.Ho * d.Wo) + 2.0;
        HN_REQUIRE(span * d.Hi * d.Wi * (double)(d.stem ? 4 : a.xstride) * 2.0 < 2147483648.0, "conv bf16: image too large for 32-bit tile offsets");
    }
    if (d.stem) {
        HN_REQUIRE(d.KH == 7 && d.KW == 7 && d.Cout == 64, "stem conv bf16: expects 7x7, Cout=64");
        HN_REQUIRE(d.Ho * d.Wo >= 128 && d.Wi % 2 == 0, "stem conv bf16: image >= one tile, even width");
        a.K = 4 * BKE;
        a.nk = 4;
        return out_f32 ? launch_cfg_h<128, 64, 2, 2, true, true>(a, s) : launch_cfg_h<128, 64, 2, 2, true, false>(a, s);
    }
    HN_REQUIRE(d.Cin % BKE == 0, "conv bf16: Cin=%d must be a multiple of %d", d.Cin, BKE);
    HN_REQUIRE(d.Cout % 32 == 0, "conv bf16: Cout=%d must be a multiple of 32", d.Cout);
    a.K = d.KH * d.KW * d.Cin;
    a.nk = a.K / BKE;
    HN_REQUIRE(256.0 * a.K * 2.0 < 2147483648.0, "conv bf16: K too large");
    if (d.bn_z) return dispatch_bn(a, d.Cout, s);
    // Split-K on the dw-reuse kernel for the 3x3 convs that give a 256-CU part only 64 .. 160 tiles of 256 x 256 at the nominal batch
    // (ghc3.0 / ghc3.1 / ghc2.1, layer4.*.conv2): 2 or 4 K slices of whole (channel chunk, filter row) steps as float32 partial
    // tiles + the ordered reduce below.  The slice count is a function of the layer's shape at the nominal batch of 32 ONLY (never of
    // the actual M), so every output bit is the same for any batch size.  HN_BF16_SPLITK=0 / HN_BF16_DWR=0 switch it off.
    if (!out_f32 && !d.res && !d.stat_sum && d.splitk_ws && d.B > 0 && !d.stem) {
        static const char* sk_env = getenv("HN_BF16_SPLITK");
        const char* dwe = getenv("HN_BF16_DWR");
        const long t32 = (long)hn_cdiv(32L * d.Ho * d.Wo, 256) * (d.Cout / 256);
        int S = t32 >= 64 && t32 <= 160 ? (int)(256 / t32) : 1;
        S = S >= 4 ? 4 : (S >= 2 ? 2 : 1);
        ConvArgsH b = a;
        b.ksplit = S;
        if (S > 1 && !(sk_env && (atoi(sk_env) == 0 || atoi(sk_env) == 2)) && !(dwe && atoi(dwe) == 0) && d.Cout % 256 == 0 && hn_conv_bf16_dwr_ok(b, 0) &&      // HN_BF16_SPLITK=2: only the 4-wave tail split below
           
            (size_t)S * a.M * d.Cout <= d.splitk_ws_floats) {
            b.y = d.splitk_ws;
            b.relu = 0;
            b.ldy = d.Cout;
            if (int rc = hn_launch_conv_bf16_dwr(b, 1, 0, s)) return rc;
            const long MN8 = (long)a.M * d.Cout / 8;
            hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid_for(MN8)), dim3(256), 0, s, d.splitk_ws, d.scale, d.shift,
                               reinterpret_cast<u16*>(d.y), MN8, d.Cout / 8, (long)a.M * d.Cout, S, d.relu);
            HN_LAUNCH_CHECK();
            return 0;
        }
    }
    // Split-K for deep-K convs with few output tiles (tails of the height-compression chains: ghc3.2 has 64 tiles of 128x128
    // with K = 9216, ghc3.3 16 tiles).  The slice count depends on the layer's shape at the nominal batch of 32 ONLY, never on
    // the actual M, so the summation order -- and with it every output bit -- is the same for any batch size.
    if (!out_f32 && !d.res && !d.stat_sum && d.splitk_ws && d.Cout % 128 == 0 && d.B > 0) {
        static const char* sk_env = getenv("HN_BF16_SPLITK");
        const long per_image = (long)d.Ho * d.Wo;
        // (unlike the float32 path -- hn_launch_conv's interactive regime -- the bf16 slice count never looks at the actual batch: a different
        //  float32 summation order flips bf16 roundings, and test_forward_bf16_batch32_consistency / test_dw_reuse_split_k pin "same bits at any
        //  batch size"; measured gain of a B = 1 regime: 2.51 -> 2.26 ms per forward, not taken)
        const long t32 = (long)hn_cdiv(32 * per_image, 128) * (d.Cout / 128);
        int S = t32 >= 192 ? 1 : (int)(256 / t32);
        if (S > 8) S = 8;
        while (S > 1 && a.nk / S < 8) --S;
        if (sk_env && atoi(sk_env) == 0) S = 1;
        if (S > 1 && (size_t)S * a.M * d.Cout <= d.splitk_ws_floats) {
            ConvArgsH b = a;
            b.ksplit = S;
            b.y = d.splitk_ws;
            b.relu = 0;
            b.ldy = d.Cout;
            if (int rc = launch_cfg_h<128, 128, 2, 2, false, true, false>(b, s)) return rc;
            const long MN8 = (long)a.M * d.Cout / 8;
            hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid_for(MN8)), dim3(256), 0, s, d.splitk_ws, d.scale, d.shift,
                               reinterpret_cast<u16*>(d.y), MN8, d.Cout / 8, (long)a.M * d.Cout, S, d.relu);
            HN_LAUNCH_CHECK();
            return 0;
        }
    }
    if (d.mask_out) {        // (the mask store lives in the 4-wave and the ping-pong kernels' epilogues)
        const char* ppe = getenv("HN_FOLD_PP");          // 0: 4-wave kernel everywhere (A/B runs)
        const long t256 = d.Cout % 256 == 0 ? (long)hn_cdiv(a.M, 256) * (d.Cout / 256) : 0;
        if (!(ppe && ppe[0] == '0') && d.KH == 1 && d.KW == 1 && t256 >= 224 && a.nk >= 2) return hn_launch_conv_bf16_pp(a, 0, 0, s);
    } else {
        bool taken = false;
        const int rc = out_f32 ? dispatch_w8<true>(a, d.Cout, s, &taken) : dispatch_w8<false>(a, d.Cout, s, &taken);
        if (taken) return rc;
    }
    return out_f32 ? dispatch<true>(a, d.Cout, s) : dispatch<false>(a, d.Cout, s);
}

// Data gradient of a forward conv on the bf16 matrix cores (see hn_launch_conv_dgrad in conv_igemm_f32.hip):
// dz_h: bf16 [B][Ho][Wo][Cout]; w_oihw: the float32 master weights (re-packed per class into w_scratch as bf16), or NULL when
// w_scratch already holds the class packings (the training step: packed once per optimiser step by hn_pack_weights_bf16);
// add (optional) and dx are float32, or -- grad_bf16 -- both bf16 (train.hip keeps the gradients between conv units in bf16).
// Needs Cout %% 64 == 0 (one K chunk = 64 bf16 channels of one tap).
int hn_launch_conv_bf16(const ConvDesc& d, int out_f32, hipStream_t s);

int hn_launch_conv_dgrad_bf16(const ConvDesc& d, const void* dz_h, const float* w_oihw, const float* add, float* dx, void* w_scratch,
                              const float* ones, const float* zeros, hipStream_t s, int grad_bf16)
{
    static const char* env = getenv("HN_XCD_SWIZZLE");
    HN_REQUIRE((d.sh == 1 || d.sh == 2) && (d.sw == 1 || d.sw == 2) && !d.stem, "conv dgrad bf16: strides must be 1 or 2");
    HN_REQUIRE(d.Cout % BKE == 0 && d.Cin % 32 == 0, "conv dgrad bf16: Cout %% 64 / Cin %% 32");
    HN_REQUIRE(d.Hi % d.sh == 0 && d.Wi % d.sw == 0, "conv dgrad bf16: input size must be a multiple of the stride");
    u16* wp = reinterpret_cast<u16*>(w_scratch);
    if (d.sh == 1 && d.sw == 1 && grad_bf16 && d.KH == d.KW && d.ph == d.KH / 2 && d.pw == d.KW / 2) {
        // Stride 1, bf16 gradients: dX = conv(dY, W') with W'[ci][dh][dw][co] = W[co][ci][KH-1-dh][KW-1-dw] and the same padding --
        // a FORWARD conv with the channel roles swapped, so it gets the forward dispatcher: the 256x256 8-wave kernel
        // (persistent where a CU has several tiles) and the taps-inner K order instead of the 128x128 data-gradient kernel
        // (0.57 PFLOP/s on the training step's data gradients against 0.9-1.2 for the forward kernels).  The identity-branch
        // gradient is the conv's bf16 residual; scale 1 / shift 0 leave the accumulator bits alone (fma(v, 1, 0) == v).
        if (w_oihw != nullptr) {
            const long total = (long)d.Cin * d.KH * d.KW * d.Cout;
            hipLaunchKernelGGL(pack_dgrad_fwd_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, s, w_oihw, wp, d.Cout, d.Cin, d.KH, d.KW);
            HN_LAUNCH_CHECK();
        }
        ConvDesc f;
        memset(&f, 0, sizeof(f));
        f.x = reinterpret_cast<const float*>(dz_h); f.w = reinterpret_cast<const float*>(wp); f.scale = ones; f.shift = zeros;
        f.res = add; f.y = dx;
        f.B = d.B; f.Hi = d.Ho; f.Wi = d.Wo; f.Cin = d.Cout; f.Ho = d.Hi; f.Wo = d.Wi; f.Cout = d.Cin;
        f.KH = d.KH; f.KW = d.KW; f.sh = 1; f.sw = 1; f.ph = d.ph; f.pw = d.pw;
        f.relu = 0; f.ldy = d.Cin;
        return hn_launch_conv_bf16(f, 0, s);
    }
    for (int ca = 0; ca < d.sh; ++ca) {
        for (int cb = 0; cb < d.sw; ++cb) {
            ConvArgsH a;
            memset(&a, 0, sizeof(a));
            a.xcd_swizzle = env ? atoi(env) : 1;
            a.x = reinterpret_cast<const u16*>(dz_h); a.scale = ones; a.shift = zeros;
            a.res = reinterpret_cast<const u16*>(add);        // float32 in OUT_F32 mode (the kernel reinterprets)
            a.y = dx;
            a.Hi = d.Ho; a.Wi = d.Wo; a.Cin = d.Cout; a.Ho = d.Hi; a.Wo = d.Wi; a.Cout = d.Cin;
            a.KW = d.KW; a.sh = d.sh; a.sw = d.sw; a.ph = d.ph; a.pw = d.pw;
            a.sh_log2 = d.sh == 2 ? 1 : 0; a.sw_log2 = d.sw == 2 ? 1 : 0;
            a.xstride = d.Cout; a.relu = 0; a.ldy = d.Cin;
            a.ca = ca; a.cb = cb; a.cHo = d.Hi / d.sh; a.cWo = d.Wi / d.sw;
            for (int t = 0; t < d.KH; ++t) if ((ca + d.ph - t) % d.sh == 0) a.tdh[a.ntdh++] = t;
            for (int t = 0; t < d.KW; ++t) if ((cb + d.pw - t) % d.sw == 0) a.tdw[a.ntdw++] = t;
            a.M = d.B * a.cHo * a.cWo;
            a.K = a.ntdh * a.ntdw * d.Cout;
            a.nk = a.K / BKE;
            if (a.K > 0 && w_oihw != nullptr) {       // (null: w_scratch already holds the class packings, hn_pack_weights_bf16)
                const long total = (long)d.Cin * a.K;
                hipLaunchKernelGGL(pack_dgrad_class_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, s, w_oihw, wp, d.Cout, d.Cin, d.KH,
                                   d.KW, a.tdh[0], a.tdh[1], a.tdh[2], a.ntdh, a.tdw[0], a.tdw[1], a.tdw[2], a.ntdw);
                HN_LAUNCH_CHECK();
            }
            a.w = wp;
            {
                const double span = 256.0 / ((double)a.cHo * a.cWo) + 2.0;
                HN_REQUIRE(span * a.Hi * a.Wi * (double)a.xstride * 2.0 < 2147483648.0, "conv dgrad bf16: image too large for 32-bit tile offsets");
                HN_REQUIRE(128.0 * a.K * 2.0 < 2147483648.0, "conv dgrad bf16: K too large");
            }
            if (int rc = grad_bf16 ? dispatch<false, true>(a, d.Cin, s) : dispatch<true, true>(a, d.Cin, s)) return rc;
            wp += (size_t)d.Cin * a.K;
        }
    }
    return 0;
}

int hn_launch_pack_conv_bf16(const float* w, void* out, int Cout, int Cin, int KH, int KW, hipStream_t s)
{
    if (KH == 7) {          // the stem: its own K layout (see the header)
        HN_REQUIRE(KW == 7 && Cin == 3, "pack bf16: the 7x7 form is the 3-channel stem only");
        hipLaunchKernelGGL(pack_stem_bf16_kernel, dim3(grid_for((long)Cout * 256)), dim3(256), 0, s, w, reinterpret_cast<u16*>(out), Cout);
        HN_LAUNCH_CHECK();
        return 0;
    }
    const long total = (long)Cout * KH * KW * Cin;
    hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, reinterpret_cast<u16*>(out), Cout, Cin, KH, KW, KW, Cin);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_f32_to_bf16(const float* in, void* out, long n, hipStream_t s)
{
    HN_REQUIRE(n % 8 == 0, "f32_to_bf16: n %% 8");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, in, reinterpret_cast<u16*>(out), n / 8);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_prep_nhwc4_bf16(const float* x, void* out, int B, int C_in, int H, int W, hipStream_t s)
{
    const long npix = (long)H * W, total = npix * B;
    HN_REQUIRE(npix % 2 == 0 && W % 2 == 0, "prep bf16: even image width required");
    hipLaunchKernelGGL(prep_nhwc4_bf16_kernel, dim3(grid_for(total / 2)), dim3(256), 0, s, x, reinterpret_cast<u16*>(out), npix, total, C_in);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_maxpool_bf16(const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s)
{
    HN_REQUIRE(C % 8 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "maxpool bf16: C%%8, even H/W required");
    const int Ho = Hi / 2, Wo = Wi / 2, C8 = C / 8;
    const long total = (long)B * Ho * Wo * C8;
    hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<const u16*>(in),
                       reinterpret_cast<u16*>(out), Hi, Wi, Ho, Wo, C8, total);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_upsample_flatten_bf16(const void* in, void* seq, int B, int hq, int Wq, int cq, int col0, hipStream_t s)
{
    HN_REQUIRE(Wq > 0 && 256 % Wq == 0 && col0 + cq * hq <= 1024, "upsample bf16: bad geometry");
    hipLaunchKernelGGL(upsample_flatten_bf16_kernel, dim3(256, B), dim3(256), 0, s, reinterpret_cast<const u16*>(in),
                       reinterpret_cast<u16*>(seq), B, hq, Wq, cq, col0, 256 / Wq);
    HN_LAUNCH_CHECK();
    return 0;
}

// y = relu(bn3(conv3(t2)) + bn_d(downsample(x))) in one launch (conv1x1_dual_bf16_kernel).  t2: bf16 [B][Ho][Wo][K1];
// x: bf16 [B][Hi2][Wi2][K2], sampled at stride s2; w1 / w2 packed bf16 [Cout][K]; y bf16 [B][Ho][Wo][Cout].
int hn_launch_conv1x1_dual_bf16(const void* t2, const void* w1, const float* scale1, const float* shift1, const void* x, const void* w2,
                                const float* scale2, const float* shift2, void* y, int B, int Ho, int Wo, int K1, int Hi2, int Wi2,
                                int K2, int s2, int Cout, hipStream_t s)
{
    static const char* env = getenv("HN_XCD_SWIZZLE");
    HN_REQUIRE(K1 % BKE == 0 && K2 % BKE == 0 && Cout % 128 == 0, "dual 1x1 conv: K1=%d K2=%d Cout=%d", K1, K2, Cout);
    HN_REQUIRE((Ho - 1) * s2 < Hi2 && (Wo - 1) * s2 < Wi2, "dual 1x1 conv: stride %d does not map the %dx%d grid into %dx%d", s2, Ho, Wo, Hi2, Wi2);
    DualArgsH a;
    a.a1 = reinterpret_cast<const u16*>(t2); a.w1 = reinterpret_cast<const u16*>(w1); a.scale1 = scale1; a.shift1 = shift1;
    a.a2 = reinterpret_cast<const u16*>(x); a.w2 = reinterpret_cast<const u16*>(w2); a.scale2 = scale2; a.shift2 = shift2;
    a.y = reinterpret_cast<u16*>(y);
    a.M = B * Ho * Wo; a.Cout = Cout; a.K1 = K1; a.K2 = K2; a.nk1 = K1 / BKE; a.nk2 = K2 / BKE;
    a.Ho = Ho; a.Wo = Wo; a.Hi2 = Hi2; a.Wi2 = Wi2; a.s2 = s2;
    a.xcd_swizzle = env ? atoi(env) : 1;
    {
        const double span = 128.0 / ((double)Ho * Wo) + 2.0;
        HN_REQUIRE(span * Hi2 * Wi2 * (double)K2 * 2.0 < 2147483648.0 && 128.0 * (K1 > K2 ? K1 : K2) * 2.0 < 2147483648.0,
                   "dual 1x1 conv: tile offsets exceed 32 bits");
    }
    const size_t lds = 2 * (size_t)(128 + 128) * ROWB;
    static bool attr_done[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_dual_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(conv1x1_dual_bf16_kernel, dim3((unsigned)(hn_cdiv(a.M, 128) * (Cout / 128))), dim3(256), lds, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

// out = relu(bn3(conv3(t2)) + x) and t1n = relu(bn1'(conv1'(out))) in one launch (layer1: 64 -> 256 -> 64 channels).
// xd != NULL: block 0 -- the residual is the downsample branch bf16(bn_d(wd . xd)) computed in the same launch (xd [M][64], stride 1).
int hn_launch_conv1x1_chain_bf16(const void* t2, const void* w3, const float* scale3, const float* shift3, const void* x, void* out,
                                 const void* w1n, const float* scale1n, const float* shift1n, void* t1n, long M, int K1, int N1, int N2,
                                 hipStream_t s, const void* xd, const void* wd, const float* scale_d, const float* shift_d)
{
    static const char* env = getenv("HN_XCD_SWIZZLE");
    HN_REQUIRE(K1 == 64 && N1 == CH_N1 && (N2 == CH_N2 || N2 == 2 * CH_N2), "chained 1x1 convs: built for 64 -> 256 -> 64 | 128 channels (got %d -> %d -> %d)", K1, N1, N2);
    HN_REQUIRE(M > 0 && M < 0x7fffffffL / 256, "chained 1x1 convs: bad M");
    ChainArgsH a;
    a.a1 = reinterpret_cast<const u16*>(t2); a.w3 = reinterpret_cast<const u16*>(w3); a.scale3 = scale3; a.shift3 = shift3;
    a.res = reinterpret_cast<const u16*>(x); a.y = reinterpret_cast<u16*>(out);
    a.ad = reinterpret_cast<const u16*>(xd); a.wd = reinterpret_cast<const u16*>(wd); a.scale_d = scale_d; a.shift_d = shift_d;
    a.w1n = reinterpret_cast<const u16*>(w1n); a.scale1n = scale1n; a.shift1n = shift1n; a.y2 = reinterpret_cast<u16*>(t1n);
    a.M = (int)M;
    a.xcd_swizzle = env ? atoi(env) : 1;
    a.n2_passes = N2 / CH_N2;
    static bool attr_done[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_chain_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_chain_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        attr_done[dev] = true;
    }
    if (xd) hipLaunchKernelGGL(conv1x1_chain_bf16_kernel<true>, dim3((unsigned)hn_cdiv(M, CH_BM)), dim3(256), CH_LDS, s, a);
    else hipLaunchKernelGGL(conv1x1_chain_bf16_kernel<false>, dim3((unsigned)hn_cdiv(M, CH_BM)), dim3(256), CH_LDS, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}
