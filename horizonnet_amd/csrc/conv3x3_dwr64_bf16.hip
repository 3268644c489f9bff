// 3x3 convolution with stride 1 along W and 64 OUTPUT CHANNELS, bf16 on the matrix cores: the dw-reuse kernel of
// conv3x3_dwr_bf16.hip on 512 x 64 tiles (layer1's three conv2 at 128 x 256 x 64 -> 64 and the 64-channel height-compression convs).
//
// Why a third shape: with 64 output channels a 128 x 64 wave tile does not exist, and the 4-wave 128 x 64 kernel these layers ran on
// moves one LDS-DMA byte per 43 FLOP (0.55 PFLOP/s measured).  Here: 8 waves = 2 groups x (2 x 2), wave tile 128 x 32 (four 32 x 32
// MFMA row tiles x ONE column tile), a chunk = TWO phases (the K halves) of 8 MFMAs each -- the same MFMA block length as the other
// ping-pong kernels -- and per step (three taps of a filter row) 64 KiB of activations + 3 x 8 KiB of weights: 143 FLOP per byte.
//
// LDS: activations 2 buffers x (A0 | A1 = the rows of group 0 | 1) x 32 KiB at 0; weights a RING OF FOUR 8 KiB regions at 128 KiB (two
// phases per chunk leave a double buffer 1.5 phases of lead, the L1 miss path needs ~1 us; the ring gives five).  The epilogue slabs
// (2 KiB per wave) live in region A1 of the activation buffer the tile has JUST finished with: the next step's activations sit in
// the other buffer and the one after is issued into A1 only two barriers after the other group has left its epilogue (model-checked).
// That keeps the kernel persistent (K is 9 chunks for layer1: a per-tile prologue would cost a third of the time).
//
// Schedule (G = chunk, t = step, dw = tap; X / Y = the two phases of a chunk):
//   X: reads A(h0) B(h0); issue B(G+3) -> ring slot (G+3)&3; dw 0: issue A0(t+1) -> other buffer;   barrier, 8 MFMAs, barrier
//   Y: reads A(h1) B(h1); dw 0: issue A1(t+1); counted wait (dw 0, 1: vmcnt(10): B(G+1) landed; dw 2: vmcnt(2): and A(t+1));
//      lgkmcnt(0) IN FRONT of the barrier (the ring slot / the activation buffer is re-filled right behind it), 8 MFMAs, barrier
//   the A cursor advances in X of (t, dw 1), its row offsets are recomputed in the shadow of that chunk's MFMA blocks.
// tools/pp_schedule_model.py (dwr64_program) checks these counts under adversarial landing before the kernel runs on a GPU.
//
// Same k order and rounding points as every other bf16 conv kernel: bit-identical to the 4-wave kernel (tests/test_gpu_bf16.py).
#include "hn_common.h"
#include "conv_bf16_args.h"
#include "conv_bf16_pp.h"
#include "stat_wave.h"

#include <type_traits>

// measurement builds only (tools/pp_ablate.sh d64_N): 1 no LDS-DMA, 2 no fragment reads after the first chunk, 4 no barriers, 8 no MFMAs,
// 16 no counted waits, 32 no epilogue (the compiler then drops the MFMAs and fragment reads too), 64 epilogue without its stores,
// 128 non-temporal stores -- the results are wrong, only the timing means something
#ifndef HN_D64_ABL
#define HN_D64_ABL 0
#endif

namespace {

constexpr int BM = 512, BN = 64;
constexpr int GROUP_ROWS = 256;
constexpr int REGION_A = GROUP_ROWS * ROWB;     // 32 KiB
constexpr int ABUF = 2 * REGION_A;
constexpr int APW = 4;                          // 1 KiB pieces per wave and A region
constexpr int AK = 8;                           // activation rows per thread (both regions)
constexpr int B_OFF = 2 * ABUF;
constexpr int BSLOT = BN * ROWB;                // 8 KiB: one piece per wave
constexpr int LDS_BYTES = B_OFF + 4 * BSLOT;
static_assert(LDS_BYTES == 163840, "LDS budget");

// STATS: the training forward -- bf16 z + the batch statistics (sum, sum of squares per channel) of the float32 accumulators it stores,
// accumulated per lane across ALL tiles of the persistent workgroup and flushed once (stat_wave.h); 174 -> 190 registers, no spills.
template <bool STATS>
__global__ __launch_bounds__(512) void conv3x3_dwr64_bf16_kernel(ConvArgsH p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2;                                // wave group = row half of the tile
    const int wm2 = (wave & 3) >> 1;                        // 128-row slice inside the group's rows
    const int wn = wave & 1;                                // 32-column slice
    const int lrow = tid >> 3;                              // loader: row 0..63 of a 64-row pass
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);
    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int NT = p.Cout / BN;
    const int total = ((p.M + BM - 1) / BM) * NT;
    const int hw_out = p.Ho * p.Wo;
    const size_t img_elems = (size_t)p.Hi * p.Wi * p.xstride;
    const int nwg = (int)gridDim.x;
    const int nsteps = p.nk / 3;

    const int wo_sh = __builtin_ctz(p.Wo);
    const int ho_sh = (p.Ho & (p.Ho - 1)) == 0 ? __builtin_ctz(p.Ho) : -1;
    const int nt_sh = (NT & (NT - 1)) == 0 ? __builtin_ctz(NT) : -1;
    auto tile_coords = [&](int bid, int& m0, int& n0) {
        if (p.xcd_swizzle) {
            const int q = total >> 3, r = total & 7;
            const int xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int mt = nt_sh >= 0 ? bid >> nt_sh : bid / NT;
        const int nt = bid - mt * NT;
        m0 = __builtin_amdgcn_readfirstlane(mt * BM);
        n0 = __builtin_amdgcn_readfirstlane(nt * BN);
    };

    // ---- loader: A cursor per step (c, dh), B cursor per chunk; both run ahead of the MFMAs, into the next tile ----
    constexpr int ROW_DEAD = -(1 << 24);
    const unsigned xs2 = (unsigned)p.xstride * 2u;
    const unsigned rowbytes = (unsigned)p.Wi * xs2;
    int avb = (int)blockIdx.x, astep = 0;
    int dh = 0, c0 = 0;
    u32x4 rsrc_a, rsrc_w;
    unsigned a_base[AK];
    int a_hi0[AK];
    unsigned a_off[AK];
    unsigned w_off = (unsigned)(lrow * p.K + lcol * 8) * 2u;    // weight row = output channel lrow of the tile
    int bvb = (int)blockIdx.x, bkc = 0;
    auto tap_rows = [&](int k0, int n) {
#pragma unroll
        for (int k = k0; k < k0 + n; ++k) {
            const int h = a_hi0[k] + dh;
            const unsigned off = __umul24((unsigned)h, rowbytes) + a_base[k];
            a_off[k] = (unsigned)h < (unsigned)p.Hi ? off : OOB;
        }
    };
    auto setup_a = [&](int vb) {
        int m0, n0;
        tile_coords(vb, m0, n0);
        const int b_first = __builtin_amdgcn_readfirstlane(ho_sh >= 0 ? m0 >> (wo_sh + ho_sh) : m0 / hw_out);
        rsrc_a = pp_rsrc(p.x + (size_t)b_first * img_elems);
#pragma unroll
        for (int k = 0; k < AK; ++k) {                      // k = APW * region + pass
            const int m = m0 + lrow + 64 * k;
            const int wo = m & (p.Wo - 1);
            const int t = m >> wo_sh;
            int ho, b;
            if (ho_sh >= 0) { ho = t & (p.Ho - 1); b = t >> ho_sh; } else { ho = t % p.Ho; b = t / p.Ho; }
            a_base[k] = (unsigned)((b - b_first) * p.Hi * p.Wi + wo) * xs2 + (unsigned)lcol * 16u;
            a_hi0[k] = m < p.M ? ho * p.sh - p.ph : ROW_DEAD;
        }
        dh = 0;
        c0 = 0;
    };
    auto advance_a = [&]() {                                // scalar part; tap_rows() follows in the shadow of MFMA blocks
        if (++astep == nsteps) {
            astep = 0;
            avb += nwg;
            if (avb < total) {
                setup_a(avb);
            } else {
#pragma unroll
                for (int k = 0; k < AK; ++k) a_hi0[k] = ROW_DEAD;
            }
        } else {
            const int dh1 = dh + 1;
            const bool wh = dh1 == 3;
            dh = wh ? 0 : dh1;
            c0 += wh ? BKE : 0;
        }
    };
    auto setup_b = [&](int vb) {
        int m0, n0;
        tile_coords(vb, m0, n0);
        rsrc_w = pp_rsrc(p.w + (size_t)n0 * p.K);
        bkc = 0;
    };
    auto advance_b = [&]() {
        if (++bkc == p.nk) {
            bvb += nwg;
            if (bvb < total) setup_b(bvb);
            else { bkc = 0; w_off = OOB; }
        }
    };
    auto issue_a = [&](int abuf, int reg) {
        if (HN_D64_ABL & 1) return;
        const unsigned dst = lds0 + (unsigned)(abuf * ABUF + reg * REGION_A + wave * 1024);
#pragma unroll
        for (int ps = 0; ps < APW; ++ps) pp_dma16<0>(rsrc_a, dst + ps * 8192, reg ? a_off[APW + ps] : a_off[ps], (unsigned)c0 * 2u);
    };
    auto issue_b = [&](int slot) {
        if (HN_D64_ABL & 1) return;
        pp_dma16<0>(rsrc_w, lds0 + (unsigned)(B_OFF + slot * BSLOT + wave * 1024), w_off, (unsigned)bkc * (unsigned)ROWB);
    };

    // ---- MFMA side ----
    int cvb = (int)blockIdx.x, cm0, cn0, cstep = 0;
    tile_coords(cvb, cm0, cn0);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 fa[4][2], fb[2];
    unsigned rd_a[3][4], rd_b[4];                           // A: inside the current activation buffer (toggled per step); B: inside ring slot 0
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int key = ((fr + dw - 1) >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rd_a[dw][s] = (unsigned)(g * REGION_A + (128 * wm2 + fr + dw - 1) * ROWB + (((2 * s + half) ^ key) * 16));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rd_b[s] = (unsigned)(B_OFF + (32 * wn + fr) * ROWB + (((2 * s + half) ^ fswz) * 16));
    const int seam_bytes = p.Wo * ROWB;
    bool abl_first = true;
    auto read_a = [&](auto dw_c, int h) {                   // see conv3x3_dwr_bf16.hip: the lane at an image-row seam reads Wo rows away
        constexpr int DW = decltype(dw_c)::value;
        if ((HN_D64_ABL & 2) && !abl_first) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t0 = 128 * wm2 + 32 * i;
            const bool seam = DW == 1 ? false : (DW == 0 ? (t0 & (p.Wo - 1)) == 0 : ((t0 + 32) & (p.Wo - 1)) == 0);
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const char* q = smem + rd_a[DW][2 * h + sh] + i * 32 * ROWB;
                if (DW != 1 && seam) {
                    if (fr == (DW == 0 ? 0 : 31)) fa[i][sh] = *reinterpret_cast<const u32x4*>(q + (DW == 0 ? seam_bytes : -seam_bytes));
                    else fa[i][sh] = *reinterpret_cast<const u32x4*>(q);
                } else {
                    fa[i][sh] = *reinterpret_cast<const u32x4*>(q);
                }
            }
        }
    };
    int bslot = 0;
    auto read_b = [&](int h) {
        if ((HN_D64_ABL & 2) && !abl_first) return;
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) fb[sh] = *reinterpret_cast<const u32x4*>(smem + rd_b[2 * h + sh] + bslot * BSLOT);
    };
    auto mm = [&](auto tap0_c, auto tapn_c) {
        constexpr int TAP0 = decltype(tap0_c)::value, TAPN = decltype(tapn_c)::value;
        if (!(HN_D64_ABL & 8))
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][sh]), __builtin_bit_cast(bf16x8, fb[sh]), acc[i], 0, 0, 0);
        if constexpr (TAPN > 0) {
            tap_rows(TAP0, TAPN);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;

    // ---- epilogue of the wave's 128 x 32 tile: 8 rounds of 16 rows through the wave's 2 KiB slab; bf16 output, folded BN, optional ReLU ----
    // scale / shift of the wave's columns stay in registers while the column tile stays the same (always, for Cout = 64): a global load
    // at the head of every epilogue is ~1 us of exposed latency per 5 us tile, and its vmcnt(0) would also wait for every piece in flight
    int sc_n0 = cn0;
    float sc = p.scale[cn0 + 32 * wn + fr];
    float sf = p.shift[cn0 + 32 * wn + fr];
    [[maybe_unused]] HnWaveStats wstat;
    if constexpr (STATS) wstat.zero();
    auto epilogue = [&](int em0, int en0, int slab_buf) {
        float* slab = reinterpret_cast<float*>(smem + slab_buf * ABUF + REGION_A + wave * 2048);
        const int row16 = lane >> 2;
        const int cb = lane & 3;
        const int f = (row16 >> 1) & 1;
        const int colg = en0 + 32 * wn + 8 * cb;
        const int mrow0 = em0 + GROUP_ROWS * g + 128 * wm2 + row16;
        if (en0 != sc_n0) {
            if constexpr (STATS) wstat.template flush<4>(p.stat_sum, p.stat_sq, p.stat_rep, p.Cout, sc_n0 + 32 * wn, lane);
            sc_n0 = en0;
            sc = p.scale[en0 + 32 * wn + fr];
            sf = p.shift[en0 + 32 * wn + fr];
        }
#pragma unroll
        for (int rd = 0; rd < 8; ++rd) {
            const int i = rd >> 1, hb = rd & 1;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int lr16 = (rr & 3) + 8 * (rr >> 2) + 4 * half;
                const int blk = (fr >> 2) ^ ((lr16 >> 1) & 1);
                slab[lr16 * 32 + blk * 4 + (fr & 3)] = acc[i][8 * hb + rr] * sc + sf;
            }
            f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + row16 * 32 + ((2 * cb) ^ f) * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + row16 * 32 + ((2 * cb + 1) ^ f) * 4);
            if (p.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
            }
            const int m = mrow0 + 16 * rd;
            if (m < p.M) {
                if constexpr (STATS) wstat.add(v0, v1);
                u32x4 o;
                o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.y) + (size_t)m * p.ldy + colg);
                if (HN_D64_ABL & 64) { if (o[0] == 0x12345678u && o[3] == 0x9abcdef0u) *dst = o; }      // (practically) no stores
                else if (HN_D64_ABL & 128) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
        }
    };

    // ---- prologue: step 0's activations, the weights of chunks 0..2 ----
    setup_a(avb);
    setup_b(bvb);
    tap_rows(0, AK);
    issue_a(0, 0); issue_a(0, 1);
    issue_b(0); advance_b();
    issue_b(1); advance_b();
    issue_b(2); advance_b();
    advance_a();
    tap_rows(0, AK);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // step 0's activations and chunk 0's weights have landed (this wave's pieces)
    pp_bar_raw();
    if (g == 1 && !(HN_D64_ABL & 4)) pp_bar_raw();         // group 1 runs one barrier behind group 0 from here on

    auto bar = [&]() { if (!(HN_D64_ABL & 4)) pp_bar_raw(); };
    int abuf = 0;
    int adelta = ABUF;
    auto chunk = [&](auto dw_c) {
        constexpr int DW = decltype(dw_c)::value;
        // ---- phase X: K half 0 ----
        read_a(dw_c, 0);
        read_b(0);
        issue_b((bslot + 3) & 3);                          // weights of chunk G+3
        if (DW == 0) issue_a(abuf ^ 1, 0);                 // A0 of the next step
        if (DW == 1) advance_a();
        bar();
        if constexpr (DW == 1) mm(I0{}, I4{}); else mm(I0{}, I0{});
        bar();
        // ---- phase Y: K half 1; the last reads of this chunk's weights (and, dw 2, of this step's activations) ----
        read_a(dw_c, 1);
        read_b(1);
        if (DW == 0) issue_a(abuf ^ 1, 1);                 // A1 of the next step
        advance_b();
        if (HN_D64_ABL & 16) {}
        else if (DW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // the next step's activations and the next chunk's weights
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");            // the next chunk's weights
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        if constexpr (DW == 1) mm(I4{}, I4{}); else mm(I0{}, I0{});
        bar();
        bslot = (bslot + 1) & 3;
        abl_first = false;
    };
    while (true) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        const int done_buf = abuf;
        abuf ^= 1;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int s = 0; s < 4; ++s) rd_a[dw][s] += (unsigned)adelta;
        adelta = -adelta;
        if (++cstep == nsteps) {                           // tile finished
            cstep = 0;
            if (g == 0) bar();                      // wait for group 1's last MFMA block (its reads of this buffer retired in front of it)
            if (!(HN_D64_ABL & 32)) epilogue(cm0, cn0, done_buf);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            cvb += nwg;
            if (cvb >= total) break;
            tile_coords(cvb, cm0, cn0);
            if (g == 1) bar();                      // re-stagger
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // surplus (out-of-range) pieces of the loader
    if constexpr (STATS) wstat.template flush<4>(p.stat_sum, p.stat_sq, p.stat_rep, p.Cout, sc_n0 + 32 * wn, lane);
}

}  // namespace

int hn_launch_conv_bf16_dwr64(const ConvArgsH& a, hipStream_t s)
{
    const bool stats = a.stat_sum != nullptr;
    HN_REQUIRE(!stats || (a.stat_sq != nullptr && a.stat_rep >= 1 && a.relu == 0), "conv bf16 (dw reuse, 64 columns): statistics come without ReLU");
    auto kern = stats ? conv3x3_dwr64_bf16_kernel<true> : conv3x3_dwr64_bf16_kernel<false>;
    static bool attr_done[2][64] = {};
    static int n_cu[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[stats][dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HN_HIP(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_done[stats][dev] = true;
    }
    const int tiles = hn_cdiv(a.M, BM) * (a.Cout / BN);
    int cus = dev < 64 && n_cu[dev] > 0 ? n_cu[dev] : 256;
    // A persistent workgroup owns a CU (160 KiB of LDS) and a FIXED share of the tiles.  In the pipelined forward layer1 runs while the
    // previous batch's recurrence holds 32 CUs: with 256 workgroups 32 of them start only when the first ones have finished, i.e. the
    // kernel takes twice as long (measured 170-181 us in the timeline against 99 alone).  Leaving those CUs out costs 1 / 8 when they
    // are free and saves 1 / 3 when they are not.  HN_D64_GRID overrides (A/B runs).
    static const char* ge = getenv("HN_D64_GRID");
    cus = ge ? atoi(ge) : cus - 32;
    if (cus < 1) cus = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(512), LDS_BYTES, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}
