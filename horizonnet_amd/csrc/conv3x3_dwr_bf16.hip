// 3x3 convolution with stride 1 along W, bf16 on the matrix cores: the ping-pong kernel of conv_igemm_bf16_pp.hip with the
// activations of the three taps of a filter ROW loaded ONCE ("dw reuse").
//
// Why: every bf16 conv kernel of this library runs at (FLOP per LDS-DMA byte) x ~10 TB/s -- the rate at which a compute unit's
// L1 miss path feeds LDS-DMA (64 KiB in flight per CU / ~1 us from issue to landed; tools/pp_stamps.py) -- long before the matrix
// pipe or the LDS read path saturate.  As an implicit GEMM a 3x3 conv fetches every activation nine times.  With the filter taps
// as the inner loop of a 64-channel chunk (K order (chunk, dh, dw), the packing of conv_igemm_bf16.hip) the A tile of tap
// (dh, dw) is the A tile of (dh, 1) ROTATED by dw - 1 pixels inside each image row (circular W padding, model.py:27-55: the
// wrap is part of the rotation, not a halo).  So a STEP = the three chunks (c, dh, 0..2): one activation load (the dw = 1
// pixels), three weight loads; the MFMA A fragments of dw = 0 / 2 are read from the same LDS rows one row up / down (the lane at
// an image-row seam reads Wo rows away instead).  LDS-DMA bytes per step: 32 + 3 x 32 KiB instead of 3 x 64 (256 x 256 tile).
//
// And a second tile shape: 512 x 128 (8 waves as 2 groups x (2 x 2), wave tile 128 x 64 as ever) for the convs with 128 output
// channels (ghc0.*, layer2.*.conv2: 1.4 ms of the B = 32 forward on the 128 x 128 4-wave kernel at 64 FLOP per DMA byte).
//
// Everything else -- two wave groups one barrier apart, hidden LDS-DMA counted by hand, loader cursors that run into the next
// tile, the per-wave epilogue -- is conv_igemm_bf16_pp.hip's; same k order, same rounding points, same bits as every other
// bf16 conv kernel (tested against the 4-wave kernel).
//
// Requirements (checked by the launcher): KH = KW = 3, stride / padding 1 along W, Wi = Wo a multiple of 32 that divides the
// rows of a wave group (128 or 256), Cin % 64 == 0.  Any H stride.
//
// Schedule of a step (chunks dw = 0, 1, 2; each four phases as in the ping-pong kernel: (j0,h0) (j1,h0) (j0,h1) (j1,h1)):
//   weights   B0 / B1 of chunk G+1 in phases 1 / 2 of chunk G (other B buffer);
//   activations A1 of step t+1 in phase 4 of (t, dw 0) (other A buffer), A0 of step t+2 in phase 4 of (t, dw 2) (this A buffer:
//             its last read was phase 3 of that chunk, which therefore waits for its reads in front of its barrier);
//   counted waits in phases 1 and 4 of every chunk (constants below); the A cursor advances in phase 2 of (t, dw 1), its row
//   offsets are recomputed in the shadow of that chunk's MFMA blocks.
// Verified against a wave-level model (random interleavings, earliest / latest landing of every piece) before it ran on a GPU.
#include "hn_common.h"
#include "conv_bf16_args.h"
#include "conv_bf16_pp.h"

#include <stdlib.h>
#include <type_traits>

namespace {

template <int BM, int BN>
struct DwrGeom {
    static constexpr int GROUP_ROWS = BM / 2;                 // rows of a wave group
    static constexpr int WGM = GROUP_ROWS / 128;              // waves of a group along M
    static constexpr int WGN = 4 / WGM;                       // ... along N
    static_assert(WGM * WGN == 4 && WGN * 64 == BN, "tile shape: 256 x 256 or 512 x 128");
    static constexpr int REGION_A = GROUP_ROWS * ROWB;        // A0 / A1: the rows of group 0 / 1
    static constexpr int REGION_B = WGN * 32 * ROWB;          // B0 / B1: left / right 32 columns of every wave column
    static constexpr int APW = GROUP_ROWS / 64;               // 1 KiB pieces per wave and A region
    static constexpr int BPW = WGN * 32 / 64;                 // ... and B region
    static constexpr int AK = 2 * APW;                        // activation rows per thread
    static constexpr int BK2 = 2 * BPW;                       // weight rows per thread
    static constexpr int ABUF = 2 * REGION_A;
    static constexpr int BBUF = 2 * REGION_B;
    static constexpr bool PERSIST = BM == 256;
    // 256 x 256: B buffers at 0, A buffers at 64 KiB, slabs at 128 KiB.  512 x 128: A buffers at 0, B buffers at 128 KiB, the
    // epilogue slabs re-use the A area (one tile per workgroup: nothing is in flight any more).
    static constexpr int A_OFF = PERSIST ? 2 * BBUF : 0;
    static constexpr int B_OFF = PERSIST ? 0 : 2 * ABUF;
    static constexpr int SLAB_OFF = PERSIST ? 2 * BBUF + 2 * ABUF : 0;
    static constexpr int LDS = 163840;
    static_assert(A_OFF % (2 * ABUF) == 0 && B_OFF % (2 * BBUF) == 0, "buffer pairs toggle by one address bit");
    static_assert(2 * ABUF + 2 * BBUF + (PERSIST ? 32768 : 0) <= LDS, "LDS budget");
    // counted waits: all but the N newest pieces of this wave have landed
    static constexpr int wait_p1(int dw) { return BPW + (dw == 2 ? 0 : APW); }       // B1 of this chunk complete
    static constexpr int wait_p4(int dw) { return BPW + (dw == 1 ? 0 : APW); }       // B0 of the next chunk (and, dw 2, the next step's A) complete
};

#define DWR_WAIT(n) do { \
        if ((n) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if ((n) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); \
        else if ((n) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if ((n) == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)

template <int BM, int BN, bool OUT_F32>
__global__ __launch_bounds__(512) void conv3x3_dwr_bf16_kernel(ConvArgsH p)
{
    using G = DwrGeom<BM, BN>;
    static_assert((G::wait_p1(0) == 4 || G::wait_p1(0) == 5) && (G::wait_p1(2) == 2 || G::wait_p1(2) == 1), "DWR_WAIT covers these counts");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2;                                // wave group = row half of the tile
    const int wm2 = (wave & 3) / G::WGN;                    // 128-row slice inside the group's rows
    const int wn = (wave & 3) % G::WGN;                     // 64-column slice
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
    // split-K (float32 partial tiles, OUT_F32 instantiation only): a work item = (tile, K slice); a slice = nsteps_s whole steps
    const int S = OUT_F32 && p.ksplit > 1 ? p.ksplit : 1;   // a power of two (launcher)
    const int s_sh = __builtin_ctz(S);
    const int nsteps_s = (p.nk / 3) >> s_sh;
    const int nk_s = 3 * nsteps_s;
    const int items = total << s_sh;

    const int wo_sh = __builtin_ctz(p.Wo);                  // Wo is a power of two here (launcher)
    const int ho_sh = (p.Ho & (p.Ho - 1)) == 0 ? __builtin_ctz(p.Ho) : -1;
    const int nt_sh = (NT & (NT - 1)) == 0 ? __builtin_ctz(NT) : -1;
    auto tile_coords = [&](int item, int& m0, int& n0, int& sl) {
        int bid = item >> s_sh;
        sl = item - (bid << s_sh);
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

    // ---- loader: A cursor per STEP (c, dh), B cursor per chunk; both run ahead of the MFMAs, into the next tile ----
    constexpr int ROW_DEAD = -(1 << 24);
    const unsigned xs2 = (unsigned)p.xstride * 2u;
    const unsigned rowbytes = (unsigned)p.Wi * xs2;         // one image row of the activation tensor (< 2^24, launcher)
    int avb = (int)blockIdx.x, astep = 0;                   // (work item, step inside its K slice)
    int dh = 0, c0 = 0;
    u32x4 rsrc_a, rsrc_w;
    unsigned a_base[G::AK];                                 // byte offset of (image, row 0, pixel wo, this lane's 16-byte piece): the dw = 1 tap
    int a_hi0[G::AK];
    unsigned a_off[G::AK], w_off[G::BK2];
    int bvb = (int)blockIdx.x, bkc = 0, bcnt = 0;           // (work item, chunk index inside the tile's K, chunks done of the slice)
#pragma unroll
    for (int kb = 0; kb < G::BK2; ++kb) {                   // kb = 2 pass + region
        const int lr = 64 * (kb >> 1) + lrow;
        const int c = 64 * (lr >> 5) + 32 * (kb & 1) + (lr & 31);
        w_off[kb] = (unsigned)(c * p.K + lcol * 8) * 2u;
    }
    auto tap_rows = [&](int k0, int n) {                    // rows k0 .. k0 + n - 1 at the cursor's filter row dh (full-rate VALU only)
#pragma unroll
        for (int k = k0; k < k0 + n; ++k) {
            const int h = a_hi0[k] + dh;
            const unsigned off = __umul24((unsigned)h, rowbytes) + a_base[k];
            a_off[k] = (unsigned)h < (unsigned)p.Hi ? off : OOB;
        }
    };
    auto setup_a = [&](int vb) {
        int m0, n0, sl;
        tile_coords(vb, m0, n0, sl);
        const int b_first = __builtin_amdgcn_readfirstlane(ho_sh >= 0 ? m0 >> (wo_sh + ho_sh) : m0 / hw_out);
        rsrc_a = pp_rsrc(p.x + (size_t)b_first * img_elems);
#pragma unroll
        for (int k = 0; k < G::AK; ++k) {                   // k = APW * region + pass
            const int m = m0 + lrow + 64 * k;
            const int wo = m & (p.Wo - 1);
            const int t = m >> wo_sh;
            int ho, b;
            if (ho_sh >= 0) { ho = t & (p.Ho - 1); b = t >> ho_sh; } else { ho = t % p.Ho; b = t / p.Ho; }
            a_base[k] = (unsigned)((b - b_first) * p.Hi * p.Wi + wo) * xs2 + (unsigned)lcol * 16u;
            a_hi0[k] = m < p.M ? ho * p.sh - p.ph : ROW_DEAD;
        }
        const int step0 = sl * nsteps_s;                    // first step of the K slice: (channel chunk, filter row)
        const int cc = step0 / 3;
        dh = step0 - 3 * cc;
        c0 = cc * BKE;
    };
    auto advance_a = [&]() {                                // scalar part; tap_rows() follows in the shadow of MFMA blocks
        if (++astep == nsteps_s) {
            astep = 0;
            avb += nwg;
            if (avb < items) {
                setup_a(avb);
            } else {
#pragma unroll
                for (int k = 0; k < G::AK; ++k) a_hi0[k] = ROW_DEAD;
            }
        } else {
            const int dh1 = dh + 1;
            const bool wh = dh1 == 3;
            dh = wh ? 0 : dh1;
            c0 += wh ? BKE : 0;
        }
    };
    auto setup_b = [&](int vb) {
        int m0, n0, sl;
        tile_coords(vb, m0, n0, sl);
        rsrc_w = pp_rsrc(p.w + (size_t)n0 * p.K);
        bkc = sl * nk_s;
    };
    auto advance_b = [&]() {
        ++bkc;
        if (++bcnt == nk_s) {
            bcnt = 0;
            bvb += nwg;
            if (bvb < items) {
                setup_b(bvb);
            } else {
#pragma unroll
                for (int kb = 0; kb < G::BK2; ++kb) w_off[kb] = OOB;
            }
        }
    };
    auto issue_a = [&](int abuf, int reg) {
        const unsigned dst = lds0 + (unsigned)(G::A_OFF + abuf * G::ABUF + reg * G::REGION_A + wave * 1024);
#pragma unroll
        for (int ps = 0; ps < G::APW; ++ps) pp_dma16<0>(rsrc_a, dst + ps * 8192, reg ? a_off[G::APW + ps] : a_off[ps], (unsigned)c0 * 2u);
    };
    auto issue_b = [&](int bbuf, int reg) {
        const unsigned dst = lds0 + (unsigned)(G::B_OFF + bbuf * G::BBUF + reg * G::REGION_B + wave * 1024);
#pragma unroll
        for (int ps = 0; ps < G::BPW; ++ps) pp_dma16<0>(rsrc_w, dst + ps * 8192, reg ? w_off[2 * ps + 1] : w_off[2 * ps], (unsigned)bkc * (unsigned)ROWB);
    };

    // ---- MFMA side ----
    int cvb = (int)blockIdx.x, cm0, cn0, csl, ckc = 0;
    tile_coords(cvb, cm0, cn0, csl);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 fa[4][2], fb[2];
    // fragment read offsets inside the CURRENT A / B buffer (toggled by ABUF per step, BBUF per chunk).  Tap dw reads the LDS rows of
    // the pixels dw - 1 to the right: row - 1 / row / row + 1 with the slot swizzle of THAT row (it depends on the row's bits 1..3
    // only: the same for all four row tiles of a wave, and unchanged by the seam correction of +- Wo rows, Wo % 32 == 0)
    unsigned rd_a[3][4], rd_b[4];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int key = ((fr + dw - 1) >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rd_a[dw][s] = (unsigned)(G::A_OFF + g * G::REGION_A + (128 * wm2 + fr + dw - 1) * ROWB + (((2 * s + half) ^ key) * 16));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rd_b[s] = (unsigned)(G::B_OFF + (32 * wn + fr) * ROWB + (((2 * s + half) ^ fswz) * 16));
    const int seam_bytes = p.Wo * ROWB;
    // K half h of the wave's four row tiles for tap dw.  A lane whose neighbour pixel lies across the seam of its image row (lane 0 of a
    // row tile that starts an image row for dw 0, lane 31 of one that ends it for dw 2) reads Wo rows further down / up instead
    auto read_a = [&](auto dw_c, int h) {
        constexpr int DW = decltype(dw_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t0 = 128 * wm2 + 32 * i;                      // first row of the tile inside the group's region (wave-uniform)
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
    auto read_b = [&](int j, int h) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) fb[sh] = *reinterpret_cast<const u32x4*>(smem + rd_b[2 * h + sh] + j * G::REGION_B);
    };
    // one MFMA block; TAPN > 0: the tap arithmetic of TAPN activation rows from row TAP0 rides in its shadow
    auto mm = [&](int j, auto tap0_c, auto tapn_c) {
        constexpr int TAP0 = decltype(tap0_c)::value, TAPN = decltype(tapn_c)::value;
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][sh]), __builtin_bit_cast(bf16x8, fb[sh]), acc[i][j], 0, 0, 0);
        if constexpr (TAPN > 0) {
            tap_rows(TAP0, TAPN);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // up to four VALU
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;

    // ---- epilogue of the wave's 128 x 64 tile: 8 rounds of 16 rows through the wave's 4 KiB slab (see conv_igemm_bf16_pp.hip) ----
    // folded-BN scale / shift of the wave's columns stay in registers across tiles: a global load at the head of the epilogue is exposed
    // latency, and the vmcnt(0) hipcc puts behind it also waits for every LDS-DMA piece in flight
    float sc[2], sf[2];
    int sc_n0 = cn0;
    auto load_scale = [&](int en0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sc[j] = p.scale[en0 + 64 * wn + 32 * j + fr];
            sf[j] = p.shift[en0 + 64 * wn + 32 * j + fr];
        }
    };
    if constexpr (G::PERSIST) load_scale(cn0);          // (one tile per workgroup otherwise: loaded in the epilogue, no registers held across the loop)
    auto epilogue = [&](int em0, int en0, int esl, auto has_res_c) {
        constexpr bool HAS_RES = decltype(has_res_c)::value;
        const bool raw = OUT_F32 && p.ksplit > 1;       // split-K partial tile: no scale / shift (the reduce kernel applies them once)
        float* slab = reinterpret_cast<float*>(smem + G::SLAB_OFF + wave * 4096);
        const int er = lane >> 3;
        const int ec = (lane & 7) * 8;
        const int colg = en0 + 64 * wn + ec;
        const int mrow0 = em0 + G::GROUP_ROWS * g + 128 * wm2 + er;
        if (!G::PERSIST || en0 != sc_n0) {              // rare: a persistent workgroup's tiles share their column tile (grid % column tiles == 0)
            sc_n0 = en0;
            load_scale(en0);
        }
        const size_t slice_off = (size_t)esl * p.M * p.ldy;
        u32x4 rres[HAS_RES ? 8 : 1];
        auto res_load = [&](int q) {
            const int m = mrow0 + 8 * q;
            const int mc = m < p.M ? m : p.M - 1;
            rres[q & 7] = *reinterpret_cast<const u32x4*>(p.res + (size_t)mc * p.Cout + colg);
        };
        if (HAS_RES) {
#pragma unroll
            for (int q = 0; q < 8; ++q) res_load(q);
        }
        const int b0 = (2 * (lane & 7)) ^ (er & 3), b1 = (2 * (lane & 7) + 1) ^ (er & 3);
#pragma unroll
        for (int rd = 0; rd < 8; ++rd) {
            const int i = rd >> 1, hb = rd & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int lr16 = (rr & 3) + 8 * (rr >> 2) + 4 * half;
                    const int blk = (8 * j + (fr >> 2)) ^ (rr & 3);
                    slab[lr16 * 64 + blk * 4 + (fr & 3)] = raw ? acc[i][j][8 * hb + rr] : acc[i][j][8 * hb + rr] * sc[j] + sf[j];
                }
#pragma unroll
            for (int sr = 0; sr < 2; ++sr) {
                const int row16 = 8 * sr + er;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + row16 * 64 + b0 * 4);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + row16 * 64 + b1 * 4);
                const int m = mrow0 + 16 * rd + 8 * sr;
                if (HAS_RES) {
                    const u32x4 rr = rres[(2 * rd + sr) & 7];
                    v0[0] += bf16_lo(rr[0]); v0[1] += bf16_hi(rr[0]); v0[2] += bf16_lo(rr[1]); v0[3] += bf16_hi(rr[1]);
                    v1[0] += bf16_lo(rr[2]); v1[1] += bf16_hi(rr[2]); v1[2] += bf16_lo(rr[3]); v1[3] += bf16_hi(rr[3]);
                    if (2 * rd + sr + 8 < 16) res_load(2 * rd + sr + 8);
                }
                if (p.relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
                }
                if (m < p.M) {
                    if (OUT_F32) {
                        float* yo = reinterpret_cast<float*>(p.y) + slice_off + (size_t)m * p.ldy + colg;
                        *reinterpret_cast<f32x4*>(yo) = v0;
                        *reinterpret_cast<f32x4*>(yo + 4) = v1;
                    } else {
                        u32x4 o;
                        o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                        o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.y) + (size_t)m * p.ldy + colg) = o;
                    }
                }
            }
        }
    };

    // ---- prologue: step 0's activations and chunk 0's weights completely, A0 of step 1 ----
    setup_a(avb);
    setup_b(bvb);
    tap_rows(0, G::AK);
    issue_a(0, 0); issue_a(0, 1); issue_b(0, 0); issue_b(0, 1);
    advance_a();
    tap_rows(0, G::AK);
    advance_b();
    issue_a(1, 0);
    DWR_WAIT(G::APW);                                      // everything but A0 of step 1 has landed (this wave's pieces)
    pp_bar_raw();
    if (g == 1) pp_bar_raw();                              // group 1 runs one barrier behind group 0 from here on

    int abuf = 0, bbuf = 0;
    int adelta = G::ABUF, bdelta = G::BBUF;                 // the fragment read offsets follow the buffers by +- one buffer (an offset may be "negative": row -1)
    // one chunk = tap DW of the current step
    auto chunk = [&](auto dw_c) {
        constexpr int DW = decltype(dw_c)::value;
        // ---- phase 1: (j 0, K half 0) ----
        read_a(dw_c, 0);
        read_b(0, 0);
        issue_b(bbuf ^ 1, 0);                              // B0 of the next chunk
        DWR_WAIT(G::wait_p1(DW));                          // B1 of THIS chunk is complete (read in phase 2)
        pp_bar_raw();
        mm(0, I0{}, I0{});
        pp_bar_raw();
        // ---- phase 2: (j 1, K half 0) ----
        read_b(1, 0);
        issue_b(bbuf ^ 1, 1);                              // B1 of the next chunk
        if (DW == 1) advance_a();                          // the A cursor moves on to the step after next (scalars)
        pp_bar_raw();
        if constexpr (DW == 1) mm(1, I0{}, std::integral_constant<int, G::AK / 2>{});                                   // + row offsets of the first half of the rows
        else mm(1, I0{}, I0{});
        pp_bar_raw();
        // ---- phase 3: (j 0, K half 1); for dw 2 the last reads of this step's activations ----
        read_a(dw_c, 1);
        read_b(0, 1);
        advance_b();
        if (DW == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // returned BEFORE the barrier: A0 is re-filled right behind it
        pp_bar_raw();
        if constexpr (DW == 1) mm(0, std::integral_constant<int, G::AK / 2>{}, std::integral_constant<int, G::AK / 2>{});   // + the other half
        else mm(0, I0{}, I0{});
        pp_bar_raw();
        // ---- phase 4: (j 1, K half 1) ----
        read_b(1, 1);
        if (DW == 0) issue_a(abuf ^ 1, 1);                 // A1 of the next step (other A buffer)
        if (DW == 2) issue_a(abuf, 0);                     // A0 of the step after next (this A buffer)
        DWR_WAIT(G::wait_p4(DW));                          // B0 of the next chunk (dw 2: and the next step's activations) complete
        pp_bar_raw();
        mm(1, I0{}, I0{});
        pp_bar_raw();
        bbuf ^= 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd_b[s] += (unsigned)bdelta;
        bdelta = -bdelta;
    };
    while (true) {
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        abuf ^= 1;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int s = 0; s < 4; ++s) rd_a[dw][s] += (unsigned)adelta;
        adelta = -adelta;
        ckc += 3;
        if (ckc == nk_s) {                                 // tile (K slice) finished
            ckc = 0;
            if (g == 0) pp_bar_raw();                      // wait for group 1's last MFMA block: both epilogues run at once
            if (!G::PERSIST) {                             // the slabs re-use the A buffers: every wave's surplus pieces must have landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pp_bar_raw();
            }
            if (!OUT_F32 && p.res != nullptr) epilogue(cm0, cn0, csl, std::true_type{}); else epilogue(cm0, cn0, csl, std::false_type{});
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            cvb += nwg;
            if (cvb >= items) break;
            tile_coords(cvb, cm0, cn0, csl);
            if (g == 1) pp_bar_raw();                      // re-stagger
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // surplus (out-of-range) pieces of the loader
}

template <int BM, int BN, bool OUT_F32>
int launch_dwr(const ConvArgsH& a, hipStream_t s)
{
    using G = DwrGeom<BM, BN>;
    auto kern = conv3x3_dwr_bf16_kernel<BM, BN, OUT_F32>;
    static bool attr_done[64] = {};   // per instantiation, per device
    static int n_cu[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        HN_HIP(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_done[dev] = true;
    }
    const int tiles = hn_cdiv(a.M, BM) * (a.Cout / BN) * (OUT_F32 && a.ksplit > 1 ? a.ksplit : 1);      // work items
    const int cus = dev < 64 && n_cu[dev] > 0 ? n_cu[dev] : 256;
    const int grid = G::PERSIST ? (tiles < cus ? tiles : cus) : tiles;      // 512 x 128: one tile per workgroup (the slabs alias the A buffers)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), G::LDS, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// shape: 0 = 256 x 256 tiles (Cout % 256 == 0, Wo in {32, 64, 128}), 1 = 512 x 128 tiles (Cout % 128 == 0, Wo in {32 .. 256}),
// 2 = 512 x 64 tiles (conv3x3_dwr64_bf16.hip: Cout % 64 == 0, bf16 output without residual, no split-K)
bool hn_conv_bf16_dwr_ok(const ConvArgsH& a, int shape)
{
    const int group_rows = shape == 0 ? 128 : 256;
    const int bn = shape == 0 ? 256 : shape == 1 ? 128 : 64;
    if (shape == 2 && (a.res != nullptr || a.ksplit > 1)) return false;
    if (a.stat_sum != nullptr && shape != 2) return false;      // only the 64-column kernel has a statistics epilogue (the others spill with it)
    return a.KH == 3 && a.KW == 3 && a.sw == 1 && a.pw == 1 && a.ph == 1 && a.Wi == a.Wo && (a.Wo & (a.Wo - 1)) == 0 && a.Wo >= 32 &&
           group_rows % a.Wo == 0 && a.Cout % bn == 0 && a.Cin % BKE == 0 &&
           (a.ksplit <= 1 || ((a.ksplit & (a.ksplit - 1)) == 0 && (a.nk / 3) % a.ksplit == 0)) &&
           (long)a.Hi * a.Wi < (1L << 24) && (long)a.Wi * a.xstride * 2 < (1L << 24) && a.Hi < (1 << 20);
}

int hn_launch_conv_bf16_dwr(const ConvArgsH& a, int out_f32, int shape, hipStream_t s)
{
    HN_REQUIRE(hn_conv_bf16_dwr_ok(a, shape), "conv bf16 (dw reuse): 3x3, stride 1 / pad 1 along W, Wo a power of two that divides the group rows");
    HN_REQUIRE(!out_f32 || a.res == nullptr, "conv bf16 (dw reuse): no residual with float32 output");
    HN_REQUIRE(a.ksplit <= 1 || out_f32, "conv bf16 (dw reuse): split-K writes float32 partial tiles");
    if (shape == 2) {
        HN_REQUIRE(!out_f32, "conv bf16 (dw reuse, 64 columns): bf16 output only");
        return hn_launch_conv_bf16_dwr64(a, s);
    }
    if (shape == 0) return out_f32 ? launch_dwr<256, 256, true>(a, s) : launch_dwr<256, 256, false>(a, s);
    return out_f32 ? launch_dwr<512, 128, true>(a, s) : launch_dwr<512, 128, false>(a, s);
}
