// Ping-pong persistent implicit-GEMM convolution for gfx950: bf16 operands, f32 accumulation, 256x256 tiles, 8 waves.
//
// Same arithmetic as conv_igemm_bf16.hip (y = act(scale * sum_k A W + shift (+ res)), v_mfma_f32_32x32x16_bf16, k ascending,
// NHWC bf16 activations, weights packed [Cout][Cin / 64][kh][kw][64], circular W padding by index arithmetic) and therefore the
// same bits; what changes is the schedule of the k loop -- the part the two-stage kernels lose 40 % of the matrix pipe on
// (reference call sites: model.py:78-81 ResNet bottlenecks, model.py:123-135 ConvCompressH; the LSTM input projections).
//
//   * Two wave GROUPS (waves 0-3 / 4-7: one wave of each on every SIMD) run the same phase sequence ONE BARRIER APART: while a
//     group issues the 8 MFMAs of a phase, its SIMD partner issues the NEXT phase's fragment reads and LDS-DMA pieces.  The
//     matrix pipe of a SIMD always has one wave feeding it and neither wave waits for its own loads in front of its own MFMAs.
//     A phase = one 32-column half (j) of the wave's 128x64 tile over one half (h) of the 64-deep K chunk: 4 row tiles x 2 k
//     steps = 8 MFMAs on FOUR independent accumulators; order (j0,h0) (j1,h0) (j0,h1) (j1,h1) keeps k ascending per
//     accumulator.  Fragment reads per phase: 8 A + 2 B, 2 B, 8 A + 2 B, 2 B (the A fragments of a K half serve both j).
//   * LDS-DMA stays in flight ACROSS the barriers: raw s_barrier (no __syncthreads, whose fence drains vmcnt) and two counted
//     s_waitcnt vmcnt(6) per chunk.  A chunk's operands live in four 16 KiB regions -- A0 / A1 (the rows of wave group 0 / 1),
//     B0 / B1 (left / right 32 columns of every wave's 64).  Measured with tools/pp_stamps.py: a piece costs its wave ~50 cycles
//     of issue and needs ~1 us (1700-1800 cycles at the sustained clock, L2 hit or not) from issue to landed, i.e. ~3 phases.  So
//     every region goes out as early as its buffer allows and is waited for as late as the read allows: phase 4 issues the
//     activations (A0, A1) of the chunk AFTER next into this buffer (last read in phase 3; waited for four phases later),
//     phases 1 and 2 the weights B0 / B1 of the next chunk (waited for three phases later).  The tap arithmetic of the
//     activation cursor (circular W padding, H range check) rides in the shadow of the MFMA blocks of phases 2 and 3.
//   * The chunk stream does not stop at a tile boundary: the persistent workgroup's loader runs two chunks ahead of its MFMAs,
//     INTO THE NEXT TILE (no prologue bubble per tile; the last tile's surplus pieces are out-of-range loads = zero fill).
//   * Epilogue without workgroup barriers: every wave transposes its own accumulators through a private 4 KiB slab (16 rows x
//     64 columns f32, XOR-swizzled 16-byte blocks: conflict-free ds_write_b32 / ds_read_b128), all 16 residual rows of the
//     wave requested up front, whole 128-byte lines stored.  The two groups re-align for it (both epilogues at once) and
//     re-stagger behind it.
//
// LDS (160 KiB): chunk buffer 0 | chunk buffer 1 (A0 A1 B0 B1 each) | 8 slabs.  Loader rows, the source-side slot swizzle
// c ^ ((r >> 1) & 7) and the fragment reads are the two-stage kernels' (conflict-free, SQ_LDS_BANK_CONFLICT = 0 there).
//
// Ordering rules the schedule below is built on (MI355X guide, "LDS-DMA stays in flight across s_barrier"):
//   RAW  a region is read one phase after the barrier that follows BOTH groups' counted waits for its pieces;
//   WAR  a region is re-filled after a barrier that follows every reader's lgkmcnt(0) (issued before the reader's own barrier).
#include "hn_common.h"
#include "conv_bf16_args.h"
#include "conv_bf16_pp.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int PP_BUF = 65536;         // one K chunk: A0 | A1 | B0 | B1
constexpr int PP_REGION = 16384;      // 128 rows x 128 bytes
constexpr int PP_BOFF = 32768;
constexpr int PP_SLAB = 131072;       // 8 wave-private epilogue slabs of 4 KiB
constexpr int PP_LDS = 163840;

// measurement builds (-DHN_PP_STAMP=1|2, tools/pp_stamps.py): s_memtime at every barrier RELEASE (1: the starts of the four load blocks
// and the four MFMA blocks of a chunk) or at every barrier ARRIVAL (2: their ends), for the first PP_STAMP_CHUNKS chunks of a
// workgroup's stream.  The counter read is asynchronous (SMEM): the eight values stay in SGPRs and are consumed ONCE per chunk.
#ifdef HN_PP_STAMP
constexpr int PP_STAMP_CHUNKS = 24;
#define PP_STAMP_AT(kind, idx) do { if (HN_PP_STAMP == (kind)) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(st[idx]) : : "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PP_STAMP_AT(kind, idx) do {} while (0)
#endif

#define pp_bar() do { if (!(abl & 4)) pp_bar_raw(); else __builtin_amdgcn_sched_barrier(0); } while (0)

// PRIO: s_setprio(1) around the MFMA block of a phase (the partner wave is in its load block meanwhile)
// MASK: the training forward's fused BatchNorm + ReLU conv (bn_fold_forward): one more 2-byte store per row piece -- its own instantiation, so the
// inference kernel's code and register allocation are exactly what they were without it
template <bool OUT_F32, bool PRIO, bool MASK = false>
__global__ __launch_bounds__(512) void conv_igemm_bf16_pp_kernel(ConvArgsH p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef HN_PP_ABL
    // measurement builds (-DHN_PP_ABL=mask, tools/pp_ablate.sh): parts of the loop switched off (WRONG results) to price them --
    // 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no barriers, 8 no MFMAs, 16 no counted waits
    constexpr int abl = HN_PP_ABL;
#else
    constexpr int abl = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2;                                // wave group = row half of the tile
    const int wn = wave & 3;                                // 64-column slice
    const int lrow = tid >> 3;                              // loader: row 0..63 of a 64-row pass
    const int lslot = tid & 7;
    const int lcol = lslot ^ ((lrow >> 1) & 7);
    const int fr = lane & 31;
    const int half = lane >> 5;
    const int fswz = (fr >> 1) & 7;

    // LDS byte address of the dynamic segment (M0 of the LDS-DMA pieces)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int NT = p.Cout >> 8;
    const int total = ((p.M + 255) >> 8) * NT;
    const int hw_out = p.Ho * p.Wo;
    const size_t img_elems = (size_t)p.Hi * p.Wi * p.xstride;
    const int nwg = (int)gridDim.x;

    // power-of-two fast paths of the index arithmetic (every grid of this network; anything else takes the divisions)
    const int wo_sh = (p.Wo & (p.Wo - 1)) == 0 ? __builtin_ctz(p.Wo) : -1;
    const int ho_sh = (p.Ho & (p.Ho - 1)) == 0 ? __builtin_ctz(p.Ho) : -1;
    const int nt_sh = (NT & (NT - 1)) == 0 ? __builtin_ctz(NT) : -1;
    auto tile_coords = [&](int vb, int& m0, int& n0) {
        int bid = vb;
        if (p.xcd_swizzle) {
            const int q = total >> 3, r = total & 7;
            const int xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int mt = nt_sh >= 0 ? bid >> nt_sh : bid / NT;
        const int nt = bid - mt * NT;
        m0 = __builtin_amdgcn_readfirstlane(mt * 256);      // (a division expands to VALU code: pin the results back to SGPRs, or every
        n0 = __builtin_amdgcn_readfirstlane(nt * 256);      //  buffer descriptor derived from them needs a waterfall loop around its DMA)
    };

    // ---- loader state: two cursors that run ahead of the MFMAs, INTO THE NEXT TILE ----
    // A cursor (activations): rows of this thread = tile rows lrow + 64 k, k = 2 region + pass (region A0 = rows 0..127 = wave
    //   group 0, A1 = group 1); advanced in phase 2 (between A1 of chunk G+1 in phase 1 and A0 of chunk G+2 in phase 4)
    // B cursor (weights): tile columns 128 pass + 32 region + 64 (lrow >> 5) + (lrow & 31), k = 2 pass + region; advanced in
    //   phase 4 (behind B1 of chunk G+1 in phase 3)
    constexpr int ROW_DEAD = -(1 << 24);                    // a_hi0 of a row beyond M (or beyond the last tile): every tap is out of range
    const unsigned xs2 = (unsigned)p.xstride * 2u;
    int avb = (int)blockIdx.x, akc = 0;
    int dh = 0, dw = 0, c0 = 0;
    u32x4 rsrc_a, rsrc_w;
    unsigned a_base[4];                                     // byte offset of (image, pixel 0, this lane's 16-byte piece) behind rsrc_a
    int a_hi0[4], a_wi0[4];
    unsigned a_off[4], w_off[4];
    int bvb = (int)blockIdx.x, bkc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 128 * (k >> 1) + 32 * (k & 1) + 64 * (lrow >> 5) + (lrow & 31);
        w_off[k] = (unsigned)(c * p.K + lcol * 8) * 2u;
    }
    auto tap_rows = [&](int k0) {                           // rows k0, k0 + 1 at the cursor's tap (dh, dw): branch-free, full-rate VALU only
#pragma unroll                                             // (this runs in the shadow of an MFMA block)
        for (int k = k0; k < k0 + 2; ++k) {
            const int h = a_hi0[k] + dh;
            int t = a_wi0[k] + dw;
            t += t < 0 ? p.Wi : 0;
            t -= t >= p.Wi ? p.Wi : 0;
            const unsigned pix = __umul24((unsigned)h, (unsigned)p.Wi) + (unsigned)t;       // < 2^24 for a valid row (checked by the launcher)
            const unsigned off = __umul24(pix, xs2) + a_base[k];
            a_off[k] = (unsigned)h < (unsigned)p.Hi ? off : OOB;
        }
    };
    auto setup_a = [&](int vb) {
        int m0, n0;
        tile_coords(vb, m0, n0);
        const int b_first = __builtin_amdgcn_readfirstlane(wo_sh >= 0 && ho_sh >= 0 ? m0 >> (wo_sh + ho_sh) : m0 / hw_out);
        rsrc_a = pp_rsrc(p.x + (size_t)b_first * img_elems);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + lrow + 64 * k;
            int wo, ho, b;
            if (wo_sh >= 0 && ho_sh >= 0) {
                wo = m & (p.Wo - 1);
                const int t = m >> wo_sh;
                ho = t & (p.Ho - 1);
                b = t >> ho_sh;
            } else {
                wo = m % p.Wo;
                const int t = m / p.Wo;
                ho = t % p.Ho;
                b = t / p.Ho;
            }
            a_base[k] = (unsigned)((b - b_first) * p.Hi * p.Wi) * xs2 + (unsigned)lcol * 16u;
            a_hi0[k] = m < p.M ? ho * p.sh - p.ph : ROW_DEAD;
            a_wi0[k] = wo * p.sw - p.pw;
        }
        dh = 0; dw = 0; c0 = 0;
    };
    auto advance_a = [&]() {                                // scalar part; tap_rows() follows in the shadow of the next two MFMA blocks
        if (++akc == p.nk) {
            akc = 0;
            avb += nwg;
            if (avb < total) {
                setup_a(avb);
            } else {                     // past the workgroup's last tile: the surplus pieces are out-of-range loads (zero fill)
#pragma unroll
                for (int k = 0; k < 4; ++k) a_hi0[k] = ROW_DEAD;
            }
        } else {                         // taps inner, 64-channel chunks outer (conv_igemm_bf16.hip fetch()); selects, not branches
            const int dw1 = dw + 1;
            const bool ww = dw1 == p.KW;
            dw = ww ? 0 : dw1;
            const int dh1 = dh + (ww ? 1 : 0);
            const bool wh = dh1 == p.KH;
            dh = wh ? 0 : dh1;
            c0 += wh ? BKE : 0;
        }
    };
    auto setup_b = [&](int vb) {
        int m0, n0;
        tile_coords(vb, m0, n0);
        rsrc_w = pp_rsrc(p.w + (size_t)n0 * p.K);
    };
    auto advance_b = [&]() {
        if (++bkc == p.nk) {
            bkc = 0;
            bvb += nwg;
            if (bvb < total) {
                setup_b(bvb);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) w_off[k] = OOB;
            }
        }
    };
    // region `reg` (0 / 1) of the cursor's chunk -> chunk buffer `buf`: two 1 KiB pieces per wave
    auto issue_a = [&](int buf, int reg) {
        const unsigned dst = lds0 + (unsigned)(buf * PP_BUF + reg * PP_REGION + wave * 1024);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) pp_dma16<HN_PP_POLICY_A>(rsrc_a, dst + ps * 8192, reg ? a_off[2 + ps] : a_off[ps], (unsigned)c0 * 2u);
    };
    auto issue_b = [&](int buf, int reg) {
        const unsigned dst = lds0 + (unsigned)(buf * PP_BUF + PP_BOFF + reg * PP_REGION + wave * 1024);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) pp_dma16<HN_PP_POLICY_B>(rsrc_w, dst + ps * 8192, reg ? w_off[1 + 2 * ps] : w_off[2 * ps], (unsigned)bkc * (unsigned)ROWB);
    };

    // ---- MFMA side ----
    int cvb = (int)blockIdx.x, cm0, cn0, ckc = 0;
    tile_coords(cvb, cm0, cn0);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 fa[4][2], fb[2];
#ifdef HN_PP_ABL
    for (int i = 0; i < 4; ++i) { fa[i][0] = fa[i][1] = u32x4{0, 0, 0, 0}; }
    fb[0] = fb[1] = u32x4{0, 0, 0, 0};
#endif
    // fragment read offsets inside the CURRENT chunk buffer (toggled by PP_BUF per chunk): k step s -> slot (2 s + half) ^ fswz
    unsigned rd_a[4], rd_b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned ks = (unsigned)(((2 * s + half) ^ fswz) * 16);
        rd_a[s] = (unsigned)(g * PP_REGION + fr * ROWB) + ks;
        rd_b[s] = (unsigned)(PP_BOFF + (32 * wn + fr) * ROWB) + ks;
    }
    auto read_a = [&](int h) {          // K half h of the wave's four row tiles
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) fa[i][sh] = *reinterpret_cast<const u32x4*>(smem + rd_a[2 * h + sh] + i * 32 * ROWB);
    };
    auto read_b = [&](int j, int h) {   // K half h of column half j
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) fb[sh] = *reinterpret_cast<const u32x4*>(smem + rd_b[2 * h + sh] + j * PP_REGION);
    };
    // one MFMA block: column half j, the K half in fa / fb.  TAPS >= 0: two rows of the A cursor's tap arithmetic ride in the shadow
    // of the MFMAs (one MFMA, then up to four VALU instructions, ...) instead of lengthening a load block
    auto mm = [&](int j, auto taps_c) {
        constexpr int TAPS = decltype(taps_c)::value;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#ifdef HN_PP_MMORD          // measurement builds: the two k steps of an accumulator back to back instead of four MFMAs apart
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][sh]), __builtin_bit_cast(bf16x8, fb[sh]), acc[i][j], 0, 0, 0);
#else
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][sh]), __builtin_bit_cast(bf16x8, fb[sh]), acc[i][j], 0, 0, 0);
#endif
        if constexpr (TAPS >= 0) {
            tap_rows(TAPS);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // up to four VALU
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    using NoTaps = std::integral_constant<int, -1>;

    // ---- epilogue of the wave's 128 x 64 tile: 8 rounds of 16 rows through the wave's slab ----
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
    load_scale(cn0);
    auto epilogue = [&](int em0, int en0, auto has_res_c) {
        // (HAS_RES is a compile-time copy of the body: with a runtime `if` around the residual loads AND around their uses hipcc
        //  assumes loads that were never consumed and makes the next chunk's fragment reads wait for them -- vmcnt(0), i.e. for the
        //  whole LDS-DMA queue)
        constexpr bool HAS_RES = decltype(has_res_c)::value;
        float* slab = reinterpret_cast<float*>(smem + PP_SLAB + wave * 4096);
        const int er = lane >> 3;                       // row (and row + 8) inside a round
        const int ec = (lane & 7) * 8;                  // 8 consecutive channels of the wave's 64
        const int colg = en0 + 64 * wn + ec;
        const int mrow0 = em0 + 128 * g + er;
        if (en0 != sc_n0) {                             // rare: a persistent workgroup's tiles share their column tile (grid % column tiles == 0)
            sc_n0 = en0;
            load_scale(en0);
        }
        // residual rows: a ring of 8 requests (4 rounds) ahead of their use
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
        const int b0 = (2 * (lane & 7)) ^ (er & 3), b1 = (2 * (lane & 7) + 1) ^ (er & 3);   // swizzled 16-byte blocks of this lane's 8 channels
#pragma unroll
        for (int rd = 0; rd < 8; ++rd) {
            const int i = rd >> 1, hb = rd & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int lr16 = (rr & 3) + 8 * (rr >> 2) + 4 * half;
                    const int blk = (8 * j + (fr >> 2)) ^ (rr & 3);
                    slab[lr16 * 64 + blk * 4 + (fr & 3)] = acc[i][j][8 * hb + rr] * sc[j] + sf[j];
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
                        float* yo = reinterpret_cast<float*>(p.y) + (size_t)m * p.ldy + colg;
                        *reinterpret_cast<f32x4*>(yo) = v0;
                        *reinterpret_cast<f32x4*>(yo + 4) = v1;
                    } else {
                        u32x4 o;
                        o[0] = pack_bf16(v0[0], v0[1]); o[1] = pack_bf16(v0[2], v0[3]);
                        o[2] = pack_bf16(v1[0], v1[1]); o[3] = pack_bf16(v1[2], v1[3]);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.y) + (size_t)m * p.ldy + colg) = o;
                        if (MASK) {            // (the training forward's fused BatchNorm + ReLU, see conv_igemm_bf16_kernel) -- one more store per row:
                                               //  the hand-counted vmcnt waits only ever see MORE operations outstanding than they assume, i.e. they over-wait
                            unsigned mk = 0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) mk |= (v0[k] > 0.f ? 1u : 0u) << k | (v1[k] > 0.f ? 1u : 0u) << (8 + k);
                            *reinterpret_cast<unsigned short*>(p.mask_out + (((size_t)m * p.ldy + colg) >> 2)) = (unsigned short)mk;
                        }
                    }
                }
            }
        }
    };

    // ---- prologue: chunk 0 completely, the activations of chunk 1 (phases 1 and 2 of chunk 0 bring its weights) ----
    setup_a(avb);
    setup_b(bvb);
    tap_rows(0); tap_rows(2);
    issue_a(0, 0); issue_a(0, 1); issue_b(0, 0); issue_b(0, 1);
    advance_a();
    tap_rows(0); tap_rows(2);
    advance_b();
    issue_a(1, 0); issue_a(1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // chunk 0 has landed (this wave's pieces)
    pp_bar();
    if (g == 1) pp_bar();                                  // group 1 runs one barrier behind group 0 from here on

    int buf = 0;
#ifdef HN_PP_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int gchunk = 0;
#endif
    while (true) {
        // ---- phase 1: (j 0, K half 0) ----
        PP_STAMP_AT(1, 0);
        if (!(abl & 2)) read_a(0);
        if (!(abl & 2)) read_b(0, 0);
        if (!(abl & 1)) issue_b(buf ^ 1, 0);                               // B0 of the next chunk
        if (!(abl & 16)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but the three newest regions: B1 of THIS chunk is complete (read in phase 2)
#ifdef HN_PP_LGKB          // measurement builds: every phase waits for its fragment reads in front of its barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PP_STAMP_AT(2, 0);
        pp_bar();                                          // (the fragment reads return behind the barrier, while the partner issues its loads)
        PP_STAMP_AT(1, 1);
        if (!(abl & 8)) mm(0, NoTaps{});
        PP_STAMP_AT(2, 1);
        pp_bar();
        // ---- phase 2: (j 1, K half 0); the A cursor moves on to the chunk after next ----
        PP_STAMP_AT(1, 2);
        if (!(abl & 2)) read_b(1, 0);
        if (!(abl & 1)) issue_b(buf ^ 1, 1);                               // B1 of the next chunk
        advance_a();                                       // (scalars; the tap arithmetic follows in the shadow of the next two MFMA blocks)
#ifdef HN_PP_LGKB
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PP_STAMP_AT(2, 2);
        pp_bar();
        PP_STAMP_AT(1, 3);
        if (!(abl & 8)) mm(1, std::integral_constant<int, 0>{});           // + tap arithmetic of rows 0, 1
        PP_STAMP_AT(2, 3);
        pp_bar();
        // ---- phase 3: (j 0, K half 1): last reads of A0, A1, B0 ----
        PP_STAMP_AT(1, 4);
        if (!(abl & 2)) read_a(1);
        if (!(abl & 2)) read_b(0, 1);
        advance_b();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // returned BEFORE the barrier: A0 / A1 are re-filled right behind it
        PP_STAMP_AT(2, 4);
        pp_bar();
        PP_STAMP_AT(1, 5);
        if (!(abl & 8)) mm(0, std::integral_constant<int, 2>{});           // + tap arithmetic of rows 2, 3
        PP_STAMP_AT(2, 5);
        pp_bar();
        // ---- phase 4: (j 1, K half 1) ----
        PP_STAMP_AT(1, 6);
        if (!(abl & 2)) read_b(1, 1);
        if (!(abl & 1)) { issue_a(buf, 0); issue_a(buf, 1); }               // the activations of the chunk after next into THIS buffer
        if (!(abl & 16)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // all but B1 of the next chunk and these four pieces: its A0, A1, B0 are complete
#ifdef HN_PP_LGKB
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PP_STAMP_AT(2, 6);
        pp_bar();
        PP_STAMP_AT(1, 7);
        if (!(abl & 8)) mm(1, NoTaps{});
        PP_STAMP_AT(2, 7);
        pp_bar();

        buf ^= 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) { rd_a[s] ^= (unsigned)PP_BUF; rd_b[s] ^= (unsigned)PP_BUF; }
#ifdef HN_PP_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p.trace && gchunk < PP_STAMP_CHUNKS && blockIdx.x < 64 && lane < 8) {
            unsigned long long v = st[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) v = lane == q ? st[q] : v;
            p.trace[(((size_t)blockIdx.x * 8 + wave) * PP_STAMP_CHUNKS + gchunk) * 8 + lane] = v;
        }
        ++gchunk;
#endif
        if (++ckc == p.nk) {                               // tile finished
            ckc = 0;
            if (g == 0) pp_bar();                          // wait for group 1's last MFMA block: both epilogues run at once
            if (!OUT_F32 && p.res != nullptr) epilogue(cm0, cn0, std::true_type{}); else epilogue(cm0, cn0, std::false_type{});
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            cvb += nwg;
            if (cvb >= total) break;
            tile_coords(cvb, cm0, cn0);
            if (g == 1) pp_bar();                          // re-stagger
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // surplus (out-of-range) pieces of the loader
}

template <bool OUT_F32, bool PRIO, bool MASK = false>
int launch_pp(const ConvArgsH& a, hipStream_t s)
{
    auto kern = conv_igemm_bf16_pp_kernel<OUT_F32, PRIO, MASK>;
    static bool attr_done[64] = {};   // per instantiation, per device
    static int n_cu[64] = {};
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
        HN_HIP(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_done[dev] = true;
    }
    const int tiles = hn_cdiv(a.M, 256) * (a.Cout / 256);
    const int cus = dev < 64 && n_cu[dev] > 0 ? n_cu[dev] : 256;
    const int grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), PP_LDS, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// variant bit 0: s_setprio around the MFMA blocks
int hn_launch_conv_bf16_pp(const ConvArgsH& a, int out_f32, int variant, hipStream_t s)
{
    HN_REQUIRE(a.Cout % 256 == 0 && a.nk >= 2 && a.ksplit <= 1 && a.stat_sum == nullptr, "conv bf16 (ping-pong): Cout %% 256, K >= 128, forward only");
    HN_REQUIRE(!out_f32 || a.res == nullptr, "conv bf16 (ping-pong): no residual with float32 output");
    if (a.mask_out) {
        HN_REQUIRE(!out_f32 && a.relu, "conv bf16 (ping-pong): the ReLU bit mask goes with a bf16 ReLU output");
        return launch_pp<false, false, true>(a, s);
    }
    if (out_f32) return (variant & 1) ? launch_pp<true, true>(a, s) : launch_pp<true, false>(a, s);
    return (variant & 1) ? launch_pp<false, true>(a, s) : launch_pp<false, false>(a, s);
}
