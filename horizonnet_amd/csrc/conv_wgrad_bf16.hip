// Weight-gradient GEMM for gfx950 on the bf16 matrix cores, f32 accumulation:
//
//   dW[n][k'] = sum_m dZ[m][n] * A[m][k']      m = output pixel (the REDUCTION index), n = output channel,
//                                              k' = (dh, dw, c) tap x input channel
//
// bf16 form of conv_wgrad_f32.hip for train_precision "bf16" (reference: autograd of the convolutions under autocast,
// train.py:51,273-280).  Same decomposition (grid = n-tiles x k'-tiles x split of the m range, float atomics into the
// packed f32 gradient), bf16 operands: dZ and the layer input both exist as NHWC bf16 copies written by the BN-adjoint
// / affine passes.
//
// The catch: the reduction index m is the SLOW index of both NHWC operands, while v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k per lane.  The tiles are therefore staged exactly as they lie in memory -- LDS images [m][n] and
// [m][k'] filled with coalesced 16-byte row segments -- and transposed on the way out of LDS by the CDNA4 transpose
// read ds_read_b64_tr_b16: within a group of 16 lanes, lane r supplies the address of 4 contiguous bf16 and lane i
// receives element (i % 4) of lanes i/4, i/4 + 4, i/4 + 8, i/4 + 12 (measured, tools/probe/tr16_probe.hip).  Pointing
// lane r at image row mb + r/4, columns nb + 4 (r % 4) .. +3 makes lane i receive rows mb..mb+3 of column nb + i:
// a 4 (m) x 16 (n) block delivered column-wise.  Two such reads give the 8 consecutive m of one MFMA operand lane.
// Image rows are 320 bytes apart (256 data + 64 pad): the 4 rows x 2 sixteen-column halves a 32-lane pass touches
// land in 8 disjoint 8-dword bank groups.
#include "hn_common.h"

#include <stdlib.h>

namespace {

constexpr int PITCH = 320;              // bytes per LDS image row of <= 128 columns (128 bf16 + 64 pad); 256 columns: 576 (same residue 16 dwords mod 64)
constexpr int pitch_of(int cols) { return cols <= 128 ? 320 : cols * 2 + 64; }
constexpr unsigned OOB = 0x80000000u;

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WgradArgsH {
    const u16* x;        // NHWC bf16 input of the forward conv [B][Hi][Wi][Cin]
    const u16* dz;       // bf16 gradient w.r.t. the conv output [M][Cout]
    float* dw;           // packed f32 [Cout][K], pre-zeroed, accumulated with atomics
    int Hi, Wi, Cin, Ho, Wo, Cout;
    int KW, sh, sw, ph, pw;
    int M, K, mchunk;
    int xstride, dzstride;   // elements between input pixels / between dz rows (>= Cin / Cout: column views of wider matrices)
    // FOLD instantiations only (the BatchNorm-folded adjoint of the 1x1 convs, train.hip: bn_fold_*): the dz operand is dy * ReLU mask
    // formed on load (bmask: 4 mask bits per byte, element e -> byte e >> 2, as affine_act_kernel stores them; null = no mask), the masked
    // rows are written back in place (dz_wb = dz, by the kt == 0 tiles; null = no write-back) and the column sums of the (masked) dz
    // operand are added to colsum_dz[Cout] (doubles, kt == 0 tiles; null = not wanted)
    const unsigned char* bmask;
    u16* dz_wb;
    double* colsum_dz;
    // ... and, for a run-to-run reproducible result (the Gram matrix behind the training FORWARD's batch statistics), `slab` != null: every
    // workgroup stores its partial tile to slab[sp][Cout][K] and its partial column sums to slab_cs[sp][Cout] (plain stores, no atomics);
    // wgrad_slab_reduce_kernel adds the m splits in index order in double precision.
    float* slab;
    float* slab_cs;
    // FUSEA instantiation (TN = Cout = 256, TK = Cin = 64: layer1's conv3 / downsample): the data-gradient conv on g that does not depend on the
    // batch sums, a_out[m][j] = sum_n g[m][n] wa[j][n] (wa = (c1 * W)^T, [Cin][Cout] bf16; a_out [M][Cin] bf16), rides in this pass over g
    const u16* wa;
    u16* a_out;
};

__device__ __forceinline__ u32x4 hbuf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}

// 8 consecutive m (image rows mb .. mb+7) of column nb + (lane & 15) [+16 for the odd 16-lane group]: one MFMA operand
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int mb, int nb, int lane, int pitch = PITCH)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int r = lane & 15;
    const int g = lane >> 4;
    const char* p = img + (mb + 8 * (g >> 1) + (r >> 2)) * pitch + (nb + 16 * (g & 1) + 4 * (r & 3)) * 2;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 4 * pitch));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
#else
    return bf16x8{};
#endif
}

// WCHB: m rows per chunk (one barrier per chunk): 32 = two k16 steps (8 MFMAs per wave and barrier for the 128x128 tile, 40 KB of
// LDS: 4 workgroups per CU), 64 = four (16 MFMAs per barrier, 80 KB: 2 workgroups per CU)
// NW = 8 (512 threads, 4 x 2 waves): the 256-wide tiles.  A 128x128 tile moves 32 KB from L2 into the CU per 64-row chunk for
// 2.1 MFLOP = 64 flop per byte, which caps it near 0.65 PFLOP/s at the ~10 TB/s the L2 -> CU path delivers chip-wide (the 61
// launches of the training step ran at 0.6); 256x256 doubles that (one workgroup per CU, 147 KB of LDS, 128 accumulator
// registers per wave).
// STEM: the 7x7 / 2 stem (TN = 64 output channels, TK = 256 = the stem's packed K: 8 filter rows x 8 window pixels x 4 channels,
// row 7 / window pixel 0 / channel 3 are padding): the A image row of output pixel (ho, wo) is, per filter row dh, the 64
// contiguous bytes of the NHWC4 input at row 2 ho - 3 + dh, pixels 2 wo - 4 .. 2 wo + 3 (circular) -- four 16-byte pieces; window
// pixel t holds filter tap dw = t - 1 (the bf16 forward's stem layout), so callers un-pack from dw_packed + 4 floats.
template <int TN, int TK, int WCHB, int NW = 4, bool STEM = false, bool FOLD = false, bool FUSEA = false>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_bf16_kernel(WgradArgsH p)
{
    static_assert(NW == 4 || NW == 8, "waves");
    static_assert(!(FOLD && STEM), "the folded adjoint is the 1x1 convs'");
    static_assert(!FUSEA || (FOLD && NW == 4 && WCHB == 32 && TK == 64 && TN % 16 == 0), "fused conv A: one 32-row MFMA tile per chunk, 64 output channels");
    static_assert(!STEM || (TN == 64 && TK == 256), "stem tile");
    constexpr int THREADS = NW * 64;
    constexpr int WAVES_N = NW / 2;                  // waves along n; 2 along k'
    constexpr int WN = TN / WAVES_N, WK = TK / 2;    // wave tile
    constexpr int TI = WN / 32, TJ = WK / 32;
    static_assert(TI >= 1 && TJ >= 1, "wave tile");
    constexpr int N_TPR = TN / 8, K_TPR = TK / 8;    // loader threads per image row (16 bytes = 8 bf16 each)
    constexpr int N_RPP = THREADS / N_TPR, K_RPP = THREADS / K_TPR;
    constexpr int N_PS = WCHB / N_RPP, K_PS = WCHB / K_RPP;
    static_assert(N_PS >= 1 && K_PS >= 1, "loader geometry");
    constexpr int PN = pitch_of(TN), PK = pitch_of(TK);
    constexpr int IMG_N = WCHB * PN, IMG_K = WCHB * PK;      // bytes per operand image

    extern __shared__ __attribute__((aligned(16))) char hsmem[];
    char* s_dz0 = hsmem;                             // [2][IMG_N]
    char* s_a0 = hsmem + 2 * IMG_N;                  // [2][IMG_K]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wi_ = wave >> 1, wj_ = wave & 1;

    const int NT = p.Cout / TN;
    const int KT = p.K / TK;
    int bid = blockIdx.x;                            // XCD-aware order, see conv_wgrad_f32.hip
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % NT;
    bid /= NT;
    const int kt = bid % KT;
    const int sp = bid / KT;
    const int n0 = nt * TN;
    const int k0 = kt * TK;
    const int mlo = sp * p.mchunk;
    const int mhi = (mlo + p.mchunk) < p.M ? (mlo + p.mchunk) : p.M;
    if (mlo >= mhi) return;

    const int tap = k0 / p.Cin;                      // the whole k' tile lies inside one filter tap (TK | Cin)
    const int c0 = k0 % p.Cin;
    const int dh = tap / p.KW;
    const int dw = tap % p.KW;

    const int hw_out = p.Ho * p.Wo;
    const int b_first = mlo / hw_out;
    const size_t img_elems = (size_t)p.Hi * p.Wi * p.xstride;
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.x + (size_t)b_first * img_elems), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_dz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.dz + (size_t)mlo * p.dzstride), 0, 0x7fffffff, 0x00020000);

    const int n_row = tid / N_TPR, n_col = (tid % N_TPR) * 8;
    const int k_row = tid / K_TPR, k_col = (tid % K_TPR) * 8;

    int r_wo[K_PS], r_ho[K_PS], r_b[K_PS];
#pragma unroll
    for (int q = 0; q < K_PS; ++q) {
        const int m = mlo + k_row + q * K_RPP;
        r_wo[q] = m % p.Wo;
        const int t = m / p.Wo;
        r_ho[q] = t % p.Ho;
        r_b[q] = t / p.Ho - b_first;
    }
    const int hw_in = p.Hi * p.Wi;
    const bool wide = p.Wo >= WCHB;

    u32x4 rdz[N_PS], ra[K_PS];
    unsigned rmk[FOLD ? N_PS : 1];       // FOLD: the 8 ReLU mask bits of rdz[q] (applied in stage(), so the loads stay in flight)
    float csum[FOLD ? 8 : 1];            // ... column sums of this thread's 8 columns over all its rows
    if (FOLD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) csum[k] = 0.f;
    }
    const bool fold_owner = FOLD && kt == 0;     // the one workgroup per (n tile, m range) that writes back / sums
    const bool has_mask = FOLD && p.bmask != nullptr;
    const bool do_wb = has_mask && p.dz_wb != nullptr;       // (dz_wb == dz: the masked rows replace the rows they were read from)
    // mask bytes of the range's first row (element e -> byte e >> 2); 2-byte loads at (dz byte offset) >> 3
    const __amdgpu_buffer_rsrc_t rsrc_mk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(FOLD && p.bmask ? p.bmask + (((size_t)mlo * p.dzstride) >> 2) : nullptr), 0, 0x7fffffff, 0x00020000);
    auto fetch = [&](int mc) {
#pragma unroll
        for (int q = 0; q < N_PS; ++q) {
            const int m = mc + n_row + q * N_RPP;
            const unsigned off = m < mhi ? (unsigned)((size_t)(m - mlo) * p.dzstride + n0 + n_col) * 2u : OOB;
            rdz[q] = hbuf_load16(rsrc_dz, off);
            if (FOLD) {     // branch-free: a row past the range reads mask 0 through the bounds check (its data is 0 anyway) and its write-back is dropped
                rmk[q] = has_mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsrc_mk, m < mhi ? (off >> 3) : OOB, 0, 0) : 0xffffu;
            }
        }
#pragma unroll
        for (int q = 0; q < K_PS; ++q) {
            const int m = mc + k_row + q * K_RPP;
            int hi, wi;
            bool ok = m < mhi;
            if (STEM) {                          // piece k_col / 8: filter row (piece >> 2), window pixels 2 (piece & 3), + 1
                const int pc = k_col >> 3;
                hi = r_ho[q] * 2 - 3 + (pc >> 2);
                wi = r_wo[q] * 2 - 4 + 2 * (pc & 3);     // even origin: a 2-pixel piece never straddles the circular wrap
                ok = ok && pc < 28;              // filter row 7 does not exist
            } else {
                hi = r_ho[q] * p.sh - p.ph + dh;
                wi = r_wo[q] * p.sw - p.pw + dw;
            }
            wi = wi < 0 ? wi + p.Wi : wi;
            wi = wi >= p.Wi ? wi - p.Wi : wi;
            const unsigned pix = (unsigned)(r_b[q] * hw_in + hi * p.Wi + wi);
            const unsigned in_off = STEM ? pix * 8u : (pix * (unsigned)p.xstride + (unsigned)(c0 + k_col)) * 2u;
            ra[q] = hbuf_load16(rsrc_x, (ok && (unsigned)hi < (unsigned)p.Hi) ? in_off : OOB);
            if (wide) {
                int wn = r_wo[q] + WCHB;
                const bool wrap_w = wn >= p.Wo;
                wn = wrap_w ? wn - p.Wo : wn;
                const int hn = r_ho[q] + (wrap_w ? 1 : 0);
                const bool wrap_h = hn >= p.Ho;
                r_ho[q] = wrap_h ? 0 : hn;
                r_b[q] += wrap_h ? 1 : 0;
                r_wo[q] = wn;
            } else {
                const int mn = m + WCHB;
                r_wo[q] = mn % p.Wo;
                const int t = mn / p.Wo;
                r_ho[q] = t % p.Ho;
                r_b[q] = t / p.Ho - b_first;
            }
        }
    };
    auto stage = [&](int buf, int smc) {      // smc: first row of the chunk being staged (its write-back offsets are recomputed here: registers)
        if (FOLD) {
#pragma unroll
            for (int q = 0; q < N_PS; ++q) {
                u32x4 v = rdz[q];
                const unsigned mk = (rmk[q] & 0xfu) | ((rmk[q] >> 4) & 0xf0u);      // two mask bytes (4 bits each) -> 8 bits
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned keep = (((mk >> (2 * k)) & 1u) ? 0x0000ffffu : 0u) | (((mk >> (2 * k + 1)) & 1u) ? 0xffff0000u : 0u);
                    v[k] &= keep;
                }
                rdz[q] = v;
                if (fold_owner) {
                    if (do_wb) {
                        const int m = smc + n_row + q * N_RPP;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_dz, m < mhi ? (unsigned)((size_t)(m - mlo) * p.dzstride + n0 + n_col) * 2u : OOB, 0, 0);
                    }
                    if (p.colsum_dz) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            csum[2 * k] += __builtin_bit_cast(float, v[k] << 16);
                            csum[2 * k + 1] += __builtin_bit_cast(float, v[k] & 0xffff0000u);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < N_PS; ++q)
            *reinterpret_cast<u32x4*>(s_dz0 + buf * IMG_N + (n_row + q * N_RPP) * PN + n_col * 2) = rdz[q];
#pragma unroll
        for (int q = 0; q < K_PS; ++q)
            *reinterpret_cast<u32x4*>(s_a0 + buf * IMG_K + (k_row + q * K_RPP) * PK + k_col * 2) = ra[q];
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // FUSEA: waves 0 / 1 hold the (c1 * W)^T fragments of their 32 output channels for the whole m range: lane (j = lane % 32, k half = lane / 32)
    // of k step ks owns wa[j][16 ks + 8 half .. + 7] -- the MFMA B operand of  a_out tile [32 rows][32 j] += g[32 rows][16 n] x wa^T
    bf16x8 wfr[FUSEA ? TN / 16 : 1];
    if (FUSEA && wave < 2) {
        const u16* wrow = p.wa + (size_t)(wave * 32 + (lane & 31)) * TN + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < TN / 16; ++ks) wfr[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wrow + 16 * ks));
    }

    fetch(mlo);
    stage(0, mlo);
    __syncthreads();
    int buf = 0;
    for (int mc = mlo; mc < mhi; mc += WCHB) {
        const bool more = mc + WCHB < mhi;
        if (more) fetch(mc + WCHB);
        const char* dz_s = s_dz0 + buf * IMG_N;
        const char* a_s = s_a0 + buf * IMG_K;
        if (FUSEA && wave < 2) {
            // the chunk's g rows are in the LDS image as staged (masked): the MFMA A operand is a plain 16-byte read, row = lane % 32
            f32x16 ca;
#pragma unroll
            for (int r = 0; r < 16; ++r) ca[r] = 0.f;
            const char* grow = dz_s + (lane & 31) * PN + 16 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < TN / 16; ++ks) {
                const bf16x8 ga = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(grow + 32 * ks));
                ca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, wfr[ks], ca, 0, 0, 0);
            }
            // accumulator r of lane l: row (r & 3) + 8 (r >> 2) + 4 (l / 32), column l % 32 -> through the wave's private 2 KiB of LDS ([32 rows][32 bf16])
            // and out as 16-byte pieces (two store instructions per wave instead of sixteen 2-byte ones)
            u16* ws_ = reinterpret_cast<u16*>(hsmem + 2 * IMG_N + 2 * IMG_K + wave * 2048);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                unsigned pk;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(ca[r]), "v"(0.f));
                ws_[row * 32 + (lane & 31)] = (u16)(pk & 0xffffu);
            }
            __builtin_amdgcn_wave_barrier();                 // (one wave: its LDS operations execute in issue order; this only pins the compiler)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pc = lane + 64 * h, row = pc >> 2, q = pc & 3;
                const u32x4 v = *reinterpret_cast<const u32x4*>(ws_ + row * 32 + q * 8);
                if (mc + row < mhi) *reinterpret_cast<u32x4*>(p.a_out + (size_t)(mc + row) * TK + wave * 32 + q * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int s = 0; s < WCHB / 16; ++s) {
            bf16x8 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = tr_frag(dz_s, 16 * s, wi_ * WN + i * 32, lane, PN);
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = tr_frag(a_s, 16 * s, wj_ * WK + j * 32, lane, PK);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage(buf ^ 1, mc + WCHB);
        __syncthreads();
        buf ^= 1;
    }

    if (FOLD && fold_owner && p.colsum_dz) {
        // the N_RPP threads that staged the same 8 columns combine through LDS (free after the loop's last barrier): one double atomic per column
        float* red = reinterpret_cast<float*>(hsmem);        // [N_RPP][TN]
        static_assert((size_t)N_RPP * TN * 4 <= 2 * (size_t)IMG_N, "column-sum scratch fits the dz images");
#pragma unroll
        for (int k = 0; k < 8; ++k) red[n_row * TN + n_col + k] = csum[k];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
            for (int r = 0; r < N_RPP; ++r) t += red[r * TN + tid];
            if (p.slab_cs) p.slab_cs[(size_t)sp * p.Cout + n0 + tid] = t;
            else atomicAdd(p.colsum_dz + n0 + tid, (double)t);
        }
    }

    const int fr = lane & 31;
    const int fh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wi_ * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int k = k0 + wj_ * WK + j * 32 + fr;
                if (FOLD && p.slab) p.slab[((size_t)sp * p.Cout + n) * p.K + k] = acc[i][j][r];
                else __hip_atomic_fetch_add(p.dw + (size_t)n * p.K + k, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
}

// ---- 3x3 convs with stride 1 along W: the three taps of a filter ROW on one pass over the operands ("tap reuse", r5) ----
// As a weight-gradient GEMM per tap, every (n tile, tap, channel tile) workgroup re-reads the same dz rows and a one-pixel-shifted copy of the same
// input rows: layer1's 64 -> 64 conv moves 18 x 0.27 GB through L2 for 155 GFLOP (0.36 PF, the L2 -> CU path at its ~10 TB/s).  Here a workgroup
// owns (n tile, filter row dh, channel tile): per chunk of WCHB output pixels -- which never straddles an image row (Wo % WCHB == 0) -- it loads
// the dz rows ONCE and the input rows of pixels wo0 - 1 .. wo0 + WCHB (circular) ONCE, WCHB + 2 LDS image rows, and the MFMA operand of tap dw is
// the same image read dw rows further down: three accumulator sets, a third of the operand traffic (190 FLOP per staged byte on 128x128 tiles
// against 128 for the 256x256 per-tap tile).  Same transposing fragment reads, same float atomics into dW[n][(dh * 3 + dw) * Cin + c].
template <int TN, int TK, int WCHB>
__global__ __launch_bounds__(256) void conv_wgrad_row3_bf16_kernel(WgradArgsH p)
{
    constexpr int THREADS = 256;
    constexpr int WN = TN / 2, WK = TK / 2;          // 2 x 2 waves
    constexpr int TI = WN / 32, TJ = WK / 32;
    static_assert(TI >= 1 && TJ >= 1, "wave tile");
    constexpr int N_TPR = TN / 8, K_TPR = TK / 8;
    constexpr int N_RPP = THREADS / N_TPR, K_RPP = THREADS / K_TPR;
    constexpr int N_PS = WCHB / N_RPP;
    constexpr int KROWS = WCHB + 2;
    constexpr int K_PS = (KROWS + K_RPP - 1) / K_RPP;
    static_assert(N_PS >= 1, "loader geometry");
    constexpr int PN = pitch_of(TN), PK = pitch_of(TK);
    constexpr int IMG_N = WCHB * PN, IMG_K = KROWS * PK;

    extern __shared__ __attribute__((aligned(16))) char hsmem[];
    char* s_dz0 = hsmem;                             // [2][IMG_N]
    char* s_a0 = hsmem + 2 * IMG_N;                  // [2][IMG_K]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wi_ = wave >> 1, wj_ = wave & 1;

    const int NT = p.Cout / TN;
    const int CT = p.Cin / TK;
    const int KT = 3 * CT;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % NT;
    bid /= NT;
    const int kt = bid % KT;
    const int sp = bid / KT;
    const int n0 = nt * TN;
    const int dh = kt / CT;
    const int c0 = (kt - dh * CT) * TK;
    const int mlo = sp * p.mchunk;
    const int mhi = (mlo + p.mchunk) < p.M ? (mlo + p.mchunk) : p.M;
    if (mlo >= mhi) return;

    const int hw_out = p.Ho * p.Wo;
    const int b_first = mlo / hw_out;
    const size_t img_elems = (size_t)p.Hi * p.Wi * p.xstride;
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.x + (size_t)b_first * img_elems), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_dz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.dz + (size_t)mlo * p.dzstride), 0, 0x7fffffff, 0x00020000);

    const int n_row = tid / N_TPR, n_col = (tid % N_TPR) * 8;
    const int k_row = tid / K_TPR, k_col = (tid % K_TPR) * 8;

    u32x4 rdz[N_PS], ra[K_PS];
    auto fetch = [&](int mc) {
#pragma unroll
        for (int q = 0; q < N_PS; ++q) {
            const int m = mc + n_row + q * N_RPP;
            rdz[q] = hbuf_load16(rsrc_dz, (unsigned)((size_t)(m - mlo) * p.dzstride + n0 + n_col) * 2u);
        }
        // the chunk's image row (wave-uniform): output pixels (b, ho, wo0 .. wo0 + WCHB - 1); input row hi = ho * sh - 1 + dh
        const int wo0 = mc % p.Wo;
        const int t = mc / p.Wo;
        const int ho = t % p.Ho;
        const int b = t / p.Ho - b_first;
        const int hi = ho * p.sh - p.ph + dh;
        const bool row_ok = (unsigned)hi < (unsigned)p.Hi;                   // zero padding above / below: the whole operand tile is zero
        const unsigned row_base = (unsigned)((b * p.Hi + hi) * p.Wi);
#pragma unroll
        for (int q = 0; q < K_PS; ++q) {
            const int r = k_row + q * K_RPP;                                 // LDS image row <-> input pixel wo0 - 1 + r (circular)
            int wi = wo0 - p.pw + r;
            wi = wi < 0 ? wi + p.Wi : wi;
            wi = wi >= p.Wi ? wi - p.Wi : wi;
            const unsigned off = ((row_base + (unsigned)wi) * (unsigned)p.xstride + (unsigned)(c0 + k_col)) * 2u;
            ra[q] = hbuf_load16(rsrc_x, (row_ok && r < KROWS) ? off : OOB);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < N_PS; ++q)
            *reinterpret_cast<u32x4*>(s_dz0 + buf * IMG_N + (n_row + q * N_RPP) * PN + n_col * 2) = rdz[q];
#pragma unroll
        for (int q = 0; q < K_PS; ++q) {
            const int r = k_row + q * K_RPP;
            if (r < KROWS) *reinterpret_cast<u32x4*>(s_a0 + buf * IMG_K + r * PK + k_col * 2) = ra[q];
        }
    };

    f32x16 acc[3][TI][TJ];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[d][i][j][r] = 0.f;

    fetch(mlo);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int mc = mlo; mc < mhi; mc += WCHB) {
        const bool more = mc + WCHB < mhi;
        if (more) fetch(mc + WCHB);
        const char* dz_s = s_dz0 + buf * IMG_N;
        const char* a_s = s_a0 + buf * IMG_K;
#pragma unroll
        for (int s = 0; s < WCHB / 16; ++s) {
            bf16x8 fa[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = tr_frag(dz_s, 16 * s, wi_ * WN + i * 32, lane, PN);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                bf16x8 fb[TJ];
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[j] = tr_frag(a_s, 16 * s + d, wj_ * WK + j * 32, lane, PK);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[d][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[d][i][j], 0, 0, 0);
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    const int fr = lane & 31;
    const int fh = lane >> 5;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wi_ * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int k = (dh * 3 + d) * p.Cin + c0 + wj_ * WK + j * 32 + fr;
                    __hip_atomic_fetch_add(p.dw + (size_t)n * p.K + k, acc[d][i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
}

template <int TN, int TK, int WCHB>
int launch_wgrad_row3(WgradArgsH a, hipStream_t s, long target)
{
    const int NT = a.Cout / TN, KT = 3 * (a.Cin / TK);
    long split = target / ((long)NT * KT);
    if (split < 1) split = 1;
    long mchunk = (a.M + split - 1) / split;
    if (mchunk < 8 * WCHB) mchunk = 8 * WCHB;
    mchunk = (mchunk + WCHB - 1) / WCHB * WCHB;
    split = (a.M + mchunk - 1) / mchunk;
    a.mchunk = (int)mchunk;
    const size_t lds = 2 * ((size_t)WCHB * pitch_of(TN) + (size_t)(WCHB + 2) * pitch_of(TK));
    auto kern = conv_wgrad_row3_bf16_kernel<TN, TK, WCHB>;
    if (lds > 65536) {
        static bool attr_done[64] = {};
        int dev = 0;
        HN_HIP(hipGetDevice(&dev));
        if (dev < 64 && !attr_done[dev]) {
            HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(NT * KT * split)), dim3(256), lds, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

// out[i] = sum_sp slab[sp][i] (float) and cs_out[c] = sum_sp slab_cs[sp][c] (double), in double precision and in a FIXED order: a workgroup
// owns 16 elements, its 16 split lanes add splits l, l + 16, ... (eight loads in flight each), the lanes are combined 0..15 through LDS.
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long n, int splits,
                                                                const float* __restrict__ slab_cs, double* __restrict__ cs_out, int C)
{
    __shared__ double red[256];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long nb = (n + 15) / 16;                       // element groups of the tiles; the column sums follow as further groups
    const long cb = cs_out ? (C + 15) / 16 : 0;
    for (long b = blockIdx.x; b < nb + cb; b += gridDim.x) {
        const bool cs = b >= nb;
        const long i = (cs ? b - nb : b) * 16 + il;
        const long lim = cs ? C : n;
        const float* src = cs ? slab_cs : slab;
        double t = 0.0;
        if (i < lim) {
            int sp = sl;
            for (; sp + 7 * 16 < splits; sp += 8 * 16) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(sp + 16 * q) * lim + i];
#pragma unroll
                for (int q = 0; q < 8; ++q) t += (double)v[q];
            }
            for (; sp < splits; sp += 16) t += (double)src[(size_t)sp * lim + i];
        }
        red[threadIdx.x] = t;
        __syncthreads();
        if (sl == 0 && i < lim) {
            for (int q = 1; q < 16; ++q) t += red[q * 16 + il];
            if (cs) cs_out[i] = t; else out[i] = (float)t;
        }
        __syncthreads();
    }
}

template <int TN, int TK, int WCHB, int NW = 4, bool STEM = false, bool FOLD = false, bool FUSEA = false>
int launch_wgrad_h_w(WgradArgsH a, hipStream_t s, long target)
{
    const int NT = a.Cout / TN, KT = a.K / TK;
    long split = target / ((long)NT * KT);
    if (split < 1) split = 1;
    long mchunk = (a.M + split - 1) / split;
    if (mchunk < 8 * WCHB) mchunk = 8 * WCHB;
    mchunk = (mchunk + WCHB - 1) / WCHB * WCHB;
    split = (a.M + mchunk - 1) / mchunk;
    a.mchunk = (int)mchunk;
    const size_t lds = 2 * (size_t)WCHB * (pitch_of(TN) + pitch_of(TK)) + (FUSEA ? 4096 : 0);      // FUSEA: + two wave-private 2 KiB output slabs
    auto kern = conv_wgrad_bf16_kernel<TN, TK, WCHB, NW, STEM, FOLD, FUSEA>;
    if (lds > 65536) {
        static bool attr_done[64] = {};   // per instantiation, per device
        int dev = 0;
        HN_HIP(hipGetDevice(&dev));
        if (dev < 64 && !attr_done[dev]) {
            HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done[dev] = true;
        }
    }
    if (FOLD && a.slab) {
        // slab[split][Cout][K] + slab_cs[split][Cout] must fit the caller's scratch (slab_cs = the float count handed in through its pointer difference)
        const size_t need = (size_t)split * a.Cout * a.K + (size_t)split * a.Cout;
        HN_REQUIRE(need <= (size_t)(a.slab_cs - a.slab), "wgrad bf16 fold: %zu floats of reduction scratch needed, %zu given", need, (size_t)(a.slab_cs - a.slab));
        a.slab_cs = a.slab + (size_t)split * a.Cout * a.K;
        // a workgroup whose m range is empty returns without storing: the ranges are [sp * mchunk, ..) with split = ceil(M / mchunk), none is empty
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(NT * KT * split)), dim3(NW * 64), lds, s, a);
    HN_LAUNCH_CHECK();
    if (FOLD && a.slab) {
        const long n = (long)a.Cout * a.K;
        long g = (n + 15) / 16 + (a.Cout + 15) / 16;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, a.slab, a.dw, n, (int)split, a.slab_cs, a.colsum_dz, a.Cout);
        HN_LAUNCH_CHECK();
    }
    return 0;
}

// Rows per chunk and the number of workgroups the m range is split into, measured on the training step at B = 32
// (tools/trace_timeline.py over prof_train_target.py, profiles/r2_wgrad_split_sweep.txt): every workgroup ends with TN x TK float
// atomics, so MORE workgroups is not better -- 512 (one resident round at 2 per CU with the 80 KB of the 64-row chunk) takes
// the 61 launches of the 128x128 tile from 9.7 ms (32 rows, 2048 workgroups) to 7.4 ms; 4096 workgroups cost 13.5 ms, 256 leave
// half the chip idle (10.3 ms).  The small 64x64 tile (layer1's 64-channel convs: tiny output, huge m) keeps 32 / 2048.
template <int TN, int TK>
int launch_wgrad_h(const WgradArgsH& a, hipStream_t s)
{
    static const char* env = getenv("HN_WGRAD_WCH");          // A/B switches (tools): rows per chunk, workgroup target
    static const char* envw = getenv("HN_WGRAD_H_WGS");
    const bool small = TN == 64 && TK == 64;
    const int wch = env ? atoi(env) : (small ? 32 : 64);
    const long target = envw ? atol(envw) : (small ? 2048 : 512);
    return wch == 64 ? launch_wgrad_h_w<TN, TK, 64>(a, s, target) : launch_wgrad_h_w<TN, TK, 32>(a, s, target);
}

}  // namespace

// x_h, dz_h: NHWC bf16; dw_packed: hn_packed_conv_weight_floats() floats, zeroed here (unless `prezeroed`) and accumulated by the kernel.
// Needs Cout %% 64 == 0 and Cin %% 64 == 0 (every conv of the network except the 7x7 stem and ghc0.3, which keep the f32
// kernel).
int hn_launch_conv_wgrad_bf16(const void* x_h, const void* dz_h, float* dw_packed, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                              int sh, int sw, hipStream_t s, int prezeroed, int xstride, int dzstride)
{
    HN_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "wgrad bf16: Cin=%d and Cout=%d must be multiples of 64", Cin, Cout);
    WgradArgsH a;
    a.x = reinterpret_cast<const u16*>(x_h); a.dz = reinterpret_cast<const u16*>(dz_h); a.dw = dw_packed;
    a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KW = KW; a.sh = sh; a.sw = sw; a.ph = KH / 2; a.pw = KW / 2;
    a.Ho = (Hi + 2 * a.ph - KH) / sh + 1;
    a.Wo = (Wi + 2 * a.pw - KW) / sw + 1;
    a.M = B * a.Ho * a.Wo;
    a.K = KH * KW * Cin;
    a.mchunk = 0;
    a.xstride = xstride ? xstride : Cin;
    a.dzstride = dzstride ? dzstride : Cout;
    a.bmask = nullptr; a.dz_wb = nullptr; a.colsum_dz = nullptr; a.slab = nullptr; a.slab_cs = nullptr; a.wa = nullptr; a.a_out = nullptr;
    HN_REQUIRE(a.xstride >= Cin && a.dzstride >= Cout && a.xstride % 8 == 0 && a.dzstride % 8 == 0, "wgrad bf16: bad strides");
    HN_REQUIRE((double)B * Hi * Wi * a.xstride * 2.0 < 2147483648.0 && (double)a.M * a.dzstride * 2.0 < 4294967296.0,
               "wgrad bf16: batch too large for 32-bit tile offsets");
    if (!prezeroed) HN_HIP(hipMemsetAsync(dw_packed, 0, (size_t)Cout * a.K * sizeof(float), s));
    {
        // 3x3 convs with stride 1 along W whose image rows are whole chunks: the tap-reuse kernel where it wins.  Per-dispatch durations of the
        // B = 64 step under both forms (tools/r5_row3.sh): layer1's 64 -> 64 convs 413 -> 291 us, ghc0.0 851 -> 709, ghc2.0 750 -> 703, ghc3.1 318 ->
        // 232, ghc3.0 1112 -> 729 (on 128x128 tiles: 192 accumulators, one workgroup per CU; the others on 128x64); the 256-channel 3x3 convs of
        // layer2..4 are a draw (210-220 us either way) and the short tails of the height-compression chains lose 15-40 % (few rows per workgroup, three
        // accumulator sets to flush), so they keep the per-tap tiles.  HN_WGRAD_ROW3 = 0: never, 1: wherever eligible (tests, A/B); read per call.
        const char* e3 = getenv("HN_WGRAD_ROW3");
        const bool eligible = KH == 3 && KW == 3 && sw == 1 && a.Wo % 32 == 0 && a.Wo == Wi && a.xstride == Cin && a.dzstride == Cout;
        const long cc = (long)Cout * Cin;
        const bool wins = (Cin == 64 && Cout == 64) || cc >= (1L << 20) || (cc >= (1L << 19) && a.M >= 32768) || (Cin == 256 && Cout == 128 && a.M >= (1L << 20));
        if (eligible && !(e3 && e3[0] == '0') && (wins || (e3 && e3[0] == '1'))) {
            const char* e3w = getenv("HN_WGRAD_ROW3_WGS");
            const char* e3t = getenv("HN_WGRAD_ROW3_TILE");                  // both channel counts % 128: 0 = 128x128, 1 = 128x64, 2 = 64x128
            const long target = e3w ? atol(e3w) : 1024;
            const int tile = e3t ? atoi(e3t) : (Cin >= 2048 ? 0 : 1);
            if (Cout % 128 == 0 && Cin % 128 == 0 && tile == 0) return launch_wgrad_row3<128, 128, 32>(a, s, target);
            if (Cout % 128 == 0 && Cin % 128 == 0 && tile == 2) return launch_wgrad_row3<64, 128, 32>(a, s, target);
            if (Cout % 128 == 0) return launch_wgrad_row3<128, 64, 32>(a, s, target);
            if (Cin % 128 == 0) return launch_wgrad_row3<64, 128, 32>(a, s, target);
            return launch_wgrad_row3<64, 64, 32>(a, s, 2 * target);
        }
    }
    {
        // 8-wave 256-wide tiles where both channel counts allow them (layer3 / layer4, the deep height-compression convs): one
        // workgroup per CU, m split into ~256 workgroups.  HN_WGRAD_W8 = 0 disables, 1 = per-shape choice between 256x256 and 256x128 (default), 2 = 256x128, 3 = 128x256, 4 = 256x256 then 256x128, 5 = 256x256 only
        static const char* e8 = getenv("HN_WGRAD_W8");
        const int w8 = e8 ? atoi(e8) : 1;
        static const char* envw = getenv("HN_WGRAD_W8_WGS");
        const long target = envw ? atol(envw) : 256;
        if (w8 == 1 && Cout % 256 == 0 && Cin % 128 == 0) {
            // per-dispatch durations of the B = 64 step under both shapes (tools/r4_run24.sh, profiles/r4_wgrad_tile_shapes.txt): the 1x1 convs
            // and the LSTM input GEMMs are 10-20 % faster on 256x128 (twice the workgroups for the same m split), the 3x3 convs 10-60 %
            // faster on 256x256 -- except where 256x256 gives a few tiles more than one resident round (ghc3.0: 288 tiles on 256 CUs)
            const long t256 = Cin % 256 == 0 ? (long)(Cout / 256) * (a.K / 256) : 0;
            const bool ragged = t256 > 256 && (double)t256 / (double)(((t256 + 255) / 256) * 256) < 0.6;
            if (t256 == 0 || KH == 1 || ragged) return launch_wgrad_h_w<256, 128, 64, 8>(a, s, target);
            return launch_wgrad_h_w<256, 256, 64, 8>(a, s, target);
        }
        if (w8 == 5 && Cout % 256 == 0 && Cin % 256 == 0) return launch_wgrad_h_w<256, 256, 64, 8>(a, s, target);      // (the round-3 rule, for A/B runs)
        if (w8 == 4 && Cout % 256 == 0 && Cin % 256 == 0) return launch_wgrad_h_w<256, 256, 64, 8>(a, s, target);
        if ((w8 == 2 || w8 == 4) && Cout % 256 == 0 && Cin % 128 == 0) return launch_wgrad_h_w<256, 128, 64, 8>(a, s, target);
        if (w8 == 3 && Cout % 128 == 0 && Cin % 256 == 0) return launch_wgrad_h_w<128, 256, 64, 8>(a, s, target);
    }
    if (Cout % 128 == 0 && Cin % 128 == 0) return launch_wgrad_h<128, 128>(a, s);
    if (Cout % 128 == 0) return launch_wgrad_h<128, 64>(a, s);
    if (Cin % 128 == 0) return launch_wgrad_h<64, 128>(a, s);
    return launch_wgrad_h<64, 64>(a, s);
}

// P[n][k] = sum_m g[m][n] * a[m][k] for the BatchNorm-folded adjoint of a 1x1 / stride-1 conv (train.hip: bn_fold_*): the weight-gradient
// GEMM above with g = dy * ReLU mask formed on load.  a_h [M][Cin] bf16, dy_h [M][Cout] bf16 (masked IN PLACE when bmask is given: the
// rows leave as g), p_out [Cout][Cin] floats (accumulated; zeroed here unless prezeroed), colsum [Cout] doubles (accumulated: sum_m g[m][n]).
// bmask == null: plain operands (the Gram matrix a^T a of an activation with its column sums: a_h == dy_h, Cin == Cout).
int hn_launch_conv_wgrad_bf16_fold(const void* a_h, void* dy_h, float* p_out, long M, int Cin, int Cout, const unsigned char* bmask,
                                   double* colsum, hipStream_t s, int prezeroed, float* slab, size_t slab_floats, const void* wa, void* a_out,
                                   int write_back)
{
    HN_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "wgrad bf16 fold: Cin=%d and Cout=%d must be multiples of 64", Cin, Cout);
    HN_REQUIRE(M > 0 && (double)M * Cin * 2.0 < 2147483648.0 && (double)M * Cout * 2.0 < 4294967296.0, "wgrad bf16 fold: batch too large for 32-bit tile offsets");
    WgradArgsH a;
    a.x = reinterpret_cast<const u16*>(a_h); a.dz = reinterpret_cast<const u16*>(dy_h); a.dw = p_out;
    a.Hi = 1; a.Wi = (int)M; a.Cin = Cin; a.Cout = Cout; a.KW = 1; a.sh = 1; a.sw = 1; a.ph = 0; a.pw = 0;
    a.Ho = 1; a.Wo = (int)M;
    a.M = (int)M;
    a.K = Cin;
    a.mchunk = 0;
    a.xstride = Cin; a.dzstride = Cout;
    a.bmask = bmask; a.dz_wb = (bmask && write_back) ? reinterpret_cast<u16*>(dy_h) : nullptr; a.colsum_dz = colsum;
    // wa / a_out (optional, Cout == 256 and Cin == 64 only: layer1): the conv-A part of the folded data gradient in the same pass (FUSEA)
    a.wa = reinterpret_cast<const u16*>(wa); a.a_out = reinterpret_cast<u16*>(a_out);
    HN_REQUIRE((wa == nullptr) == (a_out == nullptr) && (!wa || (Cout == 256 && Cin == 64)), "wgrad bf16 fold: the fused conv A is the 256 x 64 shape's");
    // slab != null (slab_floats of scratch): the reproducible form -- partial tiles stored per m split and added in order (p_out and colsum are
    // then OVERWRITTEN, not accumulated); the launcher places slab_cs behind the tiles
    a.slab = slab; a.slab_cs = slab ? slab + slab_floats : nullptr;
    // (colsum == null with a slab: the tiles only -- block 0's second unit takes the column sums of the first)
    {   // timing experiments only (results are wrong): 1 = no write-back, 2 = no mask loads, 4 = no column sums
        static const char* dbg = getenv("HN_FOLD_DEBUG");
        const int f = dbg ? atoi(dbg) : 0;
        if (f & 1) a.dz_wb = nullptr;
        if (f & 2) a.bmask = nullptr;
        if (f & 4) a.colsum_dz = nullptr;
    }
    if (!prezeroed && !slab) HN_HIP(hipMemsetAsync(p_out, 0, (size_t)Cout * Cin * sizeof(float), s));
    // These GEMMs are HBM-bound (K <= 512 per output channel, and the masked rows are written back), so they want MANY small workgroups
    // in flight, not big tiles: the 8-wave 256-wide tile (one workgroup per CU, one chunk of loads in flight) took 300 us where the
    // bytes need 120 (layer3 at B = 64); 4-wave tiles with 32-row chunks run 4 workgroups per CU.  HN_FOLD_WCH / HN_FOLD_WGS: A/B switches.
    static const char* envc = getenv("HN_FOLD_WCH");
    static const char* envw = getenv("HN_FOLD_WGS");
    const int wch = envc ? atoi(envc) : 32;
    const long target = envw ? atol(envw) : 1024;
    if (wa) return launch_wgrad_h_w<256, 64, 32, 4, false, true, true>(a, s, target);
    if (Cout % 128 == 0 && Cin % 128 == 0)
        return wch == 64 ? launch_wgrad_h_w<128, 128, 64, 4, false, true>(a, s, target) : launch_wgrad_h_w<128, 128, 32, 4, false, true>(a, s, target);
    if (Cout % 128 == 0)
        return wch == 64 ? launch_wgrad_h_w<128, 64, 64, 4, false, true>(a, s, target) : launch_wgrad_h_w<128, 64, 32, 4, false, true>(a, s, target);
    if (Cin % 128 == 0)
        return wch == 64 ? launch_wgrad_h_w<64, 128, 64, 4, false, true>(a, s, target) : launch_wgrad_h_w<64, 128, 32, 4, false, true>(a, s, target);
    return wch == 64 ? launch_wgrad_h_w<64, 64, 64, 4, false, true>(a, s, 2 * target) : launch_wgrad_h_w<64, 64, 32, 4, false, true>(a, s, 2 * target);
}

// The stem's weight gradient on the bf16 matrix cores (train_precision bf16): x4_h = the NHWC4 bf16 input the bf16 forward read
// ([B][Hi][Wi][4]), dz_h = bf16 gradient w.r.t. the stem conv output ([B][Hi/2][Wi/2][64]), dw_packed = [64][256] floats in the
// stem's packed K order (8 filter rows x 8 window pixels x 4 channels; tap dw sits at window pixel dw + 1: un-pack with
// hn_launch_unpack_conv(dw_packed + 4, ..., packed_rows = 8)), zeroed here unless `prezeroed`.  The float32 form (conv_wgrad_kernel<64, 64, true>) read 2.1 GB of float32 dz + 1 GB of float32 input per
// step at B = 64 and took 2.6 ms.
int hn_launch_stem_wgrad_bf16(const void* x4_h, const void* dz_h, float* dw_packed, int B, int Hi, int Wi, hipStream_t s, int prezeroed)
{
    HN_REQUIRE(Hi % 2 == 0 && Wi % 2 == 0, "stem wgrad bf16: even image sizes");
    WgradArgsH a;
    a.x = reinterpret_cast<const u16*>(x4_h); a.dz = reinterpret_cast<const u16*>(dz_h); a.dw = dw_packed;
    a.Hi = Hi; a.Wi = Wi; a.Cin = 4; a.Cout = 64; a.KW = 7; a.sh = 2; a.sw = 2; a.ph = 3; a.pw = 3;
    a.Ho = Hi / 2; a.Wo = Wi / 2;
    a.M = B * a.Ho * a.Wo;
    a.K = 256;
    a.mchunk = 0;
    a.xstride = 4; a.dzstride = 64;
    a.bmask = nullptr; a.dz_wb = nullptr; a.colsum_dz = nullptr; a.slab = nullptr; a.slab_cs = nullptr; a.wa = nullptr; a.a_out = nullptr;
    HN_REQUIRE((double)B * Hi * Wi * 8.0 < 2147483648.0 && (double)a.M * 128.0 < 4294967296.0, "stem wgrad bf16: batch too large for 32-bit tile offsets");
    if (!prezeroed) HN_HIP(hipMemsetAsync(dw_packed, 0, (size_t)64 * 256 * sizeof(float), s));
    return launch_wgrad_h_w<64, 256, 32, 4, true>(a, s, 512);
}
