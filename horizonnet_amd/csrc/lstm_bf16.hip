// Persistent bidirectional LSTM layer for the bf16 inference mode (hidden 512, gates i,f,g,o): the recurrent matmul on the
// bf16 matrix cores with a bf16 h_{t-1} hand-off, organised so that the per-step exchange is SMALL.
//
// Replaces the recurrent half of `nn.LSTM` (reference model.py:222-227,263-264) in hn_forward_bf16; the input projection
// gx = x @ W_ih^T + b is one bf16 GEMM launch before it.  The float32 kernel (lstm.hip) spreads one direction over 128
// workgroups, so every step is a 128-party hand-off of 64 KB (4.1 us per step, 2.1 ms per forward = 20 % of the bf16
// forward).  Here the work is partitioned by BATCH first:
//
//   * 32 groups of 8 workgroups (one per CU, 256 CUs): group = (direction, pair of panoramas).  The 8 members split the
//     512 hidden units (64 each = 256 gate rows); each member keeps its W_hh slice (256 x 512 bf16 = 256 KB) in REGISTERS for
//     the whole sequence as B fragments of v_mfma_f32_16x16x32_bf16 (wave w: 16 units x 4 gates = 64 fragments = 256 VGPRs).
//     The weights are replicated across the 16 groups of a direction -- registers are what an idle-by-construction
//     latency-bound kernel has plenty of.
//   * per step a member needs h_{t-1} of ITS TWO panoramas only: 2 x 512 bf16 = 2 KB, published by the 8 members as 512
//     eight-byte {tag, value} granules (cdna guide G16 form R2: the data is the flag -- one write-through store per
//     granule, no counter, no fence, no drain on the critical path; tag = step + 1, two slots by step parity).  One wave
//     sweeps the 512 granules until every tag matches, spreads h into LDS (every wave a quarter: 2 coalesced 8-byte sc1 loads per
//     lane), and all
//     four waves run 64 MFMAs each (A = h rows, 2 of the 16 MFMA rows live).
//   * gates, cell state (registers of the 16 lanes that own the live accumulator rows) and the float32 layer output y are
//     float32; h is rounded to bf16 only where it re-enters the matrix cores (recurrence, next layer's input GEMM).
//
// Placement independent (any workgroup -> CU / XCD map is correct), every spin is bounded and reports through the sticky
// status word of the sync scratch instead of hanging.
#include "hn_common.h"

namespace {

constexpr int LH = 512;                       // hidden size
constexpr int GRP = 8;                        // workgroups per group
constexpr int UPC = LH / GRP;                 // 64 hidden units per workgroup
constexpr int BPG = 2;                        // panoramas per group
constexpr int NGRP = 32;                      // 2 directions x 16 panorama pairs
constexpr unsigned SPIN_LIMIT_H = 1u << 22;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// Gate nonlinearities on the step's critical path (16 lanes per wave, nothing to overlap them with): hardware exp2 / rcp
// (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm's expf / tanhf / IEEE division -- a few 1e-7 of difference on values that
// are rounded to bf16 (2^-9) before they re-enter the recurrence.  Saturation is exact: exp -> inf gives rcp -> 0.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp(2.0f * x)); }
__device__ __forceinline__ unsigned bf16_rn(float x)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(0.f));
    return r & 0xffffu;
}

__global__ __launch_bounds__(256) void lstm_layer_bf16_kernel(const float* __restrict__ gx, const u16* __restrict__ whh_f,
                                                              const u16* __restrict__ whh_r, float* __restrict__ y,
                                                              u16* __restrict__ y_h, u64* xch, int T, int B, int b0, int bc,
                                                              unsigned* sync, float* __restrict__ save)
{
    __shared__ __attribute__((aligned(16))) u16 hs[2][BPG][LH];      // h_{t-1} of the group's two panoramas, by step parity

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // blockIdx b runs on XCD b % 8 (observed, speed only): keep a group's 8 members on one XCD
    const int bid = blockIdx.x;
    const int grp = (bid & 7) | ((bid >> 6) << 3);     // 0..31
    const int mem = (bid >> 3) & 7;                    // 0..7
    const int dir = grp & 1;
    const int pair = grp >> 1;                         // 0..15
    const int bfirst = pair * BPG;
    if (bfirst >= bc) return;                          // the whole group leaves together
    const int nvalid = (bc - bfirst) < BPG ? (bc - bfirst) : BPG;
    const u16* whh = dir ? whh_r : whh_f;
    const int col = lane & 15;                         // unit within the wave's 16 / MFMA column
    const int kb = lane >> 4;                          // 8-wide k block within a 32-deep MFMA
    const int u0 = mem * UPC + wave * 16;              // first hidden unit of this wave
    u64* slots = xch + (size_t)grp * LH;               // [parity][NGRP][LH] granules

    // ---- W_hh fragments: B[k][n], n = unit col of gate g -> row g*512 + u0 + col; lane holds k = ks*32 + kb*8 .. +8 ----
    u32x4 wf[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const u16* wr = whh + (size_t)(g * LH + u0 + col) * LH + kb * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[g][ks] = *reinterpret_cast<const u32x4*>(wr + ks * 32);
    }

    // accumulator rows: lane l, register r -> MFMA row (l >> 4) * 4 + r; the live rows 0 / 1 (the two panoramas) sit in
    // registers 0 / 1 of lanes 0..15 -- those 16 lanes are the gate lanes of the wave (unit u0 + col)
    const bool gate_lane = lane < 16;
    float c_state[BPG] = {0.f, 0.f};
    bool failed = false;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;

        float gxv[BPG][4];
        if (gate_lane) {
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                const int bb = b0 + bfirst + (r < nvalid ? r : 0);
                const float* gp = gx + ((size_t)t * B + bb) * 4096 + dir * 2048 + u0 + col;
#pragma unroll
                for (int g = 0; g < 4; ++g) gxv[r][g] = gp[g * LH];
            }
        }

        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            const int par = step & 1;
            {
                // every wave sweeps ITS quarter of the group's 512 granules of step-1 (tag == step) until all have landed, then
                // spreads that quarter of h into LDS: 2 coalesced 8-byte sc1 loads and 4 LDS stores per lane and sweep
                const u64* src = slots + (size_t)((step - 1) & 1) * NGRP * LH + wave * 128;
                unsigned v[2];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const u64 x = __hip_atomic_load(src + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v[k] = (unsigned)x;
                        ok = ok && (unsigned)(x >> 32) == (unsigned)step;
                    }
                    if (__all(ok) || failed) break;
                    if (++spins > SPIN_LIMIT_H) {
                        __hip_atomic_store(sync + HN_STATUS_WORD, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        failed = true;
                        break;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    hs[par][0][wave * 128 + k * 64 + lane] = (u16)(v[k] & 0xffffu);
                    hs[par][1][wave * 128 + k * 64 + lane] = (u16)(v[k] >> 16);
                }
            }
            __syncthreads();
            // A fragments: row = lane % 16 -> panorama (rows >= 2 are dead: they read row 0), k = ks*32 + kb*8 .. +8
            const u16* hrow = &hs[par][col < BPG ? col : 0][kb * 8];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(hrow + ks * 32);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wf[g][ks]),
                                                                     acc[g], 0, 0, 0);
            }
        }

        if (gate_lane) {
            unsigned packed = 0;
            float hval[BPG], gsave[BPG][4];
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                const float ig = sigm(gxv[r][0] + acc[0][r]);
                const float fg = sigm(gxv[r][1] + acc[1][r]);
                const float gg = tanh_fast(gxv[r][2] + acc[2][r]);
                const float og = sigm(gxv[r][3] + acc[3][r]);
                c_state[r] = fg * c_state[r] + ig * gg;
                hval[r] = og * tanh_fast(c_state[r]);
                gsave[r][0] = ig; gsave[r][1] = fg; gsave[r][2] = gg; gsave[r][3] = og;
                if (r < nvalid) packed |= bf16_rn(hval[r]) << (16 * r);
            }
            // publish FIRST (the peers' next step waits for it): ONE write-through 8-byte store per unit, tag = step + 1 (never 0)
            u64* dst = slots + (size_t)(step & 1) * NGRP * LH + u0 + col;
            __hip_atomic_store(dst, ((u64)(unsigned)(step + 1) << 32) | packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                if (r < nvalid) {
                    const size_t o = ((size_t)t * B + b0 + bfirst + r) * 1024 + dir * LH + u0 + col;
                    if (save) {      // training: post-activation gates + cell state for the adjoint, [t][b][dir][5][512]
                        float* sv = save + (((size_t)t * B + b0 + bfirst + r) * 2 + dir) * 5 * LH + u0 + col;
                        sv[0] = gsave[r][0]; sv[LH] = gsave[r][1]; sv[2 * LH] = gsave[r][2]; sv[3 * LH] = gsave[r][3]; sv[4 * LH] = c_state[r];
                    }
                    y[o] = hval[r];
                    if (y_h) y_h[o] = (u16)bf16_rn(hval[r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// WIDE form of the same recurrence for the pipelined forward (hn_forward_bf16_submit, engine_bf16.hip): the kernel above keeps
// 2 of the 16 MFMA rows live and therefore occupies all 256 compute units for a latency-bound chain of 256 steps.  Here a
// group owns one direction of BW = 16 (or 8) panoramas, so every MFMA row is live and a batch of 32 needs 4 groups x 8
// workgroups = 32 compute units (64 at BW = 8); the other 224 run the NEXT batch's HBM-bound stem / layer1 convolutions on the
// caller's stream while the recurrence walks its 2 x 256 steps (the point of the pipelined entry; reference model.py:263-264
// is a serial nn.LSTM call after the convolutions of the same batch).
//   * the per-step exchange grows to BW x 512 bf16 = 16 KB per group; with 8-byte {tag, 2 x bf16} granules every wave swept
//     8 KB in 16 loads per lane and the step took 6.8 us (measured, round 3).  The exchange buffer is therefore the layer's
//     bf16 OUTPUT y_h itself, one slot per time step, pre-filled with the bf16 pattern 0xFFFF (a NaN no finite h rounds to):
//     "every element differs from the sentinel" is the arrival test, there is no tag, no parity and no slot reuse.  A member
//     publishes with ONE 8-byte write-through store per lane (4 consecutive units of one panorama, built from the lanes' four
//     accumulator rows with two DPP quad swaps); a wave sweeps BW / 4 rows with one 16-byte sc1 load per lane and row.
//     Element granularity makes torn 16-byte reads harmless.  Rows of absent panoramas are neither published nor swept.
//   * all 64 lanes of a wave are gate lanes (4 accumulator rows each); the arithmetic is the narrow kernel's, so the two
//     kernels agree bit for bit (tested).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int WIDE_MAX_GROUPS = 8;            // per launch: 2 directions x 4 sets (64 panoramas at BW = 16)
constexpr int WIDE_HP = LH + 8;               // LDS row pitch (bf16): rows 4 banks apart -> conflict-free 16-byte fragment reads

__device__ __forceinline__ unsigned quad_swap1(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false); }   // lane ^ 1
__device__ __forceinline__ unsigned quad_swap2(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false); }   // lane ^ 2

template <int BW>
__global__ __launch_bounds__(256) void lstm_layer_bf16_wide_kernel(const float* __restrict__ gx, const u16* __restrict__ whh_f,
                                                                   const u16* __restrict__ whh_r, float* __restrict__ y, u16* y_h,
                                                                   int T, int B, int b0, int bc, int ngroups, int xcds_per_group,
                                                                   unsigned* sync, float* __restrict__ save)
{
    static_assert(BW == 8 || BW == 16, "rows per group");
    constexpr int RPW = BW / 4;                                       // rows swept per wave
    __shared__ __attribute__((aligned(16))) u16 hs[2][BW][WIDE_HP];   // h_{t-1} of the group's panoramas, by step parity

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // blockIdx b runs on XCD b % 8 (observed, speed only): a group's 8 members on ONE XCD (xcds_per_group 1: groups 0..7 on
    // XCDs 0..7) or 4 + 4 on two neighbouring XCDs (xcds_per_group 2: 4 compute units of every XCD at 4 groups)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    int grp, mem;
    if (xcds_per_group == 1) { grp = xcd + 8 * (slot >> 3); mem = slot & 7; }
    else { grp = (xcd >> 1) + 4 * (slot >> 2); mem = ((xcd & 1) << 2) | (slot & 3); }
    if (grp >= ngroups) return;
    const int dir = grp & 1;
    const int bfirst = (grp >> 1) * BW;
    if (bfirst >= bc) return;                          // the whole group leaves together
    const int nvalid = (bc - bfirst) < BW ? (bc - bfirst) : BW;
    const u16* whh = dir ? whh_r : whh_f;
    const int col = lane & 15;
    const int kb = lane >> 4;
    const int u0 = mem * UPC + wave * 16;
    const size_t yh_bytes = (size_t)T * B * 1024 * sizeof(u16);
    const __amdgpu_buffer_rsrc_t rsrc_h =              // h_{t-1} is read with sc1 (L1-bypassing) buffer loads: aux bit 4
        __builtin_amdgcn_make_buffer_rsrc(y_h, 0, (unsigned)(yh_bytes > 0xfffffff0u ? 0xfffffff0u : yh_bytes), 0x00020000);


    u32x4 wf[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const u16* wr = whh + (size_t)(g * LH + u0 + col) * LH + kb * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[g][ks] = *reinterpret_cast<const u32x4*>(wr + ks * 32);
    }

    // accumulator rows of this lane: kb * 4 + r, r = 0..3 (unit u0 + col); after the two quad swaps the lane publishes 4
    // consecutive units (the quad's) of row kb * 4 + pub_r
    const int row0 = kb * 4;
    const bool gate_lane = row0 < BW;
    const int j = col & 3;
    const int pub_row = row0 + ((j & 1) ? 2 : 0) + (j >> 1);
    float c_state[4] = {0.f, 0.f, 0.f, 0.f};
    bool failed = false;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;

        float gxv[4][4];
        if (gate_lane) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                const int bb = b0 + bfirst + (row < nvalid ? row : nvalid - 1);
                const float* gp = gx + ((size_t)t * B + bb) * 4096 + dir * 2048 + u0 + col;
#pragma unroll
                for (int g = 0; g < 4; ++g) gxv[r][g] = gp[g * LH];
            }
        }

        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            const int par = step & 1;
            {
                const int tprev = dir ? t + 1 : t - 1;
                u32x4 v[RPW];
                unsigned voff[RPW];
#pragma unroll
                for (int k = 0; k < RPW; ++k) {          // a row of an absent panorama re-reads the group's row 0 (its MFMA row is dead)
                    const int row = wave * RPW + k < nvalid ? wave * RPW + k : 0;
                    voff[k] = (unsigned)((((size_t)tprev * B + b0 + bfirst + row) * 1024 + dir * LH) * sizeof(u16)) + lane * 16;
                }
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int k = 0; k < RPW; ++k)
                        v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, voff[k], 0, /*sc1*/ 16));
                    u16x8 m8 = __builtin_bit_cast(u16x8, v[0]);      // running packed maximum: 0xFFFF somewhere <=> an element is still the sentinel
#pragma unroll
                    for (int k = 1; k < RPW; ++k) m8 = __builtin_elementwise_max(m8, __builtin_bit_cast(u16x8, v[k]));
                    const u16x4 m4 = __builtin_elementwise_max(m8.lo, m8.hi);
                    const u16x2 m2 = __builtin_elementwise_max(m4.lo, m4.hi);
                    const bool ok = (m2[0] != 0xffff) & (m2[1] != 0xffff);
                    if (__all(ok) || failed) break;
                    asm volatile("" ::: "memory");                       // the next sweep re-reads memory
                    if (++spins > SPIN_LIMIT_H) {
                        __hip_atomic_store(sync + HN_STATUS_WORD, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        failed = true;
                        break;
                    }
                }
#pragma unroll
                for (int k = 0; k < RPW; ++k) *reinterpret_cast<u32x4*>(&hs[par][wave * RPW + k][lane * 8]) = v[k];
            }
            __syncthreads();
            // A fragments: row = lane % 16 -> panorama (BW = 8: rows 8..15 are dead and read row 0), k = ks*32 + kb*8 .. +8
            const u16* hrow = &hs[par][col < BW ? col : 0][kb * 8];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(hrow + ks * 32);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wf[g][ks]),
                                                                     acc[g], 0, 0, 0);
            }
        }

        if (gate_lane) {
            float hval[4], gsv[4][4];
            unsigned hb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigm(gxv[r][0] + acc[0][r]);
                const float fg = sigm(gxv[r][1] + acc[1][r]);
                const float gg = tanh_fast(gxv[r][2] + acc[2][r]);
                const float og = sigm(gxv[r][3] + acc[3][r]);
                c_state[r] = fg * c_state[r] + ig * gg;
                hval[r] = og * tanh_fast(c_state[r]);
                hb[r] = bf16_rn(hval[r]);
                gsv[r][0] = ig; gsv[r][1] = fg; gsv[r][2] = gg; gsv[r][3] = og;
            }
            // publish FIRST: 4 rows x 1 unit per lane -> 1 row x 4 units per lane (a 4 x 4 transpose inside the quad of unit lanes)
            const unsigned lo = hb[0] | (hb[1] << 16), hi = hb[2] | (hb[3] << 16);
            const bool odd = j & 1;
            const unsigned mine = odd ? hi : lo;                              // even lanes keep rows 0,1; odd lanes rows 2,3
            const unsigned recv = quad_swap1(odd ? lo : hi);
            const unsigned ev = odd ? recv : mine, od = odd ? mine : recv;    // even / odd unit of the pair, two rows each
            const unsigned p0 = (ev & 0xffffu) | (od << 16);                  // first of the lane's two rows, units (2p, 2p+1)
            const unsigned p1 = (ev >> 16) | (od & 0xffff0000u);              // second row
            const bool up = j >> 1;                                           // unit pair 1 of the quad keeps the second row
            const unsigned recv2 = quad_swap2(up ? p0 : p1);
            const u64 word = up ? ((u64)recv2 | ((u64)p1 << 32)) : ((u64)p0 | ((u64)recv2 << 32));
            if (pub_row < nvalid)
                __hip_atomic_store(reinterpret_cast<u64*>(y_h + ((size_t)t * B + b0 + bfirst + pub_row) * 1024 + dir * LH + u0 + (col & ~3)), word,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < nvalid) {
                    y[((size_t)t * B + b0 + bfirst + row0 + r) * 1024 + dir * LH + u0 + col] = hval[r];
                    if (save) {      // training: post-activation gates + cell state for the adjoint, [t][b][dir][5][512]
                        float* sv = save + (((size_t)t * B + b0 + bfirst + row0 + r) * 2 + dir) * 5 * LH + u0 + col;
                        sv[0] = gsv[r][0]; sv[LH] = gsv[r][1]; sv[2 * LH] = gsv[r][2]; sv[3 * LH] = gsv[r][3]; sv[4 * LH] = c_state[r];
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The adjoint of the recurrence in the same shape (train_precision bf16; autograd of nn.LSTM under train.py:273-278):
//   dh_t = dy_t + W_hh^T dg_{t+1};  dc = dc_rec + dh o (1 - tanh^2 c);  dg_t = gate adjoints;  dc_rec = dc f
// The recurrence is independent per panorama, so again group = (direction, pair of panoramas), 8 members, member = 64
// hidden units: it computes the gate adjoints of ITS units (dh of its own units is all that needs) and the W_hh^T product
// for ITS units' dh -- which needs the dg of ALL 2048 gate rows of the group's two panoramas.  Those travel as 2048
// {tag, dg(pan 0) | dg(pan 1) << 16} bf16 granules per step (each wave sweeps 512), W_hh^T lives in registers as 64 B
// fragments per lane (k = gate row), and dg itself is written in float32 for the weight-gradient GEMMs that follow.
// Replaces 2 launches per time step (lstm_bwd_gates + lstm_bwd_dh: 1024 launches per backward pass) by one per layer.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int GR = 4 * LH;                    // gate rows

__global__ __launch_bounds__(256) void lstm_layer_bwd_bf16_kernel(const float* __restrict__ saved, const float* __restrict__ dy,
                                                                  const u16* __restrict__ whhT_f, const u16* __restrict__ whhT_r,
                                                                  float* __restrict__ dgx, u64* xch, int T, int B, int b0, int bc,
                                                                  unsigned* sync)
{
    __shared__ __attribute__((aligned(16))) u16 dgs[2][BPG][GR];      // dg of the previous step, both panoramas, by step parity

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int bid = blockIdx.x;
    const int grp = (bid & 7) | ((bid >> 6) << 3);     // a group's 8 members share an XCD (speed only)
    const int mem = (bid >> 3) & 7;
    const int dir = grp & 1;
    const int pair = grp >> 1;
    const int bfirst = pair * BPG;
    if (bfirst >= bc) return;
    const int nvalid = (bc - bfirst) < BPG ? (bc - bfirst) : BPG;
    const u16* whhT = dir ? whhT_r : whhT_f;           // [512 units][2048 gate rows]
    const int col = lane & 15;
    const int kb = lane >> 4;
    const int u0 = mem * UPC + wave * 16;
    const int unit = u0 + col;
    u64* slots = xch + (size_t)grp * GR;               // [parity][NGRP][GR] granules

    // B fragments of W_hh^T: column = unit, k = gate row ks*32 + kb*8 .. +8
    u32x4 wf[64];
    {
        const u16* wr = whhT + (size_t)unit * GR + kb * 8;
#pragma unroll
        for (int ks = 0; ks < 64; ++ks) wf[ks] = *reinterpret_cast<const u32x4*>(wr + ks * 32);
    }

    const bool gate_lane = lane < 16;
    float dc_rec[BPG] = {0.f, 0.f};
    bool failed = false;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;       // each direction's own time order, backwards
        const int tprev = dir ? t + 1 : t - 1;         // the step that ran BEFORE t in that direction's forward pass

        float sv[BPG][5], cprev[BPG], dyv[BPG];
        if (gate_lane) {
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                const int bb = b0 + bfirst + (r < nvalid ? r : 0);
                const float* sp = saved + (((size_t)t * B + bb) * 2 + dir) * 5 * LH + unit;
#pragma unroll
                for (int k = 0; k < 5; ++k) sv[r][k] = sp[k * LH];
                cprev[r] = (tprev >= 0 && tprev < T) ? saved[((((size_t)tprev * B + bb) * 2 + dir) * 5 + 4) * LH + unit] : 0.f;
                dyv[r] = dy[((size_t)t * B + bb) * 1024 + dir * LH + unit];
            }
        }

        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            const int par = step & 1;
            {
                // every wave sweeps a quarter of the group's 2048 granules of step-1 (tag == step), then spreads them into LDS
                const u64* src = slots + (size_t)((step - 1) & 1) * NGRP * GR + wave * 512;
                unsigned v[8];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const u64 x = __hip_atomic_load(src + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v[k] = (unsigned)x;
                        ok = ok && (unsigned)(x >> 32) == (unsigned)step;
                    }
                    if (__all(ok) || failed) break;
                    if (++spins > SPIN_LIMIT_H) {
                        __hip_atomic_store(sync + HN_STATUS_WORD, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        failed = true;
                        break;
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dgs[par][0][wave * 512 + k * 64 + lane] = (u16)(v[k] & 0xffffu);
                    dgs[par][1][wave * 512 + k * 64 + lane] = (u16)(v[k] >> 16);
                }
            }
            __syncthreads();
            const u16* arow = &dgs[par][col < BPG ? col : 0][kb * 8];      // A rows >= 2 are dead
#pragma unroll
            for (int ks = 0; ks < 64; ++ks) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(arow + ks * 32);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wf[ks]), acc, 0, 0, 0);
            }
        }

        if (gate_lane) {
            unsigned packed[4] = {0u, 0u, 0u, 0u};
            float g[BPG][4];
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                const float ig = sv[r][0], fg = sv[r][1], gg = sv[r][2], og = sv[r][3];
                const float dh = dyv[r] + acc[r];
                const float tc = tanh_fast(sv[r][4]);
                const float dc = dc_rec[r] + dh * og * (1.f - tc * tc);
                g[r][0] = dc * gg * ig * (1.f - ig);
                g[r][1] = dc * cprev[r] * fg * (1.f - fg);
                g[r][2] = dc * ig * (1.f - gg * gg);
                g[r][3] = dh * tc * og * (1.f - og);
                dc_rec[r] = dc * fg;
                if (r < nvalid) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) packed[k] |= bf16_rn(g[r][k]) << (16 * r);
                }
            }
            // publish FIRST (the peers' next step waits for these), then the float32 copies for the weight-gradient GEMMs
            u64* dst = slots + (size_t)(step & 1) * NGRP * GR + unit;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __hip_atomic_store(dst + k * LH, ((u64)(unsigned)(step + 1) << 32) | packed[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < BPG; ++r) {
                if (r < nvalid) {
                    float* gp = dgx + ((size_t)t * B + b0 + bfirst + r) * 4096 + dir * GR + unit;
#pragma unroll
                    for (int k = 0; k < 4; ++k) gp[k * LH] = g[r][k];
                }
            }
        }
    }
}

}  // namespace

size_t hn_lstm_bwd_bf16_xch_bytes(void) { return (size_t)2 * NGRP * GR * sizeof(u64); }

// saved: [T][B][2][5][512] float32 (i, f, g, o, c) of the forward pass; dy: [T*B][1024] gradient of the layer output;
// whhT_*: bf16 [512][2048] = W_hh transposed; dgx: [T*B][4096] float32 gate pre-activation gradients (out);
// xch: hn_lstm_bwd_bf16_xch_bytes() of scratch
int hn_launch_lstm_layer_bwd_bf16(const float* saved, const float* dy, const void* whhT_f, const void* whhT_r, float* dgx, int T, int B,
                                  void* xch, void* sync_ws, hipStream_t s)
{
    HN_REQUIRE(T >= 1 && B >= 1 && T < 0x7fffffff, "lstm bwd bf16: bad T/B");
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int bc = (B - b0) < 32 ? (B - b0) : 32;
        HN_HIP(hipMemsetAsync(xch, 0, hn_lstm_bwd_bf16_xch_bytes(), s));
        if (int rc = hn_require_coresident(lstm_layer_bwd_bf16_kernel, NGRP * GRP, 256, "bf16 LSTM adjoint")) return rc;
        hipLaunchKernelGGL(lstm_layer_bwd_bf16_kernel, dim3(NGRP * GRP), dim3(256), 0, s, saved, dy, reinterpret_cast<const u16*>(whhT_f),
                           reinterpret_cast<const u16*>(whhT_r), dgx, reinterpret_cast<u64*>(xch), T, B, b0, bc,
                           reinterpret_cast<unsigned*>(sync_ws));
        HN_LAUNCH_CHECK();
    }
    return 0;
}

size_t hn_lstm_bf16_xch_bytes(void) { return (size_t)2 * NGRP * LH * sizeof(u64); }

// gx: [T*B][4096] float32 gate pre-activations (both directions); whh_*: bf16 [2048][512]; y: float32 [T*B][1024];
// y_h: optional bf16 copy of y (the next layer's GEMM operand); xch: hn_lstm_bf16_xch_bytes() of scratch
// save: optional [T][B][2][5][512] float32 (i, f, g, o, c) for the training step's adjoint
int hn_launch_lstm_layer_bf16(const float* gx, const void* whh_f, const void* whh_r, float* y, void* y_h, int T, int B, void* xch,
                              void* sync_ws, hipStream_t s, float* save)
{
    HN_REQUIRE(T >= 1 && B >= 1 && T < 0x7fffffff, "lstm bf16: bad T/B");
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int bc = (B - b0) < 32 ? (B - b0) : 32;
        HN_HIP(hipMemsetAsync(xch, 0, hn_lstm_bf16_xch_bytes(), s));       // every tag back to 0 before every launch
        if (int rc = hn_require_coresident(lstm_layer_bf16_kernel, NGRP * GRP, 256, "bf16 LSTM recurrence")) return rc;
        hipLaunchKernelGGL(lstm_layer_bf16_kernel, dim3(NGRP * GRP), dim3(256), 0, s, gx, reinterpret_cast<const u16*>(whh_f),
                           reinterpret_cast<const u16*>(whh_r), y, reinterpret_cast<u16*>(y_h), reinterpret_cast<u64*>(xch), T, B, b0,
                           bc, reinterpret_cast<unsigned*>(sync_ws), save);
        HN_LAUNCH_CHECK();
    }
    return 0;
}

// The wide form (see lstm_layer_bf16_wide_kernel): rows_per_group 16 (default) or 8, xcds_per_group 1 or 2; <= 0 picks the default.
// y_h (bf16 [T*B][1024]) is REQUIRED: it is the kernel's exchange buffer as well as its bf16 output.
int hn_launch_lstm_layer_bf16_wide(const float* gx, const void* whh_f, const void* whh_r, float* y, void* y_h, int T, int B,
                                   void* sync_ws, int rows_per_group, int xcds_per_group, hipStream_t s, float* save)
{
    HN_REQUIRE(T >= 1 && B >= 1 && T < 0x7fffffff && y_h != nullptr, "lstm bf16 wide: bad T/B or no y_h");
    const int bw = rows_per_group == 8 ? 8 : 16;
    const int xs = xcds_per_group == 2 ? 2 : 1;
    const int chunk = bw * (WIDE_MAX_GROUPS / 2);
    HN_HIP(hipMemsetAsync(y_h, 0xFF, (size_t)T * B * 1024 * sizeof(u16), s));       // every element = the "not yet written" sentinel
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = (B - b0) < chunk ? (B - b0) : chunk;
        const int ngroups = 2 * hn_cdiv(bc, bw);
        const int grid = xs == 1 ? 64 * hn_cdiv(ngroups, 8) : 32 * hn_cdiv(ngroups, 4);
        if (bw == 16) {
            if (int rc = hn_require_coresident(lstm_layer_bf16_wide_kernel<16>, grid, 256, "bf16 LSTM recurrence (wide)")) return rc;
            hipLaunchKernelGGL(lstm_layer_bf16_wide_kernel<16>, dim3(grid), dim3(256), 0, s, gx, reinterpret_cast<const u16*>(whh_f),
                               reinterpret_cast<const u16*>(whh_r), y, reinterpret_cast<u16*>(y_h), T, B, b0, bc, ngroups, xs,
                               reinterpret_cast<unsigned*>(sync_ws), save);
        } else {
            if (int rc = hn_require_coresident(lstm_layer_bf16_wide_kernel<8>, grid, 256, "bf16 LSTM recurrence (wide)")) return rc;
            hipLaunchKernelGGL(lstm_layer_bf16_wide_kernel<8>, dim3(grid), dim3(256), 0, s, gx, reinterpret_cast<const u16*>(whh_f),
                               reinterpret_cast<const u16*>(whh_r), y, reinterpret_cast<u16*>(y_h), T, B, b0, bc, ngroups, xs,
                               reinterpret_cast<unsigned*>(sync_ws), save);
        }
        HN_LAUNCH_CHECK();
    }
    return 0;
}
