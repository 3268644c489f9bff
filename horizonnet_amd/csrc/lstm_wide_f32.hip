// WIDE float32 bidirectional LSTM recurrence for the pipelined forward (hn_forward_submit, engine.hip): the exact-f32
// counterpart of lstm_layer_bf16_wide_kernel (lstm_bf16.hip).  Replaces the recurrent half of `nn.LSTM` (reference
// model.py:222-227,263-264) like lstm_layer_kernel (lstm.hip), with the work partitioned by BATCH first:
//
//   * lstm_layer_kernel spreads one direction over 128 workgroups (a 128-party hand-off per step) and occupies all 256
//     compute units for a latency-bound chain.  Here a group owns one direction of 16 panoramas (all 16 rows of
//     v_mfma_f32_16x16x4_f32 live) and has 16 members; a member keeps the W_hh rows of ITS 32 hidden units (4 gates x 32 x 512
//     float32 = 256 KB) in registers for the whole sequence.  A batch of 32 needs 4 groups = 64 compute units; the other 192
//     run the NEXT batch's convolutions on the caller's stream (the point of the pipelined entry).
//   * wave w of a member: unit block ub = w & 1 (16 units), k half kh = w >> 1 (256 of the 512 recurrent inputs): 4 gates x 64
//     MFMAs; the two k halves meet in LDS, the kh = 0 waves apply the gates (4 accumulator rows = 4 panoramas per lane).
//   * hand-off: the layer OUTPUT y is the exchange buffer, one slot per time step, pre-filled with the bit pattern
//     0xFFFFFFFF (a NaN no finite h equals): "no element equals the sentinel" is the arrival test -- no tag, no counter, no
//     fence.  A gate lane publishes 4 consecutive units of one panorama with ONE 16-byte write-through store (4 x 4
//     transpose inside the quad of unit lanes, DPP); every wave sweeps 4 rows of h_{t-1} with 8 sc1 16-byte loads per lane.
//   * exact float32 throughout (MFMA products and sums in f32, the gate functions of lstm.hip); only the summation ORDER
//     over k differs from lstm_layer_kernel, so the two agree to rounding (tested), not bit for bit.
//
// Placement independent; every spin is bounded and reports through the sticky status word.
#include "hn_common.h"

namespace {

constexpr int LH = 512;
constexpr int WB = 16;                        // panoramas per group = MFMA rows
constexpr int MEM = 16;                       // members per group
constexpr int UPM = LH / MEM;                 // 32 hidden units per member
constexpr int HP = LH + 4;                    // LDS row pitch (floats): rows 4 banks apart -> conflict-free 16-byte fragment reads
constexpr int MAX_GROUPS = 8;                 // per launch: 2 directions x 4 sets = 64 panoramas
constexpr unsigned SPIN_LIMIT_W = 1u << 22;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fexp(2.0f * x)); }
__device__ __forceinline__ float swap1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false)); }   // lane ^ 1
__device__ __forceinline__ float swap2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false)); }   // lane ^ 2

__global__ __launch_bounds__(256) void lstm_layer_f32_wide_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                                                                  const float* __restrict__ whh_r, float* y, int T, int B, int b0, int bc,
                                                                  int ngroups, unsigned* sync)
{
    __shared__ __attribute__((aligned(16))) float hs[2][WB][HP];          // h_{t-1} of the group's panoramas, by step parity
    __shared__ __attribute__((aligned(16))) float red[2][4][4][64];       // k-half partials: [unit block][gate][acc register][lane]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // blockIdx b runs on XCD b % 8 (observed, speed only): a group's 16 members sit 8 + 8 on two neighbouring XCDs
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int grp = (xcd >> 1) + 4 * (slot >> 3);
    const int mem = ((xcd & 1) << 3) | (slot & 7);
    if (grp >= ngroups) return;
    const int dir = grp & 1;
    const int bfirst = (grp >> 1) * WB;
    if (bfirst >= bc) return;                          // the whole group leaves together
    const int nvalid = (bc - bfirst) < WB ? (bc - bfirst) : WB;
    const float* whh = dir ? whh_r : whh_f;
    const int col = lane & 15;                         // A: panorama row / B, C: unit column
    const int kq = lane >> 4;
    const int ub = wave & 1, kh = wave >> 1;
    const int u0 = mem * UPM + ub * 16;
    const size_t y_bytes = (size_t)T * B * 1024 * sizeof(float);
    const __amdgpu_buffer_rsrc_t rsrc_y =              // h_{t-1} is read (and h_t written) with sc1 buffer accesses: aux bit 4
        __builtin_amdgcn_make_buffer_rsrc(y, 0, (unsigned)(y_bytes > 0xfffffff0u ? 0xfffffff0u : y_bytes), 0x00020000);

    // W_hh fragments: B[k][n], n = unit column of gate g -> row g*512 + u0 + col; lane holds k = kh*256 + 16j + 4kq .. +4
    // (MFMA q of step j consumes k = 16j + 4kq + q on both operands: a fixed k order)
    f32x4 wf[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wr = whh + (size_t)(g * LH + u0 + col) * LH + kh * 256 + kq * 4;
#pragma unroll
        for (int j = 0; j < 16; ++j) wf[g][j] = *reinterpret_cast<const f32x4*>(wr + j * 16);
    }

    // gate role (kh == 0 waves): accumulator rows kq*4 + r of unit u0 + col; after the quad transpose the lane publishes the
    // quad's 4 units of row kq*4 + pub_r
    const bool gate_wave = kh == 0;
    const int row0 = kq * 4;
    const int jq = col & 3;
    const int pub_row = row0 + ((jq & 1) ? 2 : 0) + (jq >> 1);
    float c_state[4] = {0.f, 0.f, 0.f, 0.f};
    bool failed = false;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;

        float gxv[4][4];
        if (gate_wave) {
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
                // wave w sweeps rows 4w .. 4w+3 of h_{t-1}: 2 x 16 bytes per lane and row; a row of an absent panorama
                // re-reads the group's row 0 (its MFMA row is dead)
                const int tprev = dir ? t + 1 : t - 1;
                u32x4 v[8];
                unsigned voff[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = wave * 4 + k < nvalid ? wave * 4 + k : 0;
                    voff[k] = (unsigned)((((size_t)tprev * B + b0 + bfirst + row) * 1024 + dir * LH) * sizeof(float)) + lane * 16;
                }
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, voff[k], 0, /*sc1*/ 16));
                        v[2 * k + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, voff[k] + 1024, 0, /*sc1*/ 16));
                    }
                    u32x4 m4 = v[0];                     // running maximum: 0xFFFFFFFF somewhere <=> an element is still the sentinel
#pragma unroll
                    for (int k = 1; k < 8; ++k) m4 = __builtin_elementwise_max(m4, v[k]);
                    const unsigned mx = max(max(m4[0], m4[1]), max(m4[2], m4[3]));
                    if (__all(mx != 0xffffffffu) || failed) break;
                    asm volatile("" ::: "memory");       // the next sweep re-reads memory
                    if (++spins > SPIN_LIMIT_W) {
                        __hip_atomic_store(sync + HN_STATUS_WORD, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        failed = true;
                        break;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    *reinterpret_cast<u32x4*>(&hs[par][wave * 4 + k][lane * 4]) = v[2 * k];
                    *reinterpret_cast<u32x4*>(&hs[par][wave * 4 + k][256 + lane * 4]) = v[2 * k + 1];
                }
            }
            __syncthreads();
            const float* hrow = &hs[par][col][kh * 256 + kq * 4];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(hrow + j * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wf[g][j][q], acc[g], 0, 0, 0);
            }
            if (!gate_wave) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[ub][g][r][lane] = acc[g][r];
            }
            __syncthreads();
        }

        if (gate_wave) {
            float hval[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = gxv[r][g] + (step > 0 ? acc[g][r] + red[ub][g][r][lane] : 0.f);
                const float ig = sigm(pre[0]);
                const float fg = sigm(pre[1]);
                const float gg = tanh_(pre[2]);
                const float og = sigm(pre[3]);
                c_state[r] = fg * c_state[r] + ig * gg;
                hval[r] = og * tanh_(c_state[r]);
            }
            // 4 rows x 1 unit per lane -> 1 row x 4 units per lane (4 x 4 transpose inside the quad of unit lanes)
            const bool odd = jq & 1;
            const float ra = swap1(odd ? hval[0] : hval[2]);                  // even lanes keep rows 0,1 and receive the odd neighbour's
            const float rb = swap1(odd ? hval[1] : hval[3]);                  // rows 0,1; odd lanes keep rows 2,3
            // first / second of the lane's two rows, each as (even unit, odd unit) of the pair
            const float p0e = odd ? ra : hval[0], p0o = odd ? hval[2] : ra;
            const float p1e = odd ? rb : hval[1], p1o = odd ? hval[3] : rb;
            const bool up = jq >> 1;                                          // unit pair 1 of the quad keeps the second row
            const float se = swap2(up ? p0e : p1e), so = swap2(up ? p0o : p1o);
            const f32x4 word = up ? f32x4{se, so, p1e, p1o} : f32x4{p0e, p0o, se, so};
            if (pub_row < nvalid) {                    // publish: ONE write-through 16-byte store (units u0 + 4*(col/4) .. +3 of panorama pub_row)
                const unsigned off = (unsigned)((((size_t)t * B + b0 + bfirst + pub_row) * 1024 + dir * LH + u0 + (col & ~3)) * sizeof(float));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, word), rsrc_y, off, 0, /*sc1*/ 16);
            }
        }
    }
}

}  // namespace

// gx: [T*B][4096] float32 gate pre-activations (both directions); whh_*: float32 [2048][512]; y: float32 [T*B][1024]
// (output AND exchange buffer: pre-filled with the sentinel here)
int hn_launch_lstm_layer_f32_wide(const float* gx, const float* whh_f, const float* whh_r, float* y, int T, int B, void* sync_ws,
                                  hipStream_t s)
{
    HN_REQUIRE(T >= 1 && B >= 1, "lstm f32 wide: bad T/B");
    HN_REQUIRE((size_t)T * B * 1024 * sizeof(float) <= 0xfffffff0ull, "lstm f32 wide: T*B too large for 32-bit buffer offsets");
    HN_HIP(hipMemsetAsync(y, 0xFF, (size_t)T * B * 1024 * sizeof(float), s));       // every element = the "not yet written" sentinel
    const int chunk = WB * (MAX_GROUPS / 2);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = (B - b0) < chunk ? (B - b0) : chunk;
        const int ngroups = 2 * hn_cdiv(bc, WB);
        const int grid = 64 * hn_cdiv(ngroups, 4);
        if (int rc = hn_require_coresident(lstm_layer_f32_wide_kernel, grid, 256, "f32 LSTM recurrence (wide)")) return rc;
        hipLaunchKernelGGL(lstm_layer_f32_wide_kernel, dim3(grid), dim3(256), 0, s, gx, whh_f, whh_r, y, T, B, b0, bc, ngroups,
                           reinterpret_cast<unsigned*>(sync_ws));
        HN_LAUNCH_CHECK();
    }
    return 0;
}
