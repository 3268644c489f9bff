// Persistent bidirectional LSTM layer for gfx950 (hidden 512, gates i,f,g,o).
//
// Replaces the cuDNN/MKLDNN `nn.LSTM` delegate at reference model.py:222-227,263-264 for the
// recurrent half of a layer; the input projection gx = x @ W_ih^T + b_ih + b_hh is one implicit-
// GEMM launch (conv_igemm_f32.hip, 1x1 "conv") done before this kernel.
//
// One launch walks all T steps of BOTH directions for up to 32 panoramas:
//   * 256 workgroups (one per CU), 128 per direction; workgroup j of a direction owns hidden
//     units [4j, 4j+4) = 16 gate rows of W_hh.  Its W_hh slice (16 x 512 floats) lives in
//     REGISTERS for the whole sequence (each of the 4 waves holds its 128-wide k-range as the
//     B operand of v_mfma_f32_16x16x4_f32: 8 x float4 per lane), the cell state c lives in the
//     registers of the 128 "gate" threads.
//   * per step: h_{t-1} (all 512 units, [<=32][512]) is read straight from the layer output y
//     (L2), 4 waves x 64 MFMAs produce k-partial gate pre-activations, reduced through LDS, the
//     gate threads apply sigmoid/tanh and write h_t into y -- which IS the hand-off buffer.
//   * hand-off between workgroups (cdna guide G16, form R1): h_t is stored WRITE-THROUGH (sc1,
//     agent-scope relaxed atomic stores), every storing wave drains vmcnt, one lane arrives on a
//     monotonic counter (8 counters per direction, 16 arrivals each per step, so the atomics do
//     not serialise on one word); consumers poll the 8 words relaxed from 8 lanes, then read
//     h_{t-1} with sc1 loads (L1 bypassed) -- no release/acquire fences on the critical path.
//     Placement independent; every spin is bounded and reports through a status word instead
//     of hanging.
#include "hn_common.h"

namespace {

constexpr int LSTM_H = 512;
constexpr int LSTM_U = 4;                    // hidden units per workgroup
constexpr int LSTM_NB = LSTM_H / LSTM_U;     // 128 workgroups per direction
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int CNT_SHARDS = 8;                // arrival counters per direction
constexpr int CNT_STRIDE = 32;               // counters 128 bytes apart: word (dir*8 + shard) * 32
constexpr int STATUS_WORD = HN_STATUS_WORD;
static_assert(2 * CNT_SHARDS * CNT_STRIDE <= HN_STATUS_WORD && HN_STATUS_WORD < HN_SYNC_WORDS, "sync scratch layout");

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Gate nonlinearities sit on every step's critical path (128 gate threads, nothing to overlap them with): hardware exp2 / rcp
// (v_exp_f32, v_rcp_f32, ~1 ulp each) instead of libm's expf / tanhf and an IEEE division -- absolute error ~2e-7 per gate, the
// size of the difference between two libm implementations (torch's CPU path uses a vectorised approximation as well); the
// layer-output tap stays within 1e-5 of the reference's (tests/test_gpu_parity.py).  Saturation is exact (exp -> inf, rcp -> 0).
__device__ __forceinline__ float fast_exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp_(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp_(2.0f * x)); }

template <int BT>   // 16-row batch tiles per launch (1: <=16 panoramas, 2: <=32)
__global__ __launch_bounds__(256) void lstm_layer_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                                                         const float* __restrict__ whh_r, float* y, int T, int B,
                                                         int b0, int bc, unsigned* sync, float* __restrict__ save)
{
    __shared__ __attribute__((aligned(16))) float red[4 * BT * 16 * 17];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int dir = blockIdx.x & 1;
    const int u0 = (blockIdx.x >> 1) * LSTM_U;
    const float* whh = dir ? whh_r : whh_f;
    unsigned* cnt_mine = sync + (dir * CNT_SHARDS + ((blockIdx.x >> 1) & (CNT_SHARDS - 1))) * CNT_STRIDE;
    unsigned* cnt_poll = sync + (dir * CNT_SHARDS + (tid & (CNT_SHARDS - 1))) * CNT_STRIDE;   // lanes 0..7 of wave 0
    // h_{t-1} is read with sc1 (L1-bypassing) buffer loads: aux bit 4 = sc1 on gfx940+
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc(y, 0, (unsigned)((size_t)T * B * 1024 * sizeof(float) > 0xfffffff0u
                                                               ? 0xfffffff0u : (size_t)T * B * 1024 * sizeof(float)), 0x00020000);

    // ---- W_hh fragments: B[k][n], n = gate*4 + unit -> row gate*512 + u0 + unit --------------
    const int fn = lane & 15;
    const int kq = lane >> 4;
    const int koff = wave * 128 + kq * 4;    // + s*16 + q
    f32x4 wf[8];
    {
        const float* wr = whh + (size_t)((fn >> 2) * LSTM_H + u0 + (fn & 3)) * LSTM_H + koff;
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[s] = *reinterpret_cast<const f32x4*>(wr + s * 16);
    }

    // ---- gate-thread role ----
    const bool gate_thread = tid < BT * 16 * LSTM_U;
    const int gb = tid >> 2;                 // batch row within the chunk
    const int gu = tid & 3;                  // unit within the workgroup
    const bool gate_live = gate_thread && gb < bc;
    float c_state = 0.f;

    // A-operand rows (clamped: padding rows compute garbage that is never stored)
    int arow[BT];
#pragma unroll
    for (int tl = 0; tl < BT; ++tl) {
        const int r = tl * 16 + fn;
        arow[tl] = b0 + (r < bc ? r : 0);
    }

    bool failed = false;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;

        float gxv[4] = {0.f, 0.f, 0.f, 0.f};
        if (gate_live) {
            const float* g = gx + ((size_t)t * B + b0 + gb) * 4096 + dir * 2048 + u0 + gu;
#pragma unroll
            for (int q = 0; q < 4; ++q) gxv[q] = g[q * LSTM_H];
        }

        f32x4 acc[BT];
#pragma unroll
        for (int tl = 0; tl < BT; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            if (tid < CNT_SHARDS && !failed) {
                const unsigned want = (unsigned)(LSTM_NB / CNT_SHARDS) * (unsigned)step;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt_poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) {
                        __hip_atomic_store(sync + STATUS_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        failed = true;
                        break;
                    }
                }
            }
            __syncthreads();

            const int tp = dir ? t + 1 : t - 1;
            f32x4 hf[BT][8];
#pragma unroll
            for (int tl = 0; tl < BT; ++tl) {
                const unsigned hoff = (unsigned)((((size_t)tp * B + arow[tl]) * 1024 + dir * LSTM_H + koff) * sizeof(float));
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    hf[tl][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, hoff + s * 64, 0, /*sc1*/ 16));
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tl = 0; tl < BT; ++tl)
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf[tl][s][q], wf[s][q], acc[tl], 0, 0, 0);
        }

        // k-partials -> LDS: red[wave][tile][row b][col n], row stride 17
#pragma unroll
        for (int tl = 0; tl < BT; ++tl)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((wave * BT + tl) * 16 + kq * 4 + r) * 17 + fn] = acc[tl][r];
        __syncthreads();

        if (gate_thread) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gxv[q];
#pragma unroll
                for (int w = 0; w < 4; ++w) s += red[((w * BT + (gb >> 4)) * 16 + (gb & 15)) * 17 + q * 4 + gu];
                pre[q] = s;
            }
            const float ig = sigmoidf_(pre[0]);
            const float fg = sigmoidf_(pre[1]);
            const float gg = tanhf_(pre[2]);
            const float og = sigmoidf_(pre[3]);
            c_state = fg * c_state + ig * gg;
            const float h = og * tanhf_(c_state);
            if (gate_live)      // write-through (sc1) store: visible to every XCD once the wave's vmcnt drains
                __hip_atomic_store(y + ((size_t)t * B + b0 + gb) * 1024 + dir * LSTM_H + u0 + gu, h, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (gate_live && save) {   // training: keep (i, f, g, o, c) for the adjoint pass -- [t][b][dir][5][512]
                float* sv = save + ((((size_t)t * B + b0 + gb) * 2 + dir) * 5) * LSTM_H + u0 + gu;
                sv[0] = ig; sv[LSTM_H] = fg; sv[2 * LSTM_H] = gg; sv[3 * LSTM_H] = og; sv[4 * LSTM_H] = c_state;
            }
        }

        // publish h_t: every storing wave drains its write-through stores, then ONE lane arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Adjoint recurrence GEMM for one step: dh[b][dir*512 + k] = sum_n dg[b][dir*2048 + n] * Whh_dir[n][k]
// (reference: autograd of nn.LSTM, train.py:278) on the matrix cores.  Whh is given TRANSPOSED ([512][2048],
// k-major): it is the MFMA B operand with n as the reduction index.  Grid = 2 dirs x 16 k-tiles (32 columns)
// x 8 n-slices (256 each) = 256 workgroups; each of the 4 waves reduces 64 n with v_mfma_f32_32x32x2_f32
// (A = dg rows, <= 32 per launch row-tile), partial tiles are combined with float atomics into the pre-zeroed dh.
__global__ __launch_bounds__(256) void lstm_bwd_dh_kernel(const float* __restrict__ dg_f, const float* __restrict__ dg_r,
                                                          const float* __restrict__ whhT_f, const float* __restrict__ whhT_r,
                                                          float* __restrict__ dh, int B)
{
    const int bid = blockIdx.x;
    const int dir = bid & 1;
    const int kt = (bid >> 1) & 15;
    const int ns = bid >> 5;                               // 0..7
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, half = lane >> 5;
    const int b0 = blockIdx.y * 32;
    const int brow = (b0 + fr) < B ? (b0 + fr) : (B - 1);  // clamped: padding rows are never stored
    const int nbase = ns * 256 + wave * 64 + half * 4;     // + 8*j + q
    const float* g = (dir ? dg_r : dg_f) + (size_t)brow * 4096 + dir * 2048 + nbase;
    const float* w = (dir ? whhT_r : whhT_f) + (size_t)(kt * 32 + fr) * 2048 + nbase;
    f32x4 ga[8], wb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ga[j] = *reinterpret_cast<const f32x4*>(g + 8 * j);
        wb[j] = *reinterpret_cast<const f32x4*>(w + 8 * j);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j][q], wb[j][q], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (b < B)
            __hip_atomic_fetch_add(dh + (size_t)b * 1024 + dir * 512 + kt * 32 + fr, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C)
{
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (by + j < R && bx + tx < C) tile[j][tx] = in[(size_t)(by + j) * C + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < C && by + tx < R) out[(size_t)(bx + j) * R + by + tx] = tile[tx][j];
}

}  // namespace

// whhT_f / whhT_r: the recurrent weights transposed to [512][2048]
int hn_launch_lstm_bwd_dh(const float* dg_f, const float* dg_r, const float* whhT_f, const float* whhT_r, float* dh, int B, hipStream_t s)
{
    // dh must be zero on entry (partial tiles are accumulated with atomics): the caller zeroes it once per layer and
    // lstm_bwd_gates_kernel re-zeroes every element as it consumes it -- no memset per step
    hipLaunchKernelGGL(lstm_bwd_dh_kernel, dim3(256, (B + 31) / 32), dim3(256), 0, s, dg_f, dg_r, whhT_f, whhT_r, dh, B);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_transpose(const float* in, float* out, int R, int C, hipStream_t s)
{
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, in, out, R, C);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_lstm_layer(const float* gx, const float* whh_f, const float* whh_r, float* y, int T, int B,
                         void* sync_ws, hipStream_t s, float* save)
{
    HN_REQUIRE(T >= 1 && B >= 1, "lstm: bad T/B");
    HN_REQUIRE((size_t)T * B * 1024 * sizeof(float) <= 0xfffffff0ull, "lstm: T*B too large for 32-bit buffer offsets");
    unsigned* sync = reinterpret_cast<unsigned*>(sync_ws);
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int bc = (B - b0) < 32 ? (B - b0) : 32;
        HN_HIP(hipMemsetAsync(sync, 0, STATUS_WORD * sizeof(unsigned), s));   // counters only; status is sticky
        if (int rc = bc <= 16 ? hn_require_coresident(lstm_layer_kernel<1>, 2 * LSTM_NB, 256, "LSTM recurrence")
                              : hn_require_coresident(lstm_layer_kernel<2>, 2 * LSTM_NB, 256, "LSTM recurrence"))
            return rc;
        if (bc <= 16)
            hipLaunchKernelGGL(lstm_layer_kernel<1>, dim3(2 * LSTM_NB), dim3(256), 0, s, gx, whh_f, whh_r, y, T, B, b0, bc, sync, save);
        else
            hipLaunchKernelGGL(lstm_layer_kernel<2>, dim3(2 * LSTM_NB), dim3(256), 0, s, gx, whh_f, whh_r, y, T, B, b0, bc, sync, save);
        HN_LAUNCH_CHECK();
    }
    return 0;
}
