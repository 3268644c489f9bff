// Weight-gradient GEMM for gfx950, float32 on the matrix cores:
//
//   dW[n][k'] = sum_m dZ[m][n] * A[m][k']      m = (b, ho, wo) output pixel (the REDUCTION index here),
//                                              n = output channel, k' = (dh, dw, c) tap x input channel
//
// i.e. the adjoint of conv_igemm_f32.hip with respect to the weights (reference: autograd of the
// convolutions at model.py:73-81,123-135, called from train.py:278).  dW comes out in the engine's
// packed layout [Cout][kh][kw][Cin] and is un-packed to OIHW by unpack_conv_kernel.
//
// Design: the output is small (<= 75 MB) and the reduction is long (up to 4M rows), so the m-range is
// split over workgroups (grid = n-tiles x k'-tiles x SPLIT) and every workgroup adds its 128x128 /
// 64x64 partial tile with float atomics (global_atomic_add_f32).  Per 32-row chunk both operands are
// staged by raw buffer loads (zero rows via out-of-range offsets, like the forward kernel) into LDS as
// [m][n] / [m][k'] images; v_mfma_f32_32x32x2_f32 consumes two m-rows per instruction with
// conflict-free ds_read_b32 fragment reads (lanes 0-31 row 2j, lanes 32-63 row 2j+1).
#include "hn_common.h"

#include <stdlib.h>

namespace {

constexpr int WCH = 32;                 // m rows per chunk
constexpr unsigned OOB = 0x80000000u;

struct WgradArgs {
    const float* x;      // NHWC input of the forward conv [B][Hi][Wi][Cin]  (stem: NHWC4)
    const float* dz;     // gradient w.r.t. the conv output [M][Cout]
    float* dw;           // packed [Cout][K], pre-zeroed, accumulated with atomics
    int Hi, Wi, Cin, Ho, Wo, Cout;
    int KW, sh, sw, ph, pw;
    int M, K, mchunk;    // rows per workgroup (multiple of WCH)
    int xstride, dzstride;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wbuf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

template <int TN, int TK, bool STEM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p)
{
    constexpr int SN = TN + 4, SK = TK + 4;          // LDS row strides (floats)
    constexpr int WN = TN / 2, WK = TK / 2;          // wave tile (2 x 2 waves)
    constexpr int TI = WN / 32, TJ = WK / 32;        // MFMA tiles per wave
    static_assert(TI >= 1 && TJ >= 1, "wave tile");
    constexpr int N_TPR = TN / 4, K_TPR = TK / 4;    // loader threads per row
    constexpr int N_RPP = 256 / N_TPR, K_RPP = 256 / K_TPR;
    constexpr int N_PS = WCH / N_RPP, K_PS = WCH / K_RPP;
    static_assert(N_PS >= 1 && K_PS >= 1, "loader geometry");

    extern __shared__ __attribute__((aligned(16))) float wsmem[];
    float* s_dz0 = wsmem;                            // [2][WCH * SN]
    float* s_a0 = wsmem + 2 * WCH * SN;              // [2][WCH * SK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wi_ = wave >> 1, wj_ = wave & 1;

    const int NT = (p.Cout + TN - 1) / TN;           // Cout = 32 (ghc0.3) runs a half-empty 64-row tile
    const int KT = p.K / TK;
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own L2): XCD x takes a contiguous run of tile
    // ids, so the NT*KT workgroups that stream the SAME rows of x and dZ run side by side on ONE XCD and share them
    // through its L2.  With the plain round-robin every XCD fetched every m-range itself: 93 GB of HBM reads per
    // training step against ~31 GB algorithmic (rocprofv3 FETCH_SIZE).
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
    const int k0 = kt * TK;
    const int mlo = sp * p.mchunk;
    const int mhi = (mlo + p.mchunk) < p.M ? (mlo + p.mchunk) : p.M;
    if (mlo >= mhi) return;

    // filter tap of this k' tile (the whole tile lies inside one tap: TK | Cin; stem: two 32-wide dh rows)
    const int tap = STEM ? 0 : k0 / p.Cin;
    const int c0 = STEM ? 0 : k0 % p.Cin;
    const int dh = STEM ? 0 : tap / p.KW;
    const int dw = STEM ? 0 : tap % p.KW;

    const int hw_out = p.Ho * p.Wo;
    const int b_first = mlo / hw_out;
    const size_t img_floats = (size_t)p.Hi * p.Wi * (STEM ? 4 : p.xstride);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b_first * img_floats), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_dz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz + (size_t)mlo * p.dzstride), 0, 0x7fffffff, 0x00020000);

    const int n_row = tid / N_TPR, n_col = (tid % N_TPR) * 4;
    const int k_row = tid / K_TPR, k_col = (tid % K_TPR) * 4;

    // Output pixel (b, ho, wo) of each A row this thread stages, for the chunk about to be fetched.  Decomposed once
    // (two integer divisions per row); afterwards every chunk advances all rows by WCH with compare-and-subtract wraps,
    // so the steady-state loop has no division (they used to cost ~1/4 of the issue slots).
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

    f32x4 rdz[N_PS], ra[K_PS];
    const bool wide = p.Wo >= WCH;
    // The fetch of a chunk is split into its N_PS + K_PS row loads ("parts") so the main loop can issue one part
    // behind each group of MFMAs: address arithmetic and load issue then run in the shadow of the matrix pipe instead
    // of in front of it (in-order issue: as one block before the MFMAs they cost ~1/5 of every chunk).
    auto fetch_dz = [&](int q, int mc) {
        const int m = mc + n_row + q * N_RPP;
        const unsigned off = (m < mhi && n0 + n_col < p.Cout) ? (unsigned)((size_t)(m - mlo) * p.dzstride + n0 + n_col) * 4u : OOB;
        rdz[q] = wbuf_load16(rsrc_dz, off);
    };
    auto fetch_a = [&](int q, int mc) {  // called with mc = mlo, mlo + WCH, ... in order (advances the row's pixel state)
        const int m = mc + k_row + q * K_RPP;
        // branch-free: the address is always formed, rows past the range / in the zero padding select the OOB offset
        const int wo = r_wo[q], ho = r_ho[q];
        int hi, wi;
        if (STEM) {               // k' = kt*64 + col: dh = 2*kt + (col >= 32), horizontal tap = (col & 31) / 4
            const int sdh = 2 * kt + (k_col >= 32 ? 1 : 0);
            hi = sdh >= 7 ? -1 : ho * p.sh - p.ph + sdh;
            wi = wo * p.sw - p.pw + ((k_col & 31) >> 2);
        } else {
            hi = ho * p.sh - p.ph + dh;
            wi = wo * p.sw - p.pw + dw;
        }
        wi = wi < 0 ? wi + p.Wi : wi;
        wi = wi >= p.Wi ? wi - p.Wi : wi;
        const unsigned pix = (unsigned)(r_b[q] * hw_in + hi * p.Wi + wi);
        const unsigned in_off = STEM ? pix * 16u : (pix * (unsigned)p.xstride + (unsigned)(c0 + k_col)) * 4u;
        const unsigned off = (m < mhi && (unsigned)hi < (unsigned)p.Hi) ? in_off : OOB;
        ra[q] = wbuf_load16(rsrc_x, off);
        // advance this row to the next chunk
        if (wide) {                  // every layer of the network (WCH <= Wo): at most one wrap per level, as selects
            int wn = r_wo[q] + WCH;
            const bool wrap_w = wn >= p.Wo;
            wn = wrap_w ? wn - p.Wo : wn;
            const int hn = r_ho[q] + (wrap_w ? 1 : 0);
            const bool wrap_h = hn >= p.Ho;
            r_ho[q] = wrap_h ? 0 : hn;
            r_b[q] += wrap_h ? 1 : 0;
            r_wo[q] = wn;
        } else {                     // images narrower than a chunk (tests only): decompose again
            const int mn = m + WCH;
            r_wo[q] = mn % p.Wo;
            const int t = mn / p.Wo;
            r_ho[q] = t % p.Ho;
            r_b[q] = t / p.Ho - b_first;
        }
    };
    auto fetch_part = [&](int part, int mc) {
        if (part < N_PS) fetch_dz(part, mc);
        else if (part < N_PS + K_PS) fetch_a(part - N_PS, mc);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < N_PS; ++q)
            *reinterpret_cast<f32x4*>(s_dz0 + buf * WCH * SN + (n_row + q * N_RPP) * SN + n_col) = rdz[q];
#pragma unroll
        for (int q = 0; q < K_PS; ++q)
            *reinterpret_cast<f32x4*>(s_a0 + buf * WCH * SK + (k_row + q * K_RPP) * SK + k_col) = ra[q];
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31;
    const int fh = lane >> 5;

#pragma unroll
    for (int part = 0; part < N_PS + K_PS; ++part) fetch_part(part, mlo);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int mc = mlo; mc < mhi; mc += WCH) {
        const bool more = mc + WCH < mhi;
        static_assert(N_PS + K_PS <= WCH / 2, "one fetch part per MFMA group");
        const float* dz_s = s_dz0 + buf * WCH * SN + fh * SN + wi_ * WN + fr;
        const float* a_s = s_a0 + buf * WCH * SK + fh * SK + wj_ * WK + fr;
        // fragments double-buffered in registers: the LDS reads of step kk+1 are issued before the MFMAs of step kk
        // (without this every 4-MFMA group waited for its own ds_reads: ~1/3 of the MFMA pipe idle)
        float fa[2][TI], fb[2][TJ];
        auto ldfrag = [&](int slot, int kk) {
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[slot][i] = dz_s[kk * 2 * SN + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[slot][j] = a_s[kk * 2 * SK + j * 32];
        };
        ldfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < WCH / 2; ++kk) {
            if (kk + 1 < WCH / 2) ldfrag((kk + 1) & 1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);       // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
            if (more) fetch_part(kk, mc + WCH);      // next chunk's loads, one row per MFMA group
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // accumulate the partial tile: rows = n, cols = k'
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wi_ * WN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int k = k0 + wj_ * WK + j * 32 + fr;
                if (n < p.Cout)
                    __hip_atomic_fetch_add(p.dw + (size_t)n * p.K + k, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
}

// packed [Cout][kh][kwp][cp] -> OIHW (drops the stem's zero padding)
__global__ __launch_bounds__(256) void unpack_conv_kernel(const float* __restrict__ wp, float* __restrict__ w, int Cout, int Cin,
                                                          int KH, int KW, int KHp, int KWp, int Cp)
{
    const long total = (long)Cout * Cin * KH * KW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dw = (int)(i % KW);
        long t = i / KW;
        const int dh = (int)(t % KH);
        t /= KH;
        const int c = (int)(t % Cin);
        const int o = (int)(t / Cin);
        w[i] = wp[(((long)o * KHp + dh) * KWp + dw) * Cp + c];
    }
}

// OIHW -> data-gradient packing [Cin][kh][kw][Cout]
__global__ __launch_bounds__(256) void pack_conv_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                              int KH, int KW)
{
    const long total = (long)Cout * Cin * KH * KW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % Cout);
        long t = i / Cout;
        const int dw = (int)(t % KW);
        t /= KW;
        const int dh = (int)(t % KH);
        const int c = (int)(t / KH);
        out[i] = w[(((long)o * Cin + c) * KH + dh) * KW + dw];
    }
}

template <int TN, int TK, bool STEM>
int launch_wgrad(WgradArgs a, hipStream_t s, int max_split = 0)
{
    const int NT = (a.Cout + TN - 1) / TN, KT = a.K / TK;
    // split the reduction so that ~2048 workgroups exist, each with >= 8 chunks
    static const long target = getenv("HN_WGRAD_WGS") ? atol(getenv("HN_WGRAD_WGS")) : 2048;
    long split = target / ((long)NT * KT);
    if (split < 1) split = 1;
    // max_split = 2: at most two partial tiles meet in the float atomics of an output element, and 0 + a + b == 0 + b + a: a run-to-run
    // reproducible result (the folded BatchNorm adjoint's Q, which feeds the data gradient: bn_fold.hip)
    if (max_split > 0 && split > max_split) split = max_split;
    long mchunk = (a.M + split - 1) / split;
    if (mchunk < 8 * WCH) mchunk = 8 * WCH;
    mchunk = (mchunk + WCH - 1) / WCH * WCH;
    split = (a.M + mchunk - 1) / mchunk;
    a.mchunk = (int)mchunk;
    const size_t lds = 2 * (size_t)WCH * ((TN + 4) + (TK + 4)) * sizeof(float);
    auto kern = conv_wgrad_kernel<TN, TK, STEM>;
    static bool attr_done[64] = {};   // per instantiation, per device
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_done[dev]) {
        HN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(NT * KT * split)), dim3(256), lds, s, a);
    HN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// dw_packed must hold hn_packed_conv_weight_floats() floats; it is zeroed here (unless `prezeroed`) and accumulated by the kernel.
int hn_launch_conv_wgrad(const float* x, const float* dz, float* dw_packed, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                         int sh, int sw, int xstride, int dzstride, int stem, hipStream_t s, int prezeroed, int max_split)
{
    WgradArgs a;
    a.x = x; a.dz = dz; a.dw = dw_packed;
    a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KW = KW; a.sh = sh; a.sw = sw; a.ph = KH / 2; a.pw = KW / 2;
    a.Ho = (Hi + 2 * a.ph - KH) / sh + 1;
    a.Wo = (Wi + 2 * a.pw - KW) / sw + 1;
    a.M = B * a.Ho * a.Wo;
    a.xstride = xstride ? xstride : Cin;
    a.dzstride = dzstride ? dzstride : Cout;
    a.mchunk = 0;
    if (stem) {
        HN_REQUIRE(KH == 7 && KW == 7 && Cout == 64, "wgrad: stem expects 7x7, Cout=64");
        a.Cin = 4;
        a.K = 8 * 32;                       // scratch layout [64][8 dh (7 used)][8 dw (7 used)][4 c (3 used)]; k' tiles are 64 wide
        if (!prezeroed) HN_HIP(hipMemsetAsync(dw_packed, 0, (size_t)Cout * a.K * sizeof(float), s));
        return launch_wgrad<64, 64, true>(a, s);
    }
    HN_REQUIRE(Cin % 64 == 0 && Cout % 32 == 0, "wgrad: Cin=%d must be a multiple of 64, Cout=%d of 32", Cin, Cout);
    a.K = KH * KW * Cin;
    if (!prezeroed) HN_HIP(hipMemsetAsync(dw_packed, 0, (size_t)Cout * a.K * sizeof(float), s));
    if (Cout % 64 != 0) return launch_wgrad<64, 64, false>(a, s, max_split);
    if (Cout % 128 == 0 && Cin % 128 == 0) return launch_wgrad<128, 128, false>(a, s, max_split);
    if (Cout % 128 == 0) return launch_wgrad<128, 64, false>(a, s, max_split);
    if (Cin % 128 == 0) return launch_wgrad<64, 128, false>(a, s, max_split);
    return launch_wgrad<64, 64, false>(a, s, max_split);
}

// packed_rows: dh rows per output channel in `wp` (KH for the forward packing, 8 for the stem's wgrad scratch)
int hn_launch_unpack_conv(const float* wp, float* w_oihw, int Cout, int Cin, int KH, int KW, int packed_rows, hipStream_t s)
{
    const bool stem = KH == 7;
    const int KWp = stem ? 8 : KW, Cp = stem ? 4 : Cin;
    const long total = (long)Cout * Cin * KH * KW;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(unpack_conv_kernel, dim3((unsigned)g), dim3(256), 0, s, wp, w_oihw, Cout, Cin, KH, KW,
                       packed_rows ? packed_rows : KH, KWp, Cp);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_pack_conv_dgrad(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s)
{
    const long total = (long)Cout * Cin * KH * KW;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, s, w, out, Cout, Cin, KH, KW);
    HN_LAUNCH_CHECK();
    return 0;
}
