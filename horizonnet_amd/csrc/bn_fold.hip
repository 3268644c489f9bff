// BatchNorm-folded adjoint of the 1x1 convs (bf16 training step; reference: autograd through torchvision Bottleneck's
// conv3 -> bn3 / downsample.0 -> downsample.1, model.py:78-81, under train.py:44-58,272-281).
//
// For a 1x1 / stride-1 conv z = a W^T followed by a batch-statistics BatchNorm, the BatchNorm adjoint
//     dz = c1 * (g - S1/M - zhat * S2/M),   c1 = gamma * invstd,  zhat = (z - mu) * invstd,  S1 = sum_m g,  S2 = sum_m g * zhat
// never has to be formed per element, because z is LINEAR in a.  With the small matrices
//     P[n][k] = sum_m g[m][n] a[m][k]      (the weight-gradient GEMM on g itself: conv_wgrad_bf16_kernel<.., FOLD>)
//     G[k][k'] = sum_m a[m][k] a[m][k']    (Gram matrix of the input activation),   A[k] = sum_m a[m][k]
// everything the classical passes produce follows exactly:
//     sum_m g z           = sum_k W[n][k] P[n][k]                       -> S2[n] = invstd (that - mu S1)
//     sum_m zhat a[.][k]  = invstd ((W G)[n][k] - mu A[k])
//     dW[n][k]            = c1 (P[n][k] - S1/M A[k] - S2/M invstd ((W G)[n][k] - mu A[k]))
//     da[m][:]            = g[m][:] (c1 * W)  -  a[m][:] Q  -  r,    Q = W^T diag(e) W,  e = c1 invstd S2 / M,
//                                                                 r = sum_n (c1 S1/M - e mu) W[n][:]
// i.e. the data gradient is two plain 1x1 convs (on g with the weights c1 * W and shift -r, on a with the weights -Q and the first
// result as the residual).  The reduce pass (reads dy, z), the apply pass (reads dy, z, writes dz and the masked gradient) and the
// per-element dz tensor disappear: per conv3 unit 8.5 activation-sized transfers become ~4.
// W is the bf16 rounding the forward GEMM multiplied with, so "z = a W^T" is the forward's own z up to its f32 accumulation order.
//
// This file: the finishing step.  The two small GEMMs in it (W G: N x K x K, and Q: K x K x N) run on the float32 matrix-core kernels
// of the library (hn_launch_conv / hn_launch_conv_wgrad on W widened to float32: exact products, f32 accumulation); the rest is per
// channel / per weight arithmetic in double precision.
#include "hn_common.h"

#include <string.h>

namespace {

typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float v)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(0.f));
    return (u16)(r & 0xffffu);
}

struct FoldArgs {
    float* P;              // [N][K]: in P, out dW (same layout as the packed 1x1 weight gradient)
    const double* A;       // [K]
    const double* S1_in;   // [N]  sum_m g
    double* S1_out;        // [N]  (may alias S1_in): the BatchNorm bias gradient slot of this unit
    double* S2;            // [N]  out: the BatchNorm weight gradient
    const u16* Wh;         // [N][K] bf16, the forward GEMM's weights
    const float* mean;
    const float* invstd;
    const float* gamma;
    double M;
    double* coef;          // scratch [4][N]: c1, e, rc = c1 S1/M - e mu, q = S2 invstd / M
    float* Wf;             // [N][K] f32: W widened
    float* eW;             // [N][K] f32: e[n] * W[n][k]
    const float* WG;       // [N][K] f32: (W G)[n][k]          (second kernel)
    const float* Qf;       // [K][K] f32: Q                    (second kernel)
    u16* WA;               // [K][N] bf16: WA[j][n] = c1[n] W[n][j]
    float* shiftA;         // [K]: -r[j]
    u16* WB;               // [K][K] bf16: WB[j][k] = -Q[k][j]
    int N, K;
};

// one wave per channel n: T = sum_k W[n][k] P[n][k]  ->  S2 and the per-channel coefficients; rows n of Wf and eW
__global__ __launch_bounds__(256) void bn_fold_coef_kernel(FoldArgs p)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= p.N) return;
    double t = 0.0;
    for (int k = lane; k < p.K; k += 64) t += (double)bf2f(p.Wh[(size_t)n * p.K + k]) * (double)p.P[(size_t)n * p.K + k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
    const double is = (double)p.invstd[n], mu = (double)p.mean[n], s1 = p.S1_in[n];
    const double s2 = is * (t - mu * s1);
    const double c1 = (double)p.gamma[n] * is;
    const double e = c1 * is * s2 / p.M;
    if (lane == 0) {
        p.S2[n] = s2;
        p.S1_out[n] = s1;
        p.coef[n] = c1;
        p.coef[p.N + n] = e;
        p.coef[2 * (size_t)p.N + n] = c1 * s1 / p.M - e * mu;
        p.coef[3 * (size_t)p.N + n] = s2 * is / p.M;
    }
    for (int k = lane; k < p.K; k += 64) {
        const float w = bf2f(p.Wh[(size_t)n * p.K + k]);
        p.Wf[(size_t)n * p.K + k] = w;          // (already there when the forward was folded too: same bits)
        p.eW[(size_t)n * p.K + k] = (float)(e * (double)w);
    }
}

// ---- forward: the batch statistics of z = a W^T from G and A (no pass over z; z itself is never stored) ----
__global__ __launch_bounds__(256) void bn_fold_widen_kernel(const u16* __restrict__ wh, float* __restrict__ wf, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) wf[i] = bf2f(wh[i]);
}

// one wave per channel n: mean = W[n][:] A / M,  E[z^2] = W[n][:] G W[n][:]^T / M = (W G)[n][:] W[n][:]^T / M  ->  the affine, the saved
// mean / invstd and the running-statistics update exactly as affine_act_bn_kernel derives them from its sums (train_ops.hip)
__global__ __launch_bounds__(256) void bn_fold_stats_kernel(const float* __restrict__ Wf, const float* __restrict__ WG, const double* __restrict__ A,
                                                            double M, int N, int K, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* running_mean, float* running_var, float momentum, float* a_out, float* b_out,
                                                            float* save_mean, float* save_invstd)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double w = (double)Wf[(size_t)n * K + k];
        s1 += w * A[k];
        s2 += w * (double)WG[(size_t)n * K + k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
        const double m = s1 / M;
        double var = s2 / M - m * m;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + 1e-5));
        const float aa = gamma[n] * invstd;
        const float bb = beta[n] - (float)m * aa;
        a_out[n] = aa;
        b_out[n] = bb;
        save_mean[n] = (float)m;
        save_invstd[n] = invstd;
        if (running_mean) {
            const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
            running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)m;
            running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
        }
    }
}

// WA[j][n] = bf16(c1[n] * W[n][j]), c1 = gamma * invstd: the data-gradient weights that do not depend on the batch sums (ahead of the P-GEMM when
// conv A rides in it, conv_wgrad_bf16_kernel<.., FUSEA>); the same values bn_fold_finish_kernel writes
__global__ __launch_bounds__(256) void bn_fold_wa_kernel(const u16* __restrict__ wh, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                         u16* __restrict__ wa, int N, int K)
{
    const long total = (long)N * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        const double c1 = (double)gamma[n] * (double)invstd[n];
        wa[(size_t)k * N + n] = f2bf((float)(c1 * (double)bf2f(wh[i])));
    }
}

// blockIdx.y selects the job: 0: dW over P and WA = (c1 * W)^T (elementwise over [N][K]);  1: WB = -Q (elementwise over [K][K]);
// 2: shiftA[j] = -sum_n rc[n] W[n][j] (one workgroup per 16 columns, 16 row lanes, combined through LDS)
__global__ __launch_bounds__(256) void bn_fold_finish_kernel(FoldArgs p)
{
    __shared__ double red[256];
    const int N = p.N, K = p.K, tid = threadIdx.x;
    const int job = blockIdx.y;
    if (job == 0) {
        const long total = (long)N * K;
        for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
            const int n = (int)(i / K), k = (int)(i - (long)n * K);
            const double c1 = p.coef[n], q = p.coef[3 * (size_t)N + n], mu = (double)p.mean[n], h = p.S1_in[n] / p.M, a = p.A[k];
            p.P[i] = (float)(c1 * ((double)p.P[i] - h * a - q * ((double)p.WG[i] - mu * a)));
            p.WA[(size_t)k * N + n] = f2bf((float)(c1 * (double)p.Wf[i]));
        }
    } else if (job == 1) {
        const long total = (long)K * K;
        for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) p.WB[i] = f2bf(-p.Qf[i]);
    } else {
        const int jl = tid & 15, nl = tid >> 4;
        for (int j0 = blockIdx.x * 16; j0 < K; j0 += gridDim.x * 16) {
            double r = 0.0;
#pragma unroll 4
            for (int n = nl; n < N; n += 16) r += p.coef[2 * (size_t)N + n] * (double)p.Wf[(size_t)n * K + j0 + jl];
            red[tid] = r;
            __syncthreads();
            if (nl == 0) {
                for (int q = 1; q < 16; ++q) r += red[q * 16 + jl];
                p.shiftA[j0 + jl] = (float)(-r);
            }
            __syncthreads();
        }
    }
}

inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

// W G on the float32 matrix cores: WG[n][k] = sum_kk Wf[n][kk] G[k][kk]   (G is symmetric: its rows are the GEMM's packed weights [K][K])
int gemm_wg(const float* Wf, const float* G, float* WG, int N, int K, const float* ones, const float* zeros, hipStream_t s)
{
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = Wf; d.w = G; d.scale = ones; d.shift = zeros; d.res = nullptr; d.y = WG;
    d.B = 1; d.Hi = 1; d.Wi = N; d.Cin = K; d.Cout = K; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1; d.Ho = 1; d.Wo = N; d.relu = 0; d.ldy = K;
    return hn_launch_conv(d, s);
}

}  // namespace

// Per-unit storage kept from the forward to the backward (floats): G [K][K] | A [K] doubles | Wf [N][K] | WG [N][K]
size_t hn_bn_fold_keep_floats(int N, int K) { return (size_t)K * K + 2 * (size_t)K + 2 * (size_t)N * K; }
static float* keep_G(float* keep, int K) { (void)K; return keep; }
static double* keep_A(float* keep, int K) { return reinterpret_cast<double*>(keep + (size_t)K * K); }
static float* keep_Wf(float* keep, int N, int K) { (void)N; return keep + (size_t)K * K + 2 * (size_t)K; }
static float* keep_WG(float* keep, int N, int K) { return keep + (size_t)K * K + 2 * (size_t)K + (size_t)N * K; }

// Backward scratch of one unit behind `ws` (bytes): Qf floats (the caller zeroes this head: hn_bn_fold_zero_bytes) + eW floats + coef doubles +
// WA, WB bf16 + shiftA floats.
size_t hn_bn_fold_zero_bytes(int K) { return al256((size_t)K * K * sizeof(float)); }
size_t hn_bn_fold_scratch_bytes(int N, int K)
{
    return hn_bn_fold_zero_bytes(K) + al256((size_t)N * K * sizeof(float)) + al256(4 * (size_t)N * sizeof(double)) + al256((size_t)K * N * 2) +
           al256((size_t)K * K * 2) + al256((size_t)K * sizeof(float));
}

// G = a^T a and A = column sums of the activation a [M][K] (bf16) into the unit's keep storage.  slab (slab_floats of scratch): the run-to-run
// reproducible form (partial tiles per row range, added in order in double precision) -- the training FORWARD's batch statistics come from
// G and A, and float atomics in arrival order would make every forward differ in the last bits of every BatchNorm; null: atomics (adjoint-only use).
int hn_launch_bn_fold_gram(const void* a_h, long M, int K, float* keep, hipStream_t s, float* slab, size_t slab_floats)
{
    if (!slab) HN_HIP(hipMemsetAsync(keep, 0, ((size_t)K * K + 2 * (size_t)K) * sizeof(float), s));
    return hn_launch_conv_wgrad_bf16_fold(a_h, const_cast<void*>(a_h), keep_G(keep, K), M, K, K, nullptr, keep_A(keep, K), s, 1, slab, slab_floats);
}

// where hn_launch_bn_fold_finish keeps WA inside its scratch `ws` (so that it can be filled ahead of the P-GEMM)
void* hn_bn_fold_wa_ptr(void* ws, int N, int K)
{
    return reinterpret_cast<char*>(ws) + hn_bn_fold_zero_bytes(K) + al256((size_t)N * K * sizeof(float)) + al256(4 * (size_t)N * sizeof(double));
}

int hn_launch_bn_fold_wa(const void* w_h, const float* gamma, const float* invstd, void* wa, int N, int K, hipStream_t s)
{
    const long nk = (long)N * K;
    long g = (nk + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(bn_fold_wa_kernel, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const u16*>(w_h), gamma, invstd, reinterpret_cast<u16*>(wa), N, K);
    HN_LAUNCH_CHECK();
    return 0;
}

// Forward of a folded unit, between hn_launch_bn_fold_gram and the fused conv: Wf, WG into `keep`, then the BatchNorm affine (a_out / b_out =
// the fused conv's scale / shift), the saved mean / invstd and the running-statistics update.
int hn_launch_bn_fold_forward_stats(float* keep, const void* w_h, double M, int N, int K, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float* a_out, float* b_out, float* save_mean, float* save_invstd,
                                    const float* ones, const float* zeros, hipStream_t s)
{
    HN_REQUIRE(N % 64 == 0 && K % 64 == 0 && N > 0 && K > 0, "bn fold stats: N=%d K=%d", N, K);
    float* Wf = keep_Wf(keep, N, K);
    float* WG = keep_WG(keep, N, K);
    const long nk = (long)N * K;
    long g = (nk + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(bn_fold_widen_kernel, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const u16*>(w_h), Wf, nk);
    HN_LAUNCH_CHECK();
    if (int rc = gemm_wg(Wf, keep_G(keep, K), WG, N, K, ones, zeros, s)) return rc;
    hipLaunchKernelGGL(bn_fold_stats_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, Wf, WG, keep_A(keep, K), M, N, K, gamma, beta, running_mean,
                       running_var, momentum, a_out, b_out, save_mean, save_invstd);
    HN_LAUNCH_CHECK();
    return 0;
}

// P (in place -> dW); keep: G / A as left by hn_launch_bn_fold_gram (+ Wf / WG when wg_ready: the forward was folded); ws: scratch with a
// zeroed head; w_h: the unit's packed bf16 forward weights [N][K].  Outputs: S2 / S1_out (the BatchNorm weight / bias gradient sums), dW over
// P, and in ws the two data-gradient weight matrices + shift (returned through wa / shift_a / wb).  ones / zeros: >= K floats of 1.0 / 0.0.
int hn_launch_bn_fold_finish(float* P, float* keep, int wg_ready, const double* S1_in, double* S1_out, double* S2, const void* w_h, const float* mean,
                             const float* invstd, const float* gamma, double M, int N, int K, void* ws, const float* ones, const float* zeros,
                             const void** wa, const float** shift_a, const void** wb, hipStream_t s)
{
    HN_REQUIRE(N % 64 == 0 && K % 64 == 0 && N > 0 && K > 0 && K <= 4096, "bn fold: N=%d K=%d", N, K);
    FoldArgs p;
    char* q = reinterpret_cast<char*>(ws);
    float* Qf = reinterpret_cast<float*>(q); q += hn_bn_fold_zero_bytes(K);
    p.eW = reinterpret_cast<float*>(q); q += al256((size_t)N * K * sizeof(float));
    p.coef = reinterpret_cast<double*>(q); q += al256(4 * (size_t)N * sizeof(double));
    p.WA = reinterpret_cast<u16*>(q); q += al256((size_t)K * N * 2);
    p.WB = reinterpret_cast<u16*>(q); q += al256((size_t)K * K * 2);
    p.shiftA = reinterpret_cast<float*>(q);
    p.A = keep_A(keep, K);
    p.Wf = keep_Wf(keep, N, K);
    float* WG = keep_WG(keep, N, K);
    p.P = P; p.S1_in = S1_in; p.S1_out = S1_out; p.S2 = S2; p.Wh = reinterpret_cast<const u16*>(w_h);
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.M = M; p.N = N; p.K = K; p.WG = WG; p.Qf = Qf;
    *wa = p.WA; *shift_a = p.shiftA; *wb = p.WB;
    hipLaunchKernelGGL(bn_fold_coef_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, p);
    HN_LAUNCH_CHECK();
    if (!wg_ready)
        if (int rc = gemm_wg(p.Wf, keep_G(keep, K), WG, N, K, ones, zeros, s)) return rc;
    // Q[k][j] = sum_n eW[n][k] Wf[n][j]   (the float32 weight-gradient GEMM with the channel index n as its reduction index)
    // (at most two row splits: two float partials commute, so Q -- and with it the data gradient of every unit upstream -- repeats bit for bit)
    if (int rc = hn_launch_conv_wgrad(p.Wf, p.eW, Qf, 1, 1, N, K, K, 1, 1, 1, 1, 0, 0, 0, s, /*prezeroed=*/1, /*max_split=*/2)) return rc;
    const long nk = (long)N * K;
    long gx = (nk + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(bn_fold_finish_kernel, dim3((unsigned)gx, 3), dim3(256), 0, s, p);
    HN_LAUNCH_CHECK();
    return 0;
}
