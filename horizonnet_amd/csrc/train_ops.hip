// Training-mode helper kernels of the HorizonNet step (reference train.py:44-58,272-281 drives
// autograd through model.py:254-281): batch-statistics BatchNorm forward/backward, ReLU masks,
// column reductions, max-pool / up-sample adjoints, dropout, the Linear head's adjoint and the
// per-step LSTM gate adjoint.  All HBM-bound, NHWC / row-major float32, float4 vectorised; column
// reductions accumulate per-thread partials in f32 and combine across workgroups with f64 atomics.
#include "hn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Column reductions over a row-major [M][C] matrix.  MODE 0: sum(a), sum(a*a)           (BN statistics, bias grads)
//                                                    MODE 1: sum(g), sum(g * (z - mean) * invstd)   with g = dy * (ReLU mask | 1)
// ------------------------------------------------------------------------------------------------
// 4 consecutive z values starting at element index e: float32, or bf16 (train_precision bf16 stores the conv output z as bf16)
__device__ __forceinline__ f32x4 load_z4(const float* z, long e, int z_bf16)
{
    if (!z_bf16) return *reinterpret_cast<const f32x4*>(z + e);
    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(z) + e);
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u), __builtin_bit_cast(float, v.y << 16),
                 __builtin_bit_cast(float, v.y & 0xffff0000u)};
}

template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ a, const unsigned char* __restrict__ bmask,
                                                         const float* __restrict__ z, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, double* __restrict__ out0,
                                                         double* __restrict__ out1, long M, int C, int lda, int slab, int z_bf16,
                                                         int a_bf16)
{
    __shared__ float red[2][256 * 4];
    const int CQ = (C < 256 ? C : 256) / 4;      // column quads per workgroup
    const int RL = 256 / CQ;                     // row lanes
    const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
    const int col = blockIdx.y * 256 + cq * 4;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(mean + col);
        is = *reinterpret_cast<const f32x4*>(invstd + col);
    }
    if (rl < RL) {
        // 4 rows per trip: 4 (MODE 0) or 8-12 (MODE 1) independent 16-byte loads in flight per lane
        auto one = [&](long r, f32x4& a0, f32x4& a1) {
            f32x4 v = load_z4(a, r * lda + col, a_bf16);          // (a_bf16: the gradient tensor is kept in bf16, train.hip)
            if (MODE == 0) {
                a0 += v;
                a1 += v * v;
            } else {
                const f32x4 zz = load_z4(z, r * (long)C + col, z_bf16);
                if (bmask) {                 // ReLU mask of the forward pass, 4 bits per float4 (affine_act_kernel)
                    const unsigned mk = bmask[(r * (long)C + col) >> 2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (mk >> k) & 1u ? v[k] : 0.f;
                }
                a0 += v;
                a1 += v * ((zz - mu) * is);
            }
        };
        f32x4 p0 = s0, p1 = s1, q0 = s0, q1 = s1, w0 = s0, w1 = s1;
        long r = r0 + rl;
        for (; r + 3L * RL < r1; r += 4L * RL) {
            one(r, s0, s1);
            one(r + RL, p0, p1);
            one(r + 2L * RL, q0, q1);
            one(r + 3L * RL, w0, w1);
        }
        for (; r < r1; r += RL) one(r, s0, s1);
        s0 += p0 + q0 + w0;
        s1 += p1 + q1 + w1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][threadIdx.x * 4 + k] = s0[k];
        red[1][threadIdx.x * 4 + k] = s1[k];
    }
    __syncthreads();
    if (threadIdx.x < CQ * 4) {
        const int q = threadIdx.x / 4, k = threadIdx.x % 4;
        float t0 = 0.f, t1 = 0.f;
        for (int j = 0; j < RL; ++j) {
            t0 += red[0][(j * CQ + q) * 4 + k];
            t1 += red[1][(j * CQ + q) * 4 + k];
        }
        const int c = blockIdx.y * 256 + q * 4 + k;
        atomicAdd(out0 + c, (double)t0);
        if (out1) atomicAdd(out1 + c, (double)t1);
    }
}

// BN(train) statistics -> per-channel affine + saved mean / invstd + running-stat update (momentum, unbiased var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* sum, const double* sumsq, double n, const float* gamma,
                                                          const float* beta, float* running_mean, float* running_var,
                                                          float momentum, float* a, float* b, float* save_mean,
                                                          float* save_invstd, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sum[c] / n;
    double var = sumsq[c] / n - m * m;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    const float aa = gamma[c] * invstd;
    a[c] = aa;
    b[c] = beta[c] - (float)m * aa;
    save_mean[c] = (float)m;
    save_invstd[c] = invstd;
    if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// BN(eval) inside a training step (frozen block): the affine and the "saved" mean / invstd come from the running statistics
__global__ __launch_bounds__(256) void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* running_mean,
                                                             const float* running_var, float* a, float* b, float* save_mean,
                                                             float* save_invstd, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + 1e-5f);
    const float aa = gamma[c] * invstd;
    a[c] = aa;
    b[c] = beta[c] - running_mean[c] * aa;
    save_mean[c] = running_mean[c];
    save_invstd[c] = invstd;
}

// y = act(z * a[c] + b[c] (+ res))
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ z, const float* __restrict__ a,
                                                         const float* __restrict__ b, const float* __restrict__ res,
                                                         float* __restrict__ y, unsigned char* __restrict__ bmask,
                                                         unsigned short* __restrict__ y_h, long total4, int C4, int relu, int z_bf16,
                                                         int res_bf16)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = load_z4(z, i * 4, z_bf16);
        v = v * *reinterpret_cast<const f32x4*>(a + c) + *reinterpret_cast<const f32x4*>(b + c);
        if (res) v += load_z4(res, i * 4, res_bf16);     // identity branch: float32, or the bf16 copy (train_precision bf16)
        if (relu) {
            unsigned mk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mk |= (v[k] > 0.f ? 1u : 0u) << k;
                v[k] = fmaxf(v[k], 0.f);
            }
            if (bmask) bmask[i] = (unsigned char)mk;     // the adjoint reads 1 byte instead of 16 (y > 0)
        }
        if (y) *reinterpret_cast<f32x4*>(y + i * 4) = v;    // (null in bf16 mode where only the bf16 copy is ever read)
        if (y_h) {                   // bf16 copy for the next conv's matrix-core operand (train_precision bf16)
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
            *reinterpret_cast<uint2*>(y_h + i * 4) = make_uint2(lo, hi);
        }
    }
}

// bn_finalize_kernel + affine_act_kernel in one launch (the training forward's 69 units each had a ~5 us finalize launch
// between the conv that produced the sums and this pass): every workgroup derives the affine of ALL channels from the
// double sums into LDS with bn_finalize_kernel's arithmetic (so a / b are the same bits everywhere), workgroup 0 also stores
// a / b / mean / invstd for the adjoint and updates the running statistics; then the element pass reads a / b from LDS.
__global__ __launch_bounds__(256) void affine_act_bn_kernel(const float* __restrict__ z, const double* __restrict__ sum,
                                                            const double* __restrict__ sumsq, double n, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* running_mean, float* running_var,
                                                            float momentum, float* a_out, float* b_out, float* save_mean,
                                                            float* save_invstd, const float* __restrict__ res, float* __restrict__ y,
                                                            unsigned char* __restrict__ bmask, unsigned short* __restrict__ y_h,
                                                            long total4, int C4, int relu, int z_bf16, int res_bf16)
{
    extern __shared__ __attribute__((aligned(16))) float ab[];      // [2][C]
    const int C = C4 * 4;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double m = sum[c] / n;
        double var = sumsq[c] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + 1e-5));
        const float aa = gamma[c] * invstd;
        const float bb = beta[c] - (float)m * aa;
        ab[c] = aa;
        ab[C + c] = bb;
        if (blockIdx.x == 0) {
            a_out[c] = aa;
            b_out[c] = bb;
            save_mean[c] = (float)m;
            save_invstd[c] = invstd;
            if (running_mean) {
                const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = load_z4(z, i * 4, z_bf16);
        v = v * *reinterpret_cast<const f32x4*>(ab + c) + *reinterpret_cast<const f32x4*>(ab + C + c);
        if (res) v += load_z4(res, i * 4, res_bf16);
        if (relu) {
            unsigned mk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mk |= (v[k] > 0.f ? 1u : 0u) << k;
                v[k] = fmaxf(v[k], 0.f);
            }
            if (bmask) bmask[i] = (unsigned char)mk;
        }
        if (y) *reinterpret_cast<f32x4*>(y + i * 4) = v;
        if (y_h) {
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
            *reinterpret_cast<uint2*>(y_h + i * 4) = make_uint2(lo, hi);
        }
    }
}

// The stem of the bf16 training forward: affine_act_bn_kernel + maxpool_fwd_idx_kernel in one pass (model.py:73-76).  The two-pass form
// writes the 1.07 GB bf16 activation (B = 64) only for the pool to read it back; here a thread owns one pooled pixel x 4 channels, forms
// relu(a z + b) of its 3 x 3 window from z (bf16 rounding of every value as the stored activation had it, so the maximum and its position
// are the two-pass form's bit for bit), writes the pooled value + position word, and the ReLU bit masks of the 2 x 2 input pixels it owns
// (rows 2 ho, 2 ho + 1, columns 2 wo, 2 wo + 1: every input pixel has exactly one owner).  z 1.07 GB read, 0.27 + 0.13 + 0.13 GB
// written instead of 3.75 GB moved.
__global__ __launch_bounds__(256) void affine_act_bn_pool_kernel(const unsigned short* __restrict__ z, const double* __restrict__ sum,
                                                                 const double* __restrict__ sumsq, double n, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* running_mean, float* running_var,
                                                                 float momentum, float* a_out, float* b_out, float* save_mean, float* save_invstd,
                                                                 unsigned char* __restrict__ bmask, unsigned short* __restrict__ pool_h,
                                                                 unsigned* __restrict__ idx, int Hi, int Wi, int C4, long total)
{
    extern __shared__ __attribute__((aligned(16))) float ab[];      // [2][C]
    const int C = C4 * 4;
    for (int c = threadIdx.x; c < C; c += 256) {                    // bn_finalize_kernel's arithmetic, as in affine_act_bn_kernel
        const double m = sum[c] / n;
        double var = sumsq[c] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + 1e-5));
        const float aa = gamma[c] * invstd;
        const float bb = beta[c] - (float)m * aa;
        ab[c] = aa;
        ab[C + c] = bb;
        if (blockIdx.x == 0) {
            a_out[c] = aa;
            b_out[c] = bb;
            save_mean[c] = (float)m;
            save_invstd[c] = invstd;
            if (running_mean) {
                const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    const int Ho = Hi / 2, Wo = Wi / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long b = t / Ho;
        const f32x4 aa = *reinterpret_cast<const f32x4*>(ab + c4 * 4), bb = *reinterpret_cast<const f32x4*>(ab + C + c4 * 4);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned pos[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * 2 - 1 + dh;
            if ((unsigned)hi >= (unsigned)Hi) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wo * 2 - 1 + dw;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                const long e = ((b * Hi + hi) * Wi + wi) * (long)C4 + c4;
                f32x4 v = load_z4(reinterpret_cast<const float*>(z), e * 4, 1);
                v = v * aa + bb;
                unsigned mk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    mk |= (v[k] > 0.f ? 1u : 0u) << k;
                    v[k] = fmaxf(v[k], 0.f);
                }
                if (dh >= 1 && dw >= 1) bmask[e] = (unsigned char)mk;          // the 2 x 2 block this thread owns
                unsigned lo, hi2;                                              // the activation as the two-pass form stored it
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi2) : "v"(v[2]), "v"(v[3]));
                const f32x4 r = {__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                                 __builtin_bit_cast(float, hi2 << 16), __builtin_bit_cast(float, hi2 & 0xffff0000u)};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (r[k] > m[k]) { m[k] = r[k]; pos[k] = (unsigned)(dh * 3 + dw); }
            }
        }
        unsigned lo, hi2;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(m[0]), "v"(m[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi2) : "v"(m[2]), "v"(m[3]));
        *reinterpret_cast<uint2*>(pool_h + i * 4) = make_uint2(lo, hi2);
        idx[i] = pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24);
    }
}

// BN backward apply: g = dy * mask;  dz = gamma*invstd * (g - S1/N - zhat * S2/N);  optional dpre = g (identity branch)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ bmask,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ S1, const double* __restrict__ S2, double n,
                                                           float* __restrict__ dz, float* __restrict__ dpre,
                                                           unsigned short* __restrict__ dz_h, long total4, int C4, int z_bf16,
                                                           int dy_bf16)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 g = load_z4(dy, i * 4, dy_bf16);
        if (bmask) {
            const unsigned mk = bmask[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = (mk >> k) & 1u ? g[k] : 0.f;
        }
        const f32x4 zz = load_z4(z, i * 4, z_bf16);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float is = invstd[c + k];
            const float zh = (zz[k] - mean[c + k]) * is;
            const float m1 = (float)(S1[c + k] / n), m2 = (float)(S2[c + k] / n);
            o[k] = gamma[c + k] * is * (g[k] - m1 - zh * m2);
        }
        if (dz) *reinterpret_cast<f32x4*>(dz + i * 4) = o;       // null: only the bf16 copy is consumed (bf16 wgrad + dgrad)
        if (dz_h) {                  // bf16 copy for the data-gradient GEMM's matrix-core operand
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(o[0]), "v"(o[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(o[2]), "v"(o[3]));
            *reinterpret_cast<uint2*>(dz_h + i * 4) = make_uint2(lo, hi);
        }
        if (dpre) {                  // identity-branch gradient, in the gradient tensors' storage type
            if (dy_bf16) {
                unsigned lo, hi;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(g[0]), "v"(g[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(g[2]), "v"(g[3]));
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dpre) + i * 4) = make_uint2(lo, hi);
            } else {
                *reinterpret_cast<f32x4*>(dpre + i * 4) = g;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same two BatchNorm-adjoint passes for train_precision bf16, where dy, z, dz and the identity gradient are all bf16:
// EIGHT channels per lane (16-byte loads, 2-byte mask load), a lane keeps its channels for the whole row slab so every
// per-channel constant is computed once, and the apply pass is two FMAs per element (no f64 division in the loop).
//   workgroup = CO column octets x RL row lanes (CO = min(C / 8, 256)), grid = (row slabs, 2048-channel chunks).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(const u32x4_t v, float (&o)[8])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = __builtin_bit_cast(float, v[k] << 16);
        o[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
    }
}

__device__ __forceinline__ u32x4_t pack8(const float (&o)[8])
{
    u32x4_t r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned t;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(o[2 * k]), "v"(o[2 * k + 1]));
        r[k] = t;
    }
    return r;
}

// mask bytes hold 4 bits each (affine_act_kernel: one byte per float4) -> 8 channels = 2 bytes
__device__ __forceinline__ unsigned mask8(const unsigned char* bmask, long e)
{
    const unsigned m = *reinterpret_cast<const unsigned short*>(bmask + (e >> 2));
    return (m & 0xfu) | ((m >> 4) & 0xf0u);
}

// POOL (the stem): dy is not a stored tensor but the max-pool adjoint of the pooled gradient, formed on the fly -- the sum, in
// maxpool_bwd_idx_kernel's order and with its bf16 rounding, of the pooled gradients of the (up to four) 3x3 / 2 windows whose maximum sat at
// this input pixel.  Saves writing + twice reading the 1.07 GB (B = 64) gradient of the stem activation.
struct PoolSrc {
    const unsigned short* dpool;     // [B][Hi/2][Wi/2][C] bf16
    const unsigned* pidx;            // position words, one per 4 channels
    int Hi, Wi;
};
__device__ __forceinline__ u32x4_t pool_grad8(const PoolSrc ps, long r, int col, int C)
{
    const int Ho = ps.Hi / 2, Wo = ps.Wi / 2;
    const int wi = (int)(r % ps.Wi);
    long t = r / ps.Wi;
    const int hi = (int)(t % ps.Hi);
    const long b = t / ps.Hi;
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
        if (ho >= Ho) continue;
        for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
            if (wo >= Wo) continue;
            const long o = (b * Ho + ho) * Wo + wo;
            const uint2 a = *reinterpret_cast<const uint2*>(ps.pidx + (o * C + col) / 4);
            const unsigned me = (unsigned)((hi - (2 * ho - 1)) * 3 + (wi - (2 * wo - 1)));
            float d[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(ps.dpool + o * C + col), d);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((a.x >> (8 * k)) & 255u) == me) g[k] += d[k];
                if (((a.y >> (8 * k)) & 255u) == me) g[4 + k] += d[4 + k];
            }
        }
    }
    return pack8(g);
}

template <bool MASK, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_h8_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ bmask,
                                                               const unsigned short* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, double* __restrict__ S1,
                                                               double* __restrict__ S2, long M, int C, int slab, PoolSrc ps = PoolSrc{})
{
    __shared__ float red[2][256 * 8];
    const int C8 = C / 8;
    const int CO = C8 < 256 ? C8 : 256;
    const int RL = 256 / CO;
    const int co = threadIdx.x % CO, rl = threadIdx.x / CO;
    const int col = blockIdx.y * 2048 + co * 8;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    float za[8], zb[8], s0[8], s1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        za[k] = invstd[col + k];                  // zhat = z * za + zb
        zb[k] = -mean[col + k] * za[k];
        s0[k] = 0.f;
        s1[k] = 0.f;
    }
    auto accum = [&](const u32x4_t gv, const u32x4_t zv, unsigned mk) __attribute__((always_inline)) {
        float g[8], zz[8];
        unpack8(gv, g);
        unpack8(zv, zz);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gk = (!MASK || ((mk >> k) & 1u)) ? g[k] : 0.f;
            s0[k] += gk;
            s1[k] += gk * (zz[k] * za[k] + zb[k]);
        }
    };
    long r = r0 + rl;
    for (; r + 3L * RL < r1; r += 4L * RL) {       // 4 rows per trip: 8 sixteen-byte loads in flight per lane
        u32x4_t gv[4], zv[4];
        unsigned mk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = (r + (long)j * RL) * C + col;
            gv[j] = POOL ? pool_grad8(ps, r + (long)j * RL, col, C) : *reinterpret_cast<const u32x4_t*>(dy + e);
            zv[j] = *reinterpret_cast<const u32x4_t*>(z + e);
            if (MASK) mk[j] = mask8(bmask, e);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) accum(gv[j], zv[j], mk[j]);
    }
    for (; r < r1; r += RL) {
        const long e = r * C + col;
        accum(POOL ? pool_grad8(ps, r, col, C) : *reinterpret_cast<const u32x4_t*>(dy + e), *reinterpret_cast<const u32x4_t*>(z + e),
              MASK ? mask8(bmask, e) : 0u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][threadIdx.x * 8 + k] = s0[k];
        red[1][threadIdx.x * 8 + k] = s1[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CO * 8; t += 256) {
        const int q = t / 8, k = t % 8;
        float t0 = 0.f, t1 = 0.f;
        for (int j = 0; j < RL; ++j) {
            t0 += red[0][(j * CO + q) * 8 + k];
            t1 += red[1][(j * CO + q) * 8 + k];
        }
        const int c = blockIdx.y * 2048 + q * 8 + k;
        atomicAdd(S1 + c, (double)t0);
        atomicAdd(S2 + c, (double)t1);
    }
}

// BIAS: also accumulate the column sums of dz (the conv bias gradient of the height-compression convs) into db
template <bool MASK, bool BIAS, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_h8_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ bmask,
                                                              const unsigned short* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const double* __restrict__ S1, const double* __restrict__ S2, double n,
                                                              unsigned short* __restrict__ dpre, unsigned short* __restrict__ dz_h,
                                                              double* __restrict__ db, long M, int C, int slab, PoolSrc ps = PoolSrc{})
{
    __shared__ float red[BIAS ? 256 * 8 : 1];
    const int C8 = C / 8;
    const int CO = C8 < 256 ? C8 : 256;
    const int RL = 256 / CO;
    const int co = threadIdx.x % CO, rl = threadIdx.x / CO;
    const int col = blockIdx.y * 2048 + co * 8;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    // dz = gamma*invstd * (g - S1/n - zhat * S2/n) = ka * g + kz * z + kc   (the same f32 roundings of S1/n, S2/n as the generic kernel)
    float ka[8], kz[8], kc[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sb[k] = 0.f;
        const float is = invstd[col + k], mu = mean[col + k];
        const float m1 = (float)(S1[col + k] / n), m2 = (float)(S2[col + k] / n);
        ka[k] = gamma[col + k] * is;
        kz[k] = -ka[k] * is * m2;
        kc[k] = -ka[k] * m1 - kz[k] * mu;
    }
    auto one = [&](long e, const u32x4_t gv, const u32x4_t zv, unsigned mk) __attribute__((always_inline)) {
        float g[8], zz[8], o[8];
        unpack8(gv, g);
        unpack8(zv, zz);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MASK) g[k] = ((mk >> k) & 1u) ? g[k] : 0.f;
            o[k] = ka[k] * g[k] + (kz[k] * zz[k] + kc[k]);
            if (BIAS) sb[k] += o[k];
        }
        *reinterpret_cast<u32x4_t*>(dz_h + e) = pack8(o);
        if (dpre) *reinterpret_cast<u32x4_t*>(dpre + e) = MASK ? pack8(g) : gv;
    };
    long r = r0 + rl;
    for (; r + 3L * RL < r1; r += 4L * RL) {
        u32x4_t gv[4], zv[4];
        unsigned mk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = (r + (long)j * RL) * C + col;
            gv[j] = POOL ? pool_grad8(ps, r + (long)j * RL, col, C) : *reinterpret_cast<const u32x4_t*>(dy + e);
            zv[j] = *reinterpret_cast<const u32x4_t*>(z + e);
            if (MASK) mk[j] = mask8(bmask, e);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) one((r + (long)j * RL) * C + col, gv[j], zv[j], mk[j]);
    }
    for (; r < r1; r += RL) {
        const long e = r * C + col;
        one(e, POOL ? pool_grad8(ps, r, col, C) : *reinterpret_cast<const u32x4_t*>(dy + e), *reinterpret_cast<const u32x4_t*>(z + e),
            MASK ? mask8(bmask, e) : 0u);
    }
    if (BIAS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = sb[k];
        __syncthreads();
        for (int t = threadIdx.x; t < CO * 8; t += 256) {
            const int q = t / 8, k = t % 8;
            float t0 = 0.f;
            for (int j = 0; j < RL; ++j) t0 += red[(j * CO + q) * 8 + k];
            atomicAdd(db + blockIdx.y * 2048 + q * 8 + k, (double)t0);
        }
    }
}

// ---- block 0 of a ResNet stage: bn3 and the downsample branch's BatchNorm see the SAME masked gradient g = dOut * relu'(block output) ----
// (model.py:78-81: out = relu(bn3(conv3(t2)) + bn_d(downsample(x)))).  As two adjoints the second one read the masked gradient the
// first one had written (dpre); here ONE reduce pass and ONE apply pass serve both: dOut and the mask are read once per pass, dpre is
// not written at all (nothing else reads it in block 0).  Same per-element arithmetic as bn_bwd_{reduce,apply}_h8_kernel.
__global__ __launch_bounds__(256) void bn_bwd_reduce_dual_h8_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ bmask,
                                                                    const unsigned short* __restrict__ za_, const float* __restrict__ mean_a,
                                                                    const float* __restrict__ invstd_a, const unsigned short* __restrict__ zb_,
                                                                    const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                                                    double* __restrict__ S1a, double* __restrict__ S2a, double* __restrict__ S1b,
                                                                    double* __restrict__ S2b, long M, int C, int slab)
{
    __shared__ float red[3][256 * 8];
    const int C8 = C / 8;
    const int CO = C8 < 256 ? C8 : 256;
    const int RL = 256 / CO;
    const int co = threadIdx.x % CO, rl = threadIdx.x / CO;
    const int col = blockIdx.y * 2048 + co * 8;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    float ia[8], oa[8], ib[8], ob[8], s0[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ia[k] = invstd_a[col + k]; oa[k] = -mean_a[col + k] * ia[k];
        ib[k] = invstd_b[col + k]; ob[k] = -mean_b[col + k] * ib[k];
        s0[k] = s1[k] = s2[k] = 0.f;
    }
    auto accum = [&](const u32x4_t gv, const u32x4_t av, const u32x4_t bv, unsigned mk) __attribute__((always_inline)) {
        float g[8], a[8], b[8];
        unpack8(gv, g);
        unpack8(av, a);
        unpack8(bv, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gk = ((mk >> k) & 1u) ? g[k] : 0.f;
            s0[k] += gk;
            s1[k] += gk * (a[k] * ia[k] + oa[k]);
            s2[k] += gk * (b[k] * ib[k] + ob[k]);
        }
    };
    long r = r0 + rl;
    for (; r + 1L * RL < r1; r += 2L * RL) {        // 2 rows per trip: 6 sixteen-byte loads in flight per lane
        u32x4_t gv[2], av[2], bv[2];
        unsigned mk[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long e = (r + (long)j * RL) * C + col;
            gv[j] = *reinterpret_cast<const u32x4_t*>(dy + e);
            av[j] = *reinterpret_cast<const u32x4_t*>(za_ + e);
            bv[j] = *reinterpret_cast<const u32x4_t*>(zb_ + e);
            mk[j] = mask8(bmask, e);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) accum(gv[j], av[j], bv[j], mk[j]);
    }
    for (; r < r1; r += RL) {
        const long e = r * C + col;
        accum(*reinterpret_cast<const u32x4_t*>(dy + e), *reinterpret_cast<const u32x4_t*>(za_ + e), *reinterpret_cast<const u32x4_t*>(zb_ + e),
              mask8(bmask, e));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][threadIdx.x * 8 + k] = s0[k];
        red[1][threadIdx.x * 8 + k] = s1[k];
        red[2][threadIdx.x * 8 + k] = s2[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CO * 8; t += 256) {
        const int q = t / 8, k = t % 8;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int j = 0; j < RL; ++j) {
            t0 += red[0][(j * CO + q) * 8 + k];
            t1 += red[1][(j * CO + q) * 8 + k];
            t2 += red[2][(j * CO + q) * 8 + k];
        }
        const int c = blockIdx.y * 2048 + q * 8 + k;
        atomicAdd(S1a + c, (double)t0);
        atomicAdd(S2a + c, (double)t1);
        atomicAdd(S1b + c, (double)t0);
        atomicAdd(S2b + c, (double)t2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_dual_h8_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ bmask,
                                                                   const unsigned short* __restrict__ za_, const float* __restrict__ mean_a,
                                                                   const float* __restrict__ invstd_a, const float* __restrict__ gamma_a,
                                                                   const double* __restrict__ S1a, const double* __restrict__ S2a,
                                                                   const unsigned short* __restrict__ zb_, const float* __restrict__ mean_b,
                                                                   const float* __restrict__ invstd_b, const float* __restrict__ gamma_b,
                                                                   const double* __restrict__ S1b, const double* __restrict__ S2b, double n,
                                                                   unsigned short* __restrict__ dza_h, unsigned short* __restrict__ dzb_h, long M,
                                                                   int C, int slab)
{
    const int C8 = C / 8;
    const int CO = C8 < 256 ? C8 : 256;
    const int RL = 256 / CO;
    const int co = threadIdx.x % CO, rl = threadIdx.x / CO;
    const int col = blockIdx.y * 2048 + co * 8;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    float ka[8], kza[8], kca[8], kb[8], kzb[8], kcb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // bn_bwd_apply_h8_kernel's coefficients, per branch
        {
            const float is = invstd_a[col + k], mu = mean_a[col + k];
            const float m1 = (float)(S1a[col + k] / n), m2 = (float)(S2a[col + k] / n);
            ka[k] = gamma_a[col + k] * is;
            kza[k] = -ka[k] * is * m2;
            kca[k] = -ka[k] * m1 - kza[k] * mu;
        }
        {
            const float is = invstd_b[col + k], mu = mean_b[col + k];
            const float m1 = (float)(S1b[col + k] / n), m2 = (float)(S2b[col + k] / n);
            kb[k] = gamma_b[col + k] * is;
            kzb[k] = -kb[k] * is * m2;
            kcb[k] = -kb[k] * m1 - kzb[k] * mu;
        }
    }
    for (long r = r0 + rl; r < r1; r += RL) {
        const long e = r * C + col;
        float g[8], a[8], b[8], oa[8], ob[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(dy + e), g);
        unpack8(*reinterpret_cast<const u32x4_t*>(za_ + e), a);
        unpack8(*reinterpret_cast<const u32x4_t*>(zb_ + e), b);
        const unsigned mk = mask8(bmask, e);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gk = ((mk >> k) & 1u) ? g[k] : 0.f;
            oa[k] = ka[k] * gk + (kza[k] * a[k] + kca[k]);
            ob[k] = kb[k] * gk + (kzb[k] * b[k] + kcb[k]);
        }
        *reinterpret_cast<u32x4_t*>(dza_h + e) = pack8(oa);
        *reinterpret_cast<u32x4_t*>(dzb_h + e) = pack8(ob);
    }
}

__global__ __launch_bounds__(256) void d2f_kernel(const double* in, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<f32x4*>(y + i * 4) += *reinterpret_cast<const f32x4*>(x + i * 4);
}

// y += x on bf16 tensors (f32 add, one rounding)
__global__ __launch_bounds__(256) void axpy_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = load_z4(reinterpret_cast<const float*>(x), i * 4, 1) + load_z4(reinterpret_cast<const float*>(y), i * 4, 1);
        unsigned lo, hi;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
        *reinterpret_cast<uint2*>(y + i * 4) = make_uint2(lo, hi);
    }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<f32x4*>(out + i * 4) = load_z4(reinterpret_cast<const float*>(in), i * 4, 1);
}

// max-pool 3x3/2 pad 1 in the training step (torchvision ResNet.maxpool, reference model.py:76) and its adjoint.  The forward
// pass also records WHICH of the 9 window positions won (first maximum in row-major window order, as torch: one byte per
// output), so the adjoint is ONE deterministic gather pass with 16-byte stores: per input pixel, the <= 4 windows that contain
// it.  (It used to be an arg-max pass over the 1 GB float32 stem output plus a scalar gather: 0.70 + 1.41 ms of a 60 ms step.)
// in_bf16: `in` is a bf16 tensor; out_h (optional): the pooled tensor as bf16 (out may then be null) -- the bf16 training step
// pools the stem's bf16 activation directly (same pooled values as rounding the float32 maximum: rounding is monotonic)
__global__ __launch_bounds__(256) void maxpool_fwd_idx_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              unsigned* __restrict__ idx, int Hi, int Wi, int Ho, int Wo, int C4, long total,
                                                              int in_bf16, unsigned short* __restrict__ out_h)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long b = t / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned pos[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = ho * 2 - 1 + dh;
            if ((unsigned)hi >= (unsigned)Hi) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wo * 2 - 1 + dw;
                if ((unsigned)wi >= (unsigned)Wi) continue;
                const f32x4 v = load_z4(in, (((b * Hi + hi) * Wi + wi) * (long)C4 + c4) * 4, in_bf16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > m[k]) { m[k] = v[k]; pos[k] = (unsigned)(dh * 3 + dw); }
            }
        }
        if (out) *reinterpret_cast<f32x4*>(out + i * 4) = m;
        if (out_h) {
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(m[0]), "v"(m[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(m[2]), "v"(m[3]));
            *reinterpret_cast<uint2*>(out_h + i * 4) = make_uint2(lo, hi);
        }
        idx[i] = pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned* __restrict__ idx, const float* __restrict__ dout,
                                                              float* __restrict__ din, int Hi, int Wi, int Ho, int Wo, int C4, long total,
                                                              int dout_bf16, int din_bf16)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int wi = (int)(t % Wi);
        t /= Wi;
        const int hi = (int)(t % Hi);
        const long b = t / Hi;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {             // windows covering row hi: 2*ho-1 <= hi <= 2*ho+1
            if (ho >= Ho) continue;
            for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const long o = ((b * Ho + ho) * Wo + wo) * (long)C4 + c4;
                const unsigned a = idx[o];
                const unsigned me = (unsigned)((hi - (2 * ho - 1)) * 3 + (wi - (2 * wo - 1)));
                const f32x4 d = load_z4(dout, o * 4, dout_bf16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((a >> (8 * k)) & 255u) == me) g[k] += d[k];
            }
        }
        if (din_bf16) {          // the bf16 training step: the stem's BatchNorm adjoint reads bf16 like every other unit's
            unsigned lo, hi;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(g[0]), "v"(g[1]));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(g[2]), "v"(g[3]));
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(din) + i * 4) = make_uint2(lo, hi);
        } else {
            *reinterpret_cast<f32x4*>(din + i * 4) = g;
        }
    }
}

// adjoint of upsample_flatten: din[b][h][i][c] = sum_t w0(t)[i0(t)==i] * dseq + w1(t)[i1(t)==i] * dseq
__global__ __launch_bounds__(256) void upsample_flatten_bwd_kernel(const float* __restrict__ dseq, float* __restrict__ din, int B, int hq,
                                                                   int Wq, int cq, int col0, int f, int out_bf16)
{
    const int i = blockIdx.x;            // input column
    const int b = blockIdx.y;
    const int n = cq * hq;
    // which of the 256 output columns t touch input column i, and with which weights: computed ONCE per workgroup (one t per thread)
    // instead of by every element for every t (the float index arithmetic x 256 was the kernel: 0.63 ms per step for 17 MB of output)
    __shared__ unsigned char hit[256];
    __shared__ float wt0[256], wt1[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) {
        const float src = (1.0f / (float)f) * ((float)(t + f) + 0.5f) - 0.5f;
        const int i0p = (int)src;
        const float w1 = src - (float)i0p, w0 = 1.0f - w1;
        int i0 = i0p - 1;
        i0 = i0 < 0 ? i0 + Wq : i0;
        int i1 = i0p;
        i1 = i1 >= Wq ? i1 - Wq : i1;
        hit[t] = (unsigned char)((i0 == i ? 1 : 0) | (i1 == i ? 2 : 0));
        wt0[t] = w0;
        wt1[t] = w1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int c = e % cq, h = e / cq;
        float g = 0.f;
        for (int t = 0; t < 256; ++t) {            // same terms in the same order as before
            const unsigned m = hit[t];
            if (m == 0) continue;
            const float d = dseq[((long)t * B + b) * 1024 + col0 + c * hq + h];
            if (m & 1u) g += wt0[t] * d;
            if (m & 2u) g += wt1[t] * d;
        }
        const long o = (((long)b * hq + h) * Wq + i) * cq + c;
        if (out_bf16) {
            unsigned r;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(g), "v"(0.f));
            reinterpret_cast<unsigned short*>(din)[o] = (unsigned short)(r & 0xffffu);
        } else {
            din[o] = g;
        }
    }
}

// counter-based dropout mask: keep iff hash(seed, index) >= p * 2^32; out = in * keep / (1 - p)
__device__ __forceinline__ unsigned hash32(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (unsigned)x;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float p,
                                                      unsigned long long seed)
{
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    const float sc = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = hash32(seed * 0x9e3779b97f4a7c15ULL + (unsigned long long)i) >= thr ? in[i] * sc : 0.f;
}

// Linear head adjoint: dlin[t*B+b][12] gathered from dbon / dcor; dy[row][k] = sum_o dlin[row][o] * w[o][k]
__global__ __launch_bounds__(256) void head_bwd_dy_kernel(const float* __restrict__ dbon, const float* __restrict__ dcor,
                                                          const float* __restrict__ w, float* __restrict__ dy,
                                                          float* __restrict__ dlin, int T, int B)
{
    const long row = blockIdx.x;
    const int t = (int)(row / B), b = (int)(row % B);
    __shared__ float dl[12];
    if (threadIdx.x < 12) {
        const int ch = threadIdx.x >> 2, s = threadIdx.x & 3;
        const int W = 4 * T;
        const float v = ch == 0 ? dcor[(long)b * W + 4 * t + s] : dbon[((long)b * 2 + (ch - 1)) * W + 4 * t + s];
        dl[threadIdx.x] = v;
        dlin[row * 12 + threadIdx.x] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < 12; ++o) s += dl[o] * w[o * 1024 + k];
        dy[row * 1024 + k] = s;
    }
}
// dW[o][k] = sum_rows dlin[row][o] * y[row][k];  db[o] = sum_rows dlin[row][o].  Grid = 16 column blocks (64 k each) x
// HEAD_SLICES row slices: every workgroup reduces its rows (4 strided partitions, combined through LDS) into a partial
// [slice][12][1024] (+ [slice][12] for the bias); head_bwd_dw_reduce_kernel adds the slices in index order -- deterministic,
// and 512 workgroups instead of the 16 that used to walk all T*B rows (0.78 ms of a 60 ms step).
constexpr int HEAD_SLICES = 32;
static_assert((size_t)HEAD_SLICES * 12 * 1024 + HEAD_SLICES * 12 == HN_HEAD_BWD_SCRATCH_FLOATS, "head_bwd scratch");
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ dlin, const float* __restrict__ y,
                                                          float* __restrict__ part_w, float* __restrict__ part_b, long rows)
{
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;                 // 4 row partitions
    const int sl = blockIdx.y;
    const long r_lo = rows * sl / HEAD_SLICES, r_hi = rows * (sl + 1) / HEAD_SLICES;
    float acc[12];
#pragma unroll
    for (int o = 0; o < 12; ++o) acc[o] = 0.f;
    float bsum[12];
#pragma unroll
    for (int o = 0; o < 12; ++o) bsum[o] = 0.f;
    for (long r = r_lo + part; r < r_hi; r += 4) {
        const float yv = y[r * 1024 + k];
#pragma unroll
        for (int o = 0; o < 12; ++o) {
            const float d = dlin[r * 12 + o];
            acc[o] += d * yv;
            bsum[o] += d;
        }
    }
    __shared__ float red[4][12][64];
    __shared__ float redb[4][12];
#pragma unroll
    for (int o = 0; o < 12; ++o) red[part][o][threadIdx.x & 63] = acc[o];
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int o = 0; o < 12; ++o) redb[part][o] = bsum[o];
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int o = 0; o < 12; ++o)
            part_w[((size_t)sl * 12 + o) * 1024 + k] = red[0][o][threadIdx.x] + red[1][o][threadIdx.x] + red[2][o][threadIdx.x] + red[3][o][threadIdx.x];
    }
    if (blockIdx.x == 0 && threadIdx.x < 12)
        part_b[sl * 12 + threadIdx.x] = redb[0][threadIdx.x] + redb[1][threadIdx.x] + redb[2][threadIdx.x] + redb[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void head_bwd_dw_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                                 float* __restrict__ dw, float* __restrict__ db)
{
    const int i = blockIdx.x * 256 + threadIdx.x;      // 12 * 1024 weight entries, then 12 bias entries
    if (i < 12 * 1024) {
        float s = 0.f;
        for (int sl = 0; sl < HEAD_SLICES; ++sl) s += part_w[(size_t)sl * 12 * 1024 + i];
        dw[i] = s;
    } else if (i < 12 * 1024 + 12) {
        float s = 0.f;
        for (int sl = 0; sl < HEAD_SLICES; ++sl) s += part_b[sl * 12 + (i - 12 * 1024)];
        db[i - 12 * 1024] = s;
    }
}

// LSTM gate adjoint for ONE time index per direction (fwd at t_f, rev at t_r):
//   saved[t][b][dir][5][512] = (i, f, g, o, c) post-activation;  c_prev = c of the previous step of that direction (0 at its first step)
//   dh = dy[t][b][dir*512+u] + dh_rec[b][dir*512+u];  standard LSTM adjoint -> dgx[t][b][dir*2048 + gate*512 + u], dc_rec updated in place
__global__ __launch_bounds__(256) void lstm_bwd_gates_kernel(const float* __restrict__ saved, const float* __restrict__ dy,
                                                             float* __restrict__ dh_rec, float* __restrict__ dc_rec,
                                                             float* __restrict__ dgx, int T, int B, int step)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over B * 2 * 512
    if (idx >= B * 1024) return;
    const int u = idx & 511;
    const int dir = (idx >> 9) & 1;
    const int b = idx >> 10;
    // backward walks each direction's own time order in reverse: fwd dir at t = T-1-step, rev dir at t = step
    const int t = dir ? step : T - 1 - step;
    const int tprev = dir ? t + 1 : t - 1;                     // the step that ran BEFORE t in that direction's forward pass
    const float* sv = saved + (((size_t)t * B + b) * 2 + dir) * 5 * 512;
    const float ig = sv[u], fg = sv[512 + u], gg = sv[1024 + u], og = sv[1536 + u], c = sv[2048 + u];
    float cprev = 0.f;
    if (tprev >= 0 && tprev < T) cprev = saved[((((size_t)tprev * B + b) * 2 + dir) * 5 + 4) * 512 + u];
    const float dh = dy[((size_t)t * B + b) * 1024 + dir * 512 + u] + dh_rec[(size_t)b * 1024 + dir * 512 + u];
    dh_rec[(size_t)b * 1024 + dir * 512 + u] = 0.f;    // consumed: the next launch (lstm_bwd_dh) accumulates this step's into it with atomics
    const float tc = tanhf(c);
    const float dc = dc_rec[(size_t)b * 1024 + dir * 512 + u] + dh * og * (1.f - tc * tc);
    float* g = dgx + ((size_t)t * B + b) * 4096 + dir * 2048 + u;
    g[0] = dc * gg * ig * (1.f - ig);            // d pre-activation of i
    g[512] = dc * cprev * fg * (1.f - fg);       // f
    g[1024] = dc * ig * (1.f - gg * gg);         // g (tanh)
    g[1536] = dh * tc * og * (1.f - og);         // o
    dc_rec[(size_t)b * 1024 + dir * 512 + u] = dc * fg;
}

// Adam over ALL parameter tensors in one launch (train.py:216-225 / :279 `optimizer.step()`): thread i owns 4 consecutive
// elements of the engine's flat gradient buffer, finds the tensor they belong to in the (sorted) offset table and
// updates that tensor's parameter storage in place.  Same arithmetic as torch.optim.Adam (no amsgrad): L2 weight
// decay folded into the gradient, bias-corrected step size, denom = sqrt(v) / sqrt(1 - b2^t) + eps.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* const* __restrict__ params, const long long* __restrict__ offsets,
                                                        const long long* __restrict__ ends, const unsigned char* __restrict__ active,
                                                        int n_tensors,
                                                        const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        long long total, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, float inv_grad_scale)
{
    // A workgroup walks chunks of 4096 consecutive elements (4 x 16 bytes per lane and tensor stream).  The tensor search runs
    // once per chunk on its first element (uniform: scalar loads), lanes then step forward from there -- almost always zero
    // steps, the tensors are thousands of elements long.  Groups of 4 that lie inside one active tensor (all but the odd
    // tail: the flat layout starts every tensor on a 64-element boundary) go through 16-byte accesses.
    constexpr int CH = 4096;
    const float step_size = lr / bc1;
    auto one = [&](float gi, float pi, float& mi, float& vi) -> float {
        gi = gi * inv_grad_scale + wd * pi;
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        return pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    };
    for (long long base = (long long)blockIdx.x * CH; base < total; base += (long long)gridDim.x * CH) {
        int lo = 0, hi = n_tensors - 1;                  // last tensor whose first element is <= base
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= base) lo = mid; else hi = mid - 1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i4 = base + ((long long)u * 256 + threadIdx.x) * 4;
            if (i4 >= total) break;
            int t = lo;
            while (t + 1 < n_tensors && i4 >= offsets[t + 1]) ++t;
            const long long off = offsets[t];
            if (i4 + 3 < ends[t] && ((i4 - off) & 3) == 0) {
                if (!active[t]) continue;
                float* p = params[t] + (i4 - off);
                if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i4);
                    f32x4 pv = *reinterpret_cast<const f32x4*>(p);
                    f32x4 mv = *reinterpret_cast<const f32x4*>(m + i4);
                    f32x4 vv = *reinterpret_cast<const f32x4*>(v + i4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float mi = mv[k], vi = vv[k];
                        pv[k] = one(gv[k], pv[k], mi, vi);
                        mv[k] = mi;
                        vv[k] = vi;
                    }
                    *reinterpret_cast<f32x4*>(m + i4) = mv;
                    *reinterpret_cast<f32x4*>(v + i4) = vv;
                    *reinterpret_cast<f32x4*>(p) = pv;
                    continue;
                }
            }
            for (int k = 0; k < 4; ++k) {                // tails, gaps, unaligned storage: element by element
                const long long i = i4 + k;
                if (i >= total) break;
                while (t + 1 < n_tensors && i >= offsets[t + 1]) ++t;
                if (i >= ends[t] || !active[t]) continue;   // alignment gap between two tensors of the flat layout / frozen tensor
                float* p = params[t] + (i - offsets[t]);
                float mi = m[i], vi = v[i];
                *p = one(g[i], *p, mi, vi);
                m[i] = mi;
                v[i] = vi;
            }
        }
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

// Rows per workgroup of the column reductions: enough workgroups (>= ~2048) to fill 256 CUs several times over
// even for the deep layers (M = a few thousand rows), at most 1024 rows so the f32 partial sums stay short.
// eight-channels-per-lane kernels: C / 8 octets must tile a 256-lane workgroup (or be a multiple of it)
static bool h8_shape(int C) { return C % 8 == 0 && (C / 8 <= 256 ? 256 % (C / 8) == 0 : (C / 8) % 256 == 0); }
static int h8_slab(long M, int C)
{
    const long col_blocks = (C + 2047) / 2048;
    long slab = (M * col_blocks + 4095) / 4096;
    slab = (slab + 63) / 64 * 64;
    return (int)(slab < 64 ? 64 : (slab > 1024 ? 1024 : slab));
}

static int reduce_slab(long M, int C)
{
    const long col_blocks = (C + 255) / 256;
    long slab = (M * col_blocks + 2047) / 2048;
    slab = (slab + 63) / 64 * 64;
    return (int)(slab < 64 ? 64 : (slab > 1024 ? 1024 : slab));
}

int hn_launch_col_stats(const float* a, double* sum, double* sumsq, long M, int C, int lda, hipStream_t s)
{
    HN_REQUIRE(C % 4 == 0 && (C <= 256 ? 256 % C == 0 : C % 256 == 0), "col_stats: unsupported C=%d", C);
    const int slab = reduce_slab(M, C);
    dim3 grid((unsigned)((M + slab - 1) / slab), (unsigned)((C + 255) / 256));
    hipLaunchKernelGGL(col_reduce_kernel<0>, grid, dim3(256), 0, s, a, (const unsigned char*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, sum, sumsq, M, C, lda ? lda : C, slab, 0, 0);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_bn_bwd_reduce(const float* dy, const unsigned char* bmask, const float* z, const float* mean, const float* invstd, double* S1,
                            double* S2, long M, int C, int z_bf16, int dy_bf16, hipStream_t s)
{
    HN_REQUIRE(C % 4 == 0 && (C <= 256 ? 256 % C == 0 : C % 256 == 0), "bn_bwd_reduce: unsupported C=%d", C);
    if (z_bf16 && dy_bf16 && h8_shape(C)) {        // all-bf16 operands: the eight-channels-per-lane kernel
        const int slab8 = h8_slab(M, C);
        dim3 grid8((unsigned)((M + slab8 - 1) / slab8), (unsigned)((C + 2047) / 2048));
        const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy);
        const unsigned short* zh = reinterpret_cast<const unsigned short*>(z);
        if (bmask) hipLaunchKernelGGL(bn_bwd_reduce_h8_kernel<true>, grid8, dim3(256), 0, s, dyh, bmask, zh, mean, invstd, S1, S2, M, C, slab8);
        else hipLaunchKernelGGL(bn_bwd_reduce_h8_kernel<false>, grid8, dim3(256), 0, s, dyh, bmask, zh, mean, invstd, S1, S2, M, C, slab8);
        HN_LAUNCH_CHECK();
        return 0;
    }
    const int slab = reduce_slab(M, C);
    dim3 grid((unsigned)((M + slab - 1) / slab), (unsigned)((C + 255) / 256));
    hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, s, dy, bmask, z, mean, invstd, S1, S2, M, C, C, slab, z_bf16, dy_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

// [sum | sq][C] += its `rep` replicas (ConvDesc::stat_rep), in replica order
__global__ __launch_bounds__(256) void stat_replica_sum_kernel(double* slot, int n2, int rep)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n2) return;
    double s = slot[c];
    for (int r = 1; r <= rep; ++r) s += slot[(size_t)r * n2 + c];
    slot[c] = s;
}

int hn_launch_stat_replica_sum(double* slot, int C, int rep, hipStream_t s)
{
    hipLaunchKernelGGL(stat_replica_sum_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, slot, 2 * C, rep);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_bn_finalize(const double* sum, const double* sumsq, double n, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float* a, float* b, float* save_mean, float* save_invstd, int C,
                          hipStream_t s)
{
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sum, sumsq, n, gamma, beta, running_mean,
                       running_var, momentum, a, b, save_mean, save_invstd, C);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* a,
                              float* b, float* save_mean, float* save_invstd, int C, hipStream_t s)
{
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, running_mean, running_var, a, b,
                       save_mean, save_invstd, C);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_affine_act(const float* z, const float* a, const float* b, const float* res, float* y, unsigned char* bmask, void* y_h,
                         long M, int C, int relu, int z_bf16, int res_bf16, hipStream_t s)
{
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(total4)), dim3(256), 0, s, z, a, b, res, y, bmask,
                       reinterpret_cast<unsigned short*>(y_h), total4, C / 4, relu, z_bf16, res_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_affine_act_bn(const float* z, const double* sum, const double* sumsq, double n, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float* a, float* b, float* save_mean,
                            float* save_invstd, const float* res, float* y, unsigned char* bmask, void* y_h, long M, int C, int relu,
                            int z_bf16, int res_bf16, hipStream_t s)
{
    HN_REQUIRE(C % 4 == 0 && C <= 4096, "affine_act_bn: C=%d", C);
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(affine_act_bn_kernel, dim3(grid_for(total4)), dim3(256), 2 * (size_t)C * sizeof(float), s, z, sum, sumsq, n, gamma,
                       beta, running_mean, running_var, momentum, a, b, save_mean, save_invstd, res, y, bmask,
                       reinterpret_cast<unsigned short*>(y_h), total4, C / 4, relu, z_bf16, res_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_affine_act_bn_pool(const void* z_h, const double* sum, const double* sumsq, double n, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float* a, float* b, float* save_mean,
                                 float* save_invstd, unsigned char* bmask, void* pool_h, void* idx, int B, int Hi, int Wi, int C, hipStream_t s)
{
    HN_REQUIRE(C % 4 == 0 && C <= 4096 && Hi % 2 == 0 && Wi % 2 == 0, "affine_act_bn_pool: C=%d Hi=%d Wi=%d", C, Hi, Wi);
    const long total = (long)B * (Hi / 2) * (Wi / 2) * (C / 4);
    hipLaunchKernelGGL(affine_act_bn_pool_kernel, dim3(grid_for(total, 256L * 64)), dim3(256), 2 * (size_t)C * sizeof(float), s,
                       reinterpret_cast<const unsigned short*>(z_h), sum, sumsq, n, gamma, beta, running_mean, running_var, momentum, a, b, save_mean,
                       save_invstd, bmask, reinterpret_cast<unsigned short*>(pool_h), reinterpret_cast<unsigned*>(idx), Hi, Wi, C / 4, total);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_bn_bwd_apply(const float* dy, const unsigned char* bmask, const float* z, const float* mean, const float* invstd,
                           const float* gamma, const double* S1, const double* S2, double n, float* dz, float* dpre, void* dz_h, long M,
                           int C, int z_bf16, int dy_bf16, double* db, hipStream_t s)
{
    // db (optional, zeroed by the caller): column sums of dz, accumulated by the all-bf16 kernel only -- the float32 path
    // leaves it to a separate hn_launch_col_stats over the float32 dz
    HN_REQUIRE(!db || (z_bf16 && dy_bf16 && !dz && dz_h && h8_shape(C)), "bn_bwd_apply: db needs the all-bf16 kernel");
    if (z_bf16 && dy_bf16 && !dz && dz_h && h8_shape(C)) {
        const int slab8 = h8_slab(M, C);
        dim3 grid8((unsigned)((M + slab8 - 1) / slab8), (unsigned)((C + 2047) / 2048));
        const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy);
        const unsigned short* zh = reinterpret_cast<const unsigned short*>(z);
        unsigned short* dph = reinterpret_cast<unsigned short*>(dpre);
        unsigned short* dzh = reinterpret_cast<unsigned short*>(dz_h);
#define HN_APPLY_H8(MASK, BIAS)                                                                                                       \
    hipLaunchKernelGGL((bn_bwd_apply_h8_kernel<MASK, BIAS>), grid8, dim3(256), 0, s, dyh, bmask, zh, mean, invstd, gamma, S1, S2, n, dph, \
                       dzh, db, M, C, slab8)
        if (bmask && db) HN_APPLY_H8(true, true);
        else if (bmask) HN_APPLY_H8(true, false);
        else if (db) HN_APPLY_H8(false, true);
        else HN_APPLY_H8(false, false);
#undef HN_APPLY_H8
        HN_LAUNCH_CHECK();
        return 0;
    }
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, s, dy, bmask, z, mean, invstd, gamma, S1, S2, n, dz,
                       dpre, reinterpret_cast<unsigned short*>(dz_h), total4, C / 4, z_bf16, dy_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

// The stem's BatchNorm adjoint straight from the POOLED gradient (bf16 training step): dy = max-pool adjoint formed on the fly (PoolSrc).
// Same sums and the same dz bits as hn_launch_maxpool_bwd_idx (bf16 output) followed by hn_launch_bn_bwd_reduce / _apply.
int hn_launch_bn_bwd_pool(const void* dpool_h, const void* pidx, int B, int Hi, int Wi, const unsigned char* bmask, const void* z_h,
                          const float* mean, const float* invstd, const float* gamma, double* S1, double* S2, void* dz_h, int C, int phase,
                          hipStream_t s)
{
    HN_REQUIRE(bmask && h8_shape(C) && C % 8 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "bn_bwd_pool: unsupported shape C=%d", C);
    const long M = (long)B * Hi * Wi;
    const int slab8 = h8_slab(M, C);
    dim3 grid8((unsigned)((M + slab8 - 1) / slab8), (unsigned)((C + 2047) / 2048));
    const PoolSrc ps = {reinterpret_cast<const unsigned short*>(dpool_h), reinterpret_cast<const unsigned*>(pidx), Hi, Wi};
    const unsigned short* zh = reinterpret_cast<const unsigned short*>(z_h);
    if (phase == 0)
        hipLaunchKernelGGL((bn_bwd_reduce_h8_kernel<true, true>), grid8, dim3(256), 0, s, (const unsigned short*)nullptr, bmask, zh, mean, invstd, S1,
                           S2, M, C, slab8, ps);
    else
        hipLaunchKernelGGL((bn_bwd_apply_h8_kernel<true, false, true>), grid8, dim3(256), 0, s, (const unsigned short*)nullptr, bmask, zh, mean, invstd,
                           gamma, S1, S2, (double)M, (unsigned short*)nullptr, reinterpret_cast<unsigned short*>(dz_h), (double*)nullptr, M, C,
                           slab8, ps);
    HN_LAUNCH_CHECK();
    return 0;
}

// phase 0: the sums of both branches (S1 / S2 of each, zeroed by the caller); phase 1: both dz as bf16.  All operands bf16, M rows of C channels.
int hn_launch_bn_bwd_dual(const void* dy_h, const unsigned char* bmask, const void* za_h, const float* mean_a, const float* invstd_a,
                          const float* gamma_a, double* S1a, double* S2a, void* dza_h, const void* zb_h, const float* mean_b, const float* invstd_b,
                          const float* gamma_b, double* S1b, double* S2b, void* dzb_h, long M, int C, int phase, hipStream_t s)
{
    HN_REQUIRE(bmask && h8_shape(C), "bn_bwd_dual: unsupported shape C=%d", C);
    const int slab8 = h8_slab(M, C);
    dim3 grid8((unsigned)((M + slab8 - 1) / slab8), (unsigned)((C + 2047) / 2048));
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy_h);
    const unsigned short* zah = reinterpret_cast<const unsigned short*>(za_h);
    const unsigned short* zbh = reinterpret_cast<const unsigned short*>(zb_h);
    if (phase == 0)
        hipLaunchKernelGGL(bn_bwd_reduce_dual_h8_kernel, grid8, dim3(256), 0, s, dyh, bmask, zah, mean_a, invstd_a, zbh, mean_b, invstd_b, S1a, S2a, S1b,
                           S2b, M, C, slab8);
    else
        hipLaunchKernelGGL(bn_bwd_apply_dual_h8_kernel, grid8, dim3(256), 0, s, dyh, bmask, zah, mean_a, invstd_a, gamma_a, S1a, S2a, zbh, mean_b,
                           invstd_b, gamma_b, S1b, S2b, (double)M, reinterpret_cast<unsigned short*>(dza_h), reinterpret_cast<unsigned short*>(dzb_h),
                           M, C, slab8);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_d2f(const double* in, float* out, int n, hipStream_t s)
{
    hipLaunchKernelGGL(d2f_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_axpy(const float* x, float* y, long n, hipStream_t s)
{
    HN_REQUIRE(n % 4 == 0, "axpy: n %% 4");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, y, n / 4);
    HN_LAUNCH_CHECK();
    return 0;
}

// scratch: B * (Hi/2) * (Wi/2) * C ints
int hn_launch_axpy_bf16(const void* x, void* y, long n, hipStream_t s)
{
    HN_REQUIRE(n % 4 == 0, "axpy bf16: n %% 4");
    hipLaunchKernelGGL(axpy_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(x),
                       reinterpret_cast<unsigned short*>(y), n / 4);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_bf16_to_f32(const void* in, float* out, long n, hipStream_t s)
{
    HN_REQUIRE(n % 4 == 0, "bf16_to_f32: n %% 4");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(in), out, n / 4);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_maxpool_idx(const float* in, float* out, void* idx, int B, int Hi, int Wi, int C, hipStream_t s, int in_bf16, void* out_h)
{
    HN_REQUIRE(C % 4 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "maxpool_idx: C %% 4, even sizes");
    HN_REQUIRE(out || out_h, "maxpool_idx: no output");
    const long total = (long)B * (Hi / 2) * (Wi / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_idx_kernel, dim3(grid_for(total, 256L * 64)), dim3(256), 0, s, in, out, reinterpret_cast<unsigned*>(idx), Hi,
                       Wi, Hi / 2, Wi / 2, C / 4, total, in_bf16, reinterpret_cast<unsigned short*>(out_h));
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_maxpool_bwd_idx(const void* idx, const float* dout, float* din, int B, int Hi, int Wi, int C, int dout_bf16, hipStream_t s,
                              int din_bf16)
{
    HN_REQUIRE(C % 4 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "maxpool_bwd_idx: C %% 4, even sizes");
    const long total = (long)B * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(grid_for(total, 256L * 64)), dim3(256), 0, s, reinterpret_cast<const unsigned*>(idx), dout,
                       din, Hi, Wi, Hi / 2, Wi / 2, C / 4, total, dout_bf16, din_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_upsample_flatten_bwd(const float* dseq, float* din, int B, int hq, int Wq, int cq, int col0, int out_bf16, hipStream_t s)
{
    hipLaunchKernelGGL(upsample_flatten_bwd_kernel, dim3(Wq, B), dim3(256), 0, s, dseq, din, B, hq, Wq, cq, col0, 256 / Wq, out_bf16);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_dropout(const float* in, float* out, long n, float p, unsigned long long seed, hipStream_t s)
{
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n, p, seed);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_head_bwd(const float* dbon, const float* dcor, const float* w, const float* y, float* dy, float* dlin, float* dw,
                       float* db, int T, int B, hipStream_t s)
{
    const long rows = (long)T * B;
    hipLaunchKernelGGL(head_bwd_dy_kernel, dim3((unsigned)rows), dim3(256), 0, s, dbon, dcor, w, dy, dlin, T, B);
    HN_LAUNCH_CHECK();
    // partial sums live behind the T*B*12 floats of dlin (the caller's buffer holds HN_HEAD_BWD_SCRATCH_FLOATS more)
    float* part_w = dlin + rows * 12;
    float* part_b = part_w + (size_t)HEAD_SLICES * 12 * 1024;
    hipLaunchKernelGGL(head_bwd_dw_kernel, dim3(16, HEAD_SLICES), dim3(256), 0, s, dlin, y, part_w, part_b, rows);
    HN_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_bwd_dw_reduce_kernel, dim3((12 * 1024 + 12 + 255) / 256), dim3(256), 0, s, part_w, part_b, dw, db);
    HN_LAUNCH_CHECK();
    return 0;
}

int hn_launch_lstm_bwd_gates(const float* saved, const float* dy, float* dh_rec, float* dc_rec, float* dgx, int T, int B, int step,
                             hipStream_t s)
{
    hipLaunchKernelGGL(lstm_bwd_gates_kernel, dim3((B * 1024 + 255) / 256), dim3(256), 0, s, saved, dy, dh_rec, dc_rec, dgx, T, B, step);
    HN_LAUNCH_CHECK();
    return 0;
}

// One Adam step for every parameter tensor (see adam_flat_kernel).  params / offsets / ends / active are DEVICE arrays
// (n_tensors pointers, first / one-past-last element of every tensor inside the flat buffers in ascending order, flags);
// grads / m / v are flat float32 buffers of `total` elements in the layout of hn_grad_offset().
extern "C" int hn_adam_step(float* const* params, const long long* offsets, const long long* ends, const unsigned char* active, int n_tensors,
                            const float* grads,
                            float* m, float* v, long long total, float lr, float beta1, float beta2, float eps, float weight_decay,
                            int step, float grad_scale, void* stream)
{
    HN_REQUIRE(params && offsets && ends && active && grads && m && v, "hn_adam_step: null pointer");
    HN_REQUIRE(n_tensors >= 1 && total >= 1 && step >= 1 && grad_scale > 0.f, "hn_adam_step: bad sizes / step");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for((total + 15) / 16, 256L * 16)), dim3(256), 0, (hipStream_t)stream, params, offsets, ends, active,
                       n_tensors, grads, m, v, total, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), 1.f / grad_scale);
    HN_LAUNCH_CHECK();
    return 0;
}


// ---- the training objective in one launch (reference train.py:53-56: F.l1_loss(bon) + F.binary_cross_entropy_with_logits(cor)) ----
// ONE workgroup of 1024 lanes walks both tensors (B x 3 x 1024 elements: microseconds), so the two means are plain ordered sums
// (deterministic), and writes the two gradients of the means as it goes: d mean|a - y| = sign(a - y) / N (sign(0) = 0, as torch),
// d mean bce = (sigmoid(x) - y) / N.  losses[0] = L1, [1] = BCE, [2] = their sum.  Replaces ~12 elementwise / reduce launches.
namespace {
__global__ __launch_bounds__(1024) void loss_l1_bce_kernel(const float* __restrict__ bon, const float* __restrict__ y_bon, long n_bon,
                                                           const float* __restrict__ cor, const float* __restrict__ y_cor, long n_cor,
                                                           float* __restrict__ losses, float* __restrict__ total, float* __restrict__ dbon,
                                                           float* __restrict__ dcor)
{
    __shared__ double red[2][1024];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    const float ib = 1.f / (float)n_bon, ic = 1.f / (float)n_cor;
    for (long i = tid; i < n_bon; i += 1024) {
        const float d = bon[i] - y_bon[i];
        s0 += (double)fabsf(d);
        dbon[i] = d > 0.f ? ib : (d < 0.f ? -ib : 0.f);
    }
    for (long i = tid; i < n_cor; i += 1024) {
        const float x = cor[i], y = y_cor[i];
        s1 += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));      // torch's stable form of BCE-with-logits
        dcor[i] = (1.f / (1.f + expf(-x)) - y) * ic;
    }
    red[0][tid] = s0;
    red[1][tid] = s1;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float l0 = (float)(red[0][0] / (double)n_bon), l1 = (float)(red[1][0] / (double)n_cor);
        losses[0] = l0;
        losses[1] = l1;
        losses[2] = l0 + l1;
        if (total) total[0] = l0 + l1;
    }
}

// g[i] *= scale[0] for both gradient tensors (the adjoint of losses[2] arrives as a device scalar: no host read)
__global__ __launch_bounds__(256) void scale2_kernel(float* __restrict__ a, long na, float* __restrict__ b, long nb, const float* __restrict__ scale)
{
    const float sc = scale[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long)gridDim.x * 256) {
        if (i < na) a[i] *= sc; else b[i - na] *= sc;
    }
}
}  // namespace

extern "C" int hn_loss_l1_bce(const float* bon, const float* y_bon, long long n_bon, const float* cor, const float* y_cor, long long n_cor,
                              float* losses3, float* total, float* dbon, float* dcor, void* stream)
{
    HN_REQUIRE(bon && y_bon && cor && y_cor && losses3 && dbon && dcor && n_bon > 0 && n_cor > 0, "hn_loss_l1_bce: bad argument");
    hipLaunchKernelGGL(loss_l1_bce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bon, y_bon, (long)n_bon, cor, y_cor, (long)n_cor, losses3, total, dbon,
                       dcor);
    HN_LAUNCH_CHECK();
    return 0;
}

extern "C" int hn_scale2(float* a, long long na, float* b, long long nb, const float* scale_dev, void* stream)
{
    HN_REQUIRE(a && b && scale_dev && na >= 0 && nb >= 0, "hn_scale2: bad argument");
    long g = (na + nb + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(scale2_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, (long)na, b, (long)nb, scale_dev);
    HN_LAUNCH_CHECK();
    return 0;
}
