// Shared definitions of the bf16 implicit-GEMM convolution kernels (conv_igemm_bf16.hip, conv_igemm_bf16_pp.hip).
#pragma once
#include "hn_common.h"

constexpr int BKE = 64;                  // K chunk in elements (128 bytes)
constexpr int ROWB = 128;                // bytes per LDS tile row
constexpr unsigned OOB = 0x80000000u;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct ConvArgsH {
    const u16* x;
    const u16* w;
    const float* scale;
    const float* shift;
    const u16* res;
    void* y;
    int Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, sh, sw, ph, pw;   // K order of the forward kernels: (64-channel chunk, filter tap, channel in chunk) -- see fetch()
    int M, K, nk, relu, ldy;
    int xstride;
    int xcd_swizzle;
    int ksplit;          // > 1: split-K -- blockIdx.y = K slice, float32 partial tiles to y + slice * M * ldy (forward, OUT_F32 only)
    double* stat_sum;    // optional per-channel sum / sum of squares of the stored rows (train-mode BatchNorm statistics)
    double* stat_sq;
    int stat_rep;        // replicas of the statistics slot (power of two >= 1), see ConvDesc in hn_common.h
    // data-gradient mode (template TR), see conv_igemm_f32.hip: one launch = one stride-parity class of dX pixels
    int sh_log2, sw_log2;
    int ca, cb, cHo, cWo;
    int tdh[3], tdw[3], ntdh, ntdw;
    // optional (4-wave kernel, bf16 output; the training adjoint's conv B, train.hip: bn_fold_dgrad): the stored rows are the gradient dy of a
    // BatchNorm + ReLU unit whose z / bit mask / saved mean / invstd are given -- the epilogue also takes that unit's reduce pass: per tile
    // sum_m g and sum_m g * zhat (g = dy * mask) into bn_slab[tile row][2][Cout] (plain stores; hn_launch_slab_colsum adds the tiles in order)
    const u16* bn_z;
    const unsigned char* bn_mask;
    const float* bn_mean;
    const float* bn_invstd;
    float* bn_slab;
    unsigned char* mask_out;   // optional (4-wave kernel, bf16 output, ReLU): the ReLU bit mask of the stored rows, 4 bits per byte (element e -> byte e >> 2)
#ifdef HN_CONV_TRACE
    unsigned long long* trace = nullptr;   // throw-away measurement builds only (tools/conv_trace.py): 8 stamps per workgroup / tile
#endif
};

// Measurement builds (-DHN_CONV_TRACE, tools/conv_trace.py): s_memrealtime stamps per workgroup (per tile in the persistent
// kernel) at entry, first chunk landed, k loop done, after every epilogue band; HW_ID / XCC_ID in slot 7.  Compiled out otherwise.
#ifdef HN_CONV_TRACE
#define HN_TR_STAMP(wg, k) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)(wg) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define HN_TR_HWID(wg) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)(wg) * 8 + 7] = \
    (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); } while (0)
#else
#define HN_TR_STAMP(wg, k) do {} while (0)
#define HN_TR_HWID(wg) do {} while (0)
#endif

static __device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, unsigned voff, unsigned soff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voff, soff, 0, 0);
#endif
}

static __device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
static __device__ __forceinline__ float bf16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
static __device__ __forceinline__ float bf16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }


// ping-pong persistent 256x256 kernel (conv_igemm_bf16_pp.hip): forward convs with Cout % 256 == 0 and K >= 128
int hn_launch_conv_bf16_pp(const ConvArgsH& a, int out_f32, int variant, hipStream_t s);

// 3x3 / stride 1 along W with the activations of a filter row loaded once (conv3x3_dwr_bf16.hip); shape 0: 256 x 256 tiles, 1: 512 x 128,
// 2: 512 x 64 (conv3x3_dwr64_bf16.hip: bf16 output, no residual, no split-K)
bool hn_conv_bf16_dwr_ok(const ConvArgsH& a, int shape);
int hn_launch_conv_bf16_dwr(const ConvArgsH& a, int out_f32, int shape, hipStream_t s);
int hn_launch_conv_bf16_dwr64(const ConvArgsH& a, hipStream_t s);
