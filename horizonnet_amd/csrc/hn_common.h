// Internal helpers shared by the HIP translation units of libhorizonnet_hip.so.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

void hn_set_error(const char* fmt, ...);

#define HN_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            hn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

#define HN_REQUIRE(cond, ...)                                                         \
    do {                                                                              \
        if (!(cond)) {                                                                \
            hn_set_error(__VA_ARGS__);                                                \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

#define HN_LAUNCH_CHECK() HN_HIP(hipGetLastError())

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// device scratch of the persistent LSTM: arrival counters + sticky status word (uint32 words)
constexpr int HN_SYNC_WORDS = 1024;     // 4096 bytes
constexpr int HN_GRAD_SEGMENTS = 5;     // gradient-completion segments of the backward pass (train.hip)
constexpr int HN_STATUS_WORD = 512;

constexpr size_t HN_HEAD_BWD_SCRATCH_FLOATS = 32 * 12 * 1024 + 32 * 12;   // hn_launch_head_bwd: row-slice partials behind dlin

// The persistent LSTM kernels hand data between workgroups inside one launch: ALL workgroups of the grid must be resident at
// once (a plain launch does not check that; on a partitioned GPU -- CPX mode, fewer CUs -- they would spin into their time-out
// and report through the status word).  Checked once per kernel and device against the occupancy query, as a cooperative
// launch would, without that launch's ~17 us of host time per call.
template <class Kernel>
static inline int hn_require_coresident(Kernel kern, int grid, int threads, const char* what)
{
    static const void* ok[64][4] = {};      // kernels already checked on a device (the template is per signature, not per kernel)
    int dev = 0;
    HN_HIP(hipGetDevice(&dev));
    const void* id = reinterpret_cast<const void*>(kern);
    if (dev >= 0 && dev < 64)
        for (int i = 0; i < 4; ++i)
            if (ok[dev][i] == id) return 0;
    int per_cu = 0, cus = 0;
    HN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0));
    HN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HN_REQUIRE((long)per_cu * cus >= grid,
               "%s: the persistent kernel needs its %d workgroups resident at once, this device offers %d compute units x %d workgroups "
               "(partitioned GPU?)", what, grid, cus, per_cu);
    if (dev >= 0 && dev < 64)
        for (int i = 0; i < 4; ++i)
            if (ok[dev][i] == nullptr) { ok[dev][i] = id; break; }
    return 0;
}

static inline int hn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- internal launchers (defined in the .hip files, used by engine.hip) ----
struct ConvDesc {
    const float* x;      // NHWC input [B][Hi][Wi][Cin]   (stem: [B][Hi][Wi][4])
    const float* w;      // packed [Cout][K]
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const float* res;    // optional residual [M][Cout]
    float* y;            // [M][ldy]
    int B, Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, sh, sw, ph, pw;
    int relu;
    int ldy;             // output row stride (floats)
    int stem;            // 1 = stem addressing (NHWC4 input, K = 7 rows x (8 taps x 4))
    int xstride;         // floats between input pixels (0 = Cin)
    double* stat_sum;    // optional [Cout]: += column sums of the stored output (train-mode BatchNorm statistics, fused in the epilogue)
    double* stat_sq;     // optional [Cout]: += column sums of squares
    int stat_rep;        // R > 1 (power of two): the slot is [sum | sq][Cout] followed by R replicas of the same (hn_stat_slot_doubles);
                         // workgroup b adds into replica b % R and hn_launch_stat_replica_sum adds them into the leading [sum | sq] the
                         // readers use.  Thousands of workgroups adding into the 8 cache lines of a 64-channel slot serialise at the
                         // memory side (0.06-0.25 ms per layer1 conv).  0 / 1: straight into [sum | sq].  The slot must be zero at launch.
    float* splitk_ws;    // optional scratch for split-K partial tiles (bf16 forward; see hn_launch_conv_bf16)
    size_t splitk_ws_floats;
    const void* bn_z;          // bf16 convs only, see ConvArgsH: the reduce pass of the BatchNorm unit whose gradient this conv produces, in its epilogue
    const unsigned char* bn_mask;
    const float* bn_mean;
    const float* bn_invstd;
    float* bn_slab;            // [ceil(M / tile rows)][2][Cout] floats; the tile-row count is returned by hn_conv_bf16_bn_tile_rows
    unsigned char* mask_out;   // bf16 convs only: also store the ReLU bit mask of the output (affine_act_kernel's format); forces the 4-wave kernel
    int transposed;      // 1 = data-gradient mode: x = dY [B][Hi][Wi][Cin(=Cout of the fwd conv)], y = dX [B][Ho][Wo][Cout(=Cin fwd)],
                         //     w packed [Cin_fwd][kh][kw][Cout_fwd]; sh/sw/ph/pw are the forward conv's
};
// replicas only where thousands of workgroups meet on few channels (layer1 / layer2 at training batch sizes): the extra launch is ~5 us
inline int hn_stat_replicas(int C, long M)
{
    static const char* env = getenv("HN_STAT_REPLICAS");       // "0": never (A/B runs); "all": also for small M (tests exercise the path at B = 2)
    if (env && env[0] == '0') return 1;
    if (M < 400000 && !(env && env[0] == 'a')) return 1;
    return C <= 128 ? 16 : C <= 256 ? 8 : 1;
}
inline size_t hn_stat_slot_doubles(int C, long M) { const int r = hn_stat_replicas(C, M); return r > 1 ? 2 * (size_t)C * (1 + r) : 2 * (size_t)C; }
int hn_launch_stat_replica_sum(double* slot, int C, int rep, hipStream_t s);
int hn_launch_conv(const ConvDesc& d, hipStream_t s);
int hn_launch_conv_dgrad(const ConvDesc& fwd, const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch,
                         const float* ones, const float* zeros, hipStream_t s);
int hn_launch_prep_nhwc4(const float* x, float* out, int B, int C_in, int H, int W, hipStream_t s);
int hn_launch_maxpool(const float* in, float* out, int B, int Hi, int Wi, int C, hipStream_t s);
int hn_launch_upsample_flatten(const float* in, float* seq, int B, int hq, int Wq, int cq, int col0, hipStream_t s);
int hn_launch_lstm_layer(const float* gx, const float* whh_f, const float* whh_r, float* y, int T, int B,
                         void* sync_ws, hipStream_t s, float* save = nullptr);
int hn_launch_lstm_layer_f32_wide(const float* gx, const float* whh_f, const float* whh_r, float* y, int T, int B, void* sync_ws,
                                  hipStream_t s);
int hn_launch_lstm_bwd_dh(const float* dg_f, const float* dg_r, const float* whh_f, const float* whh_r, float* dh, int B, hipStream_t s);
int hn_launch_transpose(const float* in, float* out, int R, int C, hipStream_t s);
int hn_launch_col_stats(const float* a, double* sum, double* sumsq, long M, int C, int lda, hipStream_t s);
int hn_launch_bn_bwd_reduce(const float* dy, const unsigned char* bmask, const float* z, const float* mean, const float* invstd, double* S1,
                            double* S2, long M, int C, int z_bf16, int dy_bf16, hipStream_t s);
int hn_launch_bn_finalize(const double* sum, const double* sumsq, double n, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float* a, float* b, float* save_mean, float* save_invstd, int C,
                          hipStream_t s);
int hn_launch_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* a,
                              float* b, float* save_mean, float* save_invstd, int C, hipStream_t s);
int hn_launch_affine_act(const float* z, const float* a, const float* b, const float* res, float* y, unsigned char* bmask, void* y_h,
                         long M, int C, int relu, int z_bf16, int res_bf16, hipStream_t s);
int hn_launch_bn_bwd_dual(const void* dy_h, const unsigned char* bmask, const void* za_h, const float* mean_a, const float* invstd_a,
                          const float* gamma_a, double* S1a, double* S2a, void* dza_h, const void* zb_h, const float* mean_b, const float* invstd_b,
                          const float* gamma_b, double* S1b, double* S2b, void* dzb_h, long M, int C, int phase, hipStream_t s);
int hn_launch_bn_bwd_pool(const void* dpool_h, const void* pidx, int B, int Hi, int Wi, const unsigned char* bmask, const void* z_h,
                          const float* mean, const float* invstd, const float* gamma, double* S1, double* S2, void* dz_h, int C, int phase,
                          hipStream_t s);
int hn_launch_affine_act_bn_pool(const void* z_h, const double* sum, const double* sumsq, double n, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float* a, float* b, float* save_mean,
                                 float* save_invstd, unsigned char* bmask, void* pool_h, void* idx, int B, int Hi, int Wi, int C, hipStream_t s);
int hn_launch_affine_act_bn(const float* z, const double* sum, const double* sumsq, double n, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float* a, float* b, float* save_mean,
                            float* save_invstd, const float* res, float* y, unsigned char* bmask, void* y_h, long M, int C, int relu,
                            int z_bf16, int res_bf16, hipStream_t s);
int hn_launch_bn_bwd_apply(const float* dy, const unsigned char* bmask, const float* z, const float* mean, const float* invstd,
                           const float* gamma, const double* S1, const double* S2, double n, float* dz, float* dpre, void* dz_h, long M,
                           int C, int z_bf16, int dy_bf16, double* db, hipStream_t s);
int hn_launch_d2f(const double* in, float* out, int n, hipStream_t s);
int hn_launch_axpy(const float* x, float* y, long n, hipStream_t s);
int hn_launch_axpy_bf16(const void* x, void* y, long n, hipStream_t s);
int hn_launch_bf16_to_f32(const void* in, float* out, long n, hipStream_t s);
int hn_launch_maxpool_idx(const float* in, float* out, void* idx, int B, int Hi, int Wi, int C, hipStream_t s, int in_bf16 = 0,
                          void* out_h = nullptr);
int hn_launch_maxpool_bwd_idx(const void* idx, const float* dout, float* din, int B, int Hi, int Wi, int C, int dout_bf16, hipStream_t s,
                              int din_bf16 = 0);
int hn_launch_upsample_flatten_bwd(const float* dseq, float* din, int B, int hq, int Wq, int cq, int col0, int out_bf16, hipStream_t s);
int hn_launch_dropout(const float* in, float* out, long n, float p, unsigned long long seed, hipStream_t s);
int hn_launch_head_bwd(const float* dbon, const float* dcor, const float* w, const float* y, float* dy, float* dlin, float* dw,
                       float* db, int T, int B, hipStream_t s);
int hn_launch_lstm_bwd_gates(const float* saved, const float* dy, float* dh_rec, float* dc_rec, float* dgx, int T, int B, int step,
                             hipStream_t s);
int hn_launch_linear_head(const float* y, const float* w, const float* bias, float* bon, float* cor, int T, int B,
                          hipStream_t s);
int hn_launch_pack_conv(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s);
int hn_launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                      float* scale, float* shift, int C, hipStream_t s);
int hn_launch_conv_wgrad(const float* x, const float* dz, float* dw_packed, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                         int sh, int sw, int xstride, int dzstride, int stem, hipStream_t s, int prezeroed = 0, int max_split = 0);
int hn_launch_unpack_conv(const float* wp, float* w_oihw, int Cout, int Cin, int KH, int KW, int packed_rows, hipStream_t s);
int hn_launch_stem_conv_train_bf16(const float* x, int C_in, const void* wpk, void* z, double* stat_sum, double* stat_sq, int B, hipStream_t s);
int hn_launch_stem_wgrad_bf16(const void* x4_h, const void* dz_h, float* dw_packed, int B, int Hi, int Wi, hipStream_t s, int prezeroed);
int hn_launch_pack_conv_dgrad(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s);
// bf16 family (conv_igemm_bf16.hip): buffers are bf16 unless noted
int hn_launch_conv_bf16(const ConvDesc& d, int out_f32, hipStream_t s);
int hn_launch_pack_conv_bf16(const float* w, void* out, int Cout, int Cin, int KH, int KW, hipStream_t s);
int hn_launch_conv_wgrad_bf16(const void* x_h, const void* dz_h, float* dw_packed, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                              int sh, int sw, hipStream_t s, int prezeroed = 0, int xstride = 0, int dzstride = 0);
// BatchNorm-folded adjoint of the 1x1 convs (bn_fold.hip, conv_wgrad_bf16.hip)
int hn_launch_conv_wgrad_bf16_fold(const void* a_h, void* dy_h, float* p_out, long M, int Cin, int Cout, const unsigned char* bmask,
                                   double* colsum, hipStream_t s, int prezeroed, float* slab = nullptr, size_t slab_floats = 0,
                                   const void* wa = nullptr, void* a_out = nullptr, int write_back = 1);
int hn_conv_bf16_bn_tile_rows(int Cout, long M);                                    // rows per tile of the kernel a bn_z conv is dispatched to
int hn_launch_slab_colsum(const float* slab, int tiles, int n, double* out, hipStream_t s);   // out[i] += sum_t slab[t][i] in double (out zeroed by the caller)
int hn_launch_bn_fold_wa(const void* w_h, const float* gamma, const float* invstd, void* wa, int N, int K, hipStream_t s);
void* hn_bn_fold_wa_ptr(void* ws, int N, int K);
size_t hn_bn_fold_scratch_bytes(int N, int K);      // backward scratch of one unit (its head of hn_bn_fold_zero_bytes must be zeroed)
size_t hn_bn_fold_zero_bytes(int K);
size_t hn_bn_fold_keep_floats(int N, int K);         // per-unit storage kept from the forward to the backward: G | A | Wf | WG
int hn_launch_bn_fold_gram(const void* a_h, long M, int K, float* keep, hipStream_t s, float* slab = nullptr, size_t slab_floats = 0);
int hn_launch_bn_fold_forward_stats(float* keep, const void* w_h, double M, int N, int K, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float* a_out, float* b_out, float* save_mean, float* save_invstd,
                                    const float* ones, const float* zeros, hipStream_t s);
int hn_launch_bn_fold_finish(float* P, float* keep, int wg_ready, const double* S1_in, double* S1_out, double* S2, const void* w_h, const float* mean,
                             const float* invstd, const float* gamma, double M, int N, int K, void* ws, const float* ones, const float* zeros,
                             const void** wa, const float** shift_a, const void** wb, hipStream_t s);
int hn_launch_conv_dgrad_bf16(const ConvDesc& fwd, const void* dz_h, const float* w_oihw, const float* add, float* dx, void* w_scratch,
                              const float* ones, const float* zeros, hipStream_t s, int grad_bf16 = 0);
int hn_launch_f32_to_bf16(const float* in, void* out, long n, hipStream_t s);
int hn_launch_prep_nhwc4_bf16(const float* x, void* out, int B, int C_in, int H, int W, hipStream_t s);
int hn_launch_maxpool_bf16(const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s);
int hn_launch_stem_pool_bf16(const float* x, int C_in, const void* wpk, const float* scale, const float* shift, void* y, int B, hipStream_t s,
                             const void* w1 = nullptr, const float* scale1 = nullptr, const float* shift1 = nullptr, void* t1 = nullptr);
int hn_launch_stem_pool_f32(const float* x, int C_in, const float* wpk, const float* scale, const float* shift, float* y, int B, hipStream_t s);
int hn_launch_upsample_flatten_bf16(const void* in, void* seq, int B, int hq, int Wq, int cq, int col0, hipStream_t s);
int hn_launch_conv1x1_dual_f32(const float* t2, const float* w1, const float* scale1, const float* shift1, const float* x, const float* w2,
                               const float* scale2, const float* shift2, float* y, int B, int Ho, int Wo, int K1, int Hi2, int Wi2, int K2,
                               int s2, int Cout, hipStream_t s);
int hn_launch_conv1x1_dual_bf16(const void* t2, const void* w1, const float* scale1, const float* shift1, const void* x, const void* w2,
                                const float* scale2, const float* shift2, void* y, int B, int Ho, int Wo, int K1, int Hi2, int Wi2,
                                int K2, int s2, int Cout, hipStream_t s);
int hn_launch_conv1x1_chain_bf16(const void* t2, const void* w3, const float* scale3, const float* shift3, const void* x, void* out,
                                 const void* w1n, const float* scale1n, const float* shift1n, void* t1n, long M, int K1, int N1, int N2,
                                 hipStream_t s, const void* xd = nullptr, const void* wd = nullptr, const float* scale_d = nullptr,
                                 const float* shift_d = nullptr);
size_t hn_lstm_bf16_xch_bytes(void);
int hn_launch_lstm_layer_bf16(const float* gx, const void* whh_f, const void* whh_r, float* y, void* y_h, int T, int B, void* xch,
                              void* sync_ws, hipStream_t s, float* save = nullptr);
int hn_launch_lstm_layer_bf16_wide(const float* gx, const void* whh_f, const void* whh_r, float* y, void* y_h, int T, int B,
                                   void* sync_ws, int rows_per_group, int xcds_per_group, hipStream_t s, float* save = nullptr);
size_t hn_lstm_bwd_bf16_xch_bytes(void);
int hn_launch_lstm_layer_bwd_bf16(const float* saved, const float* dy, const void* whhT_f, const void* whhT_r, float* dgx, int T, int B,
                                  void* xch, void* sync_ws, hipStream_t s);
int hn_launch_add_vec(const float* a, const float* b, float* out, long n, hipStream_t s);
