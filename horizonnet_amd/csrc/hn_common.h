// Internal helpers shared by the HIP translation units of libhorizonnet_hip.so.
#pragma once
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
constexpr int HN_STATUS_WORD = 512;

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
};
int hn_launch_conv(const ConvDesc& d, hipStream_t s);
int hn_launch_prep_nhwc4(const float* x, float* out, int B, int C_in, int H, int W, hipStream_t s);
int hn_launch_maxpool(const float* in, float* out, int B, int Hi, int Wi, int C, hipStream_t s);
int hn_launch_upsample_flatten(const float* in, float* seq, int B, int hq, int Wq, int cq, int col0, hipStream_t s);
int hn_launch_lstm_layer(const float* gx, const float* whh_f, const float* whh_r, float* y, int T, int B,
                         void* sync_ws, hipStream_t s);
int hn_launch_linear_head(const float* y, const float* w, const float* bias, float* bon, float* cor, int T, int B,
                          hipStream_t s);
int hn_launch_pack_conv(const float* w, float* out, int Cout, int Cin, int KH, int KW, hipStream_t s);
int hn_launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                      float* scale, float* shift, int C, hipStream_t s);
int hn_launch_add_vec(const float* a, const float* b, float* out, long n, hipStream_t s);
