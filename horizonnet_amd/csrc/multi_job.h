// Table-driven batches of small index-map kernels (weight packers, gradient un-packers, double -> float copies).
//
// One training step of the reference's loop (train.py:246-286) changes all 161 weight tensors, so everything derived from
// them -- the packed f32 / bf16 conv weights, the folded BatchNorm affines, the per-class data-gradient packings -- is
// rebuilt every step, and every one of the 69 conv units ends its adjoint with an un-pack of its weight gradient and two or
// three double -> float copies of its BatchNorm sums.  Launched one by one these are ~800 launches of ~5 us per step (11 % of
// the bf16 step's kernel time, profiles/r2_train_bf16_kernel_stats.txt).  Here a host-built list of jobs is uploaded once
// (and again only when a pointer in it changes) and ONE launch walks it: workgroup b finds its job by a binary search over
// the jobs' first workgroup index, then handles 2048 consecutive output elements of that job.
#pragma once
#include "hn_common.h"

#include <vector>

enum MJKind {
    MJ_PACK_F32 = 0,        // OIHW f32 -> [Cout][KH][KWp][Cp] f32          p = {Cout, Cin, KH, KW, KWp, Cp}
    MJ_PACK_BF16,           // OIHW f32 -> [Cout][Cin / 64][KH][KW][64] bf16 (taps inside a channel chunk)   p = {Cout, Cin, KH, KW, KWp, Cp}
    MJ_PACK_STEM_BF16,      // 7x7 stem  -> [64][4 chunks][2 rows][8 taps][4 ch] bf16 (conv_igemm_bf16.hip)   p = {Cout}
    MJ_FOLD_BN,             // src = gamma (null: scale 1), aux = {beta, mean, var, bias (null: 0)} -> dst = scale, dst2 = shift
    MJ_PACK_DGRAD_BF16,     // OIHW f32 -> [Cin][ndh][ndw][Cout] bf16       p = {Cout, Cin, KH, KW, dh0, dh1, dh2, ndh, dw0, dw1, dw2, ndw}
    MJ_UNPACK,              // [Cout][KHp][KWp][Cp] f32 -> OIHW f32         p = {Cout, Cin, KH, KW, KHp, KWp, Cp}
    MJ_D2F,                 // double -> float
    MJ_F32_TO_BF16,         // float -> bf16 (round to nearest even)
    MJ_COPY_F32,            // float -> float
    MJ_ADD_VEC,             // dst = src + aux[0]
    MJ_PACK_DGRAD_FWD_BF16, // OIHW f32 -> [Cin][Cout / 64][KH][KW][64] bf16 with flipped taps: the data gradient of a stride-1 conv as a
                            // FORWARD conv (pack_dgrad_fwd_bf16_kernel)   p = {Cout, Cin, KH, KW}
};

struct MJob {
    const void* src;
    void* dst;
    void* dst2;
    const void* aux[4];
    long long total;         // output elements
    int kind;
    int first_block;         // filled by JobTable::run
    int p[12];
    int pad_[2];
};
static_assert(sizeof(MJob) == 128, "MJob is read with scalar loads: keep it a power of two");

inline MJob mj_make(int kind, const void* src, void* dst, long long total)
{
    MJob j;
    j.src = src; j.dst = dst; j.dst2 = nullptr;
    for (int i = 0; i < 4; ++i) j.aux[i] = nullptr;
    j.total = total; j.kind = kind; j.first_block = 0;
    for (int i = 0; i < 12; ++i) j.p[i] = 0;
    j.pad_[0] = j.pad_[1] = 0;
    return j;
}

// Engine-owned device copy of one job list.  `run` uploads the list only when it differs from the last one it ran (the
// pointers are stable from step to step: bound parameters, one workspace, the caching allocator's gradient buffer).
struct JobTable {
    std::vector<MJob> last;  // what the device copy holds
    void* dev = nullptr;
    void* host = nullptr;    // page-locked staging copy: the upload is a true asynchronous copy
    hipEvent_t uploaded = nullptr;   // recorded behind the last upload; waited for before the staging copy is overwritten
    size_t cap = 0;          // jobs
    int run(std::vector<MJob>& jobs, hipStream_t s);
    void release();
};

MJob mj_pack_f32(const float* w, float* out, int Cout, int Cin, int KH, int KW);
MJob mj_pack_bf16(const float* w, void* out, int Cout, int Cin, int KH, int KW);
MJob mj_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias, float* scale, float* shift,
                int C);
MJob mj_unpack(const float* wp, float* w_oihw, int Cout, int Cin, int KH, int KW, int packed_rows);
