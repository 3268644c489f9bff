// One launch for a whole list of small index-map jobs (see multi_job.h).  The element formulas are those of the single-tensor
// kernels they batch (pack_conv_kernel / fold_bn_kernel in elementwise.hip, pack_conv_bf16_kernel / pack_stem_bf16_kernel /
// pack_dgrad_class_bf16_kernel in conv_igemm_bf16.hip, unpack_conv_kernel in conv_wgrad_f32.hip, d2f_kernel in train_ops.hip):
// same reads, same rounding, so the results are bit-identical to the one-by-one launches (tests/test_gpu_train.py).
#include "multi_job.h"

#include <string.h>

namespace {

constexpr int MJ_PER_BLOCK = 2048;          // output elements per workgroup (256 threads x 8)

typedef unsigned short u16;

__device__ __forceinline__ u16 to_bf16(float v)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(0.f));
    return (u16)(r & 0xffffu);
}

__global__ __launch_bounds__(256) void multi_job_kernel(const MJob* __restrict__ jobs, int njobs)
{
    const int b = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                        // last job whose first workgroup is <= b (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const MJob* __restrict__ j = jobs + lo;
    const long long total = j->total;
    const long long base = (long long)(b - j->first_block) * MJ_PER_BLOCK;
    long long end = base + MJ_PER_BLOCK;
    if (end > total) end = total;
    const int kind = j->kind;
    const float* __restrict__ src = reinterpret_cast<const float*>(j->src);

    if (kind == MJ_PACK_BF16) {          // k = (64-channel chunk, tap, channel in chunk), see pack_conv_bf16_kernel
        const int Cin = j->p[1], KH = j->p[2], KW = j->p[3];
        const int ntap = KH * KW, nch = Cin / 64;
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int e = (int)(i % 64);
            long long t = i / 64;
            const int tap = (int)(t % ntap);
            t /= ntap;
            const int cc = (int)(t % nch);
            const int o = (int)(t / nch);
            const int dh = tap / KW, dw = tap - dh * KW;
            reinterpret_cast<u16*>(j->dst)[i] = to_bf16(src[(((long long)o * Cin + cc * 64 + e) * KH + dh) * KW + dw]);
        }
    } else if (kind == MJ_PACK_F32) {
        const int Cin = j->p[1], KH = j->p[2], KW = j->p[3], KWp = j->p[4], Cp = j->p[5];
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int c = (int)(i % Cp);
            long long t = i / Cp;
            const int dw = (int)(t % KWp);
            t /= KWp;
            const int dh = (int)(t % KH);
            const int o = (int)(t / KH);
            float v = 0.f;
            if (c < Cin && dw < KW) v = src[(((long long)o * Cin + c) * KH + dh) * KW + dw];
            reinterpret_cast<float*>(j->dst)[i] = v;
        }
    } else if (kind == MJ_PACK_STEM_BF16) {
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int ch = (int)(i & 3);
            const int t = (int)((i >> 2) & 7);
            const int dh = (int)((i >> 5) & 7);
            const int o = (int)(i >> 8);
            float v = 0.f;
            if (ch < 3 && t >= 1 && dh < 7) v = src[(((long long)o * 3 + ch) * 7 + dh) * 7 + (t - 1)];
            reinterpret_cast<u16*>(j->dst)[i] = to_bf16(v);
        }
    } else if (kind == MJ_FOLD_BN) {
        const float* beta = reinterpret_cast<const float*>(j->aux[0]);
        const float* mean = reinterpret_cast<const float*>(j->aux[1]);
        const float* var = reinterpret_cast<const float*>(j->aux[2]);
        const float* bias = reinterpret_cast<const float*>(j->aux[3]);
        float* scale = reinterpret_cast<float*>(j->dst);
        float* shift = reinterpret_cast<float*>(j->dst2);
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const float bb = bias ? bias[i] : 0.f;
            if (src) {
                const float s = src[i] / sqrtf(var[i] + 1e-5f);
                scale[i] = s;
                shift[i] = (bb - mean[i]) * s + beta[i];
            } else {
                scale[i] = 1.f;
                shift[i] = bb;
            }
        }
    } else if (kind == MJ_PACK_DGRAD_BF16) {
        const int Cout = j->p[0], Cin = j->p[1], KH = j->p[2], KW = j->p[3];
        const int dh0 = j->p[4], dh1 = j->p[5], dh2 = j->p[6], ndh = j->p[7], dw0 = j->p[8], dw1 = j->p[9], dw2 = j->p[10], ndw = j->p[11];
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int o = (int)(i % Cout);
            long long t = i / Cout;
            const int iw = (int)(t % ndw);
            t /= ndw;
            const int ih = (int)(t % ndh);
            const int c = (int)(t / ndh);
            const int dh = ih == 0 ? dh0 : (ih == 1 ? dh1 : dh2), dw = iw == 0 ? dw0 : (iw == 1 ? dw1 : dw2);
            reinterpret_cast<u16*>(j->dst)[i] = to_bf16(src[(((long long)o * Cin + c) * KH + dh) * KW + dw]);
        }
    } else if (kind == MJ_PACK_DGRAD_FWD_BF16) {
        const int Cout = j->p[0], Cin = j->p[1], KH = j->p[2], KW = j->p[3];
        const int ntap = KH * KW, nch = Cout / 64;
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int e = (int)(i % 64);
            long long t = i / 64;
            const int tap = (int)(t % ntap);
            t /= ntap;
            const int cc = (int)(t % nch);
            const int c = (int)(t / nch);
            const int dh = tap / KW, dw = tap - dh * KW;
            reinterpret_cast<u16*>(j->dst)[i] = to_bf16(src[(((long long)(cc * 64 + e) * Cin + c) * KH + (KH - 1 - dh)) * KW + (KW - 1 - dw)]);
        }
    } else if (kind == MJ_UNPACK) {
        const int Cin = j->p[1], KH = j->p[2], KW = j->p[3], KHp = j->p[4], KWp = j->p[5], Cp = j->p[6];
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            const int dw = (int)(i % KW);
            long long t = i / KW;
            const int dh = (int)(t % KH);
            t /= KH;
            const int c = (int)(t % Cin);
            const int o = (int)(t / Cin);
            reinterpret_cast<float*>(j->dst)[i] = src[(((long long)o * KHp + dh) * KWp + dw) * Cp + c];
        }
    } else if (kind == MJ_D2F) {
        const double* in = reinterpret_cast<const double*>(j->src);
        for (long long i = base + threadIdx.x; i < end; i += 256) reinterpret_cast<float*>(j->dst)[i] = (float)in[i];
    } else if (kind == MJ_F32_TO_BF16) {
        for (long long i = base + threadIdx.x; i < end; i += 256) reinterpret_cast<u16*>(j->dst)[i] = to_bf16(src[i]);
    } else if (kind == MJ_COPY_F32) {
        for (long long i = base + threadIdx.x; i < end; i += 256) reinterpret_cast<float*>(j->dst)[i] = src[i];
    } else if (kind == MJ_ADD_VEC) {
        const float* bsrc = reinterpret_cast<const float*>(j->aux[0]);
        for (long long i = base + threadIdx.x; i < end; i += 256) reinterpret_cast<float*>(j->dst)[i] = src[i] + (bsrc ? bsrc[i] : 0.f);
    }
}

}  // namespace

int JobTable::run(std::vector<MJob>& jobs, hipStream_t s)
{
    if (jobs.empty()) return 0;
    long long blocks = 0;
    for (MJob& j : jobs) {
        HN_REQUIRE(j.total > 0 && j.dst != nullptr, "multi-job: empty job / null destination (kind %d)", j.kind);
        j.first_block = (int)blocks;
        blocks += (j.total + MJ_PER_BLOCK - 1) / MJ_PER_BLOCK;
    }
    HN_REQUIRE(blocks < 2147483647LL, "multi-job: too many workgroups");
    const bool same = last.size() == jobs.size() && memcmp(last.data(), jobs.data(), jobs.size() * sizeof(MJob)) == 0;
    if (!same || dev == nullptr) {
        if (jobs.size() > cap) {
            // (hipFree waits for the device: nothing can still be reading the old table or the old staging copy)
            if (dev) HN_HIP(hipFree(dev));
            if (host) HN_HIP(hipHostFree(host));
            dev = host = nullptr;
            cap = jobs.size() + 16;
            HN_HIP(hipMalloc(&dev, cap * sizeof(MJob)));
            HN_HIP(hipHostMalloc(&host, cap * sizeof(MJob), hipHostMallocDefault));
        }
        if (uploaded == nullptr) HN_HIP(hipEventCreateWithFlags(&uploaded, hipEventDisableTiming));
        else HN_HIP(hipEventSynchronize(uploaded));       // the previous upload has read the staging copy
        memcpy(host, jobs.data(), jobs.size() * sizeof(MJob));
        last = jobs;
        // stream-ordered behind any launch still reading the previous list
        HN_HIP(hipMemcpyAsync(dev, host, jobs.size() * sizeof(MJob), hipMemcpyHostToDevice, s));
        HN_HIP(hipEventRecord(uploaded, s));
    }
    hipLaunchKernelGGL(multi_job_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const MJob*>(dev), (int)jobs.size());
    HN_LAUNCH_CHECK();
    return 0;
}

void JobTable::release()
{
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    if (uploaded) (void)hipEventDestroy(uploaded);
    dev = host = nullptr;
    uploaded = nullptr;
    cap = 0;
    last.clear();
}

MJob mj_pack_f32(const float* w, float* out, int Cout, int Cin, int KH, int KW)
{
    const bool stem = KH == 7;
    const int KWp = stem ? 8 : KW, Cp = stem ? 4 : Cin;
    MJob j = mj_make(MJ_PACK_F32, w, out, (long long)Cout * KH * KWp * Cp);
    j.p[0] = Cout; j.p[1] = Cin; j.p[2] = KH; j.p[3] = KW; j.p[4] = KWp; j.p[5] = Cp;
    return j;
}

MJob mj_pack_bf16(const float* w, void* out, int Cout, int Cin, int KH, int KW)
{
    if (KH == 7) {           // the stem's own K layout
        MJob j = mj_make(MJ_PACK_STEM_BF16, w, out, (long long)Cout * 256);
        j.p[0] = Cout;
        return j;
    }
    MJob j = mj_make(MJ_PACK_BF16, w, out, (long long)Cout * KH * KW * Cin);
    j.p[0] = Cout; j.p[1] = Cin; j.p[2] = KH; j.p[3] = KW; j.p[4] = KW; j.p[5] = Cin;
    return j;
}

MJob mj_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias, float* scale, float* shift,
                int C)
{
    MJob j = mj_make(MJ_FOLD_BN, gamma, scale, C);
    j.dst2 = shift;
    j.aux[0] = beta; j.aux[1] = mean; j.aux[2] = var; j.aux[3] = bias;
    return j;
}

MJob mj_unpack(const float* wp, float* w_oihw, int Cout, int Cin, int KH, int KW, int packed_rows)
{
    const bool stem = KH == 7;
    MJob j = mj_make(MJ_UNPACK, wp, w_oihw, (long long)Cout * Cin * KH * KW);
    j.p[0] = Cout; j.p[1] = Cin; j.p[2] = KH; j.p[3] = KW; j.p[4] = packed_rows ? packed_rows : KH; j.p[5] = stem ? 8 : KW; j.p[6] = stem ? 4 : Cin;
    return j;
}
