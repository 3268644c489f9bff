// C-ABI of libhorizonnet_hip.so: engine object, reference-state_dict binding, weight packing and
// the eval-mode forward plan of HorizonNet(resnet50, use_rnn=True) -- reference model.py:254-281.
// See include/horizonnet_hip.h for the contract of every entry point.
#include "engine_internal.h"

#include <stdarg.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

// ---- error reporting ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void hn_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hn_last_error(void) { return g_err; }
extern "C" int hn_abi_version(void) { return 1; }

// ---- architecture table ------------------------------------------------------------------------
size_t packed_w_floats(int cout, int cin, int k)
{
    if (k == 7) return (size_t)cout * 7 * 8 * 4;
    return (size_t)cout * k * k * cin;
}

const Arch& arch()
{
    static Arch A = [] {
        Arch a;
        auto add = [&](const std::string& w, const std::string& bn, int cin, int cout, int k, int bias) {
            ConvLayer c{w, bn, cin, cout, k, bias, 0, 0, 0};
            a.convs.push_back(c);
            return (int)a.convs.size() - 1;
        };
        const std::string enc = "feature_extractor.encoder.";
        a.stem = add(enc + "conv1.1", enc + "bn1", 3, 64, 7, 0);
        const int planes_l[4] = {64, 128, 256, 512};
        const int nblk[4] = {3, 4, 6, 3};
        int cin = 64;
        for (int li = 0; li < 4; ++li) {
            for (int j = 0; j < nblk[li]; ++j) {
                const std::string p = enc + "layer" + std::to_string(li + 1) + "." + std::to_string(j) + ".";
                const int pl = planes_l[li];
                a.block_first[li][j] = add(p + "conv1", p + "bn1", cin, pl, 1, 0);
                add(p + "conv2.1", p + "bn2", pl, pl, 3, 0);
                add(p + "conv3", p + "bn3", pl, pl * 4, 1, 0);
                if (j == 0) a.block_down[li] = add(p + "downsample.0", p + "downsample.1", cin, pl * 4, 1, 0);
                cin = pl * 4;
            }
        }
        const int cs[4] = {256, 512, 1024, 2048};
        for (int s = 0; s < 4; ++s) {
            const int c = cs[s];
            const int ch[5] = {c, c / 2, c / 2, c / 4, c / 8};
            for (int k = 0; k < 4; ++k) {
                const std::string p = "reduce_height_module.ghc_lst." + std::to_string(s) + ".layer." + std::to_string(k) + ".layers.";
                const int idx = add(p + "0.1", p + "1", ch[k], ch[k + 1], 3, 1);
                if (k == 0) a.ghc_first[s] = idx;
            }
        }
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };   // 256-byte aligned
        for (auto& c : a.convs) {
            c.w_off = take(packed_w_floats(c.cout, c.cin, c.k));
            c.scale_off = take(c.cout);
            c.shift_off = take(c.cout);
            a.numel[c.wkey + ".weight"] = (int64_t)c.cout * c.cin * c.k * c.k;
            if (c.has_bias) a.numel[c.wkey + ".bias"] = c.cout;
            for (const char* nm : {".weight", ".bias", ".running_mean", ".running_var"}) a.numel[c.bnkey + nm] = c.cout;
            a.numel[c.bnkey + ".num_batches_tracked"] = 1;
        }
        for (int l = 0; l < 2; ++l) {
            a.wih_off[l] = take((size_t)4096 * 1024);
            a.lbias_off[l] = take(4096);
            for (int d = 0; d < 2; ++d) {
                a.whh_off[l][d] = take((size_t)2048 * 512);
                const std::string suf = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                a.numel["bi_rnn.weight_ih" + suf] = 2048 * 1024;
                a.numel["bi_rnn.weight_hh" + suf] = 2048 * 512;
                a.numel["bi_rnn.bias_ih" + suf] = 2048;
                a.numel["bi_rnn.bias_hh" + suf] = 2048;
            }
        }
        a.ones_off = take(4096);
        a.zeros_off = take(4096);
        a.linw_off = take(12 * 1024);
        a.linb_off = take(64);
        a.numel["linear.weight"] = 12 * 1024;
        a.numel["linear.bias"] = 12;
        a.packed_floats = off;
        // flat gradient buffer: parameters in state_dict order, each 256-byte aligned
        size_t goff = 0;
        auto gtake = [&](const std::string& k, size_t n) { a.grad_off[k] = goff; goff += (n + 63) / 64 * 64; };
        for (const auto& c : a.convs) {
            gtake(c.wkey + ".weight", (size_t)c.cout * c.cin * c.k * c.k);
            if (c.has_bias) gtake(c.wkey + ".bias", c.cout);
            gtake(c.bnkey + ".weight", c.cout);
            gtake(c.bnkey + ".bias", c.cout);
        }
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < 2; ++d) {
                const std::string suf = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                gtake("bi_rnn.weight_ih" + suf, 2048 * 1024);
                gtake("bi_rnn.weight_hh" + suf, 2048 * 512);
                gtake("bi_rnn.bias_ih" + suf, 2048);
                gtake("bi_rnn.bias_hh" + suf, 2048);
            }
        gtake("linear.weight", 12 * 1024);
        gtake("linear.bias", 12);
        a.grad_floats = goff;
        return a;
    }();
    return A;
}

namespace {
// ---- workspace plan ----------------------------------------------------------------------------
struct Plan {
    size_t sync, xn, stem, pool, p0, p1, t1, t2, ds, g0, g1, seq, gx, y1, y2, sk, total;   // float offsets
    size_t c[4];      // C1..C4 (the stage outputs) in buffers of their own: the height-compression chain of scale li reads C_li on the engine's
                      // branch stream while the next stage already ping-pongs p0 / p1 on the caller's
    size_t sk2;       // split-K scratch of the caller's stream (interactive regime: the stages' deep-K convs); `sk` belongs to the branch stream
};

constexpr size_t SPLITK_WS_FLOATS_PER_PANO = 8 * 128 * 256;   // up to 8 slices of the largest split layer per panorama
// ... but never less than 8 slices of the largest tile set the interactive regime (B <= 4) splits: layer4's 3x3 convs / ghc3.0 at 4 panoramas
inline size_t splitk_ws_floats(int B)
{
    const size_t per = SPLITK_WS_FLOATS_PER_PANO * (size_t)B, floor_ = (size_t)16 * 4 * 512 * 512;       // 16 slices of layer4's 3x3 tile set at B = 4
    return per > floor_ ? per : floor_;
}

Plan make_plan(int B)
{
    Plan p;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t b = (size_t)B;
    p.sync = take(HN_SYNC_WORDS);
    p.xn = take(b * IMG_H * IMG_W * 4);
    p.stem = take(b * 256 * 512 * 64);
    p.pool = take(b * 128 * 256 * 64);
    p.p0 = take(b * 128 * 256 * 256);
    p.p1 = take(b * 128 * 256 * 256);
    p.t1 = take(b * 128 * 256 * 128);     // largest conv1 output: layer2.0 (128 planes at 128x256)
    p.t2 = take(b * 128 * 256 * 64);      // largest conv2 output: layer1 (64 planes at 128x256)
    p.ds = take(b * 128 * 256 * 256);
    p.g0 = take(b * 64 * 256 * 128);      // largest height-compression output: ghc0.0
    p.g1 = take(b * 64 * 256 * 128);
    p.seq = take((size_t)T_COLS * b * 1024);
    p.gx = take((size_t)T_COLS * b * 4096);
    p.y1 = take((size_t)T_COLS * b * 1024);
    p.y2 = take((size_t)T_COLS * b * 1024);
    p.sk = take(splitk_ws_floats((int)b));           // split-K partial tiles (conv_igemm_f32.hip)
    for (int li = 0; li < 4; ++li) p.c[li] = take(b * (128 >> li) * (256 >> li) * (size_t)(256 << li));
    p.sk2 = take(splitk_ws_floats((int)b));
    p.total = off;
    return p;
}

}  // namespace

// Engine-owned streams: plain non-blocking HIP streams of NORMAL priority.
// Round 6 (tools/stream_pool_probe.py): up to round 5 the head stream was created with the highest priority ("the few workgroups of the recurrence
// must win the dispatcher's arbitration").  The HIP runtime keeps a separate small pool of hardware queues per priority, and a process that had
// run ONE RCCL collective before the engine existed then lost 26 % of the pipelined bf16 forward (5129 -> 3776 panoramas/s at B = 32; 11 % in
// float32; unchanged by GPU_MAX_HW_QUEUES = 8 / 16, by 36 idle or 4 used torch pool streams, by destroying the process group; gone when the
// collective came AFTER the engine's first forward).  With a normal-priority head stream the same process runs at 5258 panoramas/s and a clean one
// at 5181 (5106-5129 with the priority): the priority bought nothing and made the engine's overlap depend on what the process did first.
// HN_HEAD_PRIO=high restores it (A/B runs).
int hn_make_stream(hipStream_t* out, bool high_priority)
{
    static const char* hp = getenv("HN_HEAD_PRIO");
    if (high_priority && hp && hp[0] == 'h') {
        int least = 0, greatest = 0;
        HN_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HN_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest));
    } else {
        HN_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    return 0;
}

// ---- engine API --------------------------------------------------------------------------------
extern "C" int hn_create(hn_engine** out, int device)
{
    HN_REQUIRE(out != nullptr, "hn_create: out is null");
    int n = 0;
    HN_HIP(hipGetDeviceCount(&n));
    HN_REQUIRE(device >= 0 && device < n, "hn_create: device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    HN_HIP(hipGetDeviceProperties(&prop, device));
    HN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "hn_create: device %d is %s; this library is built for gfx950 only",
               device, prop.gcnArchName);
    HN_REQUIRE(prop.multiProcessorCount >= 256, "hn_create: %d CUs visible; the persistent LSTM kernel needs 256 co-resident workgroups",
               prop.multiProcessorCount);
    hn_engine* e = new hn_engine();
    e->device = device;
    if (const char* pz = getenv("HN_POISON_WS")) {      // "1" / "ff": NaN bytes; "7f": huge finite values; "0": off (see hn_engine::poison)
        if (pz[0] == '7') e->poison = 0x7F;
        else if (pz[0] != '0' && pz[0] != 0) e->poison = 0xFF;
    }
    {
        DeviceGuard guard(device);
        HN_REQUIRE(guard.ok, "hn_create: cannot select device %d", device);
        // the height-compression branches' stream (engine-owned, see hn_make_stream)
        if (int rc = hn_make_stream(&e->branch_stream, /*high_priority=*/false)) { delete e; return rc; }
        for (int i = 0; i < 4; ++i) {
            HN_HIP(hipEventCreateWithFlags(&e->ev_fork[i], hipEventDisableTiming));
            HN_HIP(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
        }
    }
    *out = e;
    return 0;
}

extern "C" int hn_destroy(hn_engine* e)
{
    if (e) {
        for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
        for (int i = 0; i < 4; ++i) {
            if (e->ev_fork[i]) (void)hipEventDestroy(e->ev_fork[i]);
            if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
        }
        if (e->branch_stream) (void)hipStreamDestroy(e->branch_stream);
        for (int k = 0; k < 2; ++k) {
            if (e->ev_trunk[k]) (void)hipEventDestroy(e->ev_trunk[k]);
            if (e->ev_head[k]) (void)hipEventDestroy(e->ev_head[k]);
        }
        if (e->head_stream) (void)hipStreamDestroy(e->head_stream);
        DeviceGuard guard(e->device);
        e->jt_pack.release();
        e->jt_pack_h.release();
        for (JobTable& t : e->jt_bwd) t.release();
    }
    delete e;
    return 0;
}

// Host step of the Manhattan fit (see the header): two-pointer form of the reference's span matrix, float64 throughout.
extern "C" int hn_vote_scan(const double* v, int L, double tol, int32_t* best3)
{
    HN_REQUIRE(v != nullptr && best3 != nullptr && L >= 0, "hn_vote_scan: bad argument");
    int best_span = -1, best_i = -1, best_j = -1;
    int j = 0;
    for (int i = 0; i < L; ++i) {
        if (j < i) j = i;
        while (j + 1 < L && !((v[j + 1] - v[i]) + 1e-9 > tol)) ++j;      // largest j with (v_j - v_i) + 1e-9 <= tol
        if (j == i && 0.0 > tol) continue;
        const int span = j - i + 1;
        if (!((double)span < (double)L * 0.4) && span > best_span) {
            best_span = span;
            best_i = i;
            best_j = j;
        }
    }
    best3[0] = best_span;
    best3[1] = best_i;
    best3[2] = best_j;
    return 0;
}

extern "C" int hn_set_option(hn_engine* e, const char* name, int value)
{
    HN_REQUIRE(e != nullptr && name != nullptr, "hn_set_option: null argument");
    if (strcmp(name, "branch_stream") == 0) { e->use_branch_stream = value ? 1 : 0; return 0; }
    if (strcmp(name, "bf16_lstm") == 0) { e->bf16_lstm = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_downsample") == 0) { e->fuse_downsample = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_stem_conv1") == 0) { e->fuse_stem_conv1 = value ? 1 : 0; return 0; }
    if (strcmp(name, "defer_join") == 0) { e->defer_join = value != 0; return 0; }
    if (strcmp(name, "fuse_stem_pool") == 0) { e->fuse_stem_pool = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_stem_bnpool") == 0) { e->fuse_stem_bnpool = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_stem_poolbwd") == 0) { e->fuse_stem_poolbwd = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_bn_dual") == 0) { e->fuse_bn_dual = value ? 1 : 0; return 0; }
    if (strcmp(name, "fuse_bn_fold") == 0) { e->fuse_bn_fold = value == 2 ? 2 : (value ? 1 : 0); return 0; }      // 2: folded adjoint over a classical forward
    if (strcmp(name, "f32_branch") == 0) { e->f32_branch = value ? 1 : 0; return 0; }
    if (strcmp(name, "fold_deterministic") == 0) { e->fold_deterministic = value ? 1 : 0; return 0; }
    if (strcmp(name, "chain_layer1") == 0) { e->chain_layer1 = value ? 1 : 0; return 0; }
    if (strcmp(name, "lstm_wide_rows") == 0) { e->wide_rows = value == 8 ? 8 : 16; return 0; }
    if (strcmp(name, "lstm_wide_xcds") == 0) { e->wide_xcds = value == 2 ? 2 : 1; return 0; }
    if (strcmp(name, "poison_ws") == 0) { e->poison = value == 2 ? 0x7F : (value ? 0xFF : -1); return 0; }       // 0 off, 1 NaN bytes, 2 0x7F bytes
    HN_REQUIRE(false, "hn_set_option: unknown option '%s'", name);
}

extern "C" int hn_set_forward_tap(hn_engine* e, const char* name, void* dst)
{
    HN_REQUIRE(e != nullptr, "hn_set_forward_tap: null engine");
    if (name == nullptr) {
        e->taps.clear();
        return 0;
    }
    static const char* known[] = {"stem", "pool", "c1", "c2", "c3", "c4", "feature", "lstm"};
    bool ok = false;
    for (const char* k : known) ok = ok || std::string(k) == name;
    HN_REQUIRE(ok, "hn_set_forward_tap: unknown tap '%s'", name);
    if (dst) e->taps[name] = dst; else e->taps.erase(name);
    return 0;
}

extern "C" int hn_set_profiling(hn_engine* e, int on)
{
    HN_REQUIRE(e != nullptr, "hn_set_profiling: null engine");
    e->profiling = on != 0;
    return 0;
}

extern "C" int hn_profile_count(hn_engine* e) { return e ? (int)e->prof.size() : 0; }

extern "C" int hn_profile_entry(hn_engine* e, int i, char* name, int name_cap, float* ms, double* flops)
{
    HN_REQUIRE(e && name && ms && flops && i >= 0 && i < (int)e->prof.size(), "hn_profile_entry: bad index/argument");
    const ProfEntry& pe = e->prof[i];
    HN_HIP(hipEventSynchronize(pe.t1));
    HN_HIP(hipEventElapsedTime(ms, pe.t0, pe.t1));
    snprintf(name, name_cap, "%s", pe.name.c_str());
    *flops = pe.flops;
    return 0;
}

extern "C" int hn_bind_tensor(hn_engine* e, const char* name, const void* ptr, int64_t numel)
{
    HN_REQUIRE(e && name && ptr, "hn_bind_tensor: null argument");
    const Arch& a = arch();
    auto it = a.numel.find(name);
    HN_REQUIRE(it != a.numel.end(), "hn_bind_tensor: unknown state_dict key '%s'", name);
    HN_REQUIRE(it->second == numel, "hn_bind_tensor: '%s' has %lld elements, expected %lld", name, (long long)numel,
               (long long)it->second);
    e->bound[name] = ptr;
    return 0;
}

extern "C" size_t hn_packed_bytes(void) { return arch().packed_floats * sizeof(float); }

extern "C" size_t hn_workspace_bytes(int B)
{
    if (B < 1) return 0;
    return make_plan(B).total * sizeof(float);
}

extern "C" int hn_pack_weights(hn_engine* e, void* packed, size_t packed_bytes, void* stream)
{
    HN_REQUIRE(e && packed, "hn_pack_weights: null argument");
    HN_REQUIRE(packed_bytes >= hn_packed_bytes(), "hn_pack_weights: packed buffer too small (%zu < %zu)", packed_bytes,
               hn_packed_bytes());
    const Arch& a = arch();
    for (const auto& kv : a.numel) {
        if (kv.first.size() > 19 && kv.first.compare(kv.first.size() - 19, 19, "num_batches_tracked") == 0) continue;
        HN_REQUIRE(e->bound.count(kv.first) != 0, "hn_pack_weights: state_dict key '%s' was never bound", kv.first.c_str());
    }
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_pack_weights: cannot select device %d", e->device);
    hipStream_t s = (hipStream_t)stream;
    float* P = reinterpret_cast<float*>(packed);
    auto get = [&](const std::string& k) { return reinterpret_cast<const float*>(e->bound.at(k)); };
    // one table-driven launch for all 69 conv packings + BatchNorm folds + the LSTM / head copies (multi_job.h); it used to be
    // ~150 launches, repeated after every optimiser step of a training loop
    if (int rc = hn_poison(e, packed, hn_packed_bytes(), s)) return rc;
    std::vector<MJob> jobs;
    jobs.reserve(2 * a.convs.size() + 20);
    for (const auto& c : a.convs) {
        jobs.push_back(mj_pack_f32(get(c.wkey + ".weight"), P + c.w_off, c.cout, c.cin, c.k, c.k));
        jobs.push_back(mj_fold_bn(get(c.bnkey + ".weight"), get(c.bnkey + ".bias"), get(c.bnkey + ".running_mean"),
                                  get(c.bnkey + ".running_var"), c.has_bias ? get(c.wkey + ".bias") : nullptr, P + c.scale_off,
                                  P + c.shift_off, c.cout));
    }
    for (int l = 0; l < 2; ++l) {
        for (int d = 0; d < 2; ++d) {
            const std::string suf = "_l" + std::to_string(l) + (d ? "_reverse" : "");
            jobs.push_back(mj_make(MJ_COPY_F32, get("bi_rnn.weight_ih" + suf), P + a.wih_off[l] + (size_t)d * 2048 * 1024, (long long)2048 * 1024));
            jobs.push_back(mj_make(MJ_COPY_F32, get("bi_rnn.weight_hh" + suf), P + a.whh_off[l][d], (long long)2048 * 512));
            MJob add = mj_make(MJ_ADD_VEC, get("bi_rnn.bias_ih" + suf), P + a.lbias_off[l] + d * 2048, 2048);
            add.aux[0] = get("bi_rnn.bias_hh" + suf);
            jobs.push_back(add);
        }
    }
    // ones / zeros vectors: scale of the LSTM input-projection "conv" (no BN there)
    jobs.push_back(mj_fold_bn(nullptr, nullptr, nullptr, nullptr, nullptr, P + a.ones_off, P + a.zeros_off, 4096));
    jobs.push_back(mj_make(MJ_COPY_F32, get("linear.weight"), P + a.linw_off, 12 * 1024));
    jobs.push_back(mj_make(MJ_COPY_F32, get("linear.bias"), P + a.linb_off, 12));
    if (int rc = e->jt_pack.run(jobs, s)) return rc;
    e->packed = P;
    return 0;
}

namespace {

int run_conv(hn_engine* e, const float* P, const ConvLayer& c, const float* x, float* y, const float* res, int B, int Hi,
             int Wi, int sh, int sw, int relu, hipStream_t s, float* splitk_ws = nullptr, size_t splitk_ws_floats = 0)
{
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = P + c.w_off; d.scale = P + c.scale_off; d.shift = P + c.shift_off; d.res = res; d.y = y;
    d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = c.cin; d.Cout = c.cout;
    d.KH = c.k; d.KW = c.k; d.sh = sh; d.sw = sw; d.ph = c.k / 2; d.pw = c.k / 2;
    d.Ho = (Hi + 2 * d.ph - c.k) / sh + 1;
    d.Wo = (Wi + 2 * d.pw - c.k) / sw + 1;
    d.relu = relu; d.ldy = c.cout; d.stem = 0;
    d.splitk_ws = splitk_ws; d.splitk_ws_floats = splitk_ws_floats;
    ProfScope ps(e, s, c.wkey, 2.0 * B * d.Ho * d.Wo * (double)c.cout * c.cin * c.k * c.k);
    return hn_launch_conv(d, s);
}

int run_gemm_rows(const float* x, const float* w, const float* scale, const float* shift, float* y, long rows, int K, int N,
                  hipStream_t s)
{
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = w; d.scale = scale; d.shift = shift; d.res = nullptr; d.y = y;
    d.B = 1; d.Hi = 1; d.Wi = (int)rows; d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1; d.ph = 0; d.pw = 0;
    d.Ho = 1; d.Wo = (int)rows; d.relu = 0; d.ldy = N; d.stem = 0;
    return hn_launch_conv(d, s);
}

}  // namespace

namespace {

// Buffers of the recurrent head of one in-flight batch (see engine_bf16.hip): the plain forward has one set in Plan, the
// pipelined entry (hn_forward_submit) two.
struct HeadBufsF {
    float* sync;     // HN_SYNC_WORDS uint32
    float* seq;      // [T*B][1024] column features
    float* gx;       // [T*B][4096] gate pre-activations
    float* y1;       // [T*B][1024] layer-0 output
    float* y2;       // [T*B][1024] layer-1 output
};

// The convolutional trunk of hn_forward (model.py:248-252,73-81,123-179) on stream s, writing the column features to `seq`.
// The height-compression chain of scale li (model.py:138-156) depends only on C_li: like the bf16 trunk (engine_bf16.hip: run_trunk_h) it is
// forked onto the engine's branch stream and runs beside the ResNet stages that follow -- the chains are matrix-bound (ghc*.0: 4.8 GMAC each at
// 125-139 TF) with low-occupancy tails (ghc*.2 / *.3 and their split-K reduces), the stages carry the HBM-bound 1x1 convs: the two streams fill
// each other's idle matrix cycles.  defer_join (pipelined entry): the caller's stream does not wait for the chains at the end -- the recurrent
// head does -- so the NEXT batch's stem / layer1 (HBM-bound in float32) run beside this batch's last chain (ghc3: 3.2 ms of deep-K matrix work).
int run_trunk_f(hn_engine* e, const float* x, int B, int C_in, float* W, const Plan& pl, float* seq, hipStream_t s, bool defer_join = false,
                bool* forked = nullptr)
{
    const Arch& a = arch();
    const float* P = e->packed;
    int rc;
    // stem: normalise + 7x7/2 conv + BN + ReLU, then 3x3/2 max-pool  (model.py:248-252,73-76)
    {
        const ConvLayer& c = a.convs[a.stem];
        ProfScope ps(e, s, "stem(prep+conv7x7+maxpool)", 2.0 * B * 256 * 512 * 64.0 * 147);
        // one kernel (stem_pool_f32.hip: normalise + conv + BN + ReLU + max-pool out of an LDS ring of input rows) unless a parity tap
        // wants the 2.1 GB stem activation itself, which only the three-launch form writes (option "fuse_stem_pool", HN_F32_STEM_POOL=0: A/B)
        static const char* fsp = getenv("HN_F32_STEM_POOL");
        if (e->fuse_stem_pool && e->taps.empty() && !(fsp && fsp[0] == '0')) {
            if ((rc = hn_launch_stem_pool_f32(x, C_in, P + c.w_off, P + c.scale_off, P + c.shift_off, W + pl.pool, B, s))) return rc;
        } else if ((rc = hn_stem(x, B, C_in, IMG_H, IMG_W, P + c.w_off, P + c.scale_off, P + c.shift_off, W + pl.xn, W + pl.stem,
                                 W + pl.pool, (void*)s)))
            return rc;
    }
    if ((rc = e->tap("stem", W + pl.stem, (size_t)B * 256 * 512 * 64 * sizeof(float), s))) return rc;
    if ((rc = e->tap("pool", W + pl.pool, (size_t)B * 128 * 256 * 64 * sizeof(float), s))) return rc;

    // ResNet-50 stages (model.py:78-81) + per-scale height compression (model.py:138-156,172-179)
    const float* cur = W + pl.pool;
    int H = 128, Wd = 256;
    const int nblk[4] = {3, 4, 6, 3};
    // Measured in round 6 (one box).  B = 32: 846.5 panoramas/s on one stream, 843.3 forked + joined, 849.8 forked + deferred join -- unlike bf16 (whose
    // stages are HBM-bound) every float32 launch already fills the matrix pipes of every compute unit, two workgroups of 64 KiB LDS each per CU, so a
    // second stream has nothing to fill.  The interactive regime is different: at B <= 4 a launch is 32 .. 128 tiles on 256 CUs and the chains
    // (0.77 of 3.7 ms at B = 1) run beside the stages: plain forward 3.79 -> 3.57 ms (B = 1), 5.55 -> 5.16 (B = 2), 7.11 -> 6.9 (B = 4), 12.0 -> 11.9
    // (B = 8), 20.4 -> 20.7 (B = 16).  So: forked for B <= 4, one stream above (option "f32_branch" / HN_F32_BRANCH=1: forked at any batch size;
    // option "branch_stream" = 0: never).
    static const char* fenv = getenv("HN_F32_BRANCH");
    const bool fork = e->use_branch_stream && !e->profiling && e->branch_stream != nullptr && (B <= 4 || e->f32_branch || (fenv && fenv[0] == '1'));
    hipStream_t sb = fork ? e->branch_stream : s;
    if (forked) *forked = fork;
    for (int li = 0; li < 4; ++li) {
        // C_li is about to be overwritten: a chain of an EARLIER batch that was not joined (deferred) must have finished reading it
        if (e->join_deferred[li]) {
            HN_HIP(hipStreamWaitEvent(s, e->ev_join[li], 0));
            e->join_deferred[li] = false;
        }
        for (int j = 0; j < nblk[li]; ++j) {
            const int stride = (j == 0 && li > 0) ? 2 : 1;
            const ConvLayer& c1 = a.convs[a.block_first[li][j]];
            const ConvLayer& c2 = a.convs[a.block_first[li][j] + 1];
            const ConvLayer& c3 = a.convs[a.block_first[li][j] + 2];
            float* out = (j == nblk[li] - 1) ? W + pl.c[li] : ((cur == W + pl.p0) ? W + pl.p1 : W + pl.p0);
            // interactive regime (B <= 4): the deep-K convs of the stages give a 256-CU part 32 .. 128 tiles -- split-K like the height-compression
            // tails (a scratch of the caller's stream's own: the chains on the branch stream use pl.sk); at larger batches nothing changes
            float* sk = B <= 4 ? W + pl.sk2 : nullptr;
            const size_t skf = B <= 4 ? splitk_ws_floats(B) : 0;
            if ((rc = run_conv(e, P, c1, cur, W + pl.t1, nullptr, B, H, Wd, 1, 1, 1, s, sk, skf))) return rc;
            if ((rc = run_conv(e, P, c2, W + pl.t1, W + pl.t2, nullptr, B, H, Wd, stride, stride, 1, s, sk, skf))) return rc;
            const float* idt = cur;
            // block 0: downsample + conv3 + add + ReLU as ONE dual-accumulator launch (conv1x1_dual_f32_kernel): the
            // downsample output never goes to HBM; bit-identical to the two-launch form (option "fuse_downsample")
            const bool dual = j == 0 && e->fuse_downsample;
            if (j == 0 && !dual) {
                const ConvLayer& cd = a.convs[a.block_down[li]];
                if ((rc = run_conv(e, P, cd, cur, W + pl.ds, nullptr, B, H, Wd, stride, stride, 0, s))) return rc;
                idt = W + pl.ds;
            }
            const int Hin = H, Win = Wd;
            H /= stride;
            Wd /= stride;
            if (dual) {
                const ConvLayer& cd = a.convs[a.block_down[li]];
                ProfScope ps(e, s, c3.wkey + "+downsample", 2.0 * B * H * Wd * (double)c3.cout * (c3.cin + cd.cin));
                if ((rc = hn_launch_conv1x1_dual_f32(W + pl.t2, P + c3.w_off, P + c3.scale_off, P + c3.shift_off, cur, P + cd.w_off,
                                                     P + cd.scale_off, P + cd.shift_off, out, B, H, Wd, c3.cin, Hin, Win, cd.cin, stride,
                                                     c3.cout, s)))
                    return rc;
            } else if ((rc = run_conv(e, P, c3, W + pl.t2, out, idt, B, H, Wd, 1, 1, 1, s, sk, skf)))
                return rc;
            cur = out;
        }
        {
            const char* cname[4] = {"c1", "c2", "c3", "c4"};
            if ((rc = e->tap(cname[li], cur, (size_t)B * H * Wd * (256 << li) * sizeof(float), s))) return rc;
        }
        // GlobalHeightConv for this scale
        if (fork) {
            HN_HIP(hipEventRecord(e->ev_fork[li], s));
            HN_HIP(hipStreamWaitEvent(sb, e->ev_fork[li], 0));
        }
        const float* gin = cur;
        int gh = H;
        float* gbuf[2] = {W + pl.g0, W + pl.g1};
        for (int k = 0; k < 4; ++k) {
            const ConvLayer& gc = a.convs[a.ghc_first[li] + k];
            float* gout = gbuf[k & 1];
            if ((rc = run_conv(e, P, gc, gin, gout, nullptr, B, gh, Wd, 2, 1, 1, sb, W + pl.sk, splitk_ws_floats(B)))) return rc;
            gin = gout;
            gh /= 2;
        }
        const int cq = a.convs[a.ghc_first[li] + 3].cout;
        {
            ProfScope ps(e, sb, "upsample_flatten." + std::to_string(li), 0.0);
            if ((rc = hn_launch_upsample_flatten(gin, seq, B, gh, Wd, cq, 256 * li, sb))) return rc;
        }
        if (fork) HN_HIP(hipEventRecord(e->ev_join[li], sb));
    }
    if (fork) {
        if (defer_join) {
            for (int li = 0; li < 4; ++li) e->join_deferred[li] = true;
        } else {
            for (int li = 0; li < 4; ++li) HN_HIP(hipStreamWaitEvent(s, e->ev_join[li], 0));
        }
    }
    return 0;
}

// The recurrent head (model.py:263-269): bi-LSTM x2 + Linear, on stream s.  wide: the 64-compute-unit recurrence kernel of the
// pipelined entry (lstm_wide_f32.hip) instead of the 256-workgroup one (lstm.hip).
int run_head_f(hn_engine* e, int B, const HeadBufsF& hb, float* bon, float* cor, bool wide, hipStream_t s)
{
    const Arch& a = arch();
    const float* P = e->packed;
    int rc;
    HN_HIP(hipMemsetAsync(hb.sync, 0, HN_STATUS_WORD * sizeof(unsigned), s));   // arrival counters; the status word behind them is sticky (zeroed by the caller at allocation)
    const long rows = (long)T_COLS * B;
    if ((rc = e->tap("feature", hb.seq, (size_t)rows * 1024 * sizeof(float), s))) return rc;
    const float* lin = hb.seq;
    float* ybuf[2] = {hb.y1, hb.y2};
    for (int l = 0; l < 2; ++l) {
        {
            ProfScope ps(e, s, "bi_rnn.l" + std::to_string(l) + ".input_gemm", 2.0 * rows * 1024.0 * 4096);
            if ((rc = run_gemm_rows(lin, P + a.wih_off[l], P + a.ones_off, P + a.lbias_off[l], hb.gx, rows, 1024, 4096, s))) return rc;
        }
        {
            ProfScope ps(e, s, "bi_rnn.l" + std::to_string(l) + ".recurrence", 2.0 * rows * 512.0 * 2048 * 2);
            if (wide) {
                if ((rc = hn_launch_lstm_layer_f32_wide(hb.gx, P + a.whh_off[l][0], P + a.whh_off[l][1], ybuf[l], T_COLS, B, hb.sync, s))) return rc;
            } else if ((rc = hn_launch_lstm_layer(hb.gx, P + a.whh_off[l][0], P + a.whh_off[l][1], ybuf[l], T_COLS, B, hb.sync, s)))
                return rc;
        }
        lin = ybuf[l];
    }
    if ((rc = e->tap("lstm", hb.y2, (size_t)rows * 1024 * sizeof(float), s))) return rc;
    ProfScope ps(e, s, "linear", 2.0 * rows * 1024.0 * 12);
    return hn_launch_linear_head(hb.y2, P + a.linw_off, P + a.linb_off, bon, cor, T_COLS, B, s);
}

struct PlanPF {
    Plan base;
    size_t sync1, seq1, gx1, y11, y21, total;      // float offsets of the second head buffer set
};

PlanPF make_plan_pf(int B)
{
    PlanPF p;
    p.base = make_plan(B);
    size_t off = p.base.total;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t b = (size_t)B;
    p.sync1 = take(HN_SYNC_WORDS);
    p.seq1 = take((size_t)T_COLS * b * 1024);
    p.gx1 = take((size_t)T_COLS * b * 4096);
    p.y11 = take((size_t)T_COLS * b * 1024);
    p.y21 = take((size_t)T_COLS * b * 1024);
    p.total = off;
    return p;
}

}  // namespace

extern "C" int hn_forward(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                          size_t workspace_bytes, void* stream)
{
    HN_REQUIRE(e && x && bon && cor && workspace, "hn_forward: null argument");
    HN_REQUIRE(e->packed != nullptr, "hn_forward: hn_pack_weights has not been called");
    HN_REQUIRE(B >= 1 && C_in >= 3, "hn_forward: bad B=%d / C_in=%d", B, C_in);
    HN_REQUIRE((size_t)B * 128 * 256 <= 0x7fffffffull / 2, "hn_forward: batch %d too large for 32-bit row indices", B);
    const Plan pl = make_plan(B);
    HN_REQUIRE(workspace_bytes >= pl.total * sizeof(float), "hn_forward: workspace too small (%zu < %zu)", workspace_bytes,
               pl.total * sizeof(float));
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward: cannot select device %d", e->device);
    hipStream_t s = (hipStream_t)stream;
    float* W = reinterpret_cast<float*>(workspace);
    e->prof.clear();
    e->events_used = 0;
    // (everything behind the sync page is dead at entry -- unless a chain of an earlier PIPELINED batch is still un-joined on the branch stream)
    if (!(e->join_deferred[0] || e->join_deferred[1] || e->join_deferred[2] || e->join_deferred[3])) {
        if (int rc = hn_poison(e, W + pl.xn, (pl.total - pl.xn) * sizeof(float), s)) return rc;
        if (int rc = hn_poison(e, bon, (size_t)B * 2 * 1024 * sizeof(float), s)) return rc;
        if (int rc = hn_poison(e, cor, (size_t)B * 1024 * sizeof(float), s)) return rc;
    }
    if (int rc = run_trunk_f(e, x, B, C_in, W, pl, W + pl.seq, s)) return rc;
    const HeadBufsF hb = {W + pl.sync, W + pl.seq, W + pl.gx, W + pl.y1, W + pl.y2};
    return run_head_f(e, B, hb, bon, cor, false, s);
}

// ---- pipelined entry (float32): trunk of batch i+1 beside the recurrent head of batch i; see hn_forward_bf16_submit ----------
extern "C" size_t hn_workspace_pipelined_bytes(int B)
{
    if (B < 1) return 0;
    return make_plan_pf(B).total * sizeof(float);
}

int hn_ensure_head_stream(hn_engine* e)
{
    if (e->head_stream != nullptr) return 0;
    if (int rc = hn_make_stream(&e->head_stream, /*high_priority=*/true)) return rc;       // (normal priority unless HN_HEAD_PRIO=high: see hn_make_stream)
    for (int k = 0; k < 2; ++k) {
        HN_HIP(hipEventCreateWithFlags(&e->ev_trunk[k], hipEventDisableTiming));
        HN_HIP(hipEventCreateWithFlags(&e->ev_head[k], hipEventDisableTiming));
    }
    return 0;
}

extern "C" int hn_forward_submit(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                                 size_t workspace_bytes, int slot, void* stream)
{
    HN_REQUIRE(e && x && bon && cor && workspace, "hn_forward_submit: null argument");
    HN_REQUIRE(e->packed != nullptr, "hn_forward_submit: hn_pack_weights has not been called");
    HN_REQUIRE(B >= 1 && C_in >= 3, "hn_forward_submit: bad B=%d / C_in=%d", B, C_in);
    HN_REQUIRE((size_t)B * 128 * 256 <= 0x7fffffffull / 2, "hn_forward_submit: batch %d too large for 32-bit row indices", B);
    HN_REQUIRE(slot == 0 || slot == 1, "hn_forward_submit: slot must be 0 or 1 (got %d)", slot);
    HN_REQUIRE(!e->profiling && e->taps.empty(), "hn_forward_submit: profiling / taps are served by hn_forward only");
    const PlanPF pp = make_plan_pf(B);
    HN_REQUIRE(workspace_bytes >= pp.total * sizeof(float), "hn_forward_submit: workspace too small (%zu < %zu; hn_workspace_pipelined_bytes)",
               workspace_bytes, pp.total * sizeof(float));
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward_submit: cannot select device %d", e->device);
    if (int rc = hn_ensure_head_stream(e)) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* W = reinterpret_cast<float*>(workspace);
    const Plan& pl = pp.base;
    HeadBufsF hb;
    if (slot == 0) hb = {W + pl.sync, W + pl.seq, W + pl.gx, W + pl.y1, W + pl.y2};
    else hb = {W + pp.sync1, W + pp.seq1, W + pp.gx1, W + pp.y11, W + pp.y21};
    if (e->head_pending[slot]) HN_HIP(hipStreamWaitEvent(s, e->ev_head[slot], 0));   // this slot's seq was last read by the head of two submits ago
    static const char* djenv = getenv("HN_F32_DEFER_JOIN");      // "0": the caller's stream joins the four chains at the end of every trunk (A/B runs)
    const bool defer = !(djenv && djenv[0] == '0');
    bool forked = false;
    if (int rc = run_trunk_f(e, x, B, C_in, W, pl, hb.seq, s, defer, &forked)) return rc;
    HN_HIP(hipEventRecord(e->ev_trunk[slot], s));
    HN_HIP(hipStreamWaitEvent(e->head_stream, e->ev_trunk[slot], 0));
    if (forked && defer)
        for (int li = 0; li < 4; ++li) HN_HIP(hipStreamWaitEvent(e->head_stream, e->ev_join[li], 0));
    // (the 256-workgroup recurrence beside the next trunk instead of the 64-CU wide one was re-measured in round 6: 861-863 panoramas/s either way)
    if (int rc = run_head_f(e, B, hb, bon, cor, true, e->head_stream)) return rc;
    HN_HIP(hipEventRecord(e->ev_head[slot], e->head_stream));
    e->head_pending[slot] = true;
    return 0;
}

extern "C" int hn_forward_collect(hn_engine* e, int slot, void* stream)
{
    HN_REQUIRE(e != nullptr && (slot == 0 || slot == 1), "hn_forward_collect: bad argument");
    HN_REQUIRE(e->head_pending[slot], "hn_forward_collect: nothing was submitted on slot %d", slot);
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward_collect: cannot select device %d", e->device);
    HN_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_head[slot], 0));
    return 0;
}

extern "C" int hn_pipelined_status_offset_f32(int B, int slot, size_t* byte_offset)
{
    HN_REQUIRE(B >= 1 && (slot == 0 || slot == 1) && byte_offset, "hn_pipelined_status_offset_f32: bad argument");
    const PlanPF pp = make_plan_pf(B);
    *byte_offset = ((slot == 0 ? pp.base.sync : pp.sync1) + HN_STATUS_WORD) * sizeof(float);
    return 0;
}

// per-stage entry (tests): one bi-LSTM layer with the wide float32 recurrence kernel of hn_forward_submit
extern "C" int hn_lstm_layer_wide(const float* gx, const float* whh_fwd, const float* whh_rev, float* y, int T, int B, void* sync_ws,
                                  void* stream)
{
    HN_REQUIRE(gx && whh_fwd && whh_rev && y && sync_ws, "hn_lstm_layer_wide: null pointer");
    return hn_launch_lstm_layer_f32_wide(gx, whh_fwd, whh_rev, y, T, B, sync_ws, (hipStream_t)stream);
}

extern "C" int hn_check_status(hn_engine* e, void* workspace, int* status_out)
{
    HN_REQUIRE(e && workspace && status_out, "hn_check_status: null argument");
    DeviceGuard guard(e->device);
    unsigned st = 0;
    const unsigned* w = reinterpret_cast<const unsigned*>(reinterpret_cast<float*>(workspace) + make_plan(1).sync) + HN_STATUS_WORD;
    HN_HIP(hipMemcpy(&st, w, sizeof(st), hipMemcpyDeviceToHost));
    *status_out = (int)st;
    return 0;
}

// ---- per-stage entry points --------------------------------------------------------------------
extern "C" size_t hn_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW)
{
    (void)KW;
    return packed_w_floats(Cout, Cin, KH);
}

extern "C" int hn_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream)
{
    HN_REQUIRE(w_oihw && w_packed, "hn_pack_conv_weight: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3 || (KH == 7 && Cin == 3)), "hn_pack_conv_weight: unsupported %dx%d", KH, KW);
    return hn_launch_pack_conv(w_oihw, w_packed, Cout, Cin, KH, KW, (hipStream_t)stream);
}

extern "C" int hn_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                          float* scale, float* shift, int C, void* stream)
{
    HN_REQUIRE(scale && shift && C > 0, "hn_fold_bn: bad argument");
    const int n = (gamma != nullptr) + (beta != nullptr) + (mean != nullptr) + (var != nullptr);
    HN_REQUIRE(n == 0 || n == 4, "hn_fold_bn: gamma/beta/mean/var must be all set or all NULL");
    return hn_launch_fold_bn(gamma, beta, mean, var, bias, scale, shift, C, (hipStream_t)stream);
}

extern "C" int hn_conv2d_nhwc(const float* x, const float* w_packed, const float* scale, const float* shift,
                              const float* res, float* y, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int sh,
                              int sw, int relu, void* stream)
{
    HN_REQUIRE(x && w_packed && scale && shift && y, "hn_conv2d_nhwc: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_nhwc: kernel %dx%d unsupported (1x1 / 3x3)", KH, KW);
    HN_REQUIRE(B >= 1 && Hi >= 1 && Wi >= 1 && sh >= 1 && sw >= 1, "hn_conv2d_nhwc: bad geometry");
    HN_REQUIRE(KW / 2 <= Wi, "hn_conv2d_nhwc: circular pad wider than the image");
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = w_packed; d.scale = scale; d.shift = shift; d.res = res; d.y = y;
    d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = Cin; d.Cout = Cout; d.KH = KH; d.KW = KW; d.sh = sh; d.sw = sw;
    d.ph = KH / 2; d.pw = KW / 2;
    d.Ho = (Hi + 2 * d.ph - KH) / sh + 1;
    d.Wo = (Wi + 2 * d.pw - KW) / sw + 1;
    HN_REQUIRE((long)B * d.Ho * d.Wo < 0x7fffffffL, "hn_conv2d_nhwc: too many output pixels");
    d.relu = relu; d.ldy = Cout; d.stem = 0;
    return hn_launch_conv(d, (hipStream_t)stream);
}

extern "C" int hn_stem(const float* x_nchw, int B, int C_in, int H, int W, const float* w_packed, const float* scale,
                       const float* shift, float* tmp_nhwc4, float* stem_out, float* pool_out, void* stream)
{
    HN_REQUIRE(x_nchw && w_packed && scale && shift && tmp_nhwc4 && stem_out && pool_out, "hn_stem: null pointer");
    HN_REQUIRE(B >= 1 && C_in >= 3 && H % 4 == 0 && W % 4 == 0 && W >= 8, "hn_stem: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = hn_launch_prep_nhwc4(x_nchw, tmp_nhwc4, B, C_in, H, W, s))) return rc;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = tmp_nhwc4; d.w = w_packed; d.scale = scale; d.shift = shift; d.res = nullptr; d.y = stem_out;
    d.B = B; d.Hi = H; d.Wi = W; d.Cin = 4; d.Cout = 64; d.KH = 7; d.KW = 7; d.sh = 2; d.sw = 2; d.ph = 3; d.pw = 3;
    d.Ho = H / 2; d.Wo = W / 2; d.relu = 1; d.ldy = 64; d.stem = 1;
    if ((rc = hn_launch_conv(d, s))) return rc;
    return hn_launch_maxpool(stem_out, pool_out, B, H / 2, W / 2, 64, s);
}

extern "C" int hn_stem_pool_f32(const float* x_nchw, int B, int C_in, const float* w_packed, const float* scale, const float* shift, float* pool_out,
                                void* stream)
{
    HN_REQUIRE(x_nchw && w_packed && scale && shift && pool_out, "hn_stem_pool_f32: null pointer");
    return hn_launch_stem_pool_f32(x_nchw, C_in, w_packed, scale, shift, pool_out, B, (hipStream_t)stream);
}

extern "C" int hn_upsample_flatten(const float* in, float* seq, int B, int hq, int Wq, int cq, int col0, void* stream)
{
    HN_REQUIRE(in && seq && B >= 1 && hq >= 1 && cq >= 1, "hn_upsample_flatten: bad argument");
    return hn_launch_upsample_flatten(in, seq, B, hq, Wq, cq, col0, (hipStream_t)stream);
}

extern "C" int hn_lstm_layer(const float* gx, const float* whh_fwd, const float* whh_rev, float* y, int T, int B,
                             void* sync_ws, void* stream)
{
    HN_REQUIRE(gx && whh_fwd && whh_rev && y && sync_ws, "hn_lstm_layer: null pointer");
    HN_HIP(hipMemsetAsync(sync_ws, 0, HN_SYNC_WORDS * sizeof(unsigned), (hipStream_t)stream));
    return hn_launch_lstm_layer(gx, whh_fwd, whh_rev, y, T, B, sync_ws, (hipStream_t)stream);
}

extern "C" int hn_linear_head(const float* y, const float* w, const float* bias, float* bon, float* cor, int T, int B,
                              void* stream)
{
    HN_REQUIRE(y && w && bias && bon && cor && T >= 1 && B >= 1, "hn_linear_head: bad argument");
    return hn_launch_linear_head(y, w, bias, bon, cor, T, B, (hipStream_t)stream);
}

// ---- training per-stage entry points (parity tests) ----------------------------------------------
extern "C" int hn_conv2d_dgrad_nhwc(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B, int Hx,
                                    int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream)
{
    HN_REQUIRE(dz && w_oihw && dx && w_scratch, "hn_conv2d_dgrad_nhwc: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_dgrad_nhwc: kernel %dx%d unsupported", KH, KW);
    HN_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin <= 4096 && Cout <= 4096, "hn_conv2d_dgrad_nhwc: channels must be multiples of 32, <= 4096");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // ones / zeros for the epilogue live behind the packed dgrad weights in the scratch buffer
    float* ones = w_scratch + (size_t)Cout * Cin * KH * KW;
    float* zeros = ones + 4096;
    if ((rc = hn_launch_fold_bn(nullptr, nullptr, nullptr, nullptr, nullptr, ones, zeros, 4096, s))) return rc;
    ConvDesc d;                         // the FORWARD conv's geometry
    memset(&d, 0, sizeof(d));
    d.B = B; d.Hi = Hx; d.Wi = Wx; d.Cin = Cin; d.Cout = Cout; d.KH = KH; d.KW = KW; d.sh = sh; d.sw = sw; d.ph = KH / 2; d.pw = KW / 2;
    d.Ho = (Hx + 2 * d.ph - KH) / sh + 1;
    d.Wo = (Wx + 2 * d.pw - KW) / sw + 1;
    return hn_launch_conv_dgrad(d, dz, w_oihw, add, dx, w_scratch, ones, zeros, s);
}

// bf16-MFMA form of the data gradient: dz is rounded to bf16 (into the scratch), weights re-packed per class as bf16,
// add / dx stay float32.  w_scratch: Cout*Cin*KH*KW + 8192 floats (weights + ones/zeros) followed by B*Ho*Wo*Cout/2 floats.
extern "C" int hn_conv2d_dgrad_nhwc_bf16(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B,
                                         int Hx, int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream)
{
    HN_REQUIRE(dz && w_oihw && dx && w_scratch, "hn_conv2d_dgrad_nhwc_bf16: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_dgrad_nhwc_bf16: kernel %dx%d unsupported", KH, KW);
    HN_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 4096 && Cout <= 4096, "hn_conv2d_dgrad_nhwc_bf16: Cin %% 32, Cout %% 64, <= 4096");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    float* ones = w_scratch + (size_t)Cout * Cin * KH * KW;
    float* zeros = ones + 4096;
    if ((rc = hn_launch_fold_bn(nullptr, nullptr, nullptr, nullptr, nullptr, ones, zeros, 4096, s))) return rc;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.Hi = Hx; d.Wi = Wx; d.Cin = Cin; d.Cout = Cout; d.KH = KH; d.KW = KW; d.sh = sh; d.sw = sw; d.ph = KH / 2; d.pw = KW / 2;
    d.Ho = (Hx + 2 * d.ph - KH) / sh + 1;
    d.Wo = (Wx + 2 * d.pw - KW) / sw + 1;
    void* dz_h = zeros + 4096;
    const long n = (long)B * d.Ho * d.Wo * Cout;
    if ((rc = hn_launch_f32_to_bf16(dz, dz_h, n, s))) return rc;
    return hn_launch_conv_dgrad_bf16(d, dz_h, w_oihw, add, dx, w_scratch, ones, zeros, s);
}

// The form the bf16 training step actually runs: dz, the identity-branch gradient AND dx are bf16 tensors (train.hip keeps the
// gradients between conv units in bf16) -- stride-1 convs through the forward kernels with flipped taps, strided ones through the
// per-class data-gradient kernels (128x128, or the 256x256 8-wave kernel when the class has >= 224 tiles).  Host-facing buffers
// stay float32: dz / add are rounded to bf16 into the scratch, dx comes back converted.
// w_scratch: Cout*Cin*KH*KW + 8192 floats, then B*Ho*Wo*Cout/2 + 64 floats (dz), then 2 * (B*Hx*Wx*Cin/2 + 64) floats (add, dx).
extern "C" int hn_conv2d_dgrad_nhwc_bf16g(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B,
                                          int Hx, int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream)
{
    HN_REQUIRE(dz && w_oihw && dx && w_scratch, "hn_conv2d_dgrad_nhwc_bf16g: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_dgrad_nhwc_bf16g: kernel %dx%d unsupported", KH, KW);
    HN_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && Cin <= 4096 && Cout <= 4096, "hn_conv2d_dgrad_nhwc_bf16g: Cin %% 32, Cout %% 64, <= 4096");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    float* ones = w_scratch + (size_t)Cout * Cin * KH * KW;
    float* zeros = ones + 4096;
    if ((rc = hn_launch_fold_bn(nullptr, nullptr, nullptr, nullptr, nullptr, ones, zeros, 4096, s))) return rc;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.Hi = Hx; d.Wi = Wx; d.Cin = Cin; d.Cout = Cout; d.KH = KH; d.KW = KW; d.sh = sh; d.sw = sw; d.ph = KH / 2; d.pw = KW / 2;
    d.Ho = (Hx + 2 * d.ph - KH) / sh + 1;
    d.Wo = (Wx + 2 * d.pw - KW) / sw + 1;
    const long nz = (long)B * d.Ho * d.Wo * Cout, nx = (long)B * Hx * Wx * Cin;
    HN_REQUIRE(nz % 8 == 0 && nx % 8 == 0, "hn_conv2d_dgrad_nhwc_bf16g: tensor sizes must be multiples of 8");
    float* dz_h = zeros + 4096;
    float* add_h = dz_h + nz / 2 + 64;
    float* dx_h = add_h + nx / 2 + 64;
    if ((rc = hn_launch_f32_to_bf16(dz, dz_h, nz, s))) return rc;
    if (add && (rc = hn_launch_f32_to_bf16(add, add_h, nx, s))) return rc;
    if ((rc = hn_launch_conv_dgrad_bf16(d, dz_h, w_oihw, add ? add_h : nullptr, dx_h, w_scratch, ones, zeros, s, /*grad_bf16=*/1))) return rc;
    return hn_launch_bf16_to_f32(dx_h, dx, nx, s);
}

// bf16-MFMA form of the weight gradient: x and dz are rounded to bf16 into the scratch, accumulation and dw stay float32.
// scratch: Cout*KH*KW*Cin floats (packed dw) followed by (B*Hi*Wi*Cin + B*Ho*Wo*Cout) / 2 floats.
extern "C" int hn_conv2d_wgrad_nhwc_bf16(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int Hi, int Wi, int Cin,
                                         int Cout, int KH, int KW, int sh, int sw, void* stream)
{
    HN_REQUIRE(x && dz && dw_oihw && scratch, "hn_conv2d_wgrad_nhwc_bf16: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (KH == 7) {      // the stem: x is the NHWC4 float32 input [B][Hi][Wi][4] (channel 3 = 0), Cin = 3, Cout = 64, stride 2
        HN_REQUIRE(KW == 7 && Cin == 3 && Cout == 64 && sh == 2 && sw == 2, "hn_conv2d_wgrad_nhwc_bf16: the 7x7 form is the stem only");
        const long nx4 = (long)B * Hi * Wi * 4, nz4 = (long)B * (Hi / 2) * (Wi / 2) * 64;
        float* xh4 = scratch + 64 * 256;
        float* zh4 = xh4 + nx4 / 2 + 64;
        int rc4;
        if ((rc4 = hn_launch_f32_to_bf16(x, xh4, nx4, s)) || (rc4 = hn_launch_f32_to_bf16(dz, zh4, nz4, s))) return rc4;
        if ((rc4 = hn_launch_stem_wgrad_bf16(xh4, zh4, scratch, B, Hi, Wi, s, 0))) return rc4;
        return hn_launch_unpack_conv(scratch + 4, dw_oihw, 64, 3, 7, 7, 8, s);      // tap dw sits at window pixel dw + 1
    }
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_wgrad_nhwc_bf16: kernel %dx%d unsupported", KH, KW);
    const int Ho = (Hi + 2 * (KH / 2) - KH) / sh + 1, Wo = (Wi + 2 * (KW / 2) - KW) / sw + 1;
    const long nx = (long)B * Hi * Wi * Cin, nz = (long)B * Ho * Wo * Cout;
    float* xh = scratch + (size_t)Cout * KH * KW * Cin;
    float* zh = xh + (nx + 1) / 2 + 64;
    int rc;
    if ((rc = hn_launch_f32_to_bf16(x, xh, nx, s)) || (rc = hn_launch_f32_to_bf16(dz, zh, nz, s))) return rc;
    if ((rc = hn_launch_conv_wgrad_bf16(xh, zh, scratch, B, Hi, Wi, Cin, Cout, KH, KW, sh, sw, s))) return rc;
    return hn_launch_unpack_conv(scratch, dw_oihw, Cout, Cin, KH, KW, 0, s);
}

extern "C" int hn_conv2d_wgrad_nhwc(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int Hi, int Wi, int Cin,
                                    int Cout, int KH, int KW, int sh, int sw, int stem, void* stream)
{
    HN_REQUIRE(x && dz && dw_oihw && scratch, "hn_conv2d_wgrad_nhwc: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = hn_launch_conv_wgrad(x, dz, scratch, B, Hi, Wi, Cin, Cout, KH, KW, sh, sw, 0, 0, stem, s))) return rc;
    return hn_launch_unpack_conv(scratch, dw_oihw, Cout, Cin, KH, KW, stem ? 8 : 0, s);
}
