// bf16 inference mode of the engine: same forward plan as hn_forward (engine.hip) with NHWC bf16 activations and
// bf16 MFMA convolutions (conv_igemm_bf16.hip); BatchNorm stays folded in f32 scale/shift, the bi-LSTM recurrence,
// its gate pre-activations and the Linear head stay f32.  Serves the bf16 configurations of BASELINE.json (the
// reference runs net(x) under autocast, train.py:51,273); parity bar there is 3D-IoU / peak agreement, not 1e-3.
#include "engine_internal.h"

#include <string.h>

namespace {

typedef unsigned short u16;

struct HOff {
    std::vector<size_t> conv;      // element offsets of the packed bf16 conv weights, arch().convs order
    std::vector<size_t> dgrad;     // ... of the per-class data-gradient packings [Cin][taps of the class][Cout] ((size_t)-1: none)
    size_t wih[2];
    size_t whh[2][2];              // recurrent weights [layer][direction], bf16 [2048][512] (lstm_bf16.hip)
    size_t total;                  // elements
};

const HOff& hoff()
{
    static HOff H = [] {
        HOff h;
        size_t off = 0;
        auto take = [&](size_t n) { size_t o = off; off += (n + 127) / 128 * 128; return o; };
        for (const auto& c : arch().convs) h.conv.push_back(take(c.k == 7 ? (size_t)c.cout * 7 * 8 * 8 : (size_t)c.cout * c.k * c.k * c.cin));
        for (int l = 0; l < 2; ++l) h.wih[l] = take((size_t)4096 * 1024);
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < 2; ++d) h.whh[l][d] = take((size_t)2048 * 512);
        // data-gradient packings (training): the classes of one conv partition its taps, so together they are one weight tensor
        for (const auto& c : arch().convs)
            h.dgrad.push_back(c.k != 7 && c.cout % 64 == 0 && c.cin % 32 == 0 ? take((size_t)c.cout * c.k * c.k * c.cin) : (size_t)-1);
        h.total = off;
        return h;
    }();
    return H;
}

}  // namespace

size_t hn_bf16_conv_offset(int ci) { return hoff().conv[ci]; }
size_t hn_bf16_dgrad_offset(int ci) { return hoff().dgrad[ci]; }
size_t hn_bf16_wih_offset(int l) { return hoff().wih[l]; }
size_t hn_bf16_whh_offset(int l, int d) { return hoff().whh[l][d]; }

namespace {

constexpr size_t SPLITK_WS_FLOATS_PER_PANO = 2 * 256 * 1024;   // 2 slices of ghc3.0's 256 pixels x 1024 channels per panorama (= 4 of ghc3.1, 2 of layer4.*.conv2)
inline size_t splitk_floats(int B) { return SPLITK_WS_FLOATS_PER_PANO * (size_t)B; }     // floats of one split-K scratch

struct PlanH {                     // byte offsets
    size_t sync, xn, stem, pool, p0, p1, t1, t2, ds, seq, gx, y1, y1h, y2, xch, sk, sk2, total;
    size_t c[4];                   // C1..C4 (the layer outputs): own buffers, read by the next stage AND by the branch stream
    size_t ga[4], gb[4];           // per-scale ping-pong of the height-compression chain (the four chains may overlap)
};

PlanH make_plan_h(int B)
{
    PlanH p;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t b = (size_t)B;
    p.sync = take(HN_SYNC_WORDS * 4);
    p.xn = take(b * IMG_H * IMG_W * 8 * 2);
    p.stem = take(b * 256 * 512 * 64 * 2);
    p.pool = take(b * 128 * 256 * 64 * 2);
    p.p0 = take(b * 128 * 256 * 256 * 2);
    p.p1 = take(b * 128 * 256 * 256 * 2);
    p.t1 = take(b * 128 * 256 * 128 * 2);
    p.t2 = take(b * 128 * 256 * 64 * 2);
    p.ds = take(b * 128 * 256 * 256 * 2);
    for (int li = 0; li < 4; ++li) {
        const size_t hw = (size_t)(128 >> li) * (256 >> li), ch = (size_t)256 << li;
        p.c[li] = take(b * hw * ch * 2);
        p.ga[li] = take(b * (hw / 2) * (ch / 2) * 2);          // ghc.0 output; ghc.2 output is smaller
        p.gb[li] = take(b * (hw / 4) * (ch / 2) * 2);          // ghc.1 output; ghc.3 output is smaller
    }
    p.seq = take((size_t)T_COLS * b * 1024 * 2);
    p.gx = take((size_t)T_COLS * b * 4096 * 4);
    p.y1 = take((size_t)T_COLS * b * 1024 * 4);
    p.y1h = take((size_t)T_COLS * b * 1024 * 2);
    p.y2 = take((size_t)T_COLS * b * 1024 * 4);
    p.xch = take(hn_lstm_bf16_xch_bytes());
    p.sk = take(splitk_floats(B) * 4);      // split-K partial tiles of the branch stream (one user at a time)
    p.sk2 = take(splitk_floats(B) * 4);     // ... of the caller's stream (layer4's 3x3 convs run beside the height-compression chains)
    p.total = off;
    return p;
}

int run_conv_h(hn_engine* e, const ConvLayer& c, size_t woff, const void* x, void* y, const void* res, int B, int Hi, int Wi, int sh,
               int sw, int relu, hipStream_t s, float* splitk_ws = nullptr, size_t splitk_ws_floats = 0)
{
    const float* P = e->packed;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = reinterpret_cast<const float*>(x);
    d.w = reinterpret_cast<const float*>(reinterpret_cast<const u16*>(e->packed_h) + woff);
    d.scale = P + c.scale_off; d.shift = P + c.shift_off; d.res = reinterpret_cast<const float*>(res); d.y = reinterpret_cast<float*>(y);
    d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = c.cin; d.Cout = c.cout; d.KH = c.k; d.KW = c.k; d.sh = sh; d.sw = sw;
    d.ph = c.k / 2; d.pw = c.k / 2;
    d.Ho = (Hi + 2 * d.ph - c.k) / sh + 1;
    d.Wo = (Wi + 2 * d.pw - c.k) / sw + 1;
    d.relu = relu; d.ldy = c.cout;
    d.splitk_ws = splitk_ws; d.splitk_ws_floats = splitk_ws_floats;
    ProfScope ps(e, s, c.wkey, 2.0 * B * d.Ho * d.Wo * (double)c.cout * c.cin * c.k * c.k);
    return hn_launch_conv_bf16(d, 0, s);
}

}  // namespace

extern "C" size_t hn_packed_bf16_bytes(void) { return hoff().total * sizeof(u16); }

extern "C" size_t hn_workspace_bf16_bytes(int B)
{
    if (B < 1) return 0;
    return make_plan_h(B).total;
}

extern "C" int hn_pack_weights_bf16(hn_engine* e, void* packed_h, size_t bytes, void* stream)
{
    HN_REQUIRE(e && packed_h, "hn_pack_weights_bf16: null argument");
    HN_REQUIRE(e->packed != nullptr, "hn_pack_weights_bf16: call hn_pack_weights first (f32 scale/shift, LSTM and head weights come from it)");
    HN_REQUIRE(bytes >= hn_packed_bf16_bytes(), "hn_pack_weights_bf16: buffer too small (%zu < %zu)", bytes, hn_packed_bf16_bytes());
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_pack_weights_bf16: cannot select device %d", e->device);
    hipStream_t s = (hipStream_t)stream;
    const Arch& a = arch();
    const HOff& h = hoff();
    u16* H = reinterpret_cast<u16*>(packed_h);
    // one table-driven launch (multi_job.h): the forward packings, the LSTM / input-GEMM weights, and the per-class
    // data-gradient packings the bf16 training step's backward pass reads (they used to be re-packed inside every backward
    // pass, ~90 launches; as weights they only change with the optimiser step that also triggers this call)
    if (int rc = hn_poison(e, packed_h, hn_packed_bf16_bytes(), s)) return rc;
    std::vector<MJob> jobs;
    jobs.reserve(3 * a.convs.size() + 8);
    for (size_t i = 0; i < a.convs.size(); ++i) {
        const ConvLayer& c = a.convs[i];
        auto it = e->bound.find(c.wkey + ".weight");
        HN_REQUIRE(it != e->bound.end(), "hn_pack_weights_bf16: '%s.weight' was never bound", c.wkey.c_str());
        const float* w = reinterpret_cast<const float*>(it->second);
        jobs.push_back(mj_pack_bf16(w, H + h.conv[i], c.cout, c.cin, c.k, c.k));
        if (h.dgrad[i] == (size_t)-1) continue;
        int sh = 1, sw = 1;
        hn_conv_strides((int)i, &sh, &sw);
        u16* wp = H + h.dgrad[i];
        const int ph = c.k / 2, pw = c.k / 2;
        if (sh == 1 && sw == 1) {       // stride 1: the data gradient runs as a forward conv with flipped taps (hn_launch_conv_dgrad_bf16)
            MJob j = mj_make(MJ_PACK_DGRAD_FWD_BF16, w, wp, (long long)c.cin * c.k * c.k * c.cout);
            j.p[0] = c.cout; j.p[1] = c.cin; j.p[2] = c.k; j.p[3] = c.k;
            jobs.push_back(j);
            continue;
        }
        for (int ca = 0; ca < sh; ++ca) {
            for (int cb = 0; cb < sw; ++cb) {        // same class order and tap lists as hn_launch_conv_dgrad_bf16
                int tdh[3] = {0, 0, 0}, tdw[3] = {0, 0, 0}, ntdh = 0, ntdw = 0;
                for (int t = 0; t < c.k; ++t) if ((ca + ph - t) % sh == 0) tdh[ntdh++] = t;
                for (int t = 0; t < c.k; ++t) if ((cb + pw - t) % sw == 0) tdw[ntdw++] = t;
                const long long total = (long long)c.cin * ntdh * ntdw * c.cout;
                if (total == 0) continue;
                MJob j = mj_make(MJ_PACK_DGRAD_BF16, w, wp, total);
                j.p[0] = c.cout; j.p[1] = c.cin; j.p[2] = c.k; j.p[3] = c.k;
                j.p[4] = tdh[0]; j.p[5] = tdh[1]; j.p[6] = tdh[2]; j.p[7] = ntdh;
                j.p[8] = tdw[0]; j.p[9] = tdw[1]; j.p[10] = tdw[2]; j.p[11] = ntdw;
                jobs.push_back(j);
                wp += total;
            }
        }
    }
    for (int l = 0; l < 2; ++l) {
        jobs.push_back(mj_make(MJ_F32_TO_BF16, e->packed + a.wih_off[l], H + h.wih[l], (long long)4096 * 1024));
        for (int d = 0; d < 2; ++d) jobs.push_back(mj_make(MJ_F32_TO_BF16, e->packed + a.whh_off[l][d], H + h.whh[l][d], (long long)2048 * 512));
    }
    if (int rc = e->jt_pack_h.run(jobs, s)) return rc;
    e->packed_h = packed_h;
    return 0;
}

namespace {

// Buffers of the recurrent head (bi-LSTM + Linear) of one in-flight batch: the plain forward has one set inside PlanH, the
// pipelined entry two (the head of batch i runs beside the trunk of batch i+1).
struct HeadBufs {
    char* sync;      // HN_SYNC_WORDS uint32: arrival counters + sticky status word
    char* seq;       // [T*B][1024] bf16 column features (written by the trunk's up-sample / flatten launches)
    char* gx;        // [T*B][4096] float32 gate pre-activations
    char* y1;        // [T*B][1024] float32 layer-0 output
    char* y1h;       // bf16 copy of y1 (layer 1's GEMM operand)
    char* y2;        // [T*B][1024] float32 layer-1 output
    char* xch;       // hand-off granules of the narrow recurrence kernel (plain forward)
    char* y2h;       // bf16 copy of y2: the wide recurrence kernel's exchange buffer for layer 1 (pipelined entry only)
};

// The convolutional trunk (model.py:73-81,123-179): stem, the four ResNet stages, the four height-compression chains, the
// up-sample / flatten into `seq`.  Everything is enqueued on `s` (+ the engine's branch stream, joined back into `s`).
// defer_join (pipelined entry): the caller's stream does NOT wait for the four chains at the end -- the recurrent head does -- so the
// next batch's stem / layer1 (HBM-bound) start beside this batch's last chain (ghc3: deep-K matrix work + a tail of small launches).
// *forked tells the caller whether chains went to the branch stream.
int run_trunk_h(hn_engine* e, const float* x, int B, int C_in, char* W, const PlanH& pl, char* seq, hipStream_t s, bool defer_join = false,
                bool* forked = nullptr)
{
    const Arch& a = arch();
    const HOff& h = hoff();
    const float* P = e->packed;
    const u16* H = reinterpret_cast<const u16*>(e->packed_h);
    int rc;

    bool stem_c1 = false;            // layer1.0.conv1 already computed by the fused stem kernel
    {   // stem: normalise -> NHWC4 bf16, 7x7/2 conv + BN + ReLU, max-pool
        const ConvLayer& c = a.convs[a.stem];
        ProfScope ps(e, s, "stem(prep+conv7x7+maxpool)", 2.0 * B * 256 * 512 * 64.0 * 147);
        const bool fused = e->fuse_stem_pool && e->taps.empty();
        if (!fused && (rc = hn_launch_prep_nhwc4_bf16(x, W + pl.xn, B, C_in, IMG_H, IMG_W, s))) return rc;
        ConvDesc d;
        memset(&d, 0, sizeof(d));
        d.x = reinterpret_cast<const float*>(W + pl.xn); d.w = reinterpret_cast<const float*>(H + h.conv[a.stem]);
        d.scale = P + c.scale_off; d.shift = P + c.shift_off; d.y = reinterpret_cast<float*>(W + pl.stem);
        d.B = B; d.Hi = IMG_H; d.Wi = IMG_W; d.Cin = 8; d.Cout = 64; d.KH = 7; d.KW = 7; d.sh = 2; d.sw = 2; d.ph = 3; d.pw = 3;
        d.Ho = 256; d.Wo = 512; d.relu = 1; d.ldy = 64; d.stem = 1;
        if (fused) {
            // normalise + conv + BN + ReLU + max-pool in one kernel (stem_pool_bf16.hip): neither the NHWC4 copy of the input nor
            // the 537 MB stem activation exists
            // ... and layer1.0.conv1 (1x1, 64 -> 64) on every pooled half row while it is still in LDS ("fuse_stem_conv1")
            stem_c1 = e->fuse_stem_conv1 && !e->profiling;
            const int ic1 = a.block_first[0][0];
            const ConvLayer& c1 = a.convs[ic1];
            if ((rc = hn_launch_stem_pool_bf16(x, C_in, H + h.conv[a.stem], d.scale, d.shift, W + pl.pool, B, s,
                                               stem_c1 ? H + h.conv[ic1] : nullptr, P + c1.scale_off, P + c1.shift_off, W + pl.t1)))
                return rc;
        } else {
            if ((rc = hn_launch_conv_bf16(d, 0, s))) return rc;
            if ((rc = hn_launch_maxpool_bf16(W + pl.stem, W + pl.pool, B, 256, 512, 64, s))) return rc;
        }
    }
    if ((rc = e->tap("stem", W + pl.stem, (size_t)B * 256 * 512 * 64 * 2, s))) return rc;
    if ((rc = e->tap("pool", W + pl.pool, (size_t)B * 128 * 256 * 64 * 2, s))) return rc;

    // The height-compression chain of scale li depends only on C_li: it is forked onto the engine's branch stream and
    // runs beside the ResNet stages that follow (small / half-occupancy launches of the two streams fill each other's
    // idle CUs); the caller's stream joins all four before the LSTM input GEMM.  Profiling / option off: one stream.
    const bool fork = e->use_branch_stream && !e->profiling && e->branch_stream != nullptr;
    hipStream_t sb = fork ? e->branch_stream : s;
    const char* cur = W + pl.pool;
    int Hh = 128, Wd = 256;
    const int nblk[4] = {3, 4, 6, 3};
    if (forked) *forked = fork;
    for (int li = 0; li < 4; ++li) {
        // C_li is about to be overwritten: a chain of an EARLIER batch that was not joined (deferred) must have finished reading it
        if (e->join_deferred[li]) {
            HN_HIP(hipStreamWaitEvent(s, e->ev_join[li], 0));
            e->join_deferred[li] = false;
        }
        for (int j = 0; j < nblk[li]; ++j) {
            const int stride = (j == 0 && li > 0) ? 2 : 1;
            const int i1 = a.block_first[li][j];
            char* out = (j == nblk[li] - 1) ? W + pl.c[li] : ((cur == W + pl.p0) ? W + pl.p1 : W + pl.p0);
            // layer1, block j -> j+1: conv3 (+ residual + ReLU) and the NEXT block's conv1 as ONE launch (conv1x1_chain_bf16_kernel):
            // the 537 MB block output is not read back from HBM by the next conv1.  Bit-identical to the two-launch form.
            // Block 0 -> 1 likewise, with the downsample branch as a second accumulator set of the same launch (DUAL form).
            // Block 2 -> layer2.0.conv1 (128 output channels: two 64-column passes over the same LDS tile); the block output C1 is
            // still written (the downsample branch, the height compression and the next residual read it).
            const bool chain_on = e->chain_layer1 && !e->profiling;
            const bool chain_out = chain_on && li == 0 && (j >= 1 || e->fuse_downsample);
            const bool chained_in = chain_on && ((li == 0 && (j == 2 || (j == 1 && e->fuse_downsample))) || (li == 1 && j == 0));
            const bool c1_done = chained_in || (stem_c1 && li == 0 && j == 0);
            if (!c1_done && (rc = run_conv_h(e, a.convs[i1], h.conv[i1], cur, W + pl.t1, nullptr, B, Hh, Wd, 1, 1, 1, s))) return rc;
            if ((rc = run_conv_h(e, a.convs[i1 + 1], h.conv[i1 + 1], W + pl.t1, W + pl.t2, nullptr, B, Hh, Wd, stride, stride, 1, s,
                                 reinterpret_cast<float*>(W + pl.sk2), splitk_floats(B))))
                return rc;
            if (chain_out) {
                const ConvLayer& c3 = a.convs[i1 + 2];
                const int in1 = j + 1 < nblk[li] ? a.block_first[li][j + 1] : a.block_first[li + 1][0];
                const ConvLayer& c1n = a.convs[in1];
                const int id = a.block_down[li];
                const ConvLayer& cd = a.convs[id];
                if ((rc = hn_launch_conv1x1_chain_bf16(W + pl.t2, H + h.conv[i1 + 2], P + c3.scale_off, P + c3.shift_off, cur, out, H + h.conv[in1],
                                                       P + c1n.scale_off, P + c1n.shift_off, W + pl.t1, (long)B * Hh * Wd, c3.cin, c3.cout, c1n.cout, s,
                                                       j == 0 ? cur : nullptr, j == 0 ? H + h.conv[id] : nullptr, P + cd.scale_off, P + cd.shift_off)))
                    return rc;
                cur = out;
                continue;
            }
            const char* idt = cur;
            // block 0 of the HBM-bound stages (layer1, layer2): downsample + conv3 + add + ReLU in ONE launch -- the
            // downsample output never goes to HBM (conv1x1_dual_bf16_kernel; bit-identical to the two-launch form)
            const bool dual = j == 0 && li < 2 && e->fuse_downsample;
            if (j == 0 && !dual) {
                const int id = a.block_down[li];
                if ((rc = run_conv_h(e, a.convs[id], h.conv[id], cur, W + pl.ds, nullptr, B, Hh, Wd, stride, stride, 0, s))) return rc;
                idt = W + pl.ds;
            }
            const int Hin = Hh, Win = Wd;
            Hh /= stride;
            Wd /= stride;
            if (dual) {
                const int id = a.block_down[li];
                const ConvLayer& c3 = a.convs[i1 + 2];
                const ConvLayer& cd = a.convs[id];
                ProfScope ps(e, s, c3.wkey + "+downsample", 2.0 * B * Hh * Wd * (double)c3.cout * (c3.cin + cd.cin));
                if ((rc = hn_launch_conv1x1_dual_bf16(W + pl.t2, H + h.conv[i1 + 2], P + c3.scale_off, P + c3.shift_off, cur, H + h.conv[id],
                                                      P + cd.scale_off, P + cd.shift_off, out, B, Hh, Wd, c3.cin, Hin, Win, cd.cin, stride,
                                                      c3.cout, s)))
                    return rc;
            } else if ((rc = run_conv_h(e, a.convs[i1 + 2], h.conv[i1 + 2], W + pl.t2, out, idt, B, Hh, Wd, 1, 1, 1, s)))
                return rc;
            cur = out;
        }
        {
            const char* cname[4] = {"c1", "c2", "c3", "c4"};
            if ((rc = e->tap(cname[li], cur, (size_t)B * Hh * Wd * (256 << li) * 2, s))) return rc;
        }
        if (fork) {
            HN_HIP(hipEventRecord(e->ev_fork[li], s));
            HN_HIP(hipStreamWaitEvent(sb, e->ev_fork[li], 0));
        }
        const char* gin = cur;
        int gh = Hh;
        char* gbuf[2] = {W + pl.ga[li], W + pl.gb[li]};
        for (int k = 0; k < 4; ++k) {
            const int ig = a.ghc_first[li] + k;
            char* gout = gbuf[k & 1];
            if ((rc = run_conv_h(e, a.convs[ig], h.conv[ig], gin, gout, nullptr, B, gh, Wd, 2, 1, 1, sb, reinterpret_cast<float*>(W + pl.sk),
                                 splitk_floats(B))))
                return rc;
            gin = gout;
            gh /= 2;
        }
        {
            ProfScope ps(e, sb, "upsample_flatten." + std::to_string(li), 0.0);
            if ((rc = hn_launch_upsample_flatten_bf16(gin, seq, B, gh, Wd, a.convs[a.ghc_first[li] + 3].cout, 256 * li, sb))) return rc;
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

// The recurrent head (model.py:263-269): bi-LSTM x2 (bf16 input GEMMs with f32 gate pre-activations, then the recurrence)
// and the f32 Linear head, all on `s`.  wide: the few-compute-unit recurrence kernel of the pipelined entry.
int run_head_h(hn_engine* e, int B, const HeadBufs& hb, float* bon, float* cor, bool wide, hipStream_t s)
{
    const Arch& a = arch();
    const HOff& h = hoff();
    const float* P = e->packed;
    const u16* H = reinterpret_cast<const u16*>(e->packed_h);
    int rc;
    HN_HIP(hipMemsetAsync(hb.sync, 0, HN_STATUS_WORD * sizeof(unsigned), s));   // arrival counters; the status word behind them is sticky (zeroed by the caller at allocation)
    const long rows = (long)T_COLS * B;
    if ((rc = e->tap("feature", hb.seq, (size_t)rows * 1024 * 2, s))) return rc;
    const void* lin = hb.seq;
    float* ybuf[2] = {reinterpret_cast<float*>(hb.y1), reinterpret_cast<float*>(hb.y2)};
    for (int l = 0; l < 2; ++l) {
        {
            ProfScope ps(e, s, "bi_rnn.l" + std::to_string(l) + ".input_gemm", 2.0 * rows * 1024.0 * 4096);
            ConvDesc d;
            memset(&d, 0, sizeof(d));
            d.x = reinterpret_cast<const float*>(lin); d.w = reinterpret_cast<const float*>(H + h.wih[l]);
            d.scale = P + a.ones_off; d.shift = P + a.lbias_off[l]; d.y = reinterpret_cast<float*>(hb.gx);
            d.B = 1; d.Hi = 1; d.Wi = (int)rows; d.Cin = 1024; d.Cout = 4096; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1;
            d.Ho = 1; d.Wo = (int)rows; d.relu = 0; d.ldy = 4096;
            if ((rc = hn_launch_conv_bf16(d, 1, s))) return rc;
        }
        {
            ProfScope ps(e, s, "bi_rnn.l" + std::to_string(l) + ".recurrence", 2.0 * rows * 512.0 * 2048 * 2);
            if (wide) {              // 16 (8) panoramas per group: 32 (64) compute units per batch of 32 (lstm_bf16.hip, wide form)
                if ((rc = hn_launch_lstm_layer_bf16_wide(reinterpret_cast<float*>(hb.gx), H + h.whh[l][0], H + h.whh[l][1], ybuf[l],
                                                         l == 0 ? hb.y1h : hb.y2h, T_COLS, B, hb.sync, e->wide_rows, e->wide_xcds, s)))
                    return rc;
            } else if (e->bf16_lstm) {      // recurrent matmul on the bf16 matrix cores, 8-workgroup groups, granule hand-off (lstm_bf16.hip)
                if ((rc = hn_launch_lstm_layer_bf16(reinterpret_cast<float*>(hb.gx), H + h.whh[l][0], H + h.whh[l][1], ybuf[l],
                                                    l == 0 ? hb.y1h : nullptr, T_COLS, B, hb.xch, hb.sync, s)))
                    return rc;
            } else if ((rc = hn_launch_lstm_layer(reinterpret_cast<float*>(hb.gx), P + a.whh_off[l][0], P + a.whh_off[l][1], ybuf[l],
                                                  T_COLS, B, hb.sync, s)))
                return rc;
        }
        if (l == 0) {
            if (!wide && !e->bf16_lstm && (rc = hn_launch_f32_to_bf16(ybuf[0], hb.y1h, rows * 1024, s))) return rc;
            lin = hb.y1h;
        }
    }
    if ((rc = e->tap("lstm", ybuf[1], (size_t)rows * 1024 * sizeof(float), s))) return rc;
    ProfScope ps(e, s, "linear", 2.0 * rows * 1024.0 * 12);
    return hn_launch_linear_head(ybuf[1], P + a.linw_off, P + a.linb_off, bon, cor, T_COLS, B, s);
}

// Second head buffer set + the wide kernel's exchange areas, laid out behind the plain plan.
struct PlanP {
    PlanH base;
    size_t sync1, seq1, gx1, y11, y1h1, y21, y2h[2], total;
};

PlanP make_plan_p(int B)
{
    PlanP p;
    p.base = make_plan_h(B);
    size_t off = p.base.total;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t b = (size_t)B;
    p.sync1 = take(HN_SYNC_WORDS * 4);
    p.seq1 = take((size_t)T_COLS * b * 1024 * 2);
    p.gx1 = take((size_t)T_COLS * b * 4096 * 4);
    p.y11 = take((size_t)T_COLS * b * 1024 * 4);
    p.y1h1 = take((size_t)T_COLS * b * 1024 * 2);
    p.y21 = take((size_t)T_COLS * b * 1024 * 4);
    for (int k = 0; k < 2; ++k) p.y2h[k] = take((size_t)T_COLS * b * 1024 * 2);
    p.total = off;
    return p;
}

}  // namespace

extern "C" int hn_forward_bf16(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                               size_t workspace_bytes, void* stream)
{
    HN_REQUIRE(e && x && bon && cor && workspace, "hn_forward_bf16: null argument");
    HN_REQUIRE(e->packed != nullptr && e->packed_h != nullptr, "hn_forward_bf16: hn_pack_weights / hn_pack_weights_bf16 have not been called");
    HN_REQUIRE(B >= 1 && C_in >= 3, "hn_forward_bf16: bad B=%d / C_in=%d", B, C_in);
    const PlanH pl = make_plan_h(B);
    HN_REQUIRE(workspace_bytes >= pl.total, "hn_forward_bf16: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward_bf16: cannot select device %d", e->device);
    hipStream_t s = (hipStream_t)stream;
    char* W = reinterpret_cast<char*>(workspace);
    e->prof.clear();
    e->events_used = 0;
    // (everything behind the sync page is dead at entry -- unless a chain of an earlier PIPELINED batch is still un-joined on the branch stream)
    if (!(e->join_deferred[0] || e->join_deferred[1] || e->join_deferred[2] || e->join_deferred[3])) {
        if (int rc = hn_poison(e, W + pl.xn, pl.total - pl.xn, s)) return rc;
        if (int rc = hn_poison(e, bon, (size_t)B * 2 * 1024 * sizeof(float), s)) return rc;
        if (int rc = hn_poison(e, cor, (size_t)B * 1024 * sizeof(float), s)) return rc;
    }
    if (int rc = run_trunk_h(e, x, B, C_in, W, pl, W + pl.seq, s)) return rc;
    const HeadBufs hb = {W + pl.sync, W + pl.seq, W + pl.gx, W + pl.y1, W + pl.y1h, W + pl.y2, W + pl.xch, nullptr};
    return run_head_h(e, B, hb, bon, cor, false, s);
}

// ---- pipelined entry: trunk of batch i+1 beside the recurrent head of batch i ----------------------------------------------
extern "C" size_t hn_workspace_bf16_pipelined_bytes(int B)
{
    if (B < 1) return 0;
    return make_plan_p(B).total;
}

extern "C" int hn_forward_bf16_submit(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                                      size_t workspace_bytes, int slot, void* stream)
{
    HN_REQUIRE(e && x && bon && cor && workspace, "hn_forward_bf16_submit: null argument");
    HN_REQUIRE(e->packed != nullptr && e->packed_h != nullptr, "hn_forward_bf16_submit: hn_pack_weights / hn_pack_weights_bf16 have not been called");
    HN_REQUIRE(B >= 1 && C_in >= 3, "hn_forward_bf16_submit: bad B=%d / C_in=%d", B, C_in);
    HN_REQUIRE(slot == 0 || slot == 1, "hn_forward_bf16_submit: slot must be 0 or 1 (got %d)", slot);
    HN_REQUIRE(!e->profiling && e->taps.empty(), "hn_forward_bf16_submit: profiling / taps are served by hn_forward_bf16 only");
    const PlanP pp = make_plan_p(B);
    HN_REQUIRE(workspace_bytes >= pp.total, "hn_forward_bf16_submit: workspace too small (%zu < %zu; hn_workspace_bf16_pipelined_bytes)",
               workspace_bytes, pp.total);
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward_bf16_submit: cannot select device %d", e->device);
    if (int rc = hn_ensure_head_stream(e)) return rc;
    hipStream_t s = (hipStream_t)stream;
    char* W = reinterpret_cast<char*>(workspace);
    const PlanH& pl = pp.base;
    HeadBufs hb;
    if (slot == 0) hb = {W + pl.sync, W + pl.seq, W + pl.gx, W + pl.y1, W + pl.y1h, W + pl.y2, nullptr, W + pp.y2h[0]};
    else hb = {W + pp.sync1, W + pp.seq1, W + pp.gx1, W + pp.y11, W + pp.y1h1, W + pp.y21, nullptr, W + pp.y2h[1]};
    // this slot's `seq` was last read by the head submitted two calls ago: the trunk's writes into it wait for that head
    if (e->head_pending[slot]) HN_HIP(hipStreamWaitEvent(s, e->ev_head[slot], 0));
    bool forked = false;
    if (int rc = run_trunk_h(e, x, B, C_in, W, pl, hb.seq, s, e->defer_join, &forked)) return rc;
    HN_HIP(hipEventRecord(e->ev_trunk[slot], s));
    HN_HIP(hipStreamWaitEvent(e->head_stream, e->ev_trunk[slot], 0));
    if (forked && e->defer_join)
        for (int li = 0; li < 4; ++li) HN_HIP(hipStreamWaitEvent(e->head_stream, e->ev_join[li], 0));
    if (int rc = run_head_h(e, B, hb, bon, cor, true, e->head_stream)) return rc;
    HN_HIP(hipEventRecord(e->ev_head[slot], e->head_stream));
    e->head_pending[slot] = true;
    return 0;
}

extern "C" int hn_forward_bf16_collect(hn_engine* e, int slot, void* stream)
{
    HN_REQUIRE(e != nullptr && (slot == 0 || slot == 1), "hn_forward_bf16_collect: bad argument");
    HN_REQUIRE(e->head_pending[slot], "hn_forward_bf16_collect: nothing was submitted on slot %d", slot);
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_forward_bf16_collect: cannot select device %d", e->device);
    HN_HIP(hipStreamWaitEvent((hipStream_t)stream, e->ev_head[slot], 0));
    return 0;
}

extern "C" int hn_pipelined_status_offset(int B, int slot, size_t* byte_offset)
{
    HN_REQUIRE(B >= 1 && (slot == 0 || slot == 1) && byte_offset, "hn_pipelined_status_offset: bad argument");
    const PlanP pp = make_plan_p(B);
    *byte_offset = (slot == 0 ? pp.base.sync : pp.sync1) + HN_STATUS_WORD * sizeof(unsigned);
    return 0;
}

extern "C" size_t hn_lstm_bf16_exchange_bytes(void) { return hn_lstm_bf16_xch_bytes(); }

// per-stage entry (tests): one bf16-recurrence bi-LSTM layer, see lstm_bf16.hip
extern "C" int hn_lstm_layer_bf16(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, void* y_bf16, int T, int B,
                                  void* exchange, void* sync_ws, void* stream)
{
    HN_REQUIRE(gx && whh_fwd_bf16 && whh_rev_bf16 && y && exchange && sync_ws, "hn_lstm_layer_bf16: null pointer");
    return hn_launch_lstm_layer_bf16(gx, whh_fwd_bf16, whh_rev_bf16, y, y_bf16, T, B, exchange, sync_ws, (hipStream_t)stream);
}

extern "C" int hn_lstm_layer_bf16_wide(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, void* y_bf16, int T,
                                       int B, void* sync_ws, int rows_per_group, int xcds_per_group, void* stream)
{
    HN_REQUIRE(gx && whh_fwd_bf16 && whh_rev_bf16 && y && y_bf16 && sync_ws, "hn_lstm_layer_bf16_wide: null pointer");
    return hn_launch_lstm_layer_bf16_wide(gx, whh_fwd_bf16, whh_rev_bf16, y, y_bf16, T, B, sync_ws, rows_per_group, xcds_per_group,
                                          (hipStream_t)stream);
}

// per-stage entries (tests): the training step's forms of that layer -- forward also storing (i, f, g, o, c), and the adjoint
extern "C" int hn_lstm_layer_bf16_train(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, float* saved, int T,
                                        int B, void* exchange, void* sync_ws, void* stream)
{
    HN_REQUIRE(gx && whh_fwd_bf16 && whh_rev_bf16 && y && saved && exchange && sync_ws, "hn_lstm_layer_bf16_train: null pointer");
    return hn_launch_lstm_layer_bf16(gx, whh_fwd_bf16, whh_rev_bf16, y, nullptr, T, B, exchange, sync_ws, (hipStream_t)stream, saved);
}

extern "C" size_t hn_lstm_bwd_bf16_exchange_bytes(void) { return hn_lstm_bwd_bf16_xch_bytes(); }

extern "C" int hn_lstm_layer_bwd_bf16(const float* saved, const float* dy, const void* whhT_fwd_bf16, const void* whhT_rev_bf16, float* dgx,
                                      int T, int B, void* exchange, void* sync_ws, void* stream)
{
    HN_REQUIRE(saved && dy && whhT_fwd_bf16 && whhT_rev_bf16 && dgx && exchange && sync_ws, "hn_lstm_layer_bwd_bf16: null pointer");
    return hn_launch_lstm_layer_bwd_bf16(saved, dy, whhT_fwd_bf16, whhT_rev_bf16, dgx, T, B, exchange, sync_ws, (hipStream_t)stream);
}

// per-stage entry point for the parity tests: the bf16 stem (normalise, 7x7/2 conv + BN + ReLU, 3x3/2 max-pool) of a
// 512 x 1024 batch, fused (stem_pool_bf16.hip) or as the implicit GEMM + pool kernels
extern "C" int hn_stem_pool_bf16(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift, void* x4_scratch,
                                 void* w_scratch, void* stem_scratch, void* y, int B, int fused, void* stream)
{
    HN_REQUIRE(x_nchw && w_oihw && scale && shift && x4_scratch && w_scratch && y && B >= 1, "hn_stem_pool_bf16: null pointer / empty batch");
    HN_REQUIRE(fused || stem_scratch, "hn_stem_pool_bf16: the two-kernel form needs the [B][256][512][64] bf16 scratch");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = hn_launch_pack_conv_bf16(w_oihw, w_scratch, 64, 3, 7, 7, s))) return rc;
    if (fused) return hn_launch_stem_pool_bf16(x_nchw, 3, w_scratch, scale, shift, y, B, s);
    if ((rc = hn_launch_prep_nhwc4_bf16(x_nchw, x4_scratch, B, 3, IMG_H, IMG_W, s))) return rc;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = reinterpret_cast<const float*>(x4_scratch); d.w = reinterpret_cast<const float*>(w_scratch);
    d.scale = scale; d.shift = shift; d.y = reinterpret_cast<float*>(stem_scratch);
    d.B = B; d.Hi = IMG_H; d.Wi = IMG_W; d.Cin = 8; d.Cout = 64; d.KH = 7; d.KW = 7; d.sh = 2; d.sw = 2; d.ph = 3; d.pw = 3;
    d.Ho = 256; d.Wo = 512; d.relu = 1; d.ldy = 64; d.stem = 1;
    if ((rc = hn_launch_conv_bf16(d, 0, s))) return rc;
    return hn_launch_maxpool_bf16(stem_scratch, y, B, 256, 512, 64, s);
}

// per-stage entry point for the parity tests: x / w / res / y are bf16 device buffers (y f32 when out_f32)
extern "C" int hn_conv2d_nhwc_bf16_ws(const void* x, const float* w_oihw, void* w_scratch, const float* scale, const float* shift,
                                      const void* res, void* y, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int sh, int sw,
                                      int relu, int out_f32, void* splitk_ws, size_t splitk_ws_floats, void* stream)
{
    HN_REQUIRE(x && w_scratch && scale && shift && y, "hn_conv2d_nhwc_bf16: null pointer");
    HN_REQUIRE(KH == KW && (KH == 1 || KH == 3), "hn_conv2d_nhwc_bf16: kernel %dx%d unsupported", KH, KW);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (w_oihw && (rc = hn_launch_pack_conv_bf16(w_oihw, w_scratch, Cout, Cin, KH, KW, s))) return rc;   // NULL: w_scratch is already packed
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = reinterpret_cast<const float*>(x); d.w = reinterpret_cast<const float*>(w_scratch); d.scale = scale; d.shift = shift;
    d.res = reinterpret_cast<const float*>(res); d.y = reinterpret_cast<float*>(y);
    d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = Cin; d.Cout = Cout; d.KH = KH; d.KW = KW; d.sh = sh; d.sw = sw; d.ph = KH / 2; d.pw = KW / 2;
    d.Ho = (Hi + 2 * d.ph - KH) / sh + 1;
    d.Wo = (Wi + 2 * d.pw - KW) / sw + 1;
    d.relu = relu; d.ldy = Cout;
    d.splitk_ws = reinterpret_cast<float*>(splitk_ws); d.splitk_ws_floats = splitk_ws_floats;
    return hn_launch_conv_bf16(d, out_f32, s);
}

extern "C" int hn_conv2d_nhwc_bf16(const void* x, const float* w_oihw, void* w_scratch, const float* scale, const float* shift,
                                   const void* res, void* y, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int sh, int sw,
                                   int relu, int out_f32, void* stream)
{
    return hn_conv2d_nhwc_bf16_ws(x, w_oihw, w_scratch, scale, shift, res, y, B, Hi, Wi, Cin, Cout, KH, KW, sh, sw, relu, out_f32, nullptr, 0, stream);
}
