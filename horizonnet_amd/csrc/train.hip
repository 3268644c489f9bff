// Training step of HorizonNet(resnet50, use_rnn=True) on the HIP engine: train-mode forward
// (batch-statistics BatchNorm with running-stat update, dropout) and the full adjoint (all 241
// parameter gradients).  Replaces what autograd does for `net(x)` + `loss.backward()` at reference
// train.py:44-58,272-281; the losses and the optimiser stay in the caller.  float32 throughout.
//
// Every conv "unit" keeps its input x, pre-BN output z and post-activation output y resident in
// HBM (0.98 GB per panorama for z+y -- sized for the 288 GB of the MI355X), so the backward pass
// re-reads instead of recomputing.
#include "engine_internal.h"

#include <stdlib.h>

#include <string.h>

namespace {

struct Unit {
    int ci;                        // index into arch().convs
    int Hi, Wi, sh, sw, Ho, Wo;
    int relu;                      // ReLU directly after this unit's BN (conv3 / downsample: 0)
    int stem;
    size_t x, z, y, st, mk;        // float offsets: input, pre-BN, post-activation, [mean|invstd|a|b], ReLU bit mask (1 byte per 4 outputs)
    size_t xh, yh;                 // float offsets of the bf16 copies of the input / of y (train_precision bf16)
    int keep_y32;                  // bf16 mode: 1 = a float32 consumer of y exists (max-pool, up-sampling, an f32 weight-gradient fallback)
    size_t sf, sb;                 // this unit's slots in the statistics arena (doubles): forward [sum | sumsq], backward [S1 | S2 | dbias]
    size_t wg;                     // float offset of this unit's packed weight-gradient scratch (un-packed in one batch per segment)
    size_t fold;                   // float offset of the unit's BatchNorm-fold storage (bn_fold.hip: G | A | Wf | WG), 0 = the unit is never folded
    long M;
};

struct TrainPlan {
    std::vector<Unit> units;
    int blk[4][6], dsu[4], ghc[4];         // unit indices
    size_t pidx;                           // max-pool winners (one byte per pooled output), written by the forward pass
    size_t xn8, poolh, dzh[2];             // bf16: NHWC4 normalised image, max-pool output, two dz staging buffers
    size_t lxch;                           // bf16: granule exchange scratch of the persistent LSTM kernels
    size_t sync, xn, pool, seq, gx, y1, y1d, y2, y2d, save[2], dlin, G[4], DC[4], dhrec, dcrec, wsA, wsB, dstat, total;
    size_t stat_fwd_doubles, stat_bwd_first, stat_bwd_doubles, stat_lstm;   // layout of the statistics arena at `dstat` (doubles)
    size_t gmax;
    size_t wg_first, wg_floats;            // the units' weight-gradient scratches: one contiguous range (one memset per backward pass)
};

const int kBlocks[4] = {3, 4, 6, 3};

TrainPlan make_train_plan(int B)
{
    const Arch& a = arch();
    TrainPlan p;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t b = (size_t)B;
    p.sync = take(HN_SYNC_WORDS);
    p.xn = take(b * IMG_H * IMG_W * 4);
    p.xn8 = take(b * IMG_H * IMG_W * 4);          // 8 bf16 per pixel
    size_t stat_f = 0, stat_b = 0;
    auto add_unit = [&](int ci, size_t x, size_t xh, int Hi, int Wi, int sh, int sw, int relu, int stem) {
        const ConvLayer& c = a.convs[ci];
        Unit u;
        u.ci = ci; u.Hi = Hi; u.Wi = Wi; u.sh = sh; u.sw = sw; u.relu = relu; u.stem = stem; u.keep_y32 = stem;
        u.Ho = (Hi + 2 * (c.k / 2) - c.k) / sh + 1;
        u.Wo = (Wi + 2 * (c.k / 2) - c.k) / sw + 1;
        u.M = (long)B * u.Ho * u.Wo;
        u.x = x;
        u.xh = xh;
        u.z = take((size_t)u.M * c.cout);
        u.y = take((size_t)u.M * c.cout);
        u.st = take(4 * (size_t)c.cout);
        u.mk = take(((size_t)u.M * c.cout / 4 + 3) / 4);
        u.yh = take((size_t)u.M * c.cout / 2);
        u.sf = stat_f; stat_f += hn_stat_slot_doubles(c.cout, u.M);     // [sum | sq][C] (+ replicas), see ConvDesc::stat_rep
        u.sb = stat_b; stat_b += 3 * (size_t)c.cout;
        u.fold = 0;
        p.units.push_back(u);
        return (int)p.units.size() - 1;
    };
    const int us = add_unit(a.stem, p.xn, p.xn8, IMG_H, IMG_W, 2, 2, 1, 1);
    (void)us;
    p.pool = take(b * 128 * 256 * 64);
    p.poolh = take(b * 128 * 256 * 64 / 2);
    p.pidx = take(b * 128 * 256 * 64 / 4);
    size_t cur = p.pool, curh = p.poolh;
    int H = 128, W = 256;
    for (int li = 0; li < 4; ++li) {
        for (int j = 0; j < kBlocks[li]; ++j) {
            const int stride = (j == 0 && li > 0) ? 2 : 1;
            const int c1 = add_unit(a.block_first[li][j], cur, curh, H, W, 1, 1, 1, 0);
            const int c2 = add_unit(a.block_first[li][j] + 1, p.units[c1].y, p.units[c1].yh, H, W, stride, stride, 1, 0);
            if (j == 0) p.dsu[li] = add_unit(a.block_down[li], cur, curh, H, W, stride, stride, 0, 0);
            H /= stride;
            W /= stride;
            const int c3 = add_unit(a.block_first[li][j] + 2, p.units[c2].y, p.units[c2].yh, H, W, 1, 1, 0, 0);
            // conv3 of every block and a stride-1 downsample conv (layer1) may run through the BatchNorm-folded form (bn_fold.hip)
            for (int fu : {c3, (j == 0 && stride == 1) ? p.dsu[li] : -1})
                if (fu >= 0) p.units[fu].fold = take(hn_bn_fold_keep_floats(a.convs[p.units[fu].ci].cout, a.convs[p.units[fu].ci].cin));
            p.blk[li][j] = c1;                    // c2 = c1 + 1; (ds = c1 + 2 when j == 0); c3 = last
            cur = p.units[c3].y;
            curh = p.units[c3].yh;
        }
        size_t gin = cur, ginh = curh;
        int gh = H;
        for (int k = 0; k < 4; ++k) {
            const int u = add_unit(a.ghc_first[li] + k, gin, ginh, gh, W, 2, 1, 1, 0);
            if (k == 0) p.ghc[li] = u;
            // float32 y is read by the circular up-sampling (last unit of a scale) and by ghc0.3's float32 weight gradient
            // (Cout = 32 stays off the bf16 matrix-core path; its input is ghc0.2's y)
            if (k == 3 || a.convs[a.ghc_first[li] + k].cout % 64 != 0 || (k < 3 && a.convs[a.ghc_first[li] + k + 1].cout % 64 != 0))
                p.units[u].keep_y32 = 1;
            gin = p.units[u].y;
            ginh = p.units[u].yh;
            gh /= 2;
        }
    }
    p.wg_first = off;
    for (Unit& u : p.units) {
        const ConvLayer& c = a.convs[u.ci];
        u.wg = take(u.stem ? (size_t)c.cout * 8 * 32 : (size_t)c.cout * c.k * c.k * c.cin);
    }
    p.wg_floats = off - p.wg_first;
    const size_t rows = (size_t)T_COLS * b;
    p.seq = take(rows * 1024);
    p.gx = take(rows * 4096);
    p.y1 = take(rows * 1024);
    p.y1d = take(rows * 1024);
    p.y2 = take(rows * 1024);
    p.y2d = take(rows * 1024);
    p.save[0] = take(rows * 2 * 5 * 512);
    p.save[1] = take(rows * 2 * 5 * 512);
    p.dlin = take(rows * 12 + HN_HEAD_BWD_SCRATCH_FLOATS);     // d(Linear output) + the partial sums of the head's weight gradient
    p.gmax = b * 256 * 512 * 64;                 // largest gradient tensor: d(stem output)
    for (int i = 0; i < 4; ++i) p.G[i] = take(p.gmax);
    for (int i = 0; i < 2; ++i) p.dzh[i] = take(p.gmax / 2);
    const size_t csz[4] = {b * 128 * 256 * 256, b * 64 * 128 * 512, b * 32 * 64 * 1024, b * 16 * 32 * 2048};
    for (int i = 0; i < 4; ++i) p.DC[i] = take(csz[i]);
    p.lxch = take((hn_lstm_bwd_bf16_xch_bytes() > hn_lstm_bf16_xch_bytes() ? hn_lstm_bwd_bf16_xch_bytes() : hn_lstm_bf16_xch_bytes()) / sizeof(float));
    p.dhrec = take(b * 1024);
    p.dcrec = take(b * 1024);
    p.wsA = take((size_t)1024 * 18432 + 64);      // largest packed weight (ghc3.0) / wgrad scratch
    p.wsB = take((size_t)1024 * 18432 + 64);
    // statistics arena (doubles): every unit has its own forward and backward slots, so ONE memset per forward and ONE
    // per backward pass replace the two or three per unit (~160 tiny launches per step)
    p.stat_fwd_doubles = stat_f;
    p.stat_bwd_first = stat_f;
    p.stat_bwd_doubles = stat_b;
    p.stat_lstm = stat_f + stat_b;                // 4096 doubles for the LSTM bias gradients
    p.dstat = take((stat_f + stat_b + 4096) * 2);
    p.total = off;
    return p;
}

struct Ctx {
    hn_engine* e;
    const Arch& a;
    const TrainPlan& pl;
    float* W;              // workspace
    const float* P;        // packed weights (forward layout)
    float* grads;          // flat gradient buffer (may be null in forward)
    hipStream_t s;
    int B;
    std::vector<MJob>* jobs;   // backward: deferred gradient un-packs / double -> float copies of the running segment (null: launch at once)
    bool gh() const { return e->train_bf16 != 0; }      // bf16 mode: gradients BETWEEN conv units (dy, identity gradients) live in bf16
    bool bn_eval(const Unit& u) const { return (size_t)u.ci < e->bn_eval.size() && e->bn_eval[u.ci] != 0; }
    const float* bound(const std::string& k) const { return reinterpret_cast<const float*>(e->bound.at(k)); }
    float* bound_mut(const std::string& k) const { return const_cast<float*>(reinterpret_cast<const float*>(e->bound.at(k))); }
    float* grad(const std::string& k) const { return grads + a.grad_off.at(k); }
};

int conv_z(const Ctx& c, const Unit& u)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = c.W + u.x; d.w = c.P + cl.w_off; d.scale = c.P + c.a.ones_off;
    d.shift = cl.has_bias ? c.bound(cl.wkey + ".bias") : c.P + c.a.zeros_off;
    d.res = nullptr; d.y = c.W + u.z;
    d.B = c.B; d.Hi = u.Hi; d.Wi = u.Wi; d.Cin = u.stem ? 4 : cl.cin; d.Cout = cl.cout; d.KH = cl.k; d.KW = cl.k;
    d.sh = u.sh; d.sw = u.sw; d.ph = cl.k / 2; d.pw = cl.k / 2; d.Ho = u.Ho; d.Wo = u.Wo; d.relu = 0; d.ldy = cl.cout; d.stem = u.stem;
    // batch statistics of z come out of the conv epilogue (sum / sum of squares per channel, double atomics)
    if (!c.bn_eval(u)) {       // (a BatchNorm in eval() normalises with its running statistics: no batch statistics needed)
        double* ds = reinterpret_cast<double*>(c.W + c.pl.dstat) + u.sf;       // zeroed once at the start of hn_train_forward
        d.stat_sum = ds;
        d.stat_sq = ds + cl.cout;
        d.stat_rep = hn_stat_replicas(cl.cout, u.M);
    }
    if (c.e->train_bf16) {     // bf16 operands on the matrix cores (the unit's input has a bf16 copy, weights packed by
                               // hn_pack_weights_bf16), float32 accumulation, float32 z
        d.x = c.W + u.xh;
        d.w = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(c.e->packed_h) + hn_bf16_conv_offset(u.ci));
        d.Cin = u.stem ? 8 : cl.cin;
        return hn_launch_conv_bf16(d, 0, c.s);     // z itself is stored as bf16 (its statistics come from the f32 accumulators)
    }
    return hn_launch_conv(d, c.s);
}

// batch statistics (left in `dstat` by conv_z of the same unit, which must be the last conv launched) ->
// y = act(bn(z) (+res)); updates the running statistics in place
int bn_forward(const Ctx& c, const Unit& u, const float* res, int relu, float momentum, int res_bf16 = 0)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    const int C = cl.cout;
    double* ds = reinterpret_cast<double*>(c.W + c.pl.dstat) + u.sf;
    float* st = c.W + u.st;
    int rc;
    // bf16 mode: the float32 y is written only where a float32 consumer exists (u.keep_y32): 4 of the pass's bytes per element
    float* y32 = (c.e->train_bf16 && (!u.keep_y32 || u.stem)) ? nullptr : c.W + u.y;      // (the stem's max-pool reads the bf16 copy)
    void* yh = c.e->train_bf16 ? c.W + u.yh : nullptr;
    unsigned char* mk = reinterpret_cast<unsigned char*>(c.W + u.mk);
    if (!c.bn_eval(u)) {      // statistics -> affine (+ running-stat update) inside the element pass's launch
        const int rep = (u.stem && c.e->train_bf16 && c.e->fuse_stem_pool) ? 1 : hn_stat_replicas(C, u.M);     // (the fused stem kernel adds straight into [sum | sq])
        if (rep > 1 && (rc = hn_launch_stat_replica_sum(ds, C, rep, c.s))) return rc;
        return hn_launch_affine_act_bn(c.W + u.z, ds, ds + C, (double)u.M, c.bound(cl.bnkey + ".weight"), c.bound(cl.bnkey + ".bias"),
                                       c.bound_mut(cl.bnkey + ".running_mean"), c.bound_mut(cl.bnkey + ".running_var"), momentum,
                                       st + 2 * C, st + 3 * C, st, st + C, res, y32, mk, yh, u.M, C, relu, c.e->train_bf16, res_bf16, c.s);
    }
    // frozen block (train.py:245-256): running statistics, untouched
    if ((rc = hn_launch_bn_eval_affine(c.bound(cl.bnkey + ".weight"), c.bound(cl.bnkey + ".bias"), c.bound(cl.bnkey + ".running_mean"),
                                       c.bound(cl.bnkey + ".running_var"), st + 2 * C, st + 3 * C, st, st + C, C, c.s)))
        return rc;
    return hn_launch_affine_act(c.W + u.z, st + 2 * C, st + 3 * C, res, y32, mk, yh, u.M, C, relu, c.e->train_bf16, res_bf16, c.s);
}

// ---- BatchNorm-folded form of a 1x1 / stride-1 conv unit (bn_fold.hip) ----
// Eligible: bf16 step with bf16 gradients, batch statistics, no debug tap anywhere (the taps read z / dz, which this form never stores).
bool bn_fold_ok(const Ctx& c, const Unit& u)
{
    static const char* env = getenv("HN_BN_FOLD");            // "0": the classical passes everywhere (A/B runs); "b": folded adjoint only
    const ConvLayer& cl = c.a.convs[u.ci];
    return u.fold != 0 && c.e->train_bf16 && c.gh() && c.e->fuse_bn_fold && !(env && env[0] == '0') && c.e->debug_unit < 0 && c.e->debug_unit2 < 0 &&
           !c.bn_eval(u) && cl.k == 1 && u.sh == 1 && u.sw == 1 && !cl.has_bias && !u.stem && !u.keep_y32 && cl.cin % 64 == 0 && cl.cout % 64 == 0 &&
           (double)u.M * cl.cout * 2.0 < 4294967296.0;
}

// Forward: y = act(bn(conv(x)) (+ res)) WITHOUT storing z: batch statistics from the Gram matrix of the input (G, A kept for the adjoint),
// then the conv with BatchNorm, residual, ReLU and the ReLU bit mask in its epilogue.  Instead of conv (writes z) + affine pass (reads z,
// res; writes y): the unit's input is read twice more (a quarter of y's size in the backbone), z (as large as y) is neither written nor read.
int bn_fold_forward(const Ctx& c, const Unit& u, const float* res_h, int relu, float momentum)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    const int N = cl.cout, K = cl.cin;
    float* keep = c.W + u.fold;
    float* st = c.W + u.st;
    const void* wh = reinterpret_cast<const unsigned short*>(c.e->packed_h) + hn_bf16_conv_offset(u.ci);
    int rc;
    // (the two packed-weight scratches of the plan are idle during the backbone's forward: one contiguous range for the Gram partials)
    if ((rc = hn_launch_bn_fold_gram(c.W + u.xh, u.M, K, keep, c.s, c.W + c.pl.wsA, (size_t)(c.pl.wsB - c.pl.wsA) + (size_t)1024 * 18432))) return rc;
    if ((rc = hn_launch_bn_fold_forward_stats(keep, wh, (double)u.M, N, K, c.bound(cl.bnkey + ".weight"), c.bound(cl.bnkey + ".bias"),
                                              c.bound_mut(cl.bnkey + ".running_mean"), c.bound_mut(cl.bnkey + ".running_var"), momentum, st + 2 * N,
                                              st + 3 * N, st, st + N, c.P + c.a.ones_off, c.P + c.a.zeros_off, c.s)))
        return rc;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = c.W + u.xh; d.w = reinterpret_cast<const float*>(wh); d.scale = st + 2 * N; d.shift = st + 3 * N; d.res = res_h; d.y = c.W + u.yh;
    d.B = c.B; d.Hi = u.Hi; d.Wi = u.Wi; d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1; d.Ho = u.Ho; d.Wo = u.Wo; d.relu = relu; d.ldy = N;
    d.mask_out = relu ? reinterpret_cast<unsigned char*>(c.W + u.mk) : nullptr;
    return hn_launch_conv_bf16(d, 0, c.s);
}

// BN + conv-weight adjoint of one unit.  dy: gradient w.r.t. the tensor the mask refers to (ymask = post-ReLU
// tensor or null).  Produces dz (gradient w.r.t. the conv output), optional dpre (masked dy, the identity-branch
// gradient), and the parameter gradients.
// dy_bf16: dy (and dpre) are bf16 tensors -- every unit in bf16 mode except the stem, whose dy comes from the float32 max-pool adjoint.
// pool_src (the stem in the bf16 step): `dy` is not materialised -- the BatchNorm adjoint forms the max-pool adjoint of the pooled gradient
// `pool_src` (bf16 [B][Hi/2][Wi/2][C], position words at pl.pidx) on the fly (hn_launch_bn_bwd_pool)
// bn_done: the BatchNorm adjoint of this unit (its sums and its bf16 dz in staging slot dzh_slot) was already produced by
// bn_backward_dual below; only the parameter gradients remain (dy / dz / dpre are not touched)
// sums_done: S1 / S2 of this unit are already in its slots (the producing conv's epilogue took the reduce pass: bn_fold_dgrad); the apply pass and
// the parameter gradients remain
int unit_backward(const Ctx& c, const Unit& u, const float* dy, const float* ymask, float* dz, float* dpre, int dzh_slot = 0, int dy_bf16 = -1,
                  const float* pool_src = nullptr, int bn_done = 0, int sums_done = 0)
{
    if (dy_bf16 < 0) dy_bf16 = c.gh() ? 1 : 0;
    // bf16 mode: dz is also written as bf16 (slot 0 / 1 of the staging buffers) for this unit's data-gradient GEMM
    void* dz_h = (c.e->train_bf16 && dzh_slot >= 0 && c.a.convs[u.ci].cout % 64 == 0) ? c.W + c.pl.dzh[dzh_slot] : nullptr;
    // the stem in bf16 mode with a bf16 dy (from the max-pool adjoint): all-bf16 BatchNorm adjoint like every other unit, dz only as
    // bf16 -- into the caller's `dz` buffer, which nothing else reads (the stem has no data gradient) -- and the weight gradient on
    // the matrix cores (hn_launch_stem_wgrad_bf16).  The float32 form moved 2.1 GB of dy, 2.1 GB of dz and 1 GB of input per step.
    const bool stem_bf16 = u.stem && c.e->train_bf16 && dy_bf16 == 1 && dz != nullptr;
    if (stem_bf16) dz_h = dz;
    const ConvLayer& cl = c.a.convs[u.ci];
    const int C = cl.cout;
    double* ds = reinterpret_cast<double*>(c.W + c.pl.dstat) + c.pl.stat_bwd_first + u.sb;    // zeroed once per backward pass
    const float* st = c.W + u.st;
    // ymask != null: the unit ends in a ReLU; the adjoint reads the bit mask its affine_act pass stored (1/16 of y's bytes)
    const unsigned char* bmask = ymask ? reinterpret_cast<const unsigned char*>(c.W + u.mk) : nullptr;
    int rc;
    if (bn_done) {
        HN_REQUIRE(dz_h && !u.stem && !cl.has_bias && !c.bn_eval(u), "unit_backward: bn_done needs the plain bf16 unit");
    } else
    if (pool_src) {
        HN_REQUIRE(stem_bf16 && bmask && !dpre && !cl.has_bias, "unit_backward: the pooled-gradient form is the bf16 stem's");
        if ((rc = hn_launch_bn_bwd_pool(pool_src, c.W + c.pl.pidx, c.B, u.Ho, u.Wo, bmask, c.W + u.z, st, st + C, c.bound(cl.bnkey + ".weight"), ds,
                                        ds + C, dz_h, C, 0, c.s)))
            return rc;
    } else
    if (!sums_done && (rc = hn_launch_bn_bwd_reduce(dy, bmask, c.W + u.z, st, st + C, ds, ds + C, u.M, C, c.e->train_bf16, dy_bf16, c.s))) return rc;
    // S2 / S1 are the BatchNorm weight / bias gradients: double -> float into the flat gradient buffer, deferred to the
    // segment's one batched launch (the sums stay untouched until then) ...
    const bool defer = c.jobs != nullptr && !c.bn_eval(u);
    if (defer) {
        c.jobs->push_back(mj_make(MJ_D2F, ds + C, c.grad(cl.bnkey + ".weight"), C));
        c.jobs->push_back(mj_make(MJ_D2F, ds, c.grad(cl.bnkey + ".bias"), C));
    } else {
        if ((rc = hn_launch_d2f(ds + C, c.grad(cl.bnkey + ".weight"), C, c.s))) return rc;
        if ((rc = hn_launch_d2f(ds, c.grad(cl.bnkey + ".bias"), C, c.s))) return rc;
    }
    // ... except under an eval-mode BatchNorm, whose sums are cleared right away: mean / invstd are constants there, so
    // dz = gamma * invstd * g -- the batch-statistics formula with both sums zero
    if (c.bn_eval(u)) HN_HIP(hipMemsetAsync(ds, 0, 2 * (size_t)C * sizeof(double), c.s));
    // bf16 mode: when both GEMMs of this unit read the bf16 copy and nothing else needs the float32 dz (no conv bias, no
    // debug tap), it is not written at all (4 of the pass's ~16 bytes per element)
    const bool tapped = (c.e->debug_unit >= 0 && &u == &c.pl.units[c.e->debug_unit]) ||
                        (c.e->debug_unit2 >= 0 && &u == &c.pl.units[c.e->debug_unit2]);
    const bool bf16_gemms = stem_bf16 || (dz_h && !u.stem && cl.cin % 64 == 0);
    // ... and the conv bias gradient (height-compression convs) is summed inside the all-bf16 apply kernel
    const bool fused_db = cl.has_bias && bf16_gemms && dy_bf16 && !tapped && C % 8 == 0;
    const bool f32_dz_needed = !bf16_gemms || (cl.has_bias && !fused_db) || tapped;
    double* db = ds + 2 * C;
    if (bn_done) {
        HN_REQUIRE(!f32_dz_needed && !fused_db, "unit_backward: bn_done units read their bf16 dz only");
    } else
    if (pool_src) {
        HN_REQUIRE(!f32_dz_needed && !fused_db, "unit_backward: the pooled-gradient form writes bf16 dz only");
        if ((rc = hn_launch_bn_bwd_pool(pool_src, c.W + c.pl.pidx, c.B, u.Ho, u.Wo, bmask, c.W + u.z, st, st + C, c.bound(cl.bnkey + ".weight"), ds,
                                        ds + C, dz_h, C, 1, c.s)))
            return rc;
    } else
    if ((rc = hn_launch_bn_bwd_apply(dy, bmask, c.W + u.z, st, st + C, c.bound(cl.bnkey + ".weight"), ds, ds + C, (double)u.M,
                                     f32_dz_needed ? dz : nullptr, dpre, dz_h, u.M, C, c.e->train_bf16, dy_bf16, fused_db ? db : nullptr,
                                     c.s)))
        return rc;
    if (c.e->debug_unit >= 0 && &u == &c.pl.units[c.e->debug_unit]) {
        if (c.e->debug_dy) {      // taps are float32 whatever the storage type
            if (dy_bf16) { if ((rc = hn_launch_bf16_to_f32(dy, c.e->debug_dy, (long)u.M * C, c.s))) return rc; }
            else HN_HIP(hipMemcpyAsync(c.e->debug_dy, dy, (size_t)u.M * C * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        }
        if (c.e->debug_dz) HN_HIP(hipMemcpyAsync(c.e->debug_dz, dz, (size_t)u.M * C * sizeof(float), hipMemcpyDeviceToDevice, c.s));
    }
    if (c.e->debug_unit2 >= 0 && &u == &c.pl.units[c.e->debug_unit2]) {
        if (c.e->debug_dy2) {
            if (dy_bf16) { if ((rc = hn_launch_bf16_to_f32(dy, c.e->debug_dy2, (long)u.M * C, c.s))) return rc; }
            else HN_HIP(hipMemcpyAsync(c.e->debug_dy2, dy, (size_t)u.M * C * sizeof(float), hipMemcpyDeviceToDevice, c.s));
        }
        if (c.e->debug_dz2) HN_HIP(hipMemcpyAsync(c.e->debug_dz2, dz, (size_t)u.M * C * sizeof(float), hipMemcpyDeviceToDevice, c.s));
    }
    if (cl.has_bias) {
        if (!fused_db && (rc = hn_launch_col_stats(dz, db, nullptr, u.M, C, C, c.s))) return rc;
        if (c.jobs) c.jobs->push_back(mj_make(MJ_D2F, db, c.grad(cl.wkey + ".bias"), C));
        else if ((rc = hn_launch_d2f(db, c.grad(cl.wkey + ".bias"), C, c.s))) return rc;
    }
    // weight gradient (packed layout, the unit's own scratch: zeroed with all the others at the start of the pass) -> OIHW in
    // the segment's batched un-pack
    float* scratch = c.jobs ? c.W + u.wg : c.W + c.pl.wsA;
    const int prezeroed = c.jobs ? 1 : 0;
    if (stem_bf16 && !tapped) {
        if ((rc = hn_launch_stem_wgrad_bf16(c.W + u.xh, dz_h, scratch, c.B, u.Hi, u.Wi, c.s, prezeroed))) return rc;
        // tap dw sits at window pixel dw + 1 of the packed scratch (the bf16 stem layout): un-pack from 4 floats in
        if (c.jobs) {
            c.jobs->push_back(mj_unpack(scratch + 4, c.grad(cl.wkey + ".weight"), cl.cout, cl.cin, cl.k, cl.k, 8));
            return 0;
        }
        return hn_launch_unpack_conv(scratch + 4, c.grad(cl.wkey + ".weight"), cl.cout, cl.cin, cl.k, cl.k, 8, c.s);
    }
    if (dz_h && !u.stem && cl.cin % 64 == 0) {     // bf16 mode: both operands exist as bf16 copies (input: u.xh, dz: staging slot)
        if ((rc = hn_launch_conv_wgrad_bf16(c.W + u.xh, dz_h, scratch, c.B, u.Hi, u.Wi, cl.cin, cl.cout, cl.k, cl.k, u.sh, u.sw, c.s, prezeroed)))
            return rc;
    } else if ((rc = hn_launch_conv_wgrad(c.W + u.x, dz, scratch, c.B, u.Hi, u.Wi, cl.cin, cl.cout, cl.k, cl.k, u.sh, u.sw, 0, 0, u.stem,
                                          c.s, prezeroed)))
        return rc;
    if (c.jobs) {
        c.jobs->push_back(mj_unpack(scratch, c.grad(cl.wkey + ".weight"), cl.cout, cl.cin, cl.k, cl.k, u.stem ? 8 : 0));
        return 0;
    }
    return hn_launch_unpack_conv(scratch, c.grad(cl.wkey + ".weight"), cl.cout, cl.cin, cl.k, cl.k, u.stem ? 8 : 0, c.s);
}

// Block 0 of a ResNet stage (bf16 step): conv3's BatchNorm (ua, ends in the block's ReLU) and the downsample branch's (ub) see the same masked
// gradient: one reduce + one apply pass for both (hn_launch_bn_bwd_dual) -> their sums in the two units' slots, bf16 dz of ua / ub in staging
// slots slot_a / slot_b; the masked gradient itself (dpre) is not written.  unit_backward(..., bn_done = 1) then adds the parameter gradients.
bool bn_dual_ok(const Ctx& c, const Unit& ua, const Unit& ub)
{
    static const char* env = getenv("HN_FUSE_BN_DUAL");          // "0": two separate adjoints (A/B runs)
    const int C = c.a.convs[ua.ci].cout;
    auto tapped = [&](const Unit& u) {
        return (c.e->debug_unit >= 0 && &u == &c.pl.units[c.e->debug_unit]) || (c.e->debug_unit2 >= 0 && &u == &c.pl.units[c.e->debug_unit2]);
    };
    return c.e->train_bf16 && c.gh() && c.e->fuse_bn_dual && !(env && env[0] == '0') && !c.bn_eval(ua) && !c.bn_eval(ub) && !tapped(ua) && !tapped(ub) &&
           c.a.convs[ub.ci].cout == C && ua.M == ub.M && C % 64 == 0 && c.a.convs[ua.ci].cin % 64 == 0 && c.a.convs[ub.ci].cin % 64 == 0 &&
           !c.a.convs[ua.ci].has_bias && !c.a.convs[ub.ci].has_bias;
}

int bn_backward_dual(const Ctx& c, const Unit& ua, const Unit& ub, const float* dy, int slot_a, int slot_b)
{
    const ConvLayer& ca = c.a.convs[ua.ci];
    const ConvLayer& cb = c.a.convs[ub.ci];
    const int C = ca.cout;
    double* dsa = reinterpret_cast<double*>(c.W + c.pl.dstat) + c.pl.stat_bwd_first + ua.sb;
    double* dsb = reinterpret_cast<double*>(c.W + c.pl.dstat) + c.pl.stat_bwd_first + ub.sb;
    const float* sta = c.W + ua.st;
    const float* stb = c.W + ub.st;
    const unsigned char* bmask = reinterpret_cast<const unsigned char*>(c.W + ua.mk);
    for (int phase = 0; phase < 2; ++phase) {
        if (int rc = hn_launch_bn_bwd_dual(dy, bmask, c.W + ua.z, sta, sta + C, c.bound(ca.bnkey + ".weight"), dsa, dsa + C, c.W + c.pl.dzh[slot_a],
                                           c.W + ub.z, stb, stb + C, c.bound(cb.bnkey + ".weight"), dsb, dsb + C, c.W + c.pl.dzh[slot_b], ua.M, C, phase,
                                           c.s))
            return rc;
    }
    return 0;
}

// ---- BatchNorm-folded adjoint of a 1x1 / stride-1 conv unit (bn_fold.hip): no reduce pass, no apply pass, no dz tensor ----
// did hn_train_forward run this unit through bn_fold_forward (z was never stored: the adjoint MUST be the folded one)?
bool fwd_folded(const Ctx& c, const Unit& u)
{
    const size_t i = (size_t)(&u - c.pl.units.data());
    auto it = c.e->fold_fwd.find(c.W);
    return it != c.e->fold_fwd.end() && i < it->second.size() && it->second[i] != 0;
}

struct FoldOut {
    const void* wa;          // [K][N] bf16: weights of the data-gradient conv on g
    const float* shift_a;    // [K]
    const void* wb;          // [K][K] bf16: weights of the correction conv on the unit's input
};

// g_io: the gradient w.r.t. the unit's BatchNorm output side [M][N] bf16 -- with `bmask` it arrives as dy of the block output and LEAVES
// as g = dy * relu'(block output) (masked in place); without, it is read as is.  s1_src: null = this unit sums g itself (into its S1
// slot), else the slot of the unit that already did (block 0: conv3 and the downsample branch see the same g).
// ws: scratch of hn_bn_fold_scratch_bytes (one of the plan's two packed-weight scratches).
// a_part (layer1's 256 x 64 shape, HN_FOLD_FUSEA != 0): the conv-A half of the data gradient, g (c1 * W), is taken in the P-GEMM's own pass over g
// and lands there ([M][K] bf16; bn_fold_dgrad then only runs conv B on top of it); write_back = 0: g is NOT written back over dy (block 0 of layer1,
// where nothing but the two fused P-GEMMs -- both masking on load -- reads it)
bool bn_fold_fuse_a(const Ctx& c, const Unit& u)
{
    static const char* env = getenv("HN_FOLD_FUSEA");
    const ConvLayer& cl = c.a.convs[u.ci];
    return cl.cout == 256 && cl.cin == 64 && !(env && env[0] == '0');
}

int bn_fold_unit(const Ctx& c, const Unit& u, float* g_io, const unsigned char* bmask, const double* s1_src, float* ws, FoldOut* out,
                 float* a_part = nullptr, int write_back = 1)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    const int N = cl.cout, K = cl.cin;
    double* ds = reinterpret_cast<double*>(c.W + c.pl.dstat) + c.pl.stat_bwd_first + u.sb;      // [S1 | S2 | -], zeroed at the start of the pass
    const float* st = c.W + u.st;
    float* keep = c.W + u.fold;
    HN_REQUIRE(c.jobs != nullptr && hn_bn_fold_scratch_bytes(N, K) <= ((size_t)1024 * 18432 + 64) * sizeof(float), "bn_fold_unit: scratch too small");
    int rc;
    const int fwd = fwd_folded(c, u);            // the forward of this step was folded too: G, A, Wf, WG are in the unit's keep storage
    HN_HIP(hipMemsetAsync(ws, 0, hn_bn_fold_zero_bytes(K), c.s));      // Q
    static const char* slab_env = getenv("HN_FOLD_SLAB");
    const bool det = c.e->fold_deterministic && !(slab_env && slab_env[0] == '0');
    if (!fwd && (rc = hn_launch_bn_fold_gram(c.W + u.xh, u.M, K, keep, c.s, det ? c.W + c.pl.dzh[1] : nullptr, det ? c.pl.gmax / 2 : 0))) return rc;
    // P = g^T a into the unit's weight-gradient scratch (zeroed with all the others at the start of the pass)
    float* P = c.W + u.wg;
    const void* wh = reinterpret_cast<const unsigned short*>(c.e->packed_h) + hn_bf16_conv_offset(u.ci);
    void* wa_early = nullptr;
    if (a_part) {      // (c1 * W)^T does not depend on the batch sums: into its place in the scratch before the P-GEMM reads it
        wa_early = hn_bn_fold_wa_ptr(ws, N, K);
        if ((rc = hn_launch_bn_fold_wa(wh, c.bound(cl.bnkey + ".weight"), st + N, wa_early, N, K, c.s))) return rc;
    }
    // Round 6: P feeds S2, dW and BOTH data-gradient weight matrices, so float atomics in arrival order here made every gradient UPSTREAM of a folded
    // unit differ from run to run (worst gradient norm +-1e-2 on the seeded B = 1 net; the classical adjoint, whose only float atomics end in leaf
    // dW tensors, repeats to 1e-8).  Default: the slab form (per-split partial tiles with plain stores + an ordered double-precision reduce,
    // conv_wgrad_bf16.hip) in the second dz staging buffer, which no folded block uses (need <= 2.2 M floats per panorama, the buffer holds 4.2 M).
    if ((rc = hn_launch_conv_wgrad_bf16_fold(c.W + u.xh, g_io, P, u.M, K, N, bmask, s1_src ? nullptr : ds, c.s, 1, det ? c.W + c.pl.dzh[1] : nullptr,
                                             det ? c.pl.gmax / 2 : 0, wa_early, a_part, write_back)))
        return rc;
    if ((rc = hn_launch_bn_fold_finish(P, keep, fwd, s1_src ? s1_src : ds, ds, ds + N, wh, st, st + N, c.bound(cl.bnkey + ".weight"), (double)u.M, N, K, ws,
                                       c.P + c.a.ones_off, c.P + c.a.zeros_off, &out->wa, &out->shift_a, &out->wb, c.s)))
        return rc;
    c.jobs->push_back(mj_make(MJ_D2F, ds + N, c.grad(cl.bnkey + ".weight"), N));
    c.jobs->push_back(mj_make(MJ_D2F, ds, c.grad(cl.bnkey + ".bias"), N));
    c.jobs->push_back(mj_unpack(P, c.grad(cl.wkey + ".weight"), N, K, 1, 1, 0));
    return 0;
}

// da = g (c1 * W) - a Q - r (+ add): two plain 1x1 convs; tmp and out are [M][K] bf16 gradient buffers (tmp != out)
// a_done: tmp already holds g (c1 * W) (bn_fold_unit's a_part); conv B then also applies the shift -r
// red_unit (HN_FOLD_REDUCE != 0): the BatchNorm + ReLU unit whose output gradient `out` is (conv2 of the block): conv B's epilogue also takes
// that unit's reduce pass (sum g, sum g * zhat per tile into a slab behind `slab`, added by hn_launch_slab_colsum into the unit's zeroed slots)
bool bn_fold_reduce_on()
{
    static const char* env = getenv("HN_FOLD_REDUCE");
    return !(env && env[0] == '0');
}

int bn_fold_dgrad(const Ctx& c, const Unit& u, const FoldOut& f, const float* g, float* tmp, float* out, int a_done = 0, const Unit* red_unit = nullptr,
                  float* slab = nullptr)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    const int N = cl.cout, K = cl.cin;
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.B = c.B; d.Hi = u.Ho; d.Wi = u.Wo; d.Ho = u.Ho; d.Wo = u.Wo; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1; d.relu = 0;
    d.scale = c.P + c.a.ones_off;
    if (!a_done) {
        d.x = g; d.w = reinterpret_cast<const float*>(f.wa); d.shift = f.shift_a; d.res = nullptr; d.y = tmp; d.Cin = N; d.Cout = K; d.ldy = K;
        if (int rc = hn_launch_conv_bf16(d, 0, c.s)) return rc;
    }
    d.x = c.W + u.xh; d.w = reinterpret_cast<const float*>(f.wb); d.shift = a_done ? f.shift_a : c.P + c.a.zeros_off; d.res = tmp; d.y = out;
    d.Cin = K; d.Cout = K; d.ldy = K;
    if (red_unit) {
        const Unit& r = *red_unit;
        HN_REQUIRE(c.a.convs[r.ci].cout == K && r.M == u.M && r.relu && !c.bn_eval(r), "bn_fold_dgrad: the fused reduce is the block's conv2 unit's");
        const float* rst = c.W + r.st;
        d.bn_z = c.W + r.z; d.bn_mask = reinterpret_cast<const unsigned char*>(c.W + r.mk); d.bn_mean = rst; d.bn_invstd = rst + K; d.bn_slab = slab;
        if (int rc = hn_launch_conv_bf16(d, 0, c.s)) return rc;
        const int tiles = hn_cdiv(u.M, hn_conv_bf16_bn_tile_rows(K, u.M));
        double* rds = reinterpret_cast<double*>(c.W + c.pl.dstat) + c.pl.stat_bwd_first + r.sb;
        return hn_launch_slab_colsum(slab, tiles, 2 * K, rds, c.s);          // [S1 | S2] are adjacent in the unit's slot
    }
    return hn_launch_conv_bf16(d, 0, c.s);
}

// data gradient of one unit: dx = conv^T(dz) (+ add), one launch per stride-parity class
int unit_dgrad(const Ctx& c, const Unit& u, const float* dz, const float* add, float* dx, int dzh_slot = 0)
{
    const ConvLayer& cl = c.a.convs[u.ci];
    ConvDesc d;                         // the FORWARD conv's geometry
    memset(&d, 0, sizeof(d));
    d.B = c.B; d.Hi = u.Hi; d.Wi = u.Wi; d.Cin = cl.cin; d.Cout = cl.cout; d.KH = cl.k; d.KW = cl.k;
    d.sh = u.sh; d.sw = u.sw; d.ph = cl.k / 2; d.pw = cl.k / 2; d.Ho = u.Ho; d.Wo = u.Wo;
    if (c.e->train_bf16 && cl.cout % 64 == 0) {    // (ghc0.3 has Cout = 32: stays on the f32 path)
        // the per-class bf16 weight packings were written by hn_pack_weights_bf16 (once per optimiser step, one launch)
        const size_t doff = hn_bf16_dgrad_offset(u.ci);
        HN_REQUIRE(doff != (size_t)-1, "unit_dgrad: conv %d has no bf16 data-gradient packing", u.ci);
        void* wpk = const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(c.e->packed_h) + doff);
        return hn_launch_conv_dgrad_bf16(d, c.W + c.pl.dzh[dzh_slot], nullptr, add, dx, wpk, c.P + c.a.ones_off, c.P + c.a.zeros_off, c.s,
                                         /*grad_bf16=*/1);
    }
    if (c.gh()) {
        // float32 kernel inside a bf16-gradient pass (only ghc0.3, no identity branch): float32 dX into scratch, then one
        // conversion pass into the bf16 gradient tensor
        HN_REQUIRE(add == nullptr, "unit_dgrad: the float32 fallback has no bf16 identity input");
        const long n = (long)c.B * u.Hi * u.Wi * cl.cin;
        HN_REQUIRE((size_t)n <= (size_t)T_COLS * c.B * 4096, "unit_dgrad: float32 fallback scratch too small");
        float* tmp = c.W + c.pl.gx;      // the LSTM gate buffer: free once the recurrent adjoint is done
        if (int rc = hn_launch_conv_dgrad(d, dz, c.bound(cl.wkey + ".weight"), nullptr, tmp, c.W + c.pl.wsB, c.P + c.a.ones_off,
                                          c.P + c.a.zeros_off, c.s))
            return rc;
        return hn_launch_f32_to_bf16(tmp, dx, n, c.s);
    }
    return hn_launch_conv_dgrad(d, dz, c.bound(cl.wkey + ".weight"), add, dx, c.W + c.pl.wsB, c.P + c.a.ones_off, c.P + c.a.zeros_off, c.s);
}

int gemm_rows(const float* x, int xstride, const float* w, const float* scale, const float* shift, float* y, long rows, int K, int N,
              hipStream_t s)
{
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = w; d.scale = scale; d.shift = shift; d.res = nullptr; d.y = y;
    d.B = 1; d.Hi = 1; d.Wi = (int)rows; d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1;
    d.Ho = 1; d.Wo = (int)rows; d.relu = 0; d.ldy = N; d.xstride = xstride;
    return hn_launch_conv(d, s);
}

// y[rows][N] (f32) = x_h[rows][K] (bf16) @ w_h[N][K]^T (bf16) * scale + shift on the bf16 matrix cores
int gemm_rows_bf16(const void* x_h, const void* w_h, const float* scale, const float* shift, float* y, long rows, int K, int N, hipStream_t s)
{
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = reinterpret_cast<const float*>(x_h); d.w = reinterpret_cast<const float*>(w_h); d.scale = scale; d.shift = shift; d.res = nullptr; d.y = y;
    d.B = 1; d.Hi = 1; d.Wi = (int)rows; d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1; d.sh = 1; d.sw = 1;
    d.Ho = 1; d.Wo = (int)rows; d.relu = 0; d.ldy = N;
    return hn_launch_conv_bf16(d, 1, s);
}

}  // namespace

void hn_conv_strides(int ci, int* sh, int* sw)
{
    static const std::vector<Unit> units = make_train_plan(1).units;
    *sh = *sw = 1;
    for (const Unit& u : units)
        if (u.ci == ci) { *sh = u.sh; *sw = u.sw; return; }
}

extern "C" size_t hn_train_workspace_bytes(int B)
{
    if (B < 1) return 0;
    return make_train_plan(B).total * sizeof(float);
}

// Debug taps for the parity tests: geometry / workspace offsets of training unit `unit` (forward order: stem, then per
// bottleneck conv1, conv2, [downsample], conv3, then the 4 height-compression convs of the scale), and a copy-out of the
// gradients entering (dy) / leaving (dz) its BatchNorm adjoint during the next hn_train_backward.
extern "C" int hn_train_debug_unit(int B, int unit, int64_t* out8)
{
    const TrainPlan pl = make_train_plan(B);
    HN_REQUIRE(unit >= 0 && unit < (int)pl.units.size() && out8, "hn_train_debug_unit: bad unit %d (of %d)", unit, (int)pl.units.size());
    const Unit& u = pl.units[unit];
    out8[0] = u.ci; out8[1] = u.M; out8[2] = arch().convs[u.ci].cout; out8[3] = (int64_t)u.x; out8[4] = (int64_t)u.z;
    out8[5] = (int64_t)u.y; out8[6] = (int64_t)u.st; out8[7] = (int64_t)pl.units.size();
    return 0;
}

// float offset of the unit's bf16 copy of y (train_precision bf16: the float32 y of most units is not written)
extern "C" int64_t hn_train_debug_unit_yh(int B, int unit)
{
    const TrainPlan pl = make_train_plan(B);
    if (unit < 0 || unit >= (int)pl.units.size()) return -1;
    return (int64_t)pl.units[unit].yh;
}

extern "C" int hn_train_debug_set(hn_engine* e, int unit, float* dy_dst, float* dz_dst)
{
    HN_REQUIRE(e != nullptr, "hn_train_debug_set: null engine");
    e->debug_unit = unit; e->debug_dy = dy_dst; e->debug_dz = dz_dst;
    return 0;
}

extern "C" int hn_train_debug_set2(hn_engine* e, int unit, float* dy_dst, float* dz_dst)
{
    HN_REQUIRE(e != nullptr, "hn_train_debug_set2: null engine");
    e->debug_unit2 = unit; e->debug_dy2 = dy_dst; e->debug_dz2 = dz_dst;
    return 0;
}

extern "C" int hn_set_bn_eval(hn_engine* e, const char* bn_prefix, int eval)
{
    HN_REQUIRE(e != nullptr && bn_prefix != nullptr, "hn_set_bn_eval: null argument");
    const Arch& a = arch();
    if (e->bn_eval.size() != a.convs.size()) e->bn_eval.assign(a.convs.size(), 0);
    for (size_t i = 0; i < a.convs.size(); ++i)
        if (a.convs[i].bnkey == bn_prefix) {
            e->bn_eval[i] = eval ? 1 : 0;
            return 0;
        }
    HN_REQUIRE(false, "hn_set_bn_eval: no BatchNorm with prefix '%s'", bn_prefix);
}

extern "C" int hn_set_train_precision(hn_engine* e, int bf16)
{
    HN_REQUIRE(e != nullptr && (bf16 == 0 || bf16 == 1), "hn_set_train_precision: bad argument");
    e->train_bf16 = bf16;
    return 0;
}

extern "C" size_t hn_grad_floats(void) { return arch().grad_floats; }

extern "C" int64_t hn_grad_offset(const char* name)
{
    const Arch& a = arch();
    auto it = a.grad_off.find(name ? name : "");
    return it == a.grad_off.end() ? -1 : (int64_t)it->second;
}

extern "C" int hn_train_forward(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                                size_t workspace_bytes, float p_rnn, float p_head, float bn_momentum, uint64_t seed, void* stream)
{
    HN_REQUIRE(e && x && bon && cor && workspace, "hn_train_forward: null argument");
    HN_REQUIRE(e->packed != nullptr, "hn_train_forward: hn_pack_weights has not been called");
    HN_REQUIRE(B >= 1 && C_in >= 3, "hn_train_forward: bad B=%d / C_in=%d", B, C_in);
    HN_REQUIRE(p_rnn >= 0.f && p_rnn < 1.f && p_head >= 0.f && p_head < 1.f, "hn_train_forward: dropout p out of range");
    const TrainPlan pl = make_train_plan(B);
    HN_REQUIRE(workspace_bytes >= pl.total * sizeof(float), "hn_train_forward: workspace too small (%zu < %zu)", workspace_bytes,
               pl.total * sizeof(float));
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_train_forward: cannot select device %d", e->device);
    Ctx c{e, arch(), pl, reinterpret_cast<float*>(workspace), e->packed, nullptr, (hipStream_t)stream, B, nullptr};
    const Arch& a = c.a;
    float* W = c.W;
    hipStream_t s = c.s;
    int rc;
    // debug instrument (hn_engine::poison): nothing behind the sync page survives from one training forward to the next
    if ((rc = hn_poison(e, W + pl.xn, (pl.total - pl.xn) * sizeof(float), s))) return rc;
    if ((rc = hn_poison(e, bon, (size_t)B * 2 * 1024 * sizeof(float), s)) || (rc = hn_poison(e, cor, (size_t)B * 1024 * sizeof(float), s))) return rc;
    HN_HIP(hipMemsetAsync(W + pl.sync, 0, HN_STATUS_WORD * sizeof(unsigned), s));   // arrival counters; the status word behind them is sticky (zeroed by the caller at allocation)
    HN_HIP(hipMemsetAsync(W + pl.dstat, 0, pl.stat_fwd_doubles * sizeof(double), s));   // every unit's batch-statistics slot

    // stem
    if ((rc = hn_launch_prep_nhwc4(x, W + pl.xn, B, C_in, IMG_H, IMG_W, s))) return rc;          // f32: the stem's weight gradient reads it
    if (e->train_bf16) {
        HN_REQUIRE(e->packed_h != nullptr, "hn_train_forward: train_precision bf16 needs hn_pack_weights_bf16");
        if ((rc = hn_launch_prep_nhwc4_bf16(x, W + pl.xn8, B, C_in, IMG_H, IMG_W, s))) return rc;
    }
    {
        const Unit& u = pl.units[0];
        if (e->train_bf16 && e->fuse_stem_pool && !c.bn_eval(u)) {
            // the stem conv straight from the float32 planes through an LDS ring of input rows (stem_pool_bf16.hip, TRAIN mode): z as
            // bf16 + the batch statistics of the float32 accumulators, without the implicit GEMM's 12x im2col expansion
            double* ds = reinterpret_cast<double*>(W + pl.dstat) + u.sf;
            const void* wpk = reinterpret_cast<const unsigned short*>(e->packed_h) + hn_bf16_conv_offset(u.ci);
            if ((rc = hn_launch_stem_conv_train_bf16(x, C_in, wpk, W + u.z, ds, ds + a.convs[u.ci].cout, B, s))) return rc;
        } else if ((rc = conv_z(c, u))) return rc;
        // bf16 mode, batch statistics: BatchNorm + ReLU + max-pool in ONE pass over z (train_ops.hip: affine_act_bn_pool_kernel) -- the bf16
        // activation of the stem is consumed by the pool only.  HN_FUSE_STEM_BNPOOL=0: the two passes (A/B runs, the bit-equality test)
        static const char* fbp = getenv("HN_FUSE_STEM_BNPOOL");
        const bool stem_tapped_f = e->debug_unit == 0 || e->debug_unit2 == 0;      // a parity test reads this unit's tensors: keep the two-pass form (it writes the bf16 activation)
        if (e->train_bf16 && !c.bn_eval(u) && !stem_tapped_f && e->fuse_stem_bnpool && !(fbp && fbp[0] == '0')) {
            const ConvLayer& cl = a.convs[u.ci];
            const int C = cl.cout;
            double* ds = reinterpret_cast<double*>(W + pl.dstat) + u.sf;
            float* st = W + u.st;
            const int rep = (e->fuse_stem_pool) ? 1 : hn_stat_replicas(C, u.M);
            if (rep > 1 && (rc = hn_launch_stat_replica_sum(ds, C, rep, s))) return rc;
            if ((rc = hn_launch_affine_act_bn_pool(W + u.z, ds, ds + C, (double)u.M, c.bound(cl.bnkey + ".weight"), c.bound(cl.bnkey + ".bias"),
                                                   c.bound_mut(cl.bnkey + ".running_mean"), c.bound_mut(cl.bnkey + ".running_var"), bn_momentum,
                                                   st + 2 * C, st + 3 * C, st, st + C, reinterpret_cast<unsigned char*>(W + u.mk), W + pl.poolh,
                                                   W + pl.pidx, B, 256, 512, C, s)))
                return rc;
        } else {
        if ((rc = bn_forward(c, u, nullptr, 1, bn_momentum))) return rc;
        if (e->train_bf16) {
            // the stem's activation exists as bf16 only (bn_forward: no float32 y for the stem in bf16 mode) and is pooled as bf16
            // straight into the bf16 block input of layer1: 2.1 GB of float32 y written + read and a conversion pass less per step
            if ((rc = hn_launch_maxpool_idx(W + u.yh, nullptr, W + pl.pidx, B, 256, 512, 64, s, /*in_bf16=*/1, W + pl.poolh))) return rc;
        } else if ((rc = hn_launch_maxpool_idx(W + u.y, W + pl.pool, W + pl.pidx, B, 256, 512, 64, s))) return rc;
        }
    }
    // backbone + height compression
    static const char* fold_env = getenv("HN_BN_FOLD");
    const bool fold_fwd_on = !(fold_env && fold_env[0] == 'b') && e->fuse_bn_fold == 1;       // "b" / option value 2: classical forward (z stored), folded adjoint
    if (e->fold_fwd.size() > 64 && e->fold_fwd.find(workspace) == e->fold_fwd.end()) e->fold_fwd.clear();     // (re-allocated workspaces: keep the table small)
    std::vector<unsigned char>& fold_fwd = e->fold_fwd[workspace];
    fold_fwd.assign(pl.units.size(), 0);
    for (int li = 0; li < 4; ++li) {
        for (int j = 0; j < kBlocks[li]; ++j) {
            const int i1 = pl.blk[li][j];
            const Unit& u1 = pl.units[i1];
            const Unit& u2 = pl.units[i1 + 1];
            const Unit& u3 = pl.units[i1 + (j == 0 ? 3 : 2)];
            if ((rc = conv_z(c, u1)) || (rc = bn_forward(c, u1, nullptr, 1, bn_momentum))) return rc;
            if ((rc = conv_z(c, u2)) || (rc = bn_forward(c, u2, nullptr, 1, bn_momentum))) return rc;
            // identity branch of the block: float32, or (bf16 mode) the bf16 copy the previous pass wrote anyway
            const float* idt = e->train_bf16 ? W + u1.xh : W + u1.x;
            // the folded form (bn_fold.hip) where the ADJOINT will be folded: same rule as train_backward_impl
            const bool fold3 = fold_fwd_on && bn_fold_ok(c, u3) && (j > 0 || bn_fold_ok(c, pl.units[pl.dsu[li]]));
            if (j == 0) {
                const Unit& ud = pl.units[pl.dsu[li]];
                if (fold3) {
                    if ((rc = bn_fold_forward(c, ud, nullptr, 0, bn_momentum))) return rc;
                    fold_fwd[pl.dsu[li]] = 1;
                } else if ((rc = conv_z(c, ud)) || (rc = bn_forward(c, ud, nullptr, 0, bn_momentum))) return rc;
                idt = e->train_bf16 ? W + ud.yh : W + ud.y;
            }
            if (fold3) {
                if ((rc = bn_fold_forward(c, u3, idt, 1, bn_momentum))) return rc;
                fold_fwd[i1 + (j == 0 ? 3 : 2)] = 1;
            } else
            if ((rc = conv_z(c, u3)) || (rc = bn_forward(c, u3, idt, 1, bn_momentum, e->train_bf16))) return rc;
        }
        for (int k = 0; k < 4; ++k) {
            const Unit& ug = pl.units[pl.ghc[li] + k];
            if ((rc = conv_z(c, ug)) || (rc = bn_forward(c, ug, nullptr, 1, bn_momentum))) return rc;
        }
        const Unit& ul = pl.units[pl.ghc[li] + 3];
        if ((rc = hn_launch_upsample_flatten(W + ul.y, W + pl.seq, B, ul.Ho, ul.Wo, a.convs[ul.ci].cout, 256 * li, s))) return rc;
    }
    // bi-LSTM x2 (+ inter-layer dropout) + head dropout + Linear
    const long rows = (long)T_COLS * B;
    const float* lin = W + pl.seq;
    float* ybuf[2] = {W + pl.y1, W + pl.y2};
    float* ydrop[2] = {W + pl.y1d, W + pl.y2d};
    const float pdrop[2] = {p_rnn, p_head};
    for (int l = 0; l < 2; ++l) {
        if (e->train_bf16) {
            // bf16 mode (the reference runs the whole net under autocast, train.py:273): the input projection on the bf16 matrix
            // cores from a bf16 copy of the layer input and the W_ih / W_hh roundings hn_pack_weights_bf16 made for this step;
            // the batch-partitioned persistent recurrence of the bf16 forward, also storing the gates for the adjoint
            const unsigned short* H = reinterpret_cast<const unsigned short*>(e->packed_h);
            unsigned short* lin_h = reinterpret_cast<unsigned short*>(W + pl.wsB);
            HN_REQUIRE((size_t)rows * 1024 <= ((size_t)1024 * 18432) * 2, "hn_train_forward: batch too large for the bf16 LSTM input staging");
            if ((rc = hn_launch_f32_to_bf16(lin, lin_h, rows * 1024, s))) return rc;
            if ((rc = gemm_rows_bf16(lin_h, H + hn_bf16_wih_offset(l), c.P + a.ones_off, c.P + a.lbias_off[l], W + pl.gx, rows, 1024, 4096, s))) return rc;
            // the wide recurrence kernel (a group = one direction of 16 panoramas: one launch for up to 64 panoramas on 64 compute
            // units instead of one 256-workgroup launch per 32); its exchange buffer is the bf16 staging of the layer input, free
            // again once the input GEMM above has read it.  HN_TRAIN_WIDE_LSTM=0: the narrow kernel (A/B).
            static const char* wenv = getenv("HN_TRAIN_WIDE_LSTM");
            if (!(wenv && wenv[0] == '0')) {
                if ((rc = hn_launch_lstm_layer_bf16_wide(W + pl.gx, H + hn_bf16_whh_offset(l, 0), H + hn_bf16_whh_offset(l, 1), ybuf[l], lin_h,
                                                         T_COLS, B, W + pl.sync, 16, 2, s, W + pl.save[l])))
                    return rc;
            } else if ((rc = hn_launch_lstm_layer_bf16(W + pl.gx, H + hn_bf16_whh_offset(l, 0), H + hn_bf16_whh_offset(l, 1), ybuf[l], nullptr,
                                                       T_COLS, B, W + pl.lxch, W + pl.sync, s, W + pl.save[l])))
                return rc;
        } else {
            if ((rc = gemm_rows(lin, 0, c.P + a.wih_off[l], c.P + a.ones_off, c.P + a.lbias_off[l], W + pl.gx, rows, 1024, 4096, s))) return rc;
            if ((rc = hn_launch_lstm_layer(W + pl.gx, c.P + a.whh_off[l][0], c.P + a.whh_off[l][1], ybuf[l], T_COLS, B, W + pl.sync, s,
                                           W + pl.save[l])))
                return rc;
        }
        if (pdrop[l] > 0.f) {
            if ((rc = hn_launch_dropout(ybuf[l], ydrop[l], rows * 1024, pdrop[l], seed * 2 + 1 + l, s))) return rc;
        } else {
            HN_HIP(hipMemcpyAsync(ydrop[l], ybuf[l], rows * 1024 * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        lin = ydrop[l];
    }
    return hn_launch_linear_head(W + pl.y2d, c.P + a.linw_off, c.P + a.linb_off, bon, cor, T_COLS, B, s);
}

// The backward pass in gradient-completion order.  Segment boundaries are where a contiguous range of the flat gradient
// buffer becomes final, so a data-parallel caller can start that range's all-reduce while the rest still runs:
//   0: Linear + bi-LSTM   1: height compression (4 scales)   2: layer4   3: layer3   4: layer2, layer1, stem
// Launches outside [seg_lo, seg_hi] are skipped; the buffer rotation (pure pointer bookkeeping) always runs.
static int train_backward_impl(hn_engine* e, const float* dbon, const float* dcor, int B, void* workspace, size_t workspace_bytes,
                               float* grads, float p_rnn, float p_head, uint64_t seed, int seg_lo, int seg_hi, void* stream)
{
    HN_REQUIRE(e && dbon && dcor && workspace && grads, "hn_train_backward: null argument");
    HN_REQUIRE(0 <= seg_lo && seg_lo <= seg_hi && seg_hi < HN_GRAD_SEGMENTS, "hn_train_backward: bad segment range %d..%d", seg_lo, seg_hi);
    int seg = 0;
#define LIVE (seg >= seg_lo && seg <= seg_hi)
    const TrainPlan pl = make_train_plan(B);
    HN_REQUIRE(workspace_bytes >= pl.total * sizeof(float), "hn_train_backward: workspace too small");
    DeviceGuard guard(e->device);
    HN_REQUIRE(guard.ok, "hn_train_backward: cannot select device %d", e->device);
    std::vector<MJob> jobs;
    jobs.reserve(256);
    Ctx c{e, arch(), pl, reinterpret_cast<float*>(workspace), e->packed, grads, (hipStream_t)stream, B, &jobs};
    const Arch& a = c.a;
    float* W = c.W;
    hipStream_t s = c.s;
    int rc;
    const long rows = (long)T_COLS * B;
    float* G0 = W + pl.G[0];
    float* G1 = W + pl.G[1];
    float* G2 = W + pl.G[2];
    float* G3 = W + pl.G[3];
    double* ds = reinterpret_cast<double*>(W + pl.dstat) + pl.stat_lstm;      // LSTM bias-gradient sums (zeroed where used)
    if (seg_lo == 0) {    // the pass starts: every unit's [S1 | S2 | dbias] slot and every unit's weight-gradient scratch in one go each
        // debug instrument (hn_engine::poison): the ranges the adjoint treats as outputs / scratch -- gradient tensors, bf16 dz staging, d(C_s),
        // the two packed-weight scratches (the forward's Gram partials / LSTM staging are dead), the recurrent scratch, the gate buffer (the
        // adjoint overwrites it with dg; the saved gates live in pl.save), the weight-gradient scratches and the caller's flat gradient buffer
        if (e->poison >= 0) {
            for (int i = 0; i < 4; ++i) if ((rc = hn_poison(e, W + pl.G[i], pl.gmax * sizeof(float), s))) return rc;
            for (int i = 0; i < 2; ++i) if ((rc = hn_poison(e, W + pl.dzh[i], pl.gmax / 2 * sizeof(float), s))) return rc;
            if ((rc = hn_poison(e, W + pl.DC[0], (pl.lxch - pl.DC[0]) * sizeof(float), s))) return rc;
            if ((rc = hn_poison(e, W + pl.lxch, (pl.dstat - pl.lxch) * sizeof(float), s))) return rc;      // lxch, dhrec, dcrec, wsA, wsB
            if ((rc = hn_poison(e, W + pl.gx, (size_t)rows * 4096 * sizeof(float), s))) return rc;
            if ((rc = hn_poison(e, W + pl.dlin, ((size_t)rows * 12 + HN_HEAD_BWD_SCRATCH_FLOATS) * sizeof(float), s))) return rc;
            if ((rc = hn_poison(e, W + pl.wg_first, pl.wg_floats * sizeof(float), s))) return rc;
            if ((rc = hn_poison(e, reinterpret_cast<double*>(W + pl.dstat) + pl.stat_bwd_first, (pl.stat_bwd_doubles + 4096) * sizeof(double), s))) return rc;
            if ((rc = hn_poison(e, grads, arch().grad_floats * sizeof(float), s))) return rc;
        }
        HN_HIP(hipMemsetAsync(reinterpret_cast<double*>(W + pl.dstat) + pl.stat_bwd_first, 0, pl.stat_bwd_doubles * sizeof(double), s));
        HN_HIP(hipMemsetAsync(W + pl.wg_first, 0, pl.wg_floats * sizeof(float), s));
    }
    // the deferred small jobs of a segment (gradient un-packs, double -> float copies) run as ONE launch where the segment
    // ends, i.e. before its range of the flat gradient buffer is declared final
    auto flush = [&](int sg) -> int {
        if (jobs.empty()) return 0;
        const int r = e->jt_bwd[sg].run(jobs, s);
        jobs.clear();
        return r;
    };

    // ---- head: Linear + dropout ----
    if (LIVE && (rc = hn_launch_head_bwd(dbon, dcor, c.bound("linear.weight"), W + pl.y2d, G0, W + pl.dlin, c.grad("linear.weight"),
                                 c.grad("linear.bias"), T_COLS, B, s)))
        return rc;
    float* dy = G0;                                   // gradient w.r.t. the (dropped) LSTM output
    const float pdrop[2] = {p_rnn, p_head};
    const float* layer_in[2] = {W + pl.seq, W + pl.y1d};
    const float* layer_out[2] = {W + pl.y1, W + pl.y2};
    float* dx_buf[2] = {G2, G1};                      // layer 1 writes dx into G1, layer 0 into G2
    for (int l = 1; l >= 0; --l) {
        if (pdrop[l] > 0.f)
            if (LIVE && (rc = hn_launch_dropout(dy, dy, rows * 1024, pdrop[l], seed * 2 + 1 + l, s))) return rc;
        // recurrence adjoint, one time index per direction per step
        float* dgx = W + pl.gx;
        if (LIVE) HN_HIP(hipMemsetAsync(W + pl.dhrec, 0, (size_t)B * 1024 * sizeof(float), s));
        if (LIVE) HN_HIP(hipMemsetAsync(W + pl.dcrec, 0, (size_t)B * 1024 * sizeof(float), s));
        const std::string sf = "_l" + std::to_string(l), sr = sf + "_reverse";
        float* whhT = W + pl.wsB;                     // [2][512][2048]: recurrent weights, k-major, for the adjoint GEMV
        if (LIVE && (rc = hn_launch_transpose(c.bound("bi_rnn.weight_hh" + sf), whhT, 2048, 512, s))) return rc;
        if (LIVE && (rc = hn_launch_transpose(c.bound("bi_rnn.weight_hh" + sr), whhT + (size_t)512 * 2048, 2048, 512, s))) return rc;
        if (e->train_bf16) {
            // bf16 mode: ONE persistent launch per layer (W_hh^T as bf16 fragments in registers, bf16 dg hand-off)
            unsigned short* whhT_h = reinterpret_cast<unsigned short*>(whhT + (size_t)2 * 512 * 2048);
            if (LIVE && (rc = hn_launch_f32_to_bf16(whhT, whhT_h, (long)2 * 512 * 2048, s))) return rc;
            if (LIVE && (rc = hn_launch_lstm_layer_bwd_bf16(W + pl.save[l], dy, whhT_h, whhT_h + (size_t)512 * 2048, dgx, T_COLS, B,
                                                            W + pl.lxch, W + pl.sync, s)))
                return rc;
        } else
        for (int step = 0; step < T_COLS; ++step) {
            if (LIVE && (rc = hn_launch_lstm_bwd_gates(W + pl.save[l], dy, W + pl.dhrec, W + pl.dcrec, dgx, T_COLS, B, step, s))) return rc;
            const int tf = T_COLS - 1 - step, tr = step;
            if (LIVE && (rc = hn_launch_lstm_bwd_dh(dgx + (size_t)tf * B * 4096, dgx + (size_t)tr * B * 4096, whhT, whhT + (size_t)512 * 2048,
                                            W + pl.dhrec, B, s)))
                return rc;
        }
        // dW_hh = sum_t dg_t^T h_{t-1}   (fwd: rows m >= B pair with y rows m - B; rev: rows m < (T-1)B pair with y rows m + B)
        // dW_ih (both directions stacked) and the bias gradients;  dx = dg @ [W_ih_fwd; W_ih_rev]
        float* wsc = W + pl.wsA;
        float* wt = W + pl.wsB;                       // transpose of the stacked W_ih: GEMM weights [N=1024][K=4096]
        if (e->train_bf16) {
            // bf16 mode: the three weight-gradient GEMMs and the data-gradient GEMM on the bf16 matrix cores from bf16 copies of
            // dg, the layer input and the layer output (G3 is idle during the LSTM adjoint); float32 accumulation and results
            unsigned short* dg_h = reinterpret_cast<unsigned short*>(G3);                     // [rows][4096]
            unsigned short* in_h = dg_h + (size_t)rows * 4096;                                // [rows][1024]
            unsigned short* out_h = in_h + (size_t)rows * 1024;                               // [rows][1024]
            if (LIVE && (rc = hn_launch_f32_to_bf16(dgx, dg_h, rows * 4096, s))) return rc;
            if (LIVE && (rc = hn_launch_f32_to_bf16(layer_in[l], in_h, rows * 1024, s))) return rc;
            if (LIVE && (rc = hn_launch_f32_to_bf16(layer_out[l], out_h, rows * 1024, s))) return rc;
            if (LIVE && (rc = hn_launch_conv_wgrad_bf16(out_h, dg_h + (size_t)B * 4096, c.grad("bi_rnn.weight_hh" + sf), 1, 1, (int)(rows - B), 512,
                                                        2048, 1, 1, 1, 1, s, 0, 1024, 4096)))
                return rc;
            if (LIVE && (rc = hn_launch_conv_wgrad_bf16(out_h + (size_t)B * 1024 + 512, dg_h + 2048, c.grad("bi_rnn.weight_hh" + sr), 1, 1,
                                                        (int)(rows - B), 512, 2048, 1, 1, 1, 1, s, 0, 1024, 4096)))
                return rc;
            if (LIVE && (rc = hn_launch_conv_wgrad_bf16(in_h, dg_h, wsc, 1, 1, (int)rows, 1024, 4096, 1, 1, 1, 1, s))) return rc;
        } else {
            if (LIVE && (rc = hn_launch_conv_wgrad(layer_out[l], dgx + (size_t)B * 4096, c.grad("bi_rnn.weight_hh" + sf), 1, 1, (int)(rows - B), 512,
                                           2048, 1, 1, 1, 1, 1024, 4096, 0, s)))
                return rc;
            if (LIVE && (rc = hn_launch_conv_wgrad(layer_out[l] + (size_t)B * 1024 + 512, dgx + 2048, c.grad("bi_rnn.weight_hh" + sr), 1, 1,
                                           (int)(rows - B), 512, 2048, 1, 1, 1, 1, 1024, 4096, 0, s)))
                return rc;
            if (LIVE && (rc = hn_launch_conv_wgrad(layer_in[l], dgx, wsc, 1, 1, (int)rows, 1024, 4096, 1, 1, 1, 1, 1024, 4096, 0, s))) return rc;
        }
        if (LIVE) HN_HIP(hipMemcpyAsync(c.grad("bi_rnn.weight_ih" + sf), wsc, (size_t)2048 * 1024 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (LIVE) HN_HIP(hipMemcpyAsync(c.grad("bi_rnn.weight_ih" + sr), wsc + (size_t)2048 * 1024, (size_t)2048 * 1024 * sizeof(float),
                              hipMemcpyDeviceToDevice, s));
        if (LIVE) HN_HIP(hipMemsetAsync(ds, 0, 4096 * sizeof(double), s));
        if (LIVE && (rc = hn_launch_col_stats(dgx, ds, nullptr, rows, 4096, 4096, s))) return rc;
        if (LIVE && (rc = hn_launch_d2f(ds, c.grad("bi_rnn.bias_ih" + sf), 2048, s))) return rc;
        if (LIVE && (rc = hn_launch_d2f(ds, c.grad("bi_rnn.bias_hh" + sf), 2048, s))) return rc;
        if (LIVE && (rc = hn_launch_d2f(ds + 2048, c.grad("bi_rnn.bias_ih" + sr), 2048, s))) return rc;
        if (LIVE && (rc = hn_launch_d2f(ds + 2048, c.grad("bi_rnn.bias_hh" + sr), 2048, s))) return rc;
        if (LIVE && (rc = hn_launch_transpose(c.P + a.wih_off[l], wt, 4096, 1024, s))) return rc;
        if (e->train_bf16) {
            unsigned short* wt_h = reinterpret_cast<unsigned short*>(wt + (size_t)4096 * 1024);
            if (LIVE && (rc = hn_launch_f32_to_bf16(wt, wt_h, (long)4096 * 1024, s))) return rc;
            if (LIVE && (rc = gemm_rows_bf16(G3, wt_h, c.P + a.ones_off, c.P + a.zeros_off, dx_buf[l], rows, 4096, 1024, s))) return rc;
        } else if (LIVE && (rc = gemm_rows(dgx, 0, wt, c.P + a.ones_off, c.P + a.zeros_off, dx_buf[l], rows, 4096, 1024, s)))
            return rc;
        dy = dx_buf[l];
    }
    if ((rc = flush(seg))) return rc;
    seg = 1;
    float* dseq = dy;                                  // = G2: gradient w.r.t. the [256*B][1024] sequence matrix

    // ---- height compression, all four scales: leaves d(C_s) in DC[s] ----
    for (int li = 0; li < 4; ++li) {
        const Unit& ul = pl.units[pl.ghc[li] + 3];
        if (LIVE && (rc = hn_launch_upsample_flatten_bwd(dseq, G0, B, ul.Ho, ul.Wo, a.convs[ul.ci].cout, 256 * li, c.gh(), s))) return rc;
        float* gy = G0;
        float* gz = G1;
        float* gx = G3;
        for (int k = 3; k >= 0; --k) {
            const Unit& u = pl.units[pl.ghc[li] + k];
            if (LIVE && (rc = unit_backward(c, u, gy, W + u.y, gz, nullptr))) return rc;
            float* out = (k == 0) ? W + pl.DC[li] : gx;
            if (LIVE && (rc = unit_dgrad(c, u, gz, nullptr, out))) return rc;
            float* t = gy; gy = gx; gx = t;            // next unit's dy is what was just written
        }
    }

    // ---- backbone, layer4 -> layer1 ----
    if ((rc = flush(seg))) return rc;
    seg = 2;
    float* Ga = G0;
    float* Gb = G1;
    float* Gc = G2;
    float* Gd = G3;
    const size_t csz[4] = {(size_t)B * 128 * 256 * 256, (size_t)B * 64 * 128 * 512, (size_t)B * 32 * 64 * 1024, (size_t)B * 16 * 32 * 2048};
    const size_t gsz = c.gh() ? 2 : sizeof(float);      // bytes per gradient element between conv units
    if (LIVE) HN_HIP(hipMemcpyAsync(Ga, W + pl.DC[3], csz[3] * gsz, hipMemcpyDeviceToDevice, s));
    for (int li = 3; li >= 0; --li) {
        const int seg_li = li == 3 ? 2 : (li == 2 ? 3 : 4);
        if (seg_li != seg && (rc = flush(seg))) return rc;
        seg = seg_li;
        for (int j = kBlocks[li] - 1; j >= 0; --j) {
            const int i1 = pl.blk[li][j];
            const Unit& u1 = pl.units[i1];
            const Unit& u2 = pl.units[i1 + 1];
            const Unit& u3 = pl.units[i1 + (j == 0 ? 3 : 2)];
            // conv3 + bn3 + (add) + relu: mask = block output
            // BatchNorm-folded adjoint (bn_fold.hip): blocks 1.. of every stage, and block 0 where the downsample branch is stride 1 too (layer1)
            const bool fold3 = bn_fold_ok(c, u3) && (j > 0 || bn_fold_ok(c, pl.units[pl.dsu[li]]));
            HN_REQUIRE(fold3 || !fwd_folded(c, u3), "hn_train_backward: the forward ran %s through the BatchNorm-folded form (no z stored) but the adjoint "
                       "cannot (a debug tap or an option changed in between): set the option fuse_bn_fold = 0 BEFORE the forward", a.convs[u3.ci].wkey.c_str());
            HN_REQUIRE(fold3 || j > 0 || !fwd_folded(c, pl.units[pl.dsu[li]]), "hn_train_backward: the forward ran %s through the BatchNorm-folded form (no z "
                       "stored) but the adjoint cannot: set the option fuse_bn_fold = 0 BEFORE the forward", a.convs[pl.units[pl.dsu[li]].ci].wkey.c_str());
            if (fold3) {
                FoldOut f3, fd;
                const Unit* ud = j == 0 ? &pl.units[pl.dsu[li]] : nullptr;
                double* ds3 = reinterpret_cast<double*>(W + pl.dstat) + pl.stat_bwd_first + u3.sb;
                const unsigned char* mk3 = reinterpret_cast<const unsigned char*>(W + u3.mk);
                const bool fa3 = bn_fold_fuse_a(c, u3), fad = ud && bn_fold_fuse_a(c, *ud);
                // block 0 with both convs A fused (layer1): g has no other reader (no identity branch), so it is never written -- both P-GEMMs mask on load
                const bool keep_dy = ud && fa3 && fad;
                if (LIVE && (rc = bn_fold_unit(c, u3, Ga, mk3, nullptr, W + pl.wsA, &f3, fa3 ? Gb : nullptr, keep_dy ? 0 : 1))) return rc;   // Ga: dOut -> g (unless keep_dy)
                if (LIVE && ud && (rc = bn_fold_unit(c, *ud, Ga, keep_dy ? mk3 : nullptr, ds3, W + pl.wsB, &fd, fad ? Gc : nullptr, 0))) return rc;
                // conv B's epilogue takes conv2's reduce pass (its slab: behind the fold scratch of wsA, which needs < 32 MB)
                const bool red2 = bn_fold_reduce_on() && !c.bn_eval(u2) && (size_t)hn_cdiv(u3.M, 128) * 2 * a.convs[u2.ci].cout <= (size_t)1024 * 18432 - ((size_t)8 << 20);
                if (LIVE && (rc = bn_fold_dgrad(c, u3, f3, Ga, Gb, Gd, fa3 ? 1 : 0, red2 ? &u2 : nullptr, W + pl.wsA + ((size_t)8 << 20)))) return rc;   // d(t2) -> Gd
                if (LIVE && (rc = unit_backward(c, u2, Gd, W + u2.y, Gc, nullptr, 0, -1, nullptr, 0, red2 ? 1 : 0))) return rc;  // dz2 -> staging slot 0 (bf16 step: Gc is not written)
                if (LIVE && (rc = unit_dgrad(c, u2, Gc, nullptr, Gb))) return rc;               // d(t1) -> Gb
                if (LIVE && (rc = unit_backward(c, u1, Gb, W + u1.y, Gd, nullptr))) return rc;  // dz1 -> staging slot 0
                if (ud) {
                    // downsample path -> Gd   (layer1: no height-compression add); its fused conv A has been waiting in Gc (free until the next line's output)
                    HN_REQUIRE(keep_dy || !fad, "train_backward: a fused conv A of the downsample branch needs the unmasked dOut (keep_dy)");
                    if (LIVE && (rc = bn_fold_dgrad(c, *ud, fd, Ga, fad ? Gc : Gb, Gd, fad ? 1 : 0))) return rc;
                    HN_REQUIRE(li == 0, "train_backward: the folded downsample adjoint is layer1's");
                    if (LIVE && (rc = unit_dgrad(c, u1, Gd, Gd, Gc))) return rc;                // + conv1 path -> Gc
                } else {
                    if (LIVE && (rc = unit_dgrad(c, u1, Gd, Ga, Gc))) return rc;                // conv1 path + identity (g) -> Gc
                }
                float* t = Ga; Ga = Gc; Gc = t;
                continue;
            }
            const bool dual = j == 0 && bn_dual_ok(c, u3, pl.units[pl.dsu[li]]);
            if (LIVE && dual) {      // block 0: bn3 and the downsample BatchNorm in one reduce + one apply pass (slots 0 and 1)
                if ((rc = bn_backward_dual(c, u3, pl.units[pl.dsu[li]], Ga, 0, 1))) return rc;
                if ((rc = unit_backward(c, u3, Ga, W + u3.y, Gb, nullptr, 0, -1, nullptr, /*bn_done=*/1))) return rc;
            } else
            if (LIVE && (rc = unit_backward(c, u3, Ga, W + u3.y, Gb, Gc))) return rc;       // dz3 -> Gb, identity grad -> Gc
            if (LIVE && (rc = unit_dgrad(c, u3, Gb, nullptr, Gd))) return rc;               // d(t2) -> Gd
            if (LIVE && (rc = unit_backward(c, u2, Gd, W + u2.y, Ga, nullptr))) return rc;  // dz2 -> Ga
            if (LIVE && (rc = unit_dgrad(c, u2, Ga, nullptr, Gb))) return rc;               // d(t1) -> Gb
            if (LIVE && (rc = unit_backward(c, u1, Gb, W + u1.y, Gd, nullptr))) return rc;  // dz1 -> Gd
            if (j == 0) {
                const Unit& ud = pl.units[pl.dsu[li]];
                if (LIVE && dual) {
                    if ((rc = unit_backward(c, ud, Gc, nullptr, Ga, nullptr, 1, -1, nullptr, /*bn_done=*/1))) return rc;      // its dz has been in slot 1 since bn_backward_dual
                } else
                if (LIVE && (rc = unit_backward(c, ud, Gc, nullptr, Ga, nullptr, 1))) return rc;   // dz_ds -> Ga (bf16 slot 1: slot 0 still holds dz1)
                // (+ the height-compression branch's gradient of this block input, C_{li-1}: it rides as the data gradient's
                //  identity input instead of a separate read-modify-write pass over the largest gradient tensors)
                const float* dc_add = li > 0 ? W + pl.DC[li - 1] : nullptr;
                if (LIVE && (rc = unit_dgrad(c, ud, Ga, dc_add, Gb, 1))) return rc;              // ds path (+ d(C_{li-1})) -> Gb
                if (LIVE && (rc = unit_dgrad(c, u1, Gd, Gb, Gc))) return rc;                    // + conv1 path -> Gc
                float* t = Ga; Ga = Gc; Gc = t;
            } else {
                if (LIVE && (rc = unit_dgrad(c, u1, Gd, Gc, Ga))) return rc;                    // conv1 path + identity -> Ga
            }
        }
    }
    // ---- max-pool + stem ----
    {
        const Unit& u = pl.units[0];
        // bf16 mode: d(stem y) leaves the max-pool adjoint as bf16 and the stem's adjoint runs all-bf16 (unit_backward: stem_bf16);
        // with a debug tap on the stem (parity tests read float32 dy / dz) the float32 form is kept
        const bool stem_tapped = (e->debug_unit >= 0 && &u == &pl.units[e->debug_unit]) || (e->debug_unit2 >= 0 && &u == &pl.units[e->debug_unit2]);
        const int stem_h = (e->train_bf16 && c.gh() && !stem_tapped) ? 1 : 0;
        // ... and with batch statistics (no eval-mode BatchNorm: that path clears the sums between the two passes) the max-pool adjoint is
        // never materialised: the BatchNorm adjoint gathers it from the pooled gradient (HN_FUSE_STEM_POOLBWD=0: the separate pass)
        static const char* fpb = getenv("HN_FUSE_STEM_POOLBWD");
        const bool pool_fused = stem_h && !c.bn_eval(u) && e->fuse_stem_poolbwd && !(fpb && fpb[0] == '0');
        if (LIVE && pool_fused) {
            if ((rc = unit_backward(c, u, Gb, W + u.y, Gc, nullptr, -1, /*dy_bf16=*/1, /*pool_src=*/Ga))) return rc;
        } else {
        if (LIVE && (rc = hn_launch_maxpool_bwd_idx(W + pl.pidx, Ga, Gb, B, 256, 512, 64, c.gh(), s, stem_h))) return rc;     // d(stem y) -> Gb
        if (LIVE && (rc = unit_backward(c, u, Gb, W + u.y, Gc, nullptr, -1, /*dy_bf16=*/stem_h))) return rc;
        }
    }
    if ((rc = flush(seg))) return rc;
#undef LIVE
    return 0;
}

extern "C" int hn_train_backward(hn_engine* e, const float* dbon, const float* dcor, int B, void* workspace, size_t workspace_bytes,
                                 float* grads, float p_rnn, float p_head, uint64_t seed, void* stream)
{
    return train_backward_impl(e, dbon, dcor, B, workspace, workspace_bytes, grads, p_rnn, p_head, seed, 0, HN_GRAD_SEGMENTS - 1, stream);
}

extern "C" int hn_train_backward_segment(hn_engine* e, const float* dbon, const float* dcor, int B, void* workspace,
                                         size_t workspace_bytes, float* grads, float p_rnn, float p_head, uint64_t seed, int segment,
                                         void* stream)
{
    return train_backward_impl(e, dbon, dcor, B, workspace, workspace_bytes, grads, p_rnn, p_head, seed, segment, segment, stream);
}

extern "C" int hn_grad_segments(void) { return HN_GRAD_SEGMENTS; }

extern "C" int hn_grad_segment_range(int segment, int64_t* first, int64_t* count)
{
    HN_REQUIRE(segment >= 0 && segment < HN_GRAD_SEGMENTS && first && count, "hn_grad_segment_range: bad argument");
    const Arch& a = arch();
    const int64_t b_l3 = (int64_t)a.grad_off.at("feature_extractor.encoder.layer3.0.conv1.weight");
    const int64_t b_l4 = (int64_t)a.grad_off.at("feature_extractor.encoder.layer4.0.conv1.weight");
    const int64_t b_ghc = (int64_t)a.grad_off.at("reduce_height_module.ghc_lst.0.layer.0.layers.0.1.weight");
    const int64_t b_rnn = (int64_t)a.grad_off.at("bi_rnn.weight_ih_l0");
    const int64_t lo[HN_GRAD_SEGMENTS] = {b_rnn, b_ghc, b_l4, b_l3, 0};
    const int64_t hi[HN_GRAD_SEGMENTS] = {(int64_t)a.grad_floats, b_rnn, b_ghc, b_l4, b_l3};
    *first = lo[segment];
    *count = hi[segment] - lo[segment];
    return 0;
}
