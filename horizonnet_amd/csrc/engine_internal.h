// Internal declarations shared by engine.hip (eval forward) and train.hip (training step).
#pragma once
#include "hn_common.h"
#include "../../include/horizonnet_hip.h"
#include "multi_job.h"

#include <string>
#include <unordered_map>
#include <vector>

// ---- architecture table ------------------------------------------------------------------------

constexpr int T_COLS = 256;      // sequence length = 1024 / step_cols (reference model.py:194,259)
constexpr int IMG_H = 512, IMG_W = 1024;

struct ConvLayer {
    std::string wkey;            // "<...>.weight" (and ".bias" when has_bias)
    std::string bnkey;           // "<...>" BatchNorm prefix
    int cin, cout, k, has_bias;
    size_t w_off, scale_off, shift_off;   // float offsets into the packed buffer
};

struct Arch {
    std::vector<ConvLayer> convs;                 // order of the reference state_dict
    std::unordered_map<std::string, int64_t> numel;   // every bindable key -> element count
    // bottleneck index helpers
    int stem = 0;
    int block_first[4][6];                        // conv index of conv1 of layer li block j
    int block_down[4];                            // conv index of downsample of layer li (block 0)
    int ghc_first[4];                             // conv index of ghc_lst[s].layer[0]
    size_t wih_off[2], lbias_off[2], whh_off[2][2], ones_off, zeros_off, linw_off, linb_off;
    size_t packed_floats = 0;
    std::unordered_map<std::string, size_t> grad_off;   // parameter key -> float offset in the flat gradient buffer
    size_t grad_floats = 0;
};

size_t packed_w_floats(int cout, int cin, int k);
const Arch& arch();

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct ProfEntry {
    std::string name;
    double flops;                        // algorithmic FLOPs of the launch group (0 for pure data movement)
    hipEvent_t t0, t1;
};

struct hn_engine {
    int device = 0;
    std::unordered_map<std::string, const void*> bound;
    const float* packed = nullptr;       // last packed buffer (caller owned)
    const void* packed_h = nullptr;      // last bf16 weight buffer (hn_pack_weights_bf16, caller owned)
    std::vector<unsigned char> bn_eval;  // per conv layer: 1 = its BatchNorm is in eval() inside a train-mode step (hn_set_bn_eval):
                                         // normalise with the running statistics, leave them untouched (train.py:245-250 freezing)
    std::unordered_map<std::string, void*> taps;   // parity-test taps of hn_forward / hn_forward_bf16 (hn_set_forward_tap)
    int tap(const char* name, const void* src, size_t bytes, hipStream_t s) const
    {
        auto it = taps.find(name);
        if (it == taps.end() || it->second == nullptr) return 0;
        HN_HIP(hipMemcpyAsync(it->second, src, bytes, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    // branch stream: the four height-compression chains (model.py:138-156) depend only on C1..C4 and run beside the
    // following ResNet stages on an engine-owned second stream (fork / join with events)
    hipStream_t branch_stream = nullptr;
    hipEvent_t ev_fork[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool join_deferred[4] = {false, false, false, false};   // bf16 pipelined entry: ev_join[li] of the previous batch has not been waited for yet
    bool defer_join = false;                                 // option "defer_join"
    // head stream (hn_forward_bf16_submit): the recurrent head of batch i runs here, beside the trunk of batch i+1 on the
    // caller's stream; per slot one event "trunk finished" (recorded on the caller's stream) and one "head finished"
    hipStream_t head_stream = nullptr;
    hipEvent_t ev_trunk[2] = {nullptr, nullptr};
    hipEvent_t ev_head[2] = {nullptr, nullptr};
    bool head_pending[2] = {false, false};
    int wide_rows = 16;                  // panoramas per group of the wide recurrence kernel (hn_set_option "lstm_wide_rows": 8 | 16)
    int wide_xcds = 2;                   // XCDs a group's 8 workgroups are spread over ("lstm_wide_xcds": 1 | 2; 2 measured fastest: 4 CUs of every XCD)
    int fuse_downsample = 1;             // hn_forward_bf16: layer1 / layer2 block 0: downsample + conv3 in one launch
    int fuse_stem_conv1 = 1;             // ... with layer1.0.conv1 on the pooled rows while they are in LDS ("fuse_stem_conv1")
    int fuse_stem_bnpool = 1;            // bf16 training forward: the stem's BatchNorm + ReLU + max-pool in one pass ("fuse_stem_bnpool")
    int fuse_bn_dual = 1;                // bf16 training backward: block 0's bn3 and downsample BatchNorm adjoints in one reduce + one apply pass ("fuse_bn_dual")
    int fuse_bn_fold = 1;                // bf16 training backward: conv3 / stride-1 downsample units through the BatchNorm-folded adjoint (bn_fold.hip, "fuse_bn_fold"; 1 = forward and adjoint, 2 = adjoint only)
    // per training WORKSPACE (the record travels with the forward whose activations a backward reads: two forwards on two workspaces -- different batch
    // sizes, or options changed in between -- do not overwrite each other's), per training unit: 1 = that hn_train_forward ran the unit through
    // bn_fold_forward (no z stored)
    std::unordered_map<const void*, std::vector<unsigned char>> fold_fwd;
    int fold_deterministic = 1;          // bf16 training backward: the folded units' P-GEMM (g^T a, which feeds S2, dW AND the data-gradient weights) through the
                                         // reproducible slab reduce instead of float atomics in arrival order ("fold_deterministic"; HN_FOLD_SLAB=0: atomics, A/B)
    int fuse_stem_poolbwd = 1;           // bf16 training backward: the stem's BatchNorm adjoint gathers the max-pool adjoint itself ("fuse_stem_poolbwd")
    int fuse_stem_pool = 1;              // hn_forward_bf16: 7x7 conv + BN + ReLU + max-pool in one kernel ("fuse_stem_pool"; taps use the two-kernel form)
    int chain_layer1 = 1;                // hn_forward_bf16: layer1.1.conv3 (+ residual) chained into layer1.2.conv1 ("chain_layer1")
    int bf16_lstm = 1;                   // hn_forward_bf16: 1 = bf16 recurrence kernel (lstm_bf16.hip), 0 = the float32 one (lstm.hip)
    int use_branch_stream = 1;           // 0: everything on the caller's stream (hn_set_option "branch_stream")
    int f32_branch = 0;                  // float32 forward: 1 = the height-compression chains on the branch stream too (measured in round 6: no gain, DESIGN 6e.3;
                                         // option "f32_branch", env HN_F32_BRANCH=1)
    int train_bf16 = 0;                  // 1: train-mode convs (forward + data gradient) on the bf16 matrix cores (hn_set_train_precision)
    int poison = -1;                     // debug instrument (env HN_POISON_WS / option "poison_ws"): -1 off, else the byte every workspace / scratch /
                                         // packed-weight / gradient range is filled with BEFORE each engine entry writes it (0xFF = NaN in f32, bf16
                                         // and f64; 0x7F = 3.4e38): a result that changes with the pattern is a read of memory the entry never wrote
    bool profiling = false;
    int debug_unit = -1;                 // training debug tap (hn_train_debug_set): unit whose dy / dz are copied out
    float* debug_dy = nullptr;
    float* debug_dz = nullptr;
    int debug_unit2 = -1;                // second tap (hn_train_debug_set2): lets a test read two units of the SAME backward pass
    float* debug_dy2 = nullptr;
    float* debug_dz2 = nullptr;
    // engine-owned job tables of the batched small kernels (multi_job.h): weight packing (f32 / bf16 + data-gradient classes)
    // and, per gradient-completion segment of the backward pass, the deferred gradient un-packs / double -> float copies
    JobTable jt_pack, jt_pack_h, jt_bwd[HN_GRAD_SEGMENTS];
    std::vector<ProfEntry> prof;         // entries of the last profiled hn_forward
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    hipEvent_t next_event()
    {
        if (events_used == event_pool.size()) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return nullptr;
            event_pool.push_back(ev);
        }
        return event_pool[events_used++];
    }
};

// RAII bracket: records start/stop events around one launch group when profiling is on.
struct ProfScope {
    hn_engine* e;
    hipStream_t s;
    bool on;
    ProfScope(hn_engine* e_, hipStream_t s_, const std::string& name, double flops) : e(e_), s(s_), on(e_->profiling)
    {
        if (!on) return;
        ProfEntry pe{name, flops, e->next_event(), e->next_event()};
        e->prof.push_back(pe);
        (void)hipEventRecord(pe.t0, s);
    }
    ~ProfScope()
    {
        if (on) (void)hipEventRecord(e->prof.back().t1, s);
    }
};

// hn_engine::poison: fill [p, p + bytes) with the poison byte on stream s (no-op when the instrument is off)
inline int hn_poison(const hn_engine* e, void* p, size_t bytes, hipStream_t s)
{
    if (e->poison < 0 || p == nullptr || bytes == 0) return 0;
    HN_HIP(hipMemsetAsync(p, e->poison, bytes, s));
    return 0;
}
// an engine-owned stream with a hardware queue of its own (engine.hip)
int hn_make_stream(hipStream_t* out, bool high_priority);
// creates the engine's head stream + its events on first use (hn_forward_submit / hn_forward_bf16_submit)
int hn_ensure_head_stream(hn_engine* e);
// element offset of conv `ci`'s packed bf16 weights inside the hn_pack_weights_bf16 buffer (engine_bf16.hip)
size_t hn_bf16_conv_offset(int ci);
// ... and of its per-class data-gradient packing (classes in (row parity, column parity) order, as hn_launch_conv_dgrad_bf16
// walks them); (size_t)-1 for convs without a bf16 data gradient (the stem, Cout % 64 != 0)
size_t hn_bf16_dgrad_offset(int ci);
// ... of the LSTM weights in that buffer: W_ih of layer l ([4096][1024], both directions stacked), W_hh of (layer, direction)
size_t hn_bf16_wih_offset(int l);
size_t hn_bf16_whh_offset(int l, int d);
// strides of conv `ci` inside the network (train.hip's plan: the architecture fixes them)
void hn_conv_strides(int ci, int* sh, int* sw);
