/*
 * horizonnet_hip.h -- C ABI of the MI355X (gfx950) HorizonNet hot-path engine.
 *
 * The reference (sunset1995/HorizonNet) is pure Python and has no FFI of its
 * own; its hot path delegates to torch / torchvision / scipy native kernels.
 * Each entry point below names the reference call site whose native delegate
 * it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every data pointer is a DEVICE pointer owned by the caller (the PyTorch
 *     caching allocator in the Python host); nothing here allocates device
 *     memory for the caller's tensors.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work
 *     is enqueued asynchronously on it; nothing here synchronises the device
 *     except hn_check_status().
 *   - return value: 0 = ok, non-zero = error; hn_last_error() returns the
 *     message of the calling thread's last failure.  No exceptions cross the
 *     boundary.
 *   - re-entrant per (engine, stream); one engine per device
 *     (nn.DataParallel replicas, reference train.py:190-192, call from one
 *     Python thread per device).
 *   - arithmetic is float32 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 =
 *     exact fmaf chains); pano-stretch coordinates are float64 like the
 *     reference.
 */
#ifndef HORIZONNET_HIP_H
#define HORIZONNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hn_engine hn_engine;

/* ---- library ------------------------------------------------------------------------ */
const char* hn_last_error(void);
int hn_abi_version(void);

/* ---- engine: HorizonNet.forward, reference model.py:254-281 ------------------------- */

/* Create / destroy an engine bound to HIP device `device`. */
int hn_create(hn_engine** out, int device);
int hn_destroy(hn_engine* e);

/* Bind one tensor of the reference state_dict (reference misc/utils.py:49-65 defines the
 * checkpoint contract; the 448 keys are e.g. "feature_extractor.encoder.conv1.1.weight",
 * "reduce_height_module.ghc_lst.0.layer.0.layers.0.1.bias", "bi_rnn.weight_ih_l0_reverse",
 * "linear.weight").  `ptr` is a device pointer to contiguous float32 data (int64 for
 * num_batches_tracked, which is ignored), `numel` its element count (checked against the
 * architecture).  Unknown names and wrong sizes are errors. */
int hn_bind_tensor(hn_engine* e, const char* name, const void* ptr, int64_t numel);

/* Bytes the caller must provide for the packed (re-laid-out, BN-folded) weights. */
size_t hn_packed_bytes(void);

/* Pack every bound tensor into `packed` (device, hn_packed_bytes() bytes): conv weights
 * OIHW -> [Cout][kh][kw][Cin]; eval-mode BatchNorm (eps 1e-5) folded with the conv bias
 * into per-channel scale/shift (reference model.py:123-135 + torchvision Bottleneck);
 * bi-LSTM input weights of both directions stacked, bias_ih+bias_hh summed
 * (reference model.py:222-227).  Re-callable whenever the parameters change.
 * All 448 tensors (minus the 69 num_batches_tracked) must have been bound. */
int hn_pack_weights(hn_engine* e, void* packed, size_t packed_bytes, void* stream);

/* Workspace (activations) bytes for batch size B. */
size_t hn_workspace_bytes(int B);

/* Eval-mode forward.  x: [B, C_in>=3, 512, 1024] float32 NCHW in [0,1] (only the first 3
 * channels are read, reference model.py:252); bon: [B,2,1024], cor: [B,1,1024] float32.
 * Replaces the body of HorizonNet.forward (reference model.py:254-281) for
 * backbone='resnet50', use_rnn=True. */
int hn_forward(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor,
               void* workspace, size_t workspace_bytes, void* stream);

/* Pipelined form of hn_forward for back-to-back batches (a serving loop; inference.py:187-209 calls net(x) per panorama):
 * hn_forward_submit enqueues the convolutional trunk (model.py:73-81,123-179) on `stream` and the recurrent head (bi-LSTM +
 * Linear, model.py:263-269) on an engine-owned high-priority stream behind it, then returns; the head's recurrence kernel
 * occupies 64 of the 256 compute units for a batch of 32 (lstm_wide_f32.hip), so the NEXT submit's trunk runs beside it.
 * `slot` (0 | 1, alternate between consecutive submits) selects one of two head buffer sets in the workspace
 * (hn_workspace_pipelined_bytes).  bon / cor are valid on `stream` after hn_forward_collect(slot, stream) (a stream-side
 * wait, no host synchronisation).  Exact float32 like hn_forward; the recurrence sums over k in a different order, so the
 * outputs agree with hn_forward's to rounding (1e-6), not bit for bit.  hn_pipelined_status_offset_f32: byte offset of the
 * slot's sticky LSTM status word inside the workspace.  hn_lstm_layer_wide: that recurrence kernel alone (tests). */
size_t hn_workspace_pipelined_bytes(int B);
int hn_forward_submit(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace, size_t workspace_bytes,
                      int slot, void* stream);
int hn_forward_collect(hn_engine* e, int slot, void* stream);
int hn_pipelined_status_offset_f32(int B, int slot, size_t* byte_offset);
int hn_lstm_layer_wide(const float* gx, const float* whh_fwd, const float* whh_rev, float* y, int T, int B, void* sync_ws, void* stream);

/* Blocking: reads back the device-side status word of the last hn_forward / hn_lstm_layer
 * on this workspace (non-zero = the persistent LSTM kernel's bounded spin gave up). */
int hn_check_status(hn_engine* e, void* workspace, int* status_out);

/* Engine options.  "branch_stream" (default 1): hn_forward_bf16 runs the four height-compression chains
 * (model.py:138-156), which depend only on C1..C4, on an engine-owned second HIP stream beside the following ResNet
 * stages (fork / join with events: the caller's stream still orders the whole call); 0 = everything on the caller's
 * stream.  Results are identical either way.
 * "chain_layer1" (default 1): hn_forward_bf16 runs every layer1 block's conv3 (+ residual / downsample branch + ReLU) and the
 * NEXT block's conv1 (layer2.0.conv1 after the last block) as one launch (the 256-channel block output is not read back
 * from HBM; bit-identical).
 * "fuse_stem_pool" (default 1): hn_forward_bf16 runs (x - mean) / std, the 7x7/2 stem conv + BN + ReLU and the 3x3/2 max-pool
 * as ONE kernel (stem_pool_bf16.hip; bit-identical to the three-kernel form, which hn_set_forward_tap calls still use).
 * "fuse_stem_conv1" (default 1): that kernel also runs layer1.0.conv1 (1x1 + BN + ReLU) on every pooled half row while it is
 * still in LDS (bit-identical to the separate launch).
 * "defer_join" (default 0; measured +0.3 %): hn_forward_bf16_submit does not join the four chains back into the caller's stream at the end of the
 * trunk -- the recurrent head waits for them instead -- so the next batch's stem / layer1 run beside this batch's last chain
 * (every buffer a chain reads is protected by an event wait in front of its next writer; same kernels, same results).
 * "lstm_wide_rows" (16 | 8, default 16) / "lstm_wide_xcds" (1 | 2, default 2): geometry of the wide recurrence kernel of
 * hn_forward_bf16_submit (speed only).
 * "bf16_lstm" (default 1): hn_forward_bf16 runs the LSTM recurrence with bf16 W_hh / bf16 h_{t-1} on the matrix cores
 * (float32 accumulation, gates, cell state and outputs); 0 = the float32 recurrence kernel of hn_forward.
 * "fuse_downsample" (default 1): block 0 of a ResNet stage ends in ONE launch for
 * relu(bn3(conv3(t2)) + bn_d(downsample(x))) (model.py:78-81) instead of two (hn_forward: all four stages;
 * hn_forward_bf16: layer1 / layer2); bit-identical results.
 * "fuse_stem_bnpool" (default 1): the bf16 training forward runs the stem's BatchNorm + ReLU + max-pool (model.py:73-76) as ONE pass
 * over z (pooled tensor, position words and ReLU masks are the two-pass form's bit for bit).
 * "fuse_stem_poolbwd" (default 1): the bf16 training backward never materialises the max-pool adjoint: the stem's BatchNorm adjoint
 * gathers it from the pooled gradient (same dz bits).
 * "fuse_bn_dual" (default 1): bf16 training backward, block 0 of every ResNet stage: the BatchNorm adjoints of conv3 and of the
 * downsample branch (model.py:78-81: both see dOut * relu'(block output)) share one reduce and one apply pass. */
int hn_set_option(hn_engine* e, const char* name, int value);

/* Parity-test taps: during the following hn_forward / hn_forward_bf16 calls the named intermediate is copied
 * (device to device, on the call's stream) into `dst`.  Names and layouts (NHWC; float32 for hn_forward, bf16 for
 * hn_forward_bf16 except "lstm"): "stem" [B,256,512,64] (model.py:73-75), "pool" [B,128,256,64] (:76), "c1" [B,128,256,256],
 * "c2" [B,64,128,512], "c3" [B,32,64,1024], "c4" [B,16,32,2048] (:78-81), "feature" [256*B,1024] rows t*B+b (:259-262),
 * "lstm" [256*B,1024] float32 (:263-264).  dst == NULL removes the tap; name == NULL removes all. */
int hn_set_forward_tap(hn_engine* e, const char* name, void* dst);

/* Optional per-launch-group timing of hn_forward (HIP events on the caller's stream; used by
 * bench.py for the roofline numbers, never inside the timed region).  After a profiled
 * hn_forward: hn_profile_count() entries, each a name (the state_dict key of the conv, or a
 * stage label), its elapsed milliseconds and its algorithmic FLOPs. */
int hn_set_profiling(hn_engine* e, int on);
int hn_profile_count(hn_engine* e);
int hn_profile_entry(hn_engine* e, int i, char* name, int name_cap, float* ms, double* flops);

/* ---- bf16 inference mode (BASELINE configs 3-5; the reference runs net(x) under autocast, train.py:51,273) ----
 * NHWC bf16 activations, bf16 MFMA convolutions with f32 accumulation and f32 folded-BN epilogues; the LSTM
 * recurrence, its gate pre-activations and the Linear head stay f32.  Needs hn_pack_weights first. */
size_t hn_packed_bf16_bytes(void);
int hn_pack_weights_bf16(hn_engine* e, void* packed_bf16, size_t bytes, void* stream);
size_t hn_workspace_bf16_bytes(int B);
int hn_forward_bf16(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Pipelined form of hn_forward_bf16 for back-to-back batches (inference.py:187-209 calls net(x) per panorama; a serving
 * loop calls it per batch): hn_forward_bf16_submit enqueues the convolutional trunk (model.py:73-81,123-179) on `stream` and
 * the recurrent head (bi-LSTM + Linear, model.py:263-269) on an engine-owned high-priority stream behind it, then returns;
 * the head uses a recurrence kernel that occupies 32 of the 256 compute units for a batch of 32, so the NEXT submit's trunk
 * runs beside it.  `slot` (0 | 1, alternate between consecutive submits) selects one of two head buffer sets in the
 * workspace (hn_workspace_bf16_pipelined_bytes).  bon / cor are valid on `stream` after hn_forward_bf16_collect(slot, stream)
 * (a stream-side wait, no host synchronisation).  Same arithmetic as hn_forward_bf16: bit-identical outputs.
 * hn_pipelined_status_offset: byte offset of the slot's sticky LSTM status word inside the workspace. */
size_t hn_workspace_bf16_pipelined_bytes(int B);
int hn_forward_bf16_submit(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                           size_t workspace_bytes, int slot, void* stream);
int hn_forward_bf16_collect(hn_engine* e, int slot, void* stream);
int hn_pipelined_status_offset(int B, int slot, size_t* byte_offset);
/* per-stage (tests): the bf16 stem of hn_forward_bf16 on a [B][3][512][1024] float32 batch -- (x - mean) / std, 7x7/2
 * convolution (circular in W) + folded BatchNorm + ReLU, 3x3/2 max-pool (reference model.py:73-81 with torchvision's
 * resnet conv1 / bn1 / relu / maxpool and the LR padding of model.py:28-61) -> y [B][128][256][64] bf16 NHWC.
 * fused != 0: the one-kernel form the engine runs (option "fuse_stem_pool"); 0: implicit-GEMM stem + pool kernel
 * (needs stem_scratch [B][256][512][64] bf16).  x4_scratch: B*512*1024*4 bf16, w_scratch: 64*256 bf16.  Bit-identical. */
int hn_stem_pool_bf16(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift, void* x4_scratch,
                      void* w_scratch, void* stem_scratch, void* y, int B, int fused, void* stream);
/* per-stage (tests): x / res / y bf16 NHWC (y f32 when out_f32), w_oihw f32, w_scratch Cout*KH*KW*Cin bf16
 * (w_oihw == NULL: w_scratch already holds the packed weights of an earlier call) */
int hn_conv2d_nhwc_bf16(const void* x, const float* w_oihw, void* w_scratch, const float* scale, const float* shift,
                        const void* res, void* y, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int sh, int sw,
                        int relu, int out_f32, void* stream);
/* the same stage with a split-K workspace (float32, `splitk_ws_floats` elements): what the engine passes for the deep-K / few-tile
 * convs of layer4 and of the height-compression chains (model.py:123-135); the slice count depends on the per-image shape only */
int hn_conv2d_nhwc_bf16_ws(const void* x, const float* w_oihw, void* w_scratch, const float* scale, const float* shift,
                           const void* res, void* y, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int sh, int sw,
                           int relu, int out_f32, void* splitk_ws, size_t splitk_ws_floats, void* stream);

/* per-stage (tests): one bi-LSTM layer with the bf16 recurrence of hn_forward_bf16 (reference model.py:222-227,263-264).
 * gx [T*B][4096] float32 gate pre-activations (fwd gates | rev gates), whh_*_bf16 [2048][512] bf16, y [T*B][1024] float32,
 * y_bf16 optional bf16 copy of y, exchange: hn_lstm_bf16_exchange_bytes() of scratch, sync_ws: 4096 zeroed bytes (status
 * word at uint32 index 512). */
/* hn_lstm_layer_bf16_wide: the few-compute-unit form used by hn_forward_bf16_submit (a group = one direction of
 * rows_per_group = 16 | 8 panoramas, its 8 workgroups on xcds_per_group = 1 | 2 XCDs; <= 0: defaults); same arithmetic,
 * bit-identical y; y_bf16 is REQUIRED here: the bf16 output doubles as the kernel's exchange buffer (one slot per step,
 * pre-filled with a NaN sentinel by the call). */
int hn_lstm_layer_bf16_wide(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, void* y_bf16, int T, int B,
                            void* sync_ws, int rows_per_group, int xcds_per_group, void* stream);
size_t hn_lstm_bf16_exchange_bytes(void);
int hn_lstm_layer_bf16(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, void* y_bf16, int T, int B,
                       void* exchange, void* sync_ws, void* stream);

/* per-stage (tests): the same layer as the bf16 training step runs it (train_precision "bf16"; autograd of nn.LSTM under
 * reference train.py:273-278).  Forward: as hn_lstm_layer_bf16, also storing saved [T][B][2][5][512] float32 = post-activation
 * (i, f, g, o) and the cell state c per direction.  Adjoint: one persistent launch, dy [T*B][1024] float32 gradient of the
 * layer output, whhT_*_bf16 [512][2048] bf16 = W_hh transposed, dgx [T*B][4096] float32 gradient of the gate
 * pre-activations (out); exchange: hn_lstm_bwd_bf16_exchange_bytes() of scratch. */
int hn_lstm_layer_bf16_train(const float* gx, const void* whh_fwd_bf16, const void* whh_rev_bf16, float* y, float* saved, int T, int B,
                             void* exchange, void* sync_ws, void* stream);
size_t hn_lstm_bwd_bf16_exchange_bytes(void);
int hn_lstm_layer_bwd_bf16(const float* saved, const float* dy, const void* whhT_fwd_bf16, const void* whhT_rev_bf16, float* dgx, int T,
                           int B, void* exchange, void* sync_ws, void* stream);

/* ---- training step: autograd of net(x) at reference train.py:44-58,272-281 (float32) ---------- */

/* Workspace bytes for a training step at batch B (keeps every conv input / pre-BN / post-activation
 * tensor for the backward pass: about 1.5 GB per panorama). */
size_t hn_train_workspace_bytes(int B);

/* Train-mode forward: batch-statistics BatchNorm (running_mean / running_var of the bound tensors are
 * updated IN PLACE with `bn_momentum`, unbiased variance), dropout p_rnn between the LSTM layers and
 * p_head before the Linear (counter-based masks from `seed`; 0 disables).  Needs hn_pack_weights of the
 * current parameters.  num_batches_tracked is the caller's to increment. */
int hn_train_forward(hn_engine* e, const float* x, int B, int C_in, float* bon, float* cor, void* workspace,
                     size_t workspace_bytes, float p_rnn, float p_head, float bn_momentum, uint64_t seed, void* stream);

/* Per-BatchNorm mode inside a train-mode step.  Reference train.py:245-256 puts the frozen blocks'
 * modules in eval() every epoch (--freeze_earlier_blocks): such a BatchNorm normalises with its
 * RUNNING statistics and leaves them untouched, and its adjoint is the plain affine's.  `bn_prefix` is
 * the state_dict prefix of the BatchNorm ("feature_extractor.encoder.layer1.0.bn2"); eval != 0 selects
 * running statistics for hn_train_forward / hn_train_backward from now on (default 0 = batch
 * statistics).  Returns non-zero for an unknown prefix. */
int hn_set_bn_eval(hn_engine* e, const char* bn_prefix, int eval);

/* Backward of the last hn_train_forward on `workspace`: dbon [B,2,1024], dcor [B,1,1024] -> gradients of all
 * 241 parameters written (not accumulated) into the flat buffer `grads` (hn_grad_floats() floats; the
 * tensor named `key` starts at float offset hn_grad_offset(key), in its reference layout, e.g. OIHW). */
int hn_train_backward(hn_engine* e, const float* dbon, const float* dcor, int B, void* workspace, size_t workspace_bytes,
                      float* grads, float p_rnn, float p_head, uint64_t seed, void* stream);
size_t hn_grad_floats(void);
/* Train-step precision: bf16 = 1 runs the convolutions of hn_train_forward and their data gradients in
 * hn_train_backward on the bf16 matrix cores (bf16 operands, float32 accumulation): activations and dz get a bf16
 * copy written by the producing element-wise pass, weights come from hn_pack_weights_bf16 (call it after every
 * optimiser step).  BatchNorm, the weight gradients, the bi-LSTM, the head and the master weights stay float32 --
 * the mixed-precision recipe of the reference's autocast (train.py:51,273), with bf16 instead of fp16 (no GradScaler).
 * Default 0 = everything float32. */
int hn_set_train_precision(hn_engine* e, int bf16);

/* The same backward pass in 5 gradient-completion segments (0: Linear + bi-LSTM, 1: height
 * compression, 2: layer4, 3: layer3, 4: layer2 + layer1 + stem), to be called in order 0..4
 * with identical arguments.  After segment s returns (= is enqueued on `stream`) the range
 * hn_grad_segment_range(s) of `grads` is final, so a data-parallel caller can issue that
 * range's RCCL all-reduce while the later segments still run -- the "bucketed in reverse-layer
 * order, overlapped with backward" exchange of reference train.py:190-192,278-280's DataParallel
 * replacement. */
int hn_train_backward_segment(hn_engine* e, const float* dbon, const float* dcor, int B, void* workspace,
                              size_t workspace_bytes, float* grads, float p_rnn, float p_head,
                              uint64_t seed, int segment, void* stream);
int hn_grad_segments(void);
int hn_grad_segment_range(int segment, int64_t* first, int64_t* count);

/* One Adam step (torch.optim.Adam arithmetic, no amsgrad; train.py:216-225,279) for ALL parameter tensors in one launch.
 * params: DEVICE array of n_tensors float32 pointers (the parameter storages, updated in place); offsets / ends: DEVICE
 * arrays of the first / one-past-last element of every tensor inside the flat buffers (hn_grad_offset() layout, ascending;
 * the layout may leave alignment gaps between tensors); active:
 * DEVICE array of n_tensors flags (0 = frozen, skipped); grads: the flat gradient buffer of hn_train_backward; m, v: flat
 * first / second moments (total floats each, caller-owned state); step: 1-based step count; grad_scale: the loss scale
 * the gradients carry (1 if none). */
int hn_adam_step(float* const* params, const long long* offsets, const long long* ends, const unsigned char* active, int n_tensors,
                 const float* grads,
                 float* m, float* v, long long total, float lr, float beta1, float beta2, float eps, float weight_decay,
                 int step, float grad_scale, void* stream);

/* The training objective of reference train.py:53-56 in ONE launch: losses3 = {mean |bon - y_bon|, mean BCE-with-logits(cor, y_cor), their
 * sum} (device floats; `total`, optional, receives the sum once more in its own allocation) and the gradients of the two means w.r.t. bon / cor (dbon [n_bon], dcor [n_cor]); all pointers are device float32.
 * hn_scale2: a[i] *= *scale_dev, b[i] *= *scale_dev (the objective's incoming adjoint, read on the device). */
int hn_loss_l1_bce(const float* bon, const float* y_bon, long long n_bon, const float* cor, const float* y_cor, long long n_cor,
                   float* losses3, float* total, float* dbon, float* dcor, void* stream);
int hn_scale2(float* a, long long na, float* b, long long nb, const float* scale_dev, void* stream);

/* debug taps used by the parity tests (see train.hip) */
int hn_train_debug_unit(int B, int unit, int64_t* out8);
int64_t hn_train_debug_unit_yh(int B, int unit);   /* float offset of the unit's bf16 copy of y (bf16 mode) */
int hn_train_debug_set(hn_engine* e, int unit, float* dy_dst, float* dz_dst);
int hn_train_debug_set2(hn_engine* e, int unit, float* dy_dst, float* dz_dst);   /* a second unit of the same backward pass */
int64_t hn_grad_offset(const char* name);

/* ---- second boundary: misc/panostretch.py:81-102 (image half of pano_stretch) ------- */

/* Batched Pano-Stretch warp.  src/dst: [B][H][W][C] float32 (HWC per image, exactly the
 * numpy layout reference dataset.py:82 passes); kx, ky: HOST arrays of B doubles.
 * Coordinates are generated on the fly in float64 (reference misc/panostretch.py:91-96) and
 * sampled with scipy.ndimage.map_coordinates(order=1, mode='wrap') semantics (legacy wrap:
 * period len-1).  One launch for the whole batch. */
int hn_pano_stretch(const float* src, float* dst, const double* kx, const double* ky,
                    int B, int H, int W, int C, void* stream);

/* The same warp with the column and row terms supplied by the caller (DEVICE arrays of float64): col_tables [B][3][W] =
 * refx, sin(u0), sin(u) per image column (misc/panostretch.py:92,95), tan_v [H] = tan(v) per row (:17-24).  The Python host
 * computes them with numpy -- the reference's own arithmetic -- so every coordinate term except the per-pixel arctangent is
 * bit-identical to the reference's, including the case kx == ky, where refx of column 0 sits on SciPy's wrap discontinuity
 * (0 -+ 1e-13 decides between source column 0 and W-1).  tables_mirror_symmetric: the caller has verified
 * refx[W-1-x] / sin values mirror those of x exactly, which lets the symmetric kernel share one arctangent between four
 * pixels.  kx, ky: HOST arrays (ky enters the per-pixel term). */
int hn_pano_stretch_tables(const float* src, float* dst, const double* kx, const double* ky, const double* col_tables,
                           const double* tan_v, int tables_mirror_symmetric, int B, int H, int W, int C, void* stream);

/* ---- training-input pipeline: dataset.py:52,82,88-89,95-96,100-104,123 ---------------- */

/* One fused gather for a batch of B training inputs.  data: the dataset resident in HBM,
 * [n_images][H][W][3] uint8 (decoded RGB, HWC); index[b] picks the source image.  Per sample
 * (all HOST arrays of B entries, any may be NULL = augmentation off): kx, ky = Pano-Stretch
 * factors (dataset.py:70-82; the pair (1, 1) means "no stretch" and copies exactly); flip != 0
 * mirrors the columns (dataset.py:88); roll = dx of np.roll(img, dx, axis=1) (dataset.py:95);
 * gamma = exponent p of img ** p (dataset.py:100-104; float32 pow, <= 1 ulp from numpy's; 0.25 <= p <= 4).
 * dst: [B][3][H][W] float32 in [0,1] -- what dataset.py:123 returns per sample, stacked; the
 * layout hn_forward / hn_train_forward consume.  Order of operations = the reference's:
 * /255 -> stretch -> flip -> roll -> gamma -> CHW. */
int hn_augment_batch(const unsigned char* data, int n_images, const int* index, float* dst,
                     const double* kx, const double* ky, const int* flip, const int* roll,
                     const double* gamma, int B, int H, int W, void* stream);

/* Training labels on the device: the per-COLUMN half of dataset.py:108-120 (cor_2_1d),
 * dataset.py:137-169 (one row per column, np.interp at the integer columns) and
 * misc/panostretch.py:51-78 (pano_connect_points), float64 in the reference's operation order.
 * rec (device): B records of rec_floats float32 written by horizonnet_amd/labels.py
 * device_label_record -- [0] wall edges of the ceiling boundary, [1] of the floor boundary,
 * [2] visible wall-wall corners, [3] flip, [4] roll, [5..7] 0; edge[2][max_seg][8] =
 * {kind, x1, y1, dx, dy, first column, columns, z} (the per-corner scalars of
 * panostretch.py:58-70 in the reference's float32 flow; kind 1: both corners on one column);
 * corner_x[max_cor] after flip / roll (dataset.py:88-97).
 * bon: [B][2][W] float32 latitudes (dataset.py:84 after flip / roll), y_cor: [B][W] float32 =
 * p_base ** circular distance to the nearest visible corner (dataset.py:113-118).
 * status (device int32 [B]): 1 = a column of that panorama is covered by no edge (np.interp
 * would interpolate there: rasterise it on the host).  Measured <= 1 float32 ulp from the
 * reference flow (device tan / atan2 / pow against libm's). */
int hn_labels_rasterise(const float* rec, int rec_floats, int max_seg, int max_cor, int B, int H,
                        int W, double p_base, float* bon, float* y_cor, int* status, void* stream);

/* ---- corner-index extraction: inference.py:21-29 + :80 ------------------------------ */

/* For each of B signals of length n (float32): optional sigmoid (apply_sigmoid != 0, as
 * inference.py:80), periodic maximum filter of window r covering [i-r/2, i-r/2+r-1]
 * (scipy maximum_filter mode='wrap'), peak mask[i] = (max==signal[i]) && signal[i] > min_v.
 * mask: [B][n] uint8; prob: [B][n] float32 (the post-sigmoid signal) or NULL. */
int hn_find_peaks(const float* signal, int B, int n, int r, float min_v, int apply_sigmoid,
                  uint8_t* mask, float* prob, void* stream);

/* ---- host step of the Manhattan fit: misc/post_proc.py:75-98 (`vote`) ------------------- */

/* Host function (no device work): the decision half of the Manhattan fit for a BATCH of panoramas on a thread pool inside
 * the library -- misc/post_proc.py gen_ww (:337-359) with gen_ww_general (:243-334) / gen_ww_cuboid (:205-240) / vote (:75-98),
 * and the validity test + cuboid fallback of inference.py:113-126.  All pointers are HOST pointers.
 *   xs, ys      [B][W] float64: floor-plan coordinates of the ceiling boundary per image column = np_coor2xy of (column,
 *               ceiling row) (post_proc.py:29-43), computed by the caller WITH NUMPY: every transcendental value stays numpy's,
 *               this function performs only IEEE +, -, *, /, comparisons, sorting and numpy's pairwise summation, so its
 *               results are bit-identical to the reference's
 *   peak_mask   [B][W] uint8: the corner columns the fit starts from (hn_find_peaks; for the cuboid fits the caller has
 *               already reduced them to the 4 most probable, inference.py:25-28 -- ties there follow numpy's argsort)
 *   sin_u,cos_u [W+1] float64: np.sin / np.cos of np_coorx2u(column); entry W: of u = -1 (the reference's default for
 *               inferred walls)
 *   tol         [B] float64 (inference.py:111: abs(0.16 * z1 / 1.6))
 *   pts         [B][HN_FIT_MAX_CORNERS][2] float64 out: floor-plan corner points, in wall order (the input of np_xy2coor)
 *   npts, flags [B] int32 out: corner count; 0 = layout fitted, 1 = the general layout is not a valid polygon (the caller
 *               re-fits that panorama as a cuboid: force_cuboid with its 4 most probable peaks at threshold 0), 2 = one of the
 *               reference's assertions would have fired (no layout)
 * W must be 1024 (the reference's floor-plan constants). */
#define HN_FIT_MAX_CORNERS 64
int hn_layout_fit_batch(const double* xs, const double* ys, const unsigned char* peak_mask, const double* sin_u, const double* cos_u,
                        const double* tol, int B, int W, int force_cuboid, int threads, double* pts, int32_t* npts, int32_t* flags);

/* Host function: mean_percentile (post_proc.py:69-72) for B float32 rows -- out[b] = numpy's vec[(lo <= vec) & (vec <= hi)].mean()
 * (compaction in order, numpy's float32 pairwise summation, one float32 division). */
int hn_interquartile_mean_f32(const float* z, const float* lo, const float* hi, int B, int W, float* out);

/* HOST function (no device work, no stream): the decision loop of the reference's `vote` on an ascending float64 sample
 * vector -- the longest run v[i..j] with (v[j] - v[i]) + 1e-9 <= tol that covers at least 40 % of the samples, first maximum in
 * (i, j) row-major order as post_proc.py:78-90's N x N span matrix finds it.  best3 = {span, i, j} (span = -1: none).  The
 * means / medians around it stay in numpy (horizonnet_amd/postproc.py) so that every rounding is the reference's; the loop is
 * what took 2/3 of the host time of inference.py:89-141 per panorama. */
int hn_vote_scan(const double* sorted_samples, int L, double tol, int32_t* best3);

/* ---- per-stage entry points (used by the parity tests; the engine calls the same code) */

/* OIHW -> packed [Cout][kh][kw][Cin] (stem 7x7: [64][7][8][4] zero padded). */
int hn_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW,
                        void* stream);
size_t hn_packed_conv_weight_floats(int Cout, int Cin, int KH, int KW);

/* scale = gamma/sqrt(var+1e-5), shift = (bias-mean)*scale+beta.  Any of gamma/beta/mean/var
 * may be NULL together (no BN: scale=1, shift=bias); bias may be NULL. */
int hn_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
               const float* bias, float* scale, float* shift, int C, void* stream);

/* Fused conv + scale/shift (+residual) (+ReLU), NHWC float32, zero padding in H, CIRCULAR
 * padding in W (reference model.py:27-55).  x: [B][Hi][Wi][Cin]; w: packed; y/res: [B][Ho][Wo][Cout].
 * KHxKW in {1x1, 3x3}; Cin % 32 == 0; Cout % 32 == 0. */
int hn_conv2d_nhwc(const float* x, const float* w_packed, const float* scale, const float* shift,
                   const float* res, float* y, int B, int Hi, int Wi, int Cin, int Cout,
                   int KH, int KW, int sh, int sw, int relu, void* stream);

/* Stem: x NCHW [B][C_in][512][1024] -> normalise -> 7x7/2 conv (3->64) + scale/shift + ReLU ->
 * 3x3/2 max-pool (ordinary padding).  tmp_nhwc4: [B][H][W][4]; stem_out: [B][H/2][W/2][64];
 * pool_out: [B][H/4][W/4][64].  (reference model.py:248-252,73-76) */
int hn_stem(const float* x_nchw, int B, int C_in, int H, int W, const float* w_packed,
            const float* scale, const float* shift, float* tmp_nhwc4, float* stem_out,
            float* pool_out, void* stream);

/* The same stem as ONE kernel (csrc/stem_pool_f32.hip; what hn_forward runs unless a parity tap asks for the un-pooled stem activation):
 * x NCHW [B][C_in >= 3][512][1024] float32 -> pool_out [B][128][256][64] float32.  w_packed: the stem matrix hn_pack_conv_weight(64, 3, 7, 7)
 * produces, scale / shift: folded BatchNorm [64].  Agrees with hn_stem to float32 rounding (a different summation order over the 147 taps).
 * (reference model.py:248-252,73-76) */
int hn_stem_pool_f32(const float* x_nchw, int B, int C_in, const float* w_packed, const float* scale, const float* shift, float* pool_out,
                     void* stream);

/* Circular linear up-sampling along W to 256 columns + (c,h) flatten into the sequence
 * matrix (reference model.py:151-155,175-178).  in: [B][hq][Wq][cq]; seq: [256*B][1024] with
 * row = t*B + b and column = col0 + c*hq + h. */
int hn_upsample_flatten(const float* in, float* seq, int B, int hq, int Wq, int cq, int col0,
                        void* stream);

/* One bidirectional LSTM layer, hidden 512 (reference model.py:222-227,263-264).
 * gx: [T*B][4096] = x @ [W_ih_fwd; W_ih_rev]^T + (b_ih + b_hh) (columns dir*2048 + gate*512 + unit,
 * gates i,f,g,o); whh_fwd / whh_rev: [2048][512]; y: [T*B][1024] (fwd | rev).
 * sync_ws: 4096 bytes of device scratch (arrival counters; uint32 word 512 = sticky status, 0 = ok). */
int hn_lstm_layer(const float* gx, const float* whh_fwd, const float* whh_rev, float* y,
                  int T, int B, void* sync_ws, void* stream);

/* Linear(1024,12) + the (seq,step)->column interleave (reference model.py:266-269,278-279).
 * y: [T*B][1024]; bon: [B][2][4T]; cor: [B][1][4T]. */
int hn_linear_head(const float* y, const float* w, const float* bias, float* bon, float* cor,
                   int T, int B, void* stream);

/* Data gradient of hn_conv2d_nhwc (adjoint w.r.t. x): dz [B][Ho][Wo][Cout] -> dx [B][Hx][Wx][Cin] (+ add).
 * w_oihw: the ORIGINAL OIHW weights; w_scratch: Cout*Cin*KH*KW + 8192 floats of device scratch. */
int hn_conv2d_dgrad_nhwc(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B,
                         int Hx, int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream);

/* The same data gradient on the bf16 matrix cores (train_precision "bf16"): dz is rounded to bf16, the weights are
 * re-packed per stride-parity class as bf16, accumulation / add / dx stay float32.  Cout %% 64 == 0.
 * w_scratch: Cout*Cin*KH*KW + 8192 floats followed by B*Ho*Wo*Cout/2 floats (the bf16 copy of dz). */
int hn_conv2d_dgrad_nhwc_bf16(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B,
                              int Hx, int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream);

/* The data gradient as the bf16 training step runs it: dz, add and dx are bf16 tensors inside (rounded from / converted to the
 * float32 host-facing buffers here); stride-1 convs go through the forward kernels with flipped taps, strided ones through the
 * per-class data-gradient kernels (environment HN_DGRAD_W8=0: 128x128 tiles only).  w_scratch: Cout*Cin*KH*KW + 8192 floats,
 * then B*Ho*Wo*Cout/2 + 64, then 2 * (B*Hx*Wx*Cin/2 + 64) floats. */
int hn_conv2d_dgrad_nhwc_bf16g(const float* dz, const float* w_oihw, const float* add, float* dx, float* w_scratch, int B,
                               int Hx, int Wx, int Cin, int Cout, int KH, int KW, int sh, int sw, void* stream);

/* The weight gradient on the bf16 matrix cores: x and dz are rounded to bf16, products exact, accumulation and dw float32.
 * Cin %% 64 == 0, Cout %% 64 == 0.  scratch: Cout*KH*KW*Cin floats followed by (B*Hi*Wi*Cin + B*Ho*Wo*Cout) / 2 + 128 floats.
 * KH == KW == 7: the stem (Cin 3, Cout 64, stride 2), x = the NHWC4 input [B][Hi][Wi][4]; scratch 64*256 + (B*Hi*Wi*4 + B*Hi*Wi*16) / 2 + 128. */
int hn_conv2d_wgrad_nhwc_bf16(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int Hi, int Wi, int Cin,
                              int Cout, int KH, int KW, int sh, int sw, void* stream);

/* Weight gradient of hn_conv2d_nhwc / the stem conv: dw_oihw [Cout][Cin][KH][KW] = sum_m dz[m][n] * patch(x)[m][k].
 * stem != 0: x is the NHWC4 normalised image and the conv is the 7x7/2 stem.  scratch: Cout*max(KH*KW*Cin, 256) floats. */
int hn_conv2d_wgrad_nhwc(const float* x, const float* dz, float* dw_oihw, float* scratch, int B, int Hi, int Wi, int Cin,
                         int Cout, int KH, int KW, int sh, int sw, int stem, void* stream);

/* Box characterisation for the bench (no reference counterpart; SURVEY.md 8(d) "re-derive from the box ... a measured MFMA micro-benchmark;
 * report both"): launches a dense-MFMA rate kernel (dtype 0 = float32 v_mfma_f32_32x32x2_f32, 1 = bf16 v_mfma_f32_32x32x16_bf16) on
 * `workgroups` x 4 waves, `iters` rounds of 4 independent accumulator chains; *flop_out = the FLOP of the launch (time it with events on
 * `stream`).  scratch: workgroups * 256 floats. */
int hn_probe_mfma(int dtype, int workgroups, int iters, float* scratch, double* flop_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
