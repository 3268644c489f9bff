"""bench.py -- HorizonNet hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: re-executes itself under torch.distributed.run,
                                                          or is launched that way by the driver)
    python bench.py --mode train --dtype bf16 --batch 64  BASELINE configs[2] (configs[3] with --gpus N)
    python bench.py --mode layout                         BASELINE configs[4]: 1000 panoramas, 3D IoU vs the reference

One "step" = one forward of the hot path (ResNet-50 column-feature extractor + height compression +
bi-LSTM + head) over a batch of synthetic 512x1024 panoramas already resident in HBM.
Workload = BASELINE.json configs[1]: batch 32 random panoramas, forward only, fp32, per GPU.
Panoramas are independent units: each rank runs its own replica on its own batch (weak scaling,
no data-path collective); RCCL is used only for the barrier and the max-over-ranks timing.

Prints ONE JSON line (rank 0) with `roofline` (whole forward vs the fp32 MFMA roof, HIP events on the
launch stream over the timed region; per-kernel-family breakdown from a separate profiled pass) and
`cpu_baseline` (the oracle restatement of the reference forward timed on the host cores).  At --gpus 1 the same line also
carries the other BASELINE configs as extra keys (`--legs` selects them): `bf16_mode` (plain and pipelined bf16 forward,
against the bf16 MFMA peak AND the per-layer mixed roofline), `latency_b1` (configs[0]: demo.png through inference()),
`train_bf16` (configs[2]: with a MIXED compute + HBM roofline of the whole step, `train_mixed_roofline`, and the HBM-side bytes of
the committed counter passes), `layout` (configs[4]: 1000 panoramas, 3D IoU vs the reference's own inference()),
`pano_stretch`, `augment_pipeline`.  `roofline.traffic` / `train_bf16.roofline.traffic` come from the newest profiles/rN_pmc_*.json and
carry `traffic_stale` = (the kernel sources those counters were taken on != the sources this run was built from).
`--mode train --gpus N` adds `allreduce_overlap` (step with / without the exchange, a traced step with every all-reduce bucket's
start / end against the backward's kernels) and `--allreduce-dtype bf16` sends the gradients as bf16.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_FLOP_PER_PANO = 2.0 * 71_448_920_064          # BASELINE.md section 3 / SURVEY.md section 8(d)
PEAK_F32_MFMA_TFLOPS = 157.3                      # MI355X_MICROARCH.md: FP32 matrix peak (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0                    # MI355X_MICROARCH.md: BF16 dense MFMA peak (spec)


_STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))


def conv_table():
    """Every convolution of the hot path as (name, Cin, Cout, k, Hin, Win, stride_h, stride_w) -- SURVEY.md appendix A."""
    rows = [("stem", 3, 64, 7, 512, 1024, 2, 2)]
    H, W, cin = 128, 256, 64
    for li, (p, n) in enumerate(_STAGES):
        for j in range(n):
            s_ = 2 if (j == 0 and li > 0) else 1
            rows.append(("layer%d.%d.conv1" % (li + 1, j), cin, p, 1, H, W, 1, 1))
            rows.append(("layer%d.%d.conv2" % (li + 1, j), p, p, 3, H, W, s_, s_))
            rows.append(("layer%d.%d.conv3" % (li + 1, j), p, 4 * p, 1, H // s_, W // s_, 1, 1))
            if j == 0:
                rows.append(("layer%d.0.downsample" % (li + 1), cin, 4 * p, 1, H, W, s_, s_))
            cin, H, W = 4 * p, H // s_, W // s_
        ch, gh = [4 * p, 2 * p, 2 * p, p, p // 2], H
        for k in range(4):
            rows.append(("ghc%d.%d" % (li, k), ch[k], ch[k + 1], 3, gh, W, 2, 1))
            gh //= 2
    return rows


def f32_mixed_roofline(B, peak_flops=PEAK_F32_MFMA_TFLOPS * 1e12, hbm=8e12):
    """The same per-layer max(flops / MFMA peak, unfused bytes / HBM) sum for the float32 forward (4-byte activations and weights): the
    layer1 1x1 convs are HBM-bound even in float32 (1.34 GB per launch at B = 32), so the pure-MFMA bound the headline divides by is a
    little generous; printed beside it.  -> (seconds per batch, flops, bytes, seconds of the HBM-bound layers)."""
    t = fl = by = t_hbm = 0.0
    for _, cin, cout, k, H, W, sh, sw in conv_table():
        Ho, Wo = H // sh, W // sw
        f = 2.0 * Ho * Wo * cout * cin * k * k * B
        b = (H * W * cin + Ho * Wo * cout) * 4.0 * B + cout * cin * k * k * 4.0
        t += max(f / peak_flops, b / hbm)
        t_hbm += b / hbm if b / hbm > f / peak_flops else 0.0
        fl += f
        by += b
    rows = 256.0 * B
    for _ in range(2):
        f = 2.0 * rows * 1024 * 4096 + 2.0 * rows * 512 * 2048 * 2
        t += f / peak_flops
        fl += f
        by += rows * (1024 + 4096) * 4 + 4096 * 1024 * 4
    return t, fl, by, t_hbm


def bf16_mixed_roofline(B, peak_flops=PEAK_BF16_MFMA_TFLOPS * 1e12, hbm=8e12):
    """BASELINE.md section 3 / SURVEY.md 8(d): the bf16 forward's governing roofline is MIXED -- per layer
    max(flops / MFMA peak, unfused bytes / HBM bandwidth), summed (stem / layer1 / layer2 are HBM-bound in bf16).
    Bytes per conv = (input + output) x 2 B x batch + weights x 2 B; the LSTM's input GEMMs likewise (f32 gate
    pre-activations), its recurrence at the MFMA peak.  -> (seconds per batch, flops, bytes, seconds of the HBM-bound layers)."""
    t = fl = by = t_hbm = 0.0
    for _, cin, cout, k, H, W, sh, sw in conv_table():
        Ho, Wo = H // sh, W // sw
        f = 2.0 * Ho * Wo * cout * cin * k * k * B
        b = (H * W * cin + Ho * Wo * cout) * 2.0 * B + cout * cin * k * k * 2.0
        t += max(f / peak_flops, b / hbm)
        t_hbm += b / hbm if b / hbm > f / peak_flops else 0.0
        fl += f
        by += b
    rows = 256.0 * B
    for _ in range(2):
        f = 2.0 * rows * 1024 * 4096
        b = rows * 1024 * 2 + rows * 4096 * 4 + 4096 * 1024 * 2
        t += max(f / peak_flops, b / hbm)
        fl += f
        by += b
        f = 2.0 * rows * 512 * 2048 * 2
        t += f / peak_flops
        fl += f
    return t, fl, by, t_hbm


def train_mixed_roofline(B, peak_flops=PEAK_BF16_MFMA_TFLOPS * 1e12, hbm=8e12, fused_minimum=False):
    """The bf16 TRAINING step's governing roofline, the forward's model extended to the passes a training step cannot avoid
    (train.py:259-281 on model.py:216-281): per conv three GEMMs -- forward, data gradient (not for the stem: the image needs none),
    weight gradient -- each max(flops / MFMA peak, bytes / HBM) with bytes = the two activation-sized operands x 2 B + the weights
    (bf16 read, float32 gradient write); per BatchNorm the passes batch statistics force: forward normalise + ReLU (read z, write a;
    + the identity branch for conv3), backward reduce (read dy, z) and apply (read dy, z, write dz) -- 7 (8) activation-sized
    transfers of 2 B, HBM-bound; max-pool, the LSTM at 3 x its forward flops, Adam at 28 B per parameter.
    fused_minimum: the harder bound of a design that fuses every BatchNorm pass into its neighbours: per BatchNorm only
    "read z once forward, read dy + z once backward" = 3 activation-sized transfers instead of 7 / 8 -- what normalise-on-load in the
    consuming conv and dz-on-load in both gradient GEMMs would leave (the engine's BatchNorm-folded 1x1 units, bn_fold.hip, go below
    even that for conv3: they never store z).
    -> (seconds per batch, flops, bytes, seconds of the HBM-bound passes)."""
    t = fl = by = t_hbm = 0.0
    for name, cin, cout, k, H, W, sh, sw in conv_table():
        Ho, Wo = H // sh, W // sw
        f = 2.0 * Ho * Wo * cout * cin * k * k * B
        a_in, a_out, w = H * W * cin * 2.0 * B, Ho * Wo * cout * 2.0 * B, cout * cin * k * k
        gemms = [a_in + a_out + 2.0 * w, a_in + a_out + 4.0 * w] + ([] if name == "stem" else [a_in + a_out + 2.0 * w])   # forward, wgrad, dgrad
        for b in gemms:
            t += max(f / peak_flops, b / hbm)
            t_hbm += b / hbm if b / hbm > f / peak_flops else 0.0
            fl += f
            by += b
        bn = a_out * (3.0 if fused_minimum else (8.0 if name.endswith("conv3") else 7.0))
        t += bn / hbm
        t_hbm += bn / hbm
        by += bn
    pool = 64 * 256 * 512 * 2.0 * B * 1.25 * 2.0           # stem max-pool: read + write a quarter, forward and backward
    t += pool / hbm
    by += pool
    rows = 256.0 * B
    for _ in range(2):
        f = 3.0 * (2.0 * rows * 1024 * 4096 + 2.0 * rows * 512 * 2048 * 2)
        t += f / peak_flops
        fl += f
    adam = 81.57e6 * 28.0
    t += adam / hbm
    by += adam
    return t, fl, by, t_hbm


def shard_for_rank(global_units, world, rank):
    """Contiguous [start, end) slice of `global_units` independent panoramas owned by `rank`."""
    base, rem = divmod(global_units, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def cpu_baseline(seconds_budget=12.0):
    """Oracle (CPU restatement of the reference forward, torch fp32 on the host cores), bounded sample."""
    from oracle import horizonnet_ref
    from oracle.weights import make_state_dict
    from oracle.hostinfo import usable_cores
    cores = usable_cores()                           # cgroup-aware (the GPU box shows 256 CPUs, grants 16)
    torch.set_num_threads(cores)
    sd = make_state_dict(0, "random")
    x = torch.rand(4, 3, 512, 1024, generator=torch.Generator().manual_seed(0))
    horizonnet_ref.forward(x[:1], sd)                  # warm-up
    by_batch, total_n, total_s = {}, 0, 0.0
    for b in (1, 4):                                   # SURVEY 8d / BASELINE.md section 4: B = 1 and B = 4, >= 3 timed iterations each
        n, t0 = 0, time.perf_counter()
        while True:
            horizonnet_ref.forward(x[:b], sd)
            n += b
            el = time.perf_counter() - t0
            if (el > seconds_budget / 2 and n >= 3 * b) or el > 1.5 * seconds_budget:
                break
        by_batch[b] = {"panoramas_per_s": round(n / el, 3), "iterations": n // b, "seconds": round(el, 2)}
        total_n += n
        total_s += el
    best = max(v["panoramas_per_s"] for v in by_batch.values())
    return {"value": best, "unit": "panoramas/s", "cores": cores, "kind": "port", "by_batch": by_batch,
            "sample": "%d panoramas (B = 1: %d iterations, B = 4: %d iterations) through oracle.horizonnet_ref.forward, torch %s fp32, %d threads, %.1f s; "
                      "value = the better of the two batch sizes" % (total_n, by_batch[1]["iterations"], by_batch[4]["iterations"], torch.__version__, cores, total_s),
            "note": "port = the oracle's restatement of the reference forward (same torch conv / LSTM kernels, pinned to the unmodified "
                    "reference at 2.4e-7); /root/reference is absent on the GPU box.  The UNMODIFIED reference model.py measured "
                    "2.2 panoramas/s (B = 1) / 1.6 (B = 4) on the build container's 8 vCPUs (SURVEY.md section 6)"}


def pano_stretch_leg(dev, n_img=64, iters=10):
    """Second boundary (misc/panostretch.py:81-102): batched HIP gather vs the oracle on one host core.
    Algorithmic bytes: 512*1024*3*(4+4) = 12,582,912 per image (f32 HWC read + f32 HWC write).
    Every timed iteration warps the batch with FRESH per-image factors, as dataset.py:70-82 draws them per sample, so nothing a real
    batch pays is outside the timed region: `device_coordinates` (hn_pano_stretch -- north_star's form: lon / lat generated on the fly
    in the kernel, nothing but 2 x 64 doubles cross the boundary) and `host_tables` (hn_pano_stretch_tables -- the bit-exact variant:
    per-column terms from numpy, 64 x 1024 arctan2 / sin + a 1.5 MB upload per batch, all inside the timed region; wall-clock, since
    host work is part of it).  `kernel_only_same_factors` is the round-4 figure (tables cached across iterations) for comparison."""
    from horizonnet_amd import pano_stretch_batch
    from oracle import panostretch_ref
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(n_img, 512, 1024, 3, generator=g).to(dev)
    rng = np.random.RandomState(5)
    draws = [(rng.uniform(0.5, 2.0, n_img), rng.uniform(0.5, 2.0, n_img)) for _ in range(iters + 1)]
    out = torch.empty_like(imgs)
    bytes_per_batch = n_img * 12_582_912

    def timed(host_tables, fresh):
        kx, ky = draws[0]
        pano_stretch_batch(imgs, kx, ky, out=out, host_tables=host_tables)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for it in range(iters):
            kx, ky = draws[1 + it] if fresh else draws[0]
            pano_stretch_batch(imgs, kx, ky, out=out, host_tables=host_tables)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / iters * 1e3
        gbs = bytes_per_batch / (ms * 1e-3) / 1e9
        return {"images_per_s": round(n_img / (ms * 1e-3), 1), "ms_per_batch": round(ms, 4),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}}

    dev_form = timed(False, True)
    host_form = timed(True, True)
    cached = timed(True, False)
    kx, ky = draws[iters]
    img = imgs[0].cpu().numpy()
    t0 = time.perf_counter()
    ref, _ = panostretch_ref.pano_stretch(img, np.zeros((1, 2), np.float32), float(kx[0]), float(ky[0]))
    cpu_s = time.perf_counter() - t0
    chk = pano_stretch_batch(imgs[:1], kx[:1], ky[:1], host_tables=True)[0].cpu().numpy()
    err_host = float(np.abs(chk - ref).max())
    chk = pano_stretch_batch(imgs[:1], kx[:1], ky[:1], host_tables=False)[0].cpu().numpy()
    err_dev = float(np.abs(chk - ref).max())
    res = dict(dev_form)
    res.update({"batch": n_img, "form": "device_coordinates (hn_pano_stretch), fresh factors every iteration, wall-clock",
                "host_tables": dict(host_form, form="hn_pano_stretch_tables, fresh factors every iteration: numpy column terms + upload inside the timed region"),
                "kernel_only_same_factors": dict(cached, form="hn_pano_stretch_tables, the same factors every iteration (tables cached): the kernel alone"),
                "cpu_baseline": {"value": round(1.0 / cpu_s, 2), "unit": "images/s", "cores": 1, "kind": "port",
                                 "sample": "1 image 512x1024x3 through oracle.panostretch_ref.pano_stretch (numpy, 1 thread)"},
                "max_abs_vs_oracle": err_dev, "max_abs_vs_oracle_host_tables": err_host})
    return res


def augment_leg(dev, n_img=64, iters=10):
    """Training-input pipeline (dataset.py:52-123 image half): one fused launch per batch of 64, every augmentation on.
    Algorithmic bytes: 512*1024*3*(1+4) = 7,864,320 per image (uint8 HWC read + f32 CHW write)."""
    from horizonnet_amd.dataset import augment_images
    from oracle import dataset_ref
    g = torch.Generator().manual_seed(6)
    data = torch.randint(0, 256, (n_img, 512, 1024, 3), generator=g, dtype=torch.uint8).to(dev)
    rng = np.random.RandomState(6)
    augs = [{"kx": float(rng.uniform(0.5, 2)), "ky": float(rng.uniform(0.5, 2)), "flip": int(rng.randint(2)), "rotate": True,
             "roll": int(rng.randint(1024)), "gamma": float(rng.uniform(0.5, 2))} for _ in range(n_img)]
    idx = list(rng.permutation(n_img))
    out = torch.empty(n_img, 3, 512, 1024, device=dev)
    augment_images(data, idx, augs, out=out)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        augment_images(data, idx, augs, out=out)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    gbs = n_img * 7_864_320 / (ms * 1e-3) / 1e9
    a = augs[0]
    src = data[idx[0]].cpu().numpy()
    t0 = time.perf_counter()
    ref = dataset_ref.augment_image(src, a["kx"], a["ky"], a["flip"], a["roll"], a["gamma"])
    cpu_s = time.perf_counter() - t0
    got = out[0].cpu().numpy()
    ulp = int(np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max())
    return {"images_per_s": round(n_img / (ms * 1e-3), 1), "ms_per_batch": round(ms, 4), "batch": n_img,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)},
            "cpu_baseline": {"value": round(1.0 / cpu_s, 2), "unit": "images/s", "cores": 1, "kind": "port",
                             "sample": "1 image through oracle.dataset_ref.augment_image (numpy, 1 thread)"},
            "max_ulp_vs_oracle": ulp}


def rccl_report(dist, dev, world, rank, mbytes=326, iters=5):
    """Which devices RCCL sees, and the bus bandwidth of the training step's exchange: an in-place float32
    all-reduce of the flat gradient buffer's size (81.57 M elements = 326 MB), 64 MB buckets as in
    horizonnet_amd.parallel.  Ring all-reduce moves 2(N-1)/N of the buffer over each rank's links; xGMI is
    ~153 GB/s per link per direction (MI355X_MICROARCH / task statement)."""
    from horizonnet_amd.parallel import allreduce_sum_async
    names = [None] * world
    dist.all_gather_object(names, "rank %d: %s (cuda:%d)" % (rank, torch.cuda.get_device_name(dev), dev.index))
    buf = torch.ones(mbytes * 1_000_000 // 4, dtype=torch.float32, device=dev)
    for _ in range(2):
        for w in allreduce_sum_async(buf):
            w.wait()
        buf.fill_(1.0)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        for w in allreduce_sum_async(buf):
            w.wait()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / iters
    nbytes = buf.numel() * 4
    busbw = nbytes * (2.0 * (world - 1) / world) / dt / 1e9 if world > 1 else 0.0
    return {"backend": dist.get_backend(), "world_size_seen_by_rccl": dist.get_world_size(), "devices": names,
            "allreduce_bytes": nbytes, "allreduce_ms": round(dt * 1e3, 3), "algbw_GBps": round(nbytes / dt / 1e9, 1),
            "busbw_GBps": round(busbw, 1), "xgmi_link_GBps": 153.0,
            "busbw_vs_one_link": round(busbw / 153.0, 3) if world > 1 else None}


def seeded_net(seed=0):
    """HorizonNet with seeded random-init weights (there is no checkpoint offline): the module's own initialisation
    under torch.manual_seed, plus randomised BatchNorm affine / running statistics so that the folded-BN epilogues and
    the activations are not the trivial identity case.  Self-contained: the benchmarked legs do not touch oracle/."""
    from horizonnet_amd import HorizonNet
    torch.manual_seed(seed)
    net = HorizonNet("resnet50", True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return net


def loss_curve_check(dev, dtype):
    """configs[2] agreement check inside the bench: tests/golden/traincurve_seed41.npz holds the losses of the UNMODIFIED
    reference's train.py loop (5 Adam steps, B=4, every augmentation through the reference's own dataset class, dropout
    off) on the committed synthetic dataset; the same seeds through DeviceBatcher + the HIP training step."""
    import argparse
    import torch.nn.functional as F
    from horizonnet_amd import utils as hutils
    from horizonnet_amd.dataset import DeviceBatcher
    from horizonnet_amd.train import objective
    g = np.load(os.path.join(ROOT, "tests", "golden", "traincurve_seed41.npz"))
    net = seeded_net_from_state(int(g["weights_seed"])).to(dev).train()
    net.bi_rnn.dropout, net.drop_out.p = 0.0, 0.0
    net.train_precision = dtype
    data = DeviceBatcher(root_dir=os.path.join(ROOT, "tests", "golden", "synth_ds"), device=dev, flip=True, rotate=True, gamma=True, stretch=True)
    a = argparse.Namespace(lr=1e-4, warmup_lr=1e-6, warmup_iters=0, max_iters=int(g["max_iters"]), lr_pow=0.9, cur_iter=0, running_lr=1e-4)
    opt = torch.optim.Adam(net.parameters(), lr=a.lr, betas=(0.9, 0.999))
    got = []
    for k in range(len(g["curve"])):
        hutils.adjust_learning_rate(opt, a)
        a.cur_iter += 1
        np.random.seed(int(g["data_seed0"]) + k)
        x, y_bon, y_cor = data.batch(np.random.randint(len(data), size=int(g["B"])))
        opt.zero_grad()
        bon, cor = net(x)
        loss = objective(bon, y_bon, cor, y_cor)["total"]
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    want = (g["curve"][:, 0] + g["curve"][:, 1]).tolist()
    return {"engine": [round(v, 6) for v in got], "reference_cpu_f32": [round(v, 6) for v in want],
            "max_rel_diff": round(max(abs(p - q) / q for p, q in zip(got, want)), 6),
            "note": "first K=5 steps of train.py:246-286 (poly LR, Adam, B=4, all augmentations, dropout off); step 0 is pure forward "
                    "parity, later steps carry Adam's sign noise on near-zero gradients"}


def seeded_net_from_state(seed):
    """HorizonNet with the seeded weights of the golden fixtures (oracle/weights.py is the CHECKER's generator: used by the
    loss-curve check only, never by a timed leg)."""
    from horizonnet_amd import HorizonNet
    from oracle.weights import make_state_dict
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(seed, "random"))
    return net


def train_leg(dev, world, rank, B, steps, warmup, dist, dtype, rccl, rooms, with_curve=True, allreduce_dtype="f32"):
    """BASELINE configs[2] (and configs[3] at world > 1): the loop of reference train.py:246-286 on the engine.  Every step
    draws its batch through the device data pipeline INSIDE the timed region -- DeviceBatcher: Pano-Stretch + flip + roll +
    gamma in one fused launch (dataset.py:70-105) plus the host label rasterisation (dataset.py:85,108-120) -- then
    train-mode forward, L1(bon) + BCE-with-logits(cor) (train.py:53-54), backward (all 241 gradients), the data-parallel
    all-reduce when world > 1, Adam (lr 1e-4, poly decay).  Data: synthetic PanoContext-shaped cuboid rooms rendered at
    start-up (no dataset offline), uint8 resident in HBM."""
    import torch.nn.functional as F
    from horizonnet_amd import broadcast_module_
    from horizonnet_amd.dataset import DeviceBatcher
    imgs, cors = rooms                                    # rendered by main() BEFORE the first GPU call (fork pool, HIP-free)
    data = DeviceBatcher(images=imgs, corners=cors, device=dev, flip=True, rotate=True, gamma=True, stretch=True)
    net = seeded_net(0).to(dev).train()
    net.train_precision = dtype          # bf16: the three conv GEMMs on the bf16 matrix cores, rest f32
    net.allreduce_dtype = allreduce_dtype   # "bf16": the gradient ranges travel as bf16 (163 MB instead of 326 MB over xGMI)
    broadcast_module_(net)
    from horizonnet_amd.optim import FusedAdam
    from horizonnet_amd.train import objective
    opt = FusedAdam(net, lr=1e-4, betas=(0.9, 0.999))     # one hn_adam_step launch over the flat gradient buffer (train.py:216-225,279)
    rng = np.random.RandomState(2000 + rank)
    total_iters = max(1, warmup + steps + max(2, steps // 2) + 1)
    state = {"it": 0, "host_s": 0.0}

    def draws():                                          # the sampler: random indices, then that batch's augmentation draws
        while True:
            yield rng.randint(len(data), size=B)

    # host half (augmentation draws + label rasterisation) two batches ahead in a thread, like the reference's DataLoader
    # workers (train.py:153-160); the device half (fused augmentation launch + label upload) runs in the step
    batches = data.stream(draws(), rng, depth=2)

    def step():
        for g_ in opt.param_groups:                       # misc/utils.py:35-46, power 0.9
            g_["lr"] = 1e-4 * max(0.0, 1.0 - state["it"] / float(total_iters)) ** 0.9       # (the overlap block runs past total_iters)
        state["it"] += 1
        th = time.perf_counter()
        x, y_bon, y_cor = next(batches)
        state["host_s"] += time.perf_counter() - th
        bon, cor = net(x)
        loss = objective(bon, y_bon, cor, y_cor)["total"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    def timed(n):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        state["host_s"] = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        return max_over_ranks(wall, dist, dev), loss

    for _ in range(warmup):
        step()
    wall, loss = timed(steps)
    host_s = state["host_s"]
    assert net.hip_status(dev) == 0 and bool(torch.isfinite(loss))
    overlap = None
    close_batches = batches.close
    if world > 1:                                          # exposed all-reduce time = step with exchange - step without
        net.sync_gradients = False
        local_wall, _ = timed(max(2, steps // 2))
        net.sync_gradients = True
        exposed_ms = max(0.0, wall / steps - local_wall / max(2, steps // 2)) * 1e3
        overlap = {"ms_per_step_with_allreduce": round(wall / steps * 1e3, 2), "ms_per_step_without": round(local_wall / max(2, steps // 2) * 1e3, 2),
                   "allreduce_alone_ms": rccl["allreduce_ms"] if rccl else None, "exposed_ms": round(exposed_ms, 2),
                   "overlap_fraction": round(1.0 - exposed_ms / rccl["allreduce_ms"], 3) if rccl and rccl["allreduce_ms"] > 0 else None,
                   "note": "backward runs in 5 gradient-completion segments; each finished range of the flat gradient buffer starts its "
                           "RCCL all-reduce at once (64 MB buckets) while the remaining segments compute"}
        overlap.update(traced_backward(net, step, dev))
    close_batches()
    host_half_ms = {}
    for label_mode in (True, False):                       # one batch's host half (draws + labels), labels on the device / on the host
        keep, data.device_labels = data.device_labels, label_mode
        th = time.perf_counter()
        data.host_half(list(np.random.RandomState(5).randint(len(data), size=B)), np.random.RandomState(6))
        host_half_ms["device_labels" if label_mode else "host_labels"] = round((time.perf_counter() - th) * 1e3, 2)
        data.device_labels = keep
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    flop = 3.0 * FWD_FLOP_PER_PANO * B * world * steps          # fwd + dgrad + wgrad
    out = {"metric": "training panoramas/s (512x1024, data pipeline + fwd + bwd + Adam)", "value": round(B * world * steps / wall, 2),
           "unit": "panoramas/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(wall / steps * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
           "config": {"workload": "BASELINE configs[%d]: train.py loop on synthetic PanoContext-shaped rooms, pano-stretch + flip + roll + gamma "
                                  "augmentation and label rasterisation inside the timed step, L1 + BCE losses, Adam; %s"
                                  % (2 if world == 1 else 3, "float32 engine" if dtype == "f32" else
                                     "bf16 MFMA conv GEMMs (forward, data and weight gradients), f32 accumulation / BN / LSTM / master weights"),
                      "batch_per_gpu": B, "global_batch": B * world, "rooms_per_gpu": len(cors),
                      "parallelism": "dp%d (one process per GPU, RCCL all-reduce of the flat 326 MB gradient buffer overlapped with backward)" % world,
                      "allreduce_dtype": allreduce_dtype, "defer_grad_mean": bool(getattr(net, "defer_grad_mean", False)),
                      "labels": "device (hn_labels_rasterise)" if data.device_labels else "host (labels.py)"},
           "roofline": train_roofline(dtype, B, wall / steps, flop / wall / 1e12 / world, peak),
           "host_data_pipeline_ms_per_step": round(host_s / steps * 1e3, 2),       # time the step waited for its batch (host thread two batches ahead)
           "host_half_ms_per_batch": host_half_ms,                                # one thread's work per batch of B: what 8 ranks per host must fit 8 x of
           "final_loss": float(loss.detach())}
    if rccl is not None:
        out["rccl"] = rccl
    if overlap is not None:
        out["allreduce_overlap"] = overlap
    if world == 1 and rank == 0 and with_curve:
        out["loss_curve_check"] = loss_curve_check(dev, dtype)
    del net, opt, data
    torch.cuda.empty_cache()
    return out


def train_roofline(dtype, B, s_per_step, achieved_tflops, peak):
    """The training leg's roofline block: bf16 against the MIXED per-pass bound (train_mixed_roofline) with the HBM-side bytes per step
    from the committed counter passes (profiles/r4_pmc_train.json, tools/profile_train.sh), tied to the kernel sources by their hash;
    float32 against the fp32 matrix peak as before."""
    if dtype != "bf16":
        return {"bound": "mfma", "kernel": "whole training step (3 x forward flop: forward, data gradient, weight gradient GEMMs)",
                "achieved": round(achieved_tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved_tflops / peak, 4), "traffic": None,
                "note": "per GPU, against the dense fp32 matrix peak (157.3 TF)"}
    t_mixed, fl, by, t_hbm = train_mixed_roofline(B)
    traffic = stale = note = None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r6_pmc_train.json", "r5_pmc_train.json", "r4_pmc_train.json")) if os.path.exists(q)), "")
    if B == 64 and tpath:
        from horizonnet_amd import _lib as _l
        rec = json.load(open(tpath))
        traffic = rec["total_bytes"]
        measured = (rec.get("measured_on") or {}).get("csrc_sha256")
        stale = measured is None or measured != _l.source_fingerprint()
        note = ("bytes per step, rocprofv3 FETCH_SIZE (x2, gfx950) %.1f GB + WRITE_SIZE %.1f GB, separate --pmc passes over tools/prof_train_target.py "
                "bf16 64 (tools/profile_train.sh); the model's unavoidable bytes %.1f GB -> counter / model = %.2f"
                % (rec["fetch_bytes"] / 1e9, rec["write_bytes"] / 1e9, by / 1e9, traffic / by))
    return {"bound": "mixed (per pass max(flops / 2.5 PFLOP/s, bytes / 8 TB/s), summed: three GEMMs per conv, the BatchNorm passes batch "
                     "statistics force, LSTM x 3, Adam)",
            "achieved": round(B / s_per_step, 1), "peak": round(B / t_mixed, 1), "unit": "panoramas/s", "frac": round(t_mixed / s_per_step, 4),
            "mixed_roofline_ms_per_step": round(t_mixed * 1e3, 2), "hbm_bound_passes_ms": round(t_hbm * 1e3, 2),
            "fused_minimum": (lambda tf: {"peak": round(B / tf[0], 1), "frac": round(tf[0] / s_per_step, 4), "ms_per_step": round(tf[0] * 1e3, 2),
                                          "model_bytes_per_step": tf[2],
                                          "note": "BatchNorm = read z once forward, read dy + z once backward (3 transfers instead of 7 / 8)"})(
                                              train_mixed_roofline(B, fused_minimum=True)),
            "model_flop_per_step": fl, "model_bytes_per_step": by, "traffic": traffic, "traffic_stale": stale, "traffic_note": note,
            "mfma_only": {"achieved_tflops": round(achieved_tflops, 2), "peak": peak, "frac": round(achieved_tflops / peak, 4)}}


def traced_backward(net, step, dev):
    """ONE more step with net.allreduce_trace on (model.py: device events at the start of the backward, after every gradient segment's
    kernels, at the end of every all-reduce bucket) -> when each bucket could start, when it was done, and how much of the exchange
    ran under the backward's own kernels: the diagnosis of an N-GPU run in the bench line itself."""
    net.allreduce_trace = []
    step()
    torch.cuda.synchronize(dev)
    tr, net.allreduce_trace = net.allreduce_trace, None
    t0 = [e for e in tr if e["kind"] == "start"][-1]["event"]
    segs = [e for e in tr if e["kind"] == "segment"]
    seg_end = {e["segment"]: t0.elapsed_time(e["event"]) for e in segs}
    compute_end = max(seg_end.values()) if seg_end else 0.0
    rows, prev_done, busy, hidden = [], 0.0, 0.0, 0.0
    for e in [e for e in tr if e["kind"] == "bucket"]:
        done = t0.elapsed_time(e["event"])
        begin = max(seg_end[e["segment"]], prev_done)                   # buckets queue behind each other on RCCL's stream
        rows.append({"segment": e["segment"], "bucket": e["bucket"], "MB": round(e["bytes"] / 1e6, 1), "may_start_ms": round(seg_end[e["segment"]], 2),
                     "done_ms": round(done, 2), "GBps": round(e["bytes"] / max(done - begin, 1e-3) / 1e6, 1)})
        busy += max(0.0, done - begin)
        hidden += max(0.0, min(done, compute_end) - min(begin, compute_end))
        prev_done = max(prev_done, done)
    return {"traced_step": {"segment_kernels_done_ms": [round(seg_end[k], 2) for k in sorted(seg_end)], "backward_kernels_done_ms": round(compute_end, 2),
                            "last_bucket_done_ms": round(prev_done, 2), "exchange_busy_ms": round(busy, 2),
                            "exchange_under_backward_kernels_ms": round(hidden, 2),
                            "overlap_fraction_traced": round(hidden / busy, 3) if busy > 0 else None, "buckets": rows,
                            "note": "ms from the start of the backward on the device clock; a bucket may start when its segment's kernels are "
                                    "done and the previous bucket has left RCCL's stream"}}


def box_probe(dev):
    """What THIS box sustains on the two rooflines ("re-derive from the box ... a measured MFMA micro-benchmark; report both"):
    dense MFMA rate of the library's rate kernel (hn_probe_mfma, ~4 ms per dtype, every CU, 4 independent accumulator chains per wave) and the
    bandwidth of a 1 GiB device copy.  Round 6 saw the same library at 849 (fp32 forward) on one box and 748 on another: the fraction of the
    guide's peak is the contract's number, the fraction of the box's own rate says whether the CODE or the BOX moved."""
    import ctypes
    from horizonnet_amd import _lib
    lib = _lib.load()
    out = {}
    wgs = 2048
    scratch = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    flop = ctypes.c_double()
    with torch.cuda.device(dev):
        for name, dt, iters in (("mfma_f32_tflops", 0, 4000), ("mfma_bf16_tflops", 1, 8000)):
            best = 0.0
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.hn_probe_mfma(dt, wgs, iters, _lib.ptr(scratch), ctypes.byref(flop), _lib.stream_ptr(dev)), "hn_probe_mfma")
                e1.record()
                torch.cuda.synchronize(dev)
                best = max(best, flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            out[name] = round(best, 1)
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        best = 0.0
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize(dev)
            best = max(best, 2.0 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        out["hbm_copy_GBps"] = round(best, 0)
        del a, b, scratch
    out["note"] = ("measured on this box in this run (best of 3-4): dense v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 issue rate on all CUs, "
                   "read + write bytes of a 1 GiB device copy; the guide's peaks are 157.3 TF / ~2500 TF / ~8000 GB/s")
    return out


def max_over_ranks(value, dist, dev):
    """MAX over the ranks of a host scalar (the contract's max-over-ranks timing); device tensor for RCCL, host tensor for gloo."""
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def layout_leg(dev, n, batch, rooms, pool, dist=None, world=1, rank=0, precisions=("bf16", "f32")):
    """BASELINE configs[4]: inference.py general-layout path over n Structured3D-shaped synthetic panoramas, engine bf16 (and
    f32) vs the fixtures of the UNMODIFIED reference inference() (tools/c5_layout.py).  world > 1: the panoramas are
    sharded over the ranks; rank 0 returns the merged line, the others None."""
    from tools.c5_layout import run_layout_eval
    r = run_layout_eval(dev, n=n, batch=batch, precisions=precisions, timing_repeats=1, rooms=rooms, pool=pool, dist=dist,
                        world=world, rank=rank)
    if r is None:
        return None
    b = r["bf16"]
    assert b["iou3d_failed"] == 0 and all(r[p_]["iou3d_failed"] == 0 for p_ in precisions), "layout_metrics failed on a predicted layout"
    return {"metric": "layout panoramas/s (inference.py general layout, end to end incl. post-processing) + 3D-IoU parity vs reference",
            "value": b["panoramas_per_s_end_to_end"], "unit": "panoramas/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": round(b["seconds"] * 1e3, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %d seeded synthetic Structured3D-shaped panoramas (corner mix 63/17/8/12 %% of 4/6/8/10), "
                                   "briefly-trained checkpoint tests/golden/config5/ckpt_q.npz, uint8 panoramas resident in HBM -> "
                                   "inference() general layout (engine forward + device peaks + host Manhattan fit), batch %d, "
                                   "sharded over %d rank(s)" % (r["panoramas"], batch, world)},
            "iou3d_parity_vs_reference_inference": {p_: r[p_] for p_ in precisions},
            "render_crc_mismatches": r["render_crc_mismatches"], "corner_mix_reference": r["corner_mix_reference"],
            "host_cores": r["host_cores"], "host_cores_per_rank": r["host_cores_per_rank"]}


def bf16_leg(net, x, dev, steps, f32_out):
    """The engine's bf16 inference mode on the headline inputs: the plain forward (one stream-ordered call per batch) and the
    pipelined entry (forward_async: recurrent head of batch i on 32 compute units beside the trunk of batch i+1); both
    against the bf16 MFMA peak and against the per-layer MIXED roofline that governs this mode."""
    B = int(x.shape[0])
    net.precision = "bf16"
    bon, cor = f32_out
    steps = max(int(steps), 60)                           # 0.4 s: the pipelined entry's fill + drain (one recurrent head, ~2 ms) is < 1 % of the region

    def run_plain(n):
        for _ in range(n):
            out = net(x)
        return out

    def run_piped(n):
        pend = None
        for _ in range(n):
            nxt = net.forward_async(x)
            if pend is not None:
                pend.result()
            pend = nxt
        return pend.result()                              # the last batch's head is INSIDE the timed region (pipeline drained)

    res = {}
    with torch.no_grad():
        for name, fn in (("plain", run_plain), ("pipelined", run_piped)):
            hb, hc = fn(2)
            torch.cuda.synchronize(dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            ev0.record()
            hb, hc = fn(steps)
            ev1.record()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t1) / steps
            res[name] = {"value": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "ms_per_step_hip_events": round(ev0.elapsed_time(ev1) / steps, 3),
                         "max_abs_vs_f32_outputs": round(float(max((hb - bon).abs().max(), (hc - cor).abs().max())), 6)}
            res[name + "_out"] = (hb, hc)
    identical = bool(torch.equal(res["plain_out"][0], res["pipelined_out"][0]) and torch.equal(res["plain_out"][1], res["pipelined_out"][1]))
    del res["plain_out"], res["pipelined_out"]
    assert net.hip_status(dev) == 0, "persistent LSTM kernel reported a spin time-out (bf16 leg)"
    net.precision = "f32"
    t_mixed, fl, by, t_hbm = bf16_mixed_roofline(B)
    best = max(res["plain"]["value"], res["pipelined"]["value"])
    ms_best = B / best * 1e3
    tf = FWD_FLOP_PER_PANO * B / (ms_best * 1e-3) / 1e12
    return {"value": best, "unit": "panoramas/s", "ms_per_step": round(ms_best, 3), "steps": steps,
            "plain": res["plain"], "pipelined": res["pipelined"], "pipelined_equals_plain_bitwise": identical,
            "f32_output_max_abs": round(float(max(bon.abs().max(), cor.abs().max())), 3),
            "roofline": {"bound": "mixed (per layer max(flops / 2.5 PFLOP/s, unfused bytes / 8 TB/s), summed)",
                         "achieved": best, "peak": round(B / t_mixed, 1), "unit": "panoramas/s", "frac": round(best / (B / t_mixed), 4),
                         "mixed_roofline_ms_per_batch": round(t_mixed * 1e3, 3), "hbm_bound_layers_ms": round(t_hbm * 1e3, 3),
                         "algorithmic_flop_per_batch": fl, "unfused_bytes_per_batch": by,
                         "mfma_only": {"achieved": round(tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_MFMA_TFLOPS, 4)}},
            "note": "bf16 MFMA convs, f32 accumulate / gates / head; single GPU; NOT the fp32 configuration the headline value is quoted on. "
                    "value = the faster of the two entries; `pipelined` times K submits AND K collects (the last head inside the region)"}


def latency_leg(dev):
    """BASELINE configs[0]: ONE panorama (assets/demo.png, committed as tests/golden/demo_input_u8.npz) through inference()
    (inference.py:196-209: forward + general-layout post-processing, no test-time augmentation), latency in ms, beside the
    CPU oracle's forward on the same input (bounded: 3 forwards).  Seeded random weights: the layout itself is meaningless,
    the work is the same."""
    from horizonnet_amd import inference
    from oracle import horizonnet_ref
    from oracle.weights import make_state_dict
    img = np.load(os.path.join(ROOT, "tests", "golden", "demo_input_u8.npz"))["img"]
    x = torch.from_numpy(img.transpose(2, 0, 1)[None].astype(np.float32) / 255.0)
    net = seeded_net(0).to(dev).eval()
    out = {}
    with torch.no_grad():
        for prec in ("f32", "bf16"):
            net.precision = prec
            times_fwd, times_e2e = [], []
            for it in range(7):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                bon, cor = net(x.to(dev))
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                try:
                    inference(net, x, dev, force_raw=False)
                except Exception:                         # random weights may give < 4 peaks: the raw path has the same device work
                    inference(net, x, dev, force_raw=True)
                t2 = time.perf_counter()
                if it >= 2:
                    times_fwd.append(t1 - t0)
                    times_e2e.append(t2 - t1)
            out[prec] = {"forward_ms": round(1e3 * float(np.median(times_fwd)), 3), "inference_end_to_end_ms": round(1e3 * float(np.median(times_e2e)), 3)}
    sd = make_state_dict(0, "random")
    torch.set_num_threads(cpu_cores())
    horizonnet_ref.forward(x, sd)
    t0 = time.perf_counter()
    for _ in range(3):
        horizonnet_ref.forward(x, sd)
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    del net
    return {"workload": "BASELINE configs[0]: demo.png, B=1, host tensor in -> layout out (H2D copy included)", "engine": out,
            "cpu_oracle_forward_ms": round(cpu_ms, 1), "cpu_cores": cpu_cores(),
            "reference_cpu_note": "BASELINE.md section 4: unmodified reference inference() on 8 vCPU 0.55-1.11 s"}


def cpu_cores():
    from horizonnet_amd.hostcores import usable_cores
    return usable_cores()


def prepare_rooms(args, world, rank, legs):
    """Host-side rendering of the synthetic rooms for the train / layout legs in a pool forked BEFORE the first GPU call
    (forking a process that already holds a HIP runtime and helper threads is undefined behaviour).  The pool is
    kept for the IoU evaluation of the layout leg (its workers never touch the GPU)."""
    import multiprocessing as mp
    from horizonnet_amd.hostcores import rank_cores
    from tools import c5_common as c5
    need_train = args.mode == "train" or (args.mode == "forward" and "train" in legs)
    need_layout = args.mode == "layout" or (args.mode == "forward" and "layout" in legs)
    if not (need_train or need_layout):
        return None, {}
    pool = mp.get_context("fork").Pool(rank_cores(world))
    rooms = {}
    t0 = time.perf_counter()
    if need_train:
        rooms["train"] = c5.make_rooms(args.rooms, 20_000 + 1000 * rank, pool, mix=[4])
    if need_layout:
        n = min(args.panoramas, 1000)
        if world > 1:                                     # every rank renders only its own shard
            lo, hi = shard_for_rank(n, world, rank)
            imgs, gt = c5.make_rooms(hi - lo, c5.VAL_SEED0, pool, first=lo)
            rooms["layout_shard"] = (lo, hi, imgs, gt)
        else:
            rooms["layout"] = c5.make_rooms(n, c5.VAL_SEED0, pool)
    rooms["render_s"] = round(time.perf_counter() - t0, 1)
    return pool, rooms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="panoramas per GPU per step (forward: 32 = configs[1]; train: 64 = configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (default, the parity path); bf16 = the engine's bf16 inference mode")
    ap.add_argument("--force-rccl", action="store_true",
                    help="initialise the RCCL process group also at world size 1 (exercises the nccl branch on one GPU)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for --gpus N (nccl = RCCL, the product path; gloo only to exercise the N > 1 "
                         "code path on a box with one GPU, together with --share-gpu)")
    ap.add_argument("--allreduce-dtype", choices=["f32", "bf16"], default="f32",
                    help="train mode, N > 1: wire format of the gradient all-reduce (bf16 = 163 MB instead of 326 MB per step over xGMI; the sum is "
                         "rounded to bf16 -- net.allreduce_dtype)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (test rigs with fewer GPUs than ranks)")
    ap.add_argument("--mode", choices=["forward", "train", "layout"], default="forward",
                    help="forward = the headline metric (default, configs[1]); train = configs[2]/[3]: one optimisation step per step "
                         "incl. the data pipeline; layout = configs[4]: inference.py general-layout path + 3D-IoU vs the reference")
    ap.add_argument("--legs", default="all",
                    help="--mode forward, --gpus 1: extra legs carried by the same JSON line, comma separated out of "
                         "bf16,latency,train,layout,stretch,augment,cpu ('all', 'none')")
    ap.add_argument("--panoramas", type=int, default=1000, help="layout leg: panoramas evaluated")
    ap.add_argument("--rooms", type=int, default=96, help="train leg: synthetic rooms rendered per rank")
    ap.add_argument("--plain", action="store_true", help="forward mode: time the plain stream-ordered forward instead of the pipelined entry")
    ap.add_argument("--train-steps", type=int, default=10, help="train LEG of the default run: timed steps (B=64, bf16)")
    args = ap.parse_args()
    all_legs = ["bf16", "latency", "train", "layout", "stretch", "augment", "cpu"]
    legs = all_legs if args.legs == "all" else [] if args.legs == "none" else [l for l in args.legs.split(",") if l]
    for l in legs:
        if l not in all_legs:
            raise SystemExit("unknown leg %r (choose from %s)" % (l, ",".join(all_legs)))
    if args.no_cpu_baseline and "cpu" in legs:
        legs.remove("cpu")
    if args.batch is None:
        args.batch = 64 if args.mode == "train" else 32

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run
        # (the form the driver uses itself for N > 1), rendezvous on 127.0.0.1
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if world > 1 or args.mode != "forward":
        legs = [] if args.mode == "forward" else legs      # the extra legs belong to the single-GPU default run

    from horizonnet_amd.hostcores import pin_rank_affinity, rank_cores
    pin_rank_affinity(local_rank, world)                   # each rank's host threads / pools stay on its own slice of the CPUs
    pool, rooms = prepare_rooms(args, world, rank, legs)   # fork pool + rendering BEFORE the first GPU call

    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    rccl = None
    if world > 1 or args.force_rccl:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            # NOT bound with device_id=: torch's eager communicator init costs the HBM-bound bf16 legs 4-5 % and the B = 1 bf16 latency 0.4 ms on this
            # part (tools/stream_pool_probe.py nccl_bench vs nccl_bench_nodev); torch.cuda.set_device(dev) above tells RCCL the device
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        rccl = rccl_report(dist, dev, world, rank, mbytes=326 if args.backend == "nccl" else 8)   # who is in the job + all-reduce bandwidth

    if args.mode == "layout":
        if world > 1:                                      # stitch the shards' images back into one index space for the evaluator
            lo, hi, imgs, gt = rooms["layout_shard"]
            n = min(args.panoramas, 1000)
            full = np.zeros((n,) + imgs.shape[1:], np.uint8)
            full[lo:hi] = imgs
            parts = [None] * world
            dist.all_gather_object(parts, (lo, gt))
            gts = [g for _, gl in sorted(parts, key=lambda t_: t_[0]) for g in gl]
            lay_rooms = (full, gts)
        else:
            lay_rooms = rooms["layout"]
        out = layout_leg(dev, args.panoramas, args.batch, lay_rooms, pool, dist, world, rank)
        if rank == 0:
            out["render_s"] = rooms["render_s"]
            if rccl is not None:
                out["rccl"] = rccl
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == "train":
        out = train_leg(dev, world, rank, args.batch, args.steps, args.warmup, dist, args.dtype, rccl, rooms["train"], allreduce_dtype=args.allreduce_dtype)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    net = seeded_net(0)                              # seeded random-init weights (no checkpoint offline)
    net = net.to(dev).eval()
    net.precision = args.dtype

    B = args.batch
    lo, hi = shard_for_rank(B * world, world, rank)   # weak scaling: B panoramas per rank
    assert hi - lo == B
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    x = torch.rand(B, 3, 512, 1024, generator=g).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run_plain(n):
        for _ in range(n):
            out_ = net(x)
        return out_

    def run(n):
        """n steps through the pipelined entry (forward_async): the recurrent head of step i runs on 64 (f32) / 32 (bf16)
        compute units beside the convolutional trunk of step i+1; EVERY step's head, the last one included, completes inside
        the caller's timed region (result() of the last handle + the synchronisation that ends the region)."""
        if args.plain:
            return run_plain(n)
        pend = None
        for _ in range(n):
            nxt = net.forward_async(x)
            if pend is not None:
                pend.result()
            pend = nxt
        return pend.result()

    with torch.no_grad():
        if args.warmup > 0:
            run(args.warmup)
        barrier()
        stream = torch.cuda.current_stream(dev)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)                              # HIP events on the stream the kernels are launched on
        bon, cor = run(args.steps)
        ev1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    assert net.hip_status(dev) == 0, "persistent LSTM kernel reported a spin time-out"
    assert bool(torch.isfinite(bon).all()) and bool(torch.isfinite(cor).all())
    wall_max = max_over_ranks(wall, dist, dev)

    plain_rec = None
    if not args.plain:                                   # the plain (one stream-ordered call per batch) forward beside it
        with torch.no_grad():
            run_plain(2)
            barrier()
            t0 = time.perf_counter()
            pb, pc = run_plain(args.steps)
            barrier()
            pw = max_over_ranks(time.perf_counter() - t0, dist, dev)
        plain_rec = {"value": round(B * world * args.steps / pw, 2), "ms_per_step": round(pw / args.steps * 1e3, 3),
                     "max_abs_diff_pipelined_vs_plain_outputs": float(max((pb - bon).abs().max(), (pc - cor).abs().max()))}
    if rank == 0:
        total_panos = B * world * args.steps
        value = total_panos / wall_max
        # ---- roofline: whole forward as the unit, HIP-event time of the timed region ----
        ms_per_fwd = dev_ms / args.steps
        achieved_tflops = FWD_FLOP_PER_PANO * B / (ms_per_fwd * 1e-3) / 1e12
        with torch.no_grad():
            _, _, entries = net.profile_forward(x)
        fam = {}
        for name, ms, fl in entries:
            if "ghc_lst" in name:
                key = "height_compress_convs(conv_igemm)"
            elif "encoder.layer" in name:
                key = "resnet_stage_convs(conv_igemm)"
            elif name.startswith("stem"):
                key = "stem(prep+conv_igemm+maxpool)"
            elif "input_gemm" in name:
                key = "lstm_input_gemm(conv_igemm)"
            elif "recurrence" in name:
                key = "lstm_recurrence(persistent)"
            else:
                key = "other(upsample,linear)"
            a = fam.setdefault(key, [0.0, 0.0])
            a[0] += ms
            a[1] += fl
        prof_total = sum(v[0] for v in fam.values())
        breakdown = {k: {"ms": round(v[0], 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0.0,
                         "share": round(v[0] / prof_total, 3)} for k, v in fam.items()}
        igemm_ms = sum(v[0] for k, v in fam.items() if "conv_igemm" in k)
        igemm_fl = sum(v[1] for k, v in fam.items() if "conv_igemm" in k)
        traffic, traffic_note = None, None               # HBM-side bytes per forward from the committed PMC passes of the newest round's kernels
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        traffic_stale = None
        for tname in ("r6_pmc_forward.json", "r5_pmc_forward.json", "r4_pmc_forward.json", "r3_pmc_forward.json", "r2_pmc_forward.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if B == 32 and os.path.exists(tpath):
                whole = json.load(open(tpath))
                rec = whole["precisions"].get(args.dtype)
                if rec is None:
                    continue
                traffic = rec["total_bytes"]
                # the counters are from a separate rocprofv3 run: tie them to the code that runs NOW (hash of the loaded library)
                from horizonnet_amd import _lib as _l
                measured = (whole.get("measured_on") or {}).get("csrc_sha256")
                traffic_stale = measured is None or measured != _l.source_fingerprint()      # (kernel sources: a library hash is not reproducible)
                traffic_note = ("bytes per forward, rocprofv3 FETCH_SIZE (x2, gfx950) %.1f GB + WRITE_SIZE %.1f GB, separate --pmc passes, profiles/%s "
                                "(tools/profile_forward.sh); Infinity-Cache hits included; algorithmic minimum %.1f GB -> counter / algorithmic = %.2f; "
                                "matrix pipe busy %s %% of the conv kernels' cycles" % (
                                    rec["fetch_bytes"] / 1e9, rec["write_bytes"] / 1e9, tname, rec["algorithmic_min_bytes"] / 1e9, rec["counter_over_algorithmic"],
                                    "/".join(str(v["mfma_busy_pct"]) for k, v in rec["by_kernel_family"].items() if k.startswith("conv") and "mfma_busy_pct" in v)))
                break
        out = {
            "metric": "panoramas/s (512x1024 fwd)",
            "value": round(value, 2),
            "unit": "panoramas/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch=32 random 512x1024 panos per GPU, ResNet-50 + height-compression + bi-LSTM forward, %s, seeded random-init weights"
                                   % ("fp32" if args.dtype == "f32" else "bf16 MFMA convs (f32 accumulate, f32 gates / head) -- NOT the fp32 config"),
                       "entry": "plain forward (one stream-ordered call per step)" if args.plain else
                                "pipelined forward (forward_async: the recurrent head of step i beside the trunk of step i+1; all K heads complete inside the timed region)",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "dp%d (independent replicas, no data-path collective)" % world},
            "roofline": {"bound": "mfma", "kernel": "%s (all launches of one forward; implicit-GEMM conv family = %.0f%% of device time)" % (
                             ("hn_forward" if args.plain else "hn_forward_submit") if args.dtype == "f32" else ("hn_forward_bf16" if args.plain else "hn_forward_bf16_submit"), 100.0 * igemm_ms / prof_total),
                         "achieved": round(achieved_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved_tflops / peak, 4), "traffic": traffic,
                         "traffic_stale": traffic_stale,
                         "traffic_note": traffic_note,
                         "algorithmic_flop_per_launch": FWD_FLOP_PER_PANO * B,
                         "launch_ms_hip_events": round(ms_per_fwd, 3),
                         "conv_igemm": {"ms": round(igemm_ms, 3), "tflops": round(igemm_fl / (igemm_ms * 1e-3) / 1e12, 2),
                                            "frac": round(igemm_fl / (igemm_ms * 1e-3) / 1e12 / peak, 4)},
                         "breakdown": breakdown},
        }
        if args.dtype == "bf16":
            t_mixed = bf16_mixed_roofline(B)[0]
            out["roofline"]["mixed_per_layer"] = {"peak_panoramas_per_s": round(B / t_mixed, 1), "frac": round(B / (ms_per_fwd * 1e-3) / (B / t_mixed), 4),
                                                  "note": "the governing roofline of the bf16 mode (BASELINE.md section 3)"}
        else:
            t_mixed, _, _, t_hbm = f32_mixed_roofline(B)
            out["roofline"]["mixed_per_layer"] = {"peak_panoramas_per_s": round(B / t_mixed, 1), "frac": round(t_mixed / (ms_per_fwd * 1e-3), 4),
                                                  "hbm_bound_layers_ms": round(t_hbm * 1e3, 3),
                                                  "note": "per layer max(flops / 157.3 TF, unfused float32 bytes / 8 TB/s), summed: the bound that also charges the "
                                                          "HBM-bound layer1 1x1 convs their bytes; `frac` above stays the pure-MFMA figure SURVEY 8(d) names"}
        try:
            box = box_probe(dev)
            out["box"] = box
            key = "mfma_f32_tflops" if args.dtype == "f32" else "mfma_bf16_tflops"
            out["roofline"]["frac_of_box_measured_mfma"] = round(achieved_tflops / box[key], 4)
        except Exception as exc:                          # noqa: BLE001  (a probe must never cost the headline line)
            out["box"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if plain_rec is not None:
            out["plain_forward"] = plain_rec
        if rccl is not None:
            out["rccl"] = rccl
        if world == 1:
            leg_s = {}

            def timed_leg(key, fn):
                """An extra leg must never cost the headline line: a failure is reported under the leg's key."""
                t_ = time.perf_counter()
                try:
                    out[key] = fn()
                except Exception as exc:                  # noqa: BLE001  (reported, not swallowed: the key carries the error)
                    import traceback
                    out[key] = {"error": "%s: %s" % (type(exc).__name__, exc), "traceback_tail": traceback.format_exc()[-1500:]}
                    try:
                        torch.cuda.synchronize(dev)
                        torch.cuda.empty_cache()
                    except Exception:                     # noqa: BLE001
                        pass
                leg_s[key] = round(time.perf_counter() - t_, 1)

            if args.dtype == "f32" and "bf16" in legs:    # informational: the engine's bf16 mode on the same inputs (configs[1] is fp32)
                timed_leg("bf16_mode", lambda: bf16_leg(net, x, dev, args.steps, (bon, cor)))
            del net
            torch.cuda.empty_cache()
            if "cpu" in legs:                             # the CPU reference leg runs on rank 0 of the single-GPU job only
                timed_leg("cpu_baseline", cpu_baseline)
            if "latency" in legs:
                timed_leg("latency_b1", lambda: latency_leg(dev))
            if "stretch" in legs:
                timed_leg("pano_stretch", lambda: pano_stretch_leg(dev))
            if "augment" in legs:
                timed_leg("augment_pipeline", lambda: augment_leg(dev))
            if "train" in legs:                           # configs[2]: B = 64, bf16, data pipeline inside the step
                # (5 untimed steps: the ring of four pinned staging slots of the batch stream and FusedAdam's state are first-use allocations)
                timed_leg("train_bf16", lambda: train_leg(dev, 1, 0, 64, args.train_steps, 5, None, "bf16", None, rooms["train"]))
            if "layout" in legs:                          # configs[4]: 1000 panoramas, end to end, IoU vs the reference's inference()
                timed_leg("layout", lambda: layout_leg(dev, args.panoramas, 32, rooms["layout"], pool))
            out["leg_seconds"] = dict(leg_s, render_rooms=rooms.get("render_s"))
            # the same legs against the rooflines THIS box measured in this run (out["box"]): the contract's fractions above use the guide's peaks
            # (2.5 PF, 8 TB/s); a device copy reaches 5.2-5.5 TB/s and the dense bf16 MFMA rate 2.4 PF on the boxes of round 6
            bx = out.get("box") or {}
            if "mfma_bf16_tflops" in bx and "hbm_copy_GBps" in bx:
                pf, bw = bx["mfma_bf16_tflops"] * 1e12, bx["hbm_copy_GBps"] * 1e9
                r = (out.get("bf16_mode") or {}).get("roofline")
                if isinstance(r, dict):
                    tb = bf16_mixed_roofline(B, pf, bw)[0]
                    r["frac_of_box_measured_mixed"] = round(out["bf16_mode"]["value"] / (B / tb), 4)
                r = (out.get("train_bf16") or {}).get("roofline")
                if isinstance(r, dict) and isinstance(out["train_bf16"].get("ms_per_step"), (int, float)):
                    st_ = out["train_bf16"]["ms_per_step"] * 1e-3
                    r["frac_of_box_measured_mixed"] = round(train_mixed_roofline(64, pf, bw)[0] / st_, 4)
                    r["fused_minimum"]["frac_of_box_measured"] = round(train_mixed_roofline(64, pf, bw, fused_minimum=True)[0] / st_, 4)
                ps_ = out.get("pano_stretch")
                if isinstance(ps_, dict) and isinstance(ps_.get("roofline"), dict):
                    ps_["roofline"]["frac_of_box_copy_rate"] = round(ps_["roofline"]["achieved"] / bx["hbm_copy_GBps"], 4)
        print(json.dumps(out))
    if pool is not None:
        pool.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
