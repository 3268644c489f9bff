"""BASELINE configs[2] + configs[4] in one GPU job (ROUND-1 MEASUREMENT TOOL -- test infrastructure, may use oracle/;
the round-2 forms are `bench.py --mode train` / `--mode layout`, which compare with the UNMODIFIED reference's outputs).

There is no dataset and no trained checkpoint offline, so:

1. render a synthetic Structured3D-shaped set of Manhattan rooms (tools/synth_rooms.py) -- panoramas + label_cor;
2. train HorizonNet from random init on the MI355X engine for a few hundred steps with ALL of the reference's
   augmentations (flip, rotate, gamma, Pano-Stretch) through the fused device pipeline (horizonnet_amd.dataset
   .DeviceBatcher), losses / optimiser of train.py:44-58,272-281 (float32, or --train-precision bf16 = the bf16 matrix-core step);
3. run the general-layout inference path (inference.py:65-141 = horizonnet_amd.inference) on held-out panoramas with
   (a) the engine in float32, (b) the engine in bf16, (c) the CPU float32 oracle of the reference forward, and report
   the per-image 3D IoU (eval_general.py:56-95 = horizonnet_amd.evaluation) between the predicted layouts -- the
   "3D-IoU parity vs reference" half of BASELINE.json's metric -- and of each against the ground truth.

Writes one JSON document (default gpurun_out/config5.json).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.hostinfo import usable_cores  # noqa: E402
from tools.c5_common import make_rooms, make_room as _make_room, ST3D_MIX  # noqa: E402,F401


def layouts(bon, cor_prob, peaks_fn):
    """-> list of cor_id in pixels (or None where the Manhattan fit fails) for a batch of signals."""
    from horizonnet_amd.inference import layout_from_signals
    out = []
    for b in range(bon.shape[0]):
        try:
            cid, _, _ = layout_from_signals(bon[b].copy(), cor_prob[b, 0].copy(), peaks_fn=peaks_fn)
            out.append(cid * np.array([[1024, 512]], np.float32))
        except Exception:                                   # too few corner peaks etc. (asserts of post_proc.py)
            out.append(None)
    return out


def pair_iou(a, b):
    from horizonnet_amd.evaluation import layout_metrics
    if a is None or b is None:
        return None
    try:
        m = layout_metrics(a, b)
        return None if m is None else float(m["iou3d"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-rooms", type=int, default=384)
    ap.add_argument("--val-rooms", type=int, default=96)
    ap.add_argument("--oracle-rooms", type=int, default=32, help="held-out panoramas also run through the CPU oracle")
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--train-precision", choices=["f32", "bf16"], default="f32",
                    help="bf16 = forward + data-gradient convs on the bf16 matrix cores (net.train_precision)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--time-limit", type=float, default=150.0, help="seconds of training after which the loop stops")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "config5.json"))
    args = ap.parse_args()

    import multiprocessing as mp
    from horizonnet_amd import HorizonNet, find_N_peaks
    from horizonnet_amd.dataset import DeviceBatcher
    from oracle import horizonnet_ref
    dev = torch.device("cuda:0")
    cores = usable_cores()
    torch.set_num_threads(cores)
    res = {"config": vars(args), "host_cores": cores}

    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        tr_img, tr_cor = make_rooms(args.train_rooms, 10_000, pool)
        va_img, va_cor = make_rooms(args.val_rooms, 50_000, pool)
    res["render_s"] = round(time.perf_counter() - t0, 1)

    # ---- configs[2]: training with every augmentation on ------------------------------------------------------------
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    net = HorizonNet("resnet50", True).to(dev).train()
    net.train_precision = args.train_precision
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.999), fused=True)
    train = DeviceBatcher(images=tr_img, corners=tr_cor, device=dev, flip=True, rotate=True, gamma=True, stretch=True)
    curve, n_seen = [], 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_data = 0.0
    for step in range(args.steps):
        td = time.perf_counter()
        idx = np.random.randint(len(train), size=args.batch)
        x, y_bon, y_cor = train.batch(idx)
        t_data += time.perf_counter() - td
        for g_ in opt.param_groups:                       # polynomial decay of train.py (misc/utils.py:35-46, power 0.9)
            g_["lr"] = args.lr * (1.0 - step / float(args.steps)) ** 0.9
        bon, cor = net(x)
        l_bon, l_cor = F.l1_loss(bon, y_bon), F.binary_cross_entropy_with_logits(cor, y_cor)
        loss = l_bon + l_cor
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        n_seen += args.batch
        if step % 25 == 0 or step == args.steps - 1:
            curve.append([step, round(float(l_bon), 4), round(float(l_cor), 4)])
            if time.perf_counter() - t0 > args.time_limit:
                break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert net.hip_status(dev) == 0
    res["train"] = {"steps": step + 1, "batch": args.batch, "seconds": round(wall, 1),
                    "panoramas_per_s_incl_data_pipeline": round(n_seen / wall, 1),
                    "host_label_seconds": round(t_data, 1), "dtype": args.train_precision,
                    "augmentations": "flip+rotate+gamma+pano-stretch (hn_augment_batch)", "loss_curve_step_bon_cor": curve}
    print("train:", json.dumps(res["train"])[:400], flush=True)

    # ---- configs[4]: general-layout inference, engine f32 / bf16 vs the CPU oracle ----------------------------------
    net.eval()
    val = DeviceBatcher(images=va_img, corners=va_cor, device=dev)
    peaks = lambda s, r, min_v, N: find_N_peaks(s, r=r, min_v=min_v, N=N, device=dev)   # noqa: E731
    sig = {}
    with torch.no_grad():
        for prec in ("f32", "bf16"):
            net.precision = prec
            bons, cors = [], []
            for i in range(0, len(val), 8):
                x, _, _ = val.batch(list(range(i, min(i + 8, len(val)))))
                b, c = net(x)
                bons.append(b.cpu().numpy())
                cors.append(torch.sigmoid(c).cpu().numpy())
            sig[prec] = (np.concatenate(bons), np.concatenate(cors))
        net.precision = "f32"
        n_or = min(args.oracle_rooms, len(val))
        x, _, _ = val.batch(list(range(n_or)))
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        t0 = time.perf_counter()
        ob, oc = horizonnet_ref.forward(x.cpu(), sd)
        res["oracle_forward_panoramas_per_s"] = round(n_or / (time.perf_counter() - t0), 2)
        sig["oracle"] = (ob.numpy(), torch.sigmoid(oc).numpy())
    res["signal_max_abs_vs_oracle"] = {
        p: {"bon": float(np.abs(sig[p][0][:n_or] - sig["oracle"][0]).max()),
            "cor_prob": float(np.abs(sig[p][1][:n_or] - sig["oracle"][1]).max())} for p in ("f32", "bf16")}

    gt = [np.asarray(c, np.float64) for c in va_cor]
    lay = {p: layouts(sig[p][0], sig[p][1], peaks) for p in sig}

    def summary(vals):
        ok = [v for v in vals if v is not None]
        if not ok:
            return {"n": 0, "failed": len(vals)}
        return {"n": len(ok), "failed": len(vals) - len(ok), "mean": round(float(np.mean(ok)), 5),
                "min": round(float(np.min(ok)), 5), "median": round(float(np.median(ok)), 5)}

    res["iou3d_vs_ground_truth"] = {p: summary([pair_iou(lay[p][i], gt[i]) for i in range(len(lay[p]))]) for p in lay}
    res["iou3d_engine_vs_oracle_layout"] = {
        p: summary([pair_iou(lay[p][i], lay["oracle"][i]) for i in range(n_or)]) for p in ("f32", "bf16")}
    res["iou3d_bf16_vs_f32_engine_layout"] = summary([pair_iou(lay["bf16"][i], lay["f32"][i]) for i in range(len(val))])
    res["same_corner_count_as_oracle"] = {
        p: int(sum(1 for i in range(n_or) if lay[p][i] is not None and lay["oracle"][i] is not None
                   and len(lay[p][i]) == len(lay["oracle"][i]))) for p in ("f32", "bf16")}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "train"}, indent=1))


if __name__ == "__main__":
    main()
