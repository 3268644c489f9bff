"""Train the config-5 model on the MI355X engine and export it as the compact checkpoint of tools/c5_common.py.

MEASUREMENT TOOL (runs on the GPU box; writes gpurun_out/c5/ckpt_q.npz + gpurun_out/c5/train_export.json).

  render synthetic rooms -> train from random init (all augmentations, bf16 matrix cores) -> quantise every >= 2-D
  tensor onto the codec's grid -> re-estimate the BatchNorm running statistics (train-mode forwards, no optimiser) ->
  fine-tune the 1-D parameters only (BatchNorm affine, biases; the quantised tensors stay on their grid) -> evaluate
  3D IoU vs ground truth before / after -> save.  The saved file decodes to exactly the evaluated state_dict.
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

from tools import c5_common as c5  # noqa: E402
from tools.config5 import layouts, pair_iou  # noqa: E402


usable_cores = c5.usable_cores


def evaluate(net, val, gt, dev, precisions=("f32",)):
    from horizonnet_amd import find_N_peaks
    peaks = lambda s, r, min_v, N: find_N_peaks(s, r=r, min_v=min_v, N=N, device=dev)   # noqa: E731
    net.eval()
    out = {}
    with torch.no_grad():
        for prec in precisions:
            net.precision = prec
            bons, cors = [], []
            for i in range(0, len(val), 16):
                x, _, _ = val.batch(list(range(i, min(i + 16, len(val)))))
                b, c = net(x)
                bons.append(b.cpu().numpy())
                cors.append(torch.sigmoid(c).cpu().numpy())
            lay = layouts(np.concatenate(bons), np.concatenate(cors), peaks)
            ious = [pair_iou(lay[i], gt[i]) for i in range(len(lay))]
            ok = [v for v in ious if v is not None]
            out[prec] = {"n": len(ok), "failed": len(ious) - len(ok), "mean": round(float(np.mean(ok)), 5) if ok else None,
                         "min": round(float(np.min(ok)), 5) if ok else None,
                         "corner_count_ok": int(sum(1 for i in range(len(lay)) if lay[i] is not None and len(lay[i]) == len(gt[i])))}
    net.precision = "f32"
    return out


def run_steps(net, opt, train, steps, batch, lr, time_limit, curve, tag):
    net.train()
    t0 = time.perf_counter()
    for step in range(steps):
        idx = np.random.randint(len(train), size=batch)
        x, y_bon, y_cor = train.batch(idx)
        for g_ in opt.param_groups:
            g_["lr"] = lr * (1.0 - step / float(steps)) ** 0.9
        bon, cor = net(x)
        l_bon, l_cor = F.l1_loss(bon, y_bon), F.binary_cross_entropy_with_logits(cor, y_cor)
        opt.zero_grad(set_to_none=True)
        (l_bon + l_cor).backward()
        opt.step()
        if step % 50 == 0 or step == steps - 1:
            curve.append([tag, step, round(float(l_bon), 4), round(float(l_cor), 4)])
            if time.perf_counter() - t0 > time_limit:
                break
    torch.cuda.synchronize()
    return step + 1, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-rooms", type=int, default=512)
    ap.add_argument("--val-rooms", type=int, default=96)
    ap.add_argument("--steps", type=int, default=2400)
    ap.add_argument("--finetune-steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--time-limit", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "c5"))
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)

    import multiprocessing as mp
    from horizonnet_amd import HorizonNet
    from horizonnet_amd.dataset import DeviceBatcher
    dev = torch.device("cuda:0")
    res = {"config": vars(args)}
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(usable_cores()) as pool:
        tr_img, tr_cor = c5.make_rooms(args.train_rooms, c5.TRAIN_SEED0, pool)
        va_img, va_cor = c5.make_rooms(args.val_rooms, c5.VAL_SEED0, pool)
    res["render_s"] = round(time.perf_counter() - t0, 1)
    gt = [np.asarray(c, np.float64) for c in va_cor]

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    net = HorizonNet("resnet50", True).to(dev)
    net.train_precision = "bf16"
    train = DeviceBatcher(images=tr_img, corners=tr_cor, device=dev, flip=True, rotate=True, gamma=True, stretch=True)
    val = DeviceBatcher(images=va_img, corners=va_cor, device=dev)
    curve = []
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.999), fused=True)
    n, s = run_steps(net, opt, train, args.steps, args.batch, args.lr, args.time_limit, curve, "train")
    res["train"] = {"steps": n, "seconds": round(s, 1), "panoramas_per_s": round(n * args.batch / s, 1)}
    res["iou3d_vs_gt_trained"] = evaluate(net, val, gt, dev)
    print(json.dumps(res["iou3d_vs_gt_trained"]), flush=True)

    # ---- onto the codec's grid --------------------------------------------------------------------------------------
    with torch.no_grad():
        for k, p in net.named_parameters():
            if p.dim() >= 2:
                q, sc = c5.quantize_tensor(p.detach().cpu().numpy(), c5.bits_for(p.shape))
                p.copy_(torch.from_numpy(c5.dequantize_tensor(q, sc)).to(dev))
                p.requires_grad_(False)
    res["iou3d_vs_gt_quantised_raw"] = evaluate(net, val, gt, dev)
    print(json.dumps(res["iou3d_vs_gt_quantised_raw"]), flush=True)
    net.train()
    with torch.no_grad():                                   # BatchNorm running statistics of the quantised network
        for _ in range(60):
            x, _, _ = train.batch(np.random.randint(len(train), size=args.batch))
            net(x)
    res["iou3d_vs_gt_quantised_bn_recalibrated"] = evaluate(net, val, gt, dev)
    print(json.dumps(res["iou3d_vs_gt_quantised_bn_recalibrated"]), flush=True)
    small = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam(small, lr=1e-4, betas=(0.9, 0.999))
    n, s = run_steps(net, opt, train, args.finetune_steps, args.batch, 1e-4, 60.0, curve, "finetune-1d")
    res["finetune"] = {"steps": n, "seconds": round(s, 1), "tensors": len(small), "elements": int(sum(p.numel() for p in small))}
    assert net.hip_status(dev) == 0

    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    path = os.path.join(args.out_dir, "ckpt_q.npz")
    res["checkpoint_bytes"] = c5.save_checkpoint(sd, path)
    back = c5.decode_state_dict(path)
    worst = max(float((back[k].float() - sd[k].float()).abs().max()) for k in sd)
    res["decode_max_abs_vs_evaluated_state_dict"] = worst                 # must be 0: the file IS the model
    net.load_state_dict(back)
    net.to(dev)
    res["iou3d_vs_gt_final"] = evaluate(net, val, gt, dev, ("f32", "bf16"))
    res["loss_curve"] = curve
    with open(os.path.join(args.out_dir, "train_export.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "loss_curve"}, indent=1))


if __name__ == "__main__":
    main()
