"""BASELINE configs[4] ("config 5") measured side: the engine's general-layout inference (inference.py:65-141 =
horizonnet_amd.inference) in float32 and bf16 over the seeded synthetic Structured3D-shaped panoramas, compared per image
with the layouts the UNMODIFIED reference ``inference()`` produced for the same pixels and weights in the build container
(tests/golden/config5/reference_layouts.npz, oracle/gen_config5.py): 3D IoU between the two predicted layouts
(eval_general.py:56-95 semantics = horizonnet_amd.evaluation.layout_metrics), corner-count agreement, signal deviation,
and end-to-end panoramas/s including the host Manhattan fit.

MEASUREMENT TOOL: used by bench.py --mode layout and tests/test_gpu_integration.py.  Imports nothing from oracle/.
"""
import multiprocessing as mp
import os
import time

import numpy as np
import torch

from tools import c5_common as c5

FIXTURE = os.path.join(c5.ROOT, "tests", "golden", "config5", "reference_layouts.npz")


def _cores():
    return c5.usable_cores()


def load_reference():
    z = np.load(FIXTURE)
    off = np.concatenate([[0], np.cumsum(z["count"])])
    cor = [z["cor_id"][off[i]:off[i + 1]] for i in range(len(z["count"]))]
    return z, cor


def _iou(a, b):
    from horizonnet_amd.evaluation import layout_metrics
    try:
        m = layout_metrics(a, b)
        return None if m is None else float(m["iou3d"])
    except Exception:
        return None


def seam_aware_max_abs(a, b, width=1024.0):
    """Largest corner displacement between two layouts with the SAME number of corners, in pixels, with the panorama's 0 / `width`
    column seam taken into account: the corner list is ordered by column, so a corner that crosses the seam (x = 1023.9 -> 0.2) moves to
    the other end of the list -- the raw |a - b| then reads ~`width` for a sub-pixel move.  Corners come as (ceiling, floor) row pairs:
    try every cyclic shift by whole pairs, measure x circularly, keep the best shift."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape and a.shape[0] % 2 == 0
    best = None
    for sft in range(0, a.shape[0], 2):
        c = np.roll(b, sft, axis=0)
        dx = np.abs(a[:, 0] - c[:, 0])
        dx = np.minimum(dx, width - dx)
        v = float(max(dx.max(), np.abs(a[:, 1] - c[:, 1]).max()))
        best = v if best is None or v < best else best
    return best


def _pair_iou(args):
    return _iou(*args)


def _shard(n, world, rank):
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_layout_eval(dev, n=1000, batch=32, precisions=("f32", "bf16"), timing_repeats=1, rooms=None, pool=None, dist=None,
                    world=1, rank=0):
    """rooms: (imgs uint8 [n,512,1024,3], ground-truth corner lists) rendered by the caller (bench.py renders them BEFORE the
    first GPU call, in a pool forked while the process is still single-threaded and HIP-free); pool: that pool, reused for
    the IoU evaluation.  world > 1: the panoramas are sharded contiguously over the ranks (independent units, no data-path
    collective); every rank times its own stream, the layouts are gathered on rank 0, which evaluates all n of them and
    reports n / max-over-ranks time.  Ranks > 0 return None."""
    from horizonnet_amd import HorizonNet
    from horizonnet_amd.inference import inference_stream
    from horizonnet_amd.dataset import images_to_input
    from horizonnet_amd.hostcores import rank_cores
    z, ref_cor = load_reference()
    n = min(n, int(z["n"]))
    own_pool = None
    if rooms is None:
        own_pool = pool = mp.get_context("fork").Pool(_cores())
        t0 = time.perf_counter()
        rooms = c5.make_rooms(n, int(z["seed0"]), pool)
    imgs, gt = rooms
    lo, hi = _shard(n, world, rank)
    res = {"panoramas": n, "batch": batch, "host_cores": _cores(), "host_cores_per_rank": rank_cores(world), "ranks": world,
           "corner_mix_reference": {int(k): int(v) for k, v in zip(*np.unique(z["count"][:n] // 2, return_counts=True))}}
    res["render_crc_mismatches"] = int(sum(1 for i in range(lo, hi) if c5.image_crc(imgs[i]) != int(z["crc"][i])))
    net = HorizonNet("resnet50", True)
    net.load_state_dict(c5.decode_state_dict(), strict=True)
    net = net.to(dev).eval()
    lut = torch.from_numpy((np.arange(256) / 255).astype(np.float32)).to(dev)      # inference.py:199-200: img / 255 -> FloatTensor
    data = torch.from_numpy(np.ascontiguousarray(imgs[lo:hi])).to(dev)             # uint8 [shard,512,1024,3] resident in HBM
    m = hi - lo
    scale = np.array([[1024, 512]], np.float32)
    nfull = min(m, max(0, min(n, z["bon"].shape[0]) - lo))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for prec in precisions:
            net.precision = prec
            layouts, sig_err, times = None, 0.0, []
            for rep in range(timing_repeats + 1):                                    # first pass = warm-up (weight packing, workspaces)
                barrier()
                t0 = time.perf_counter()
                out = []
                # inference.py:199-200 (img / 255 -> FloatTensor [B,3,H,W]) as one device pass per batch (dataset.images_to_input);
                # the signal comparisons below use the torch LUT form of the same conversion, so both are exercised
                feed = (images_to_input(data, list(range(a, min(a + batch, m)))) for a in range(0, m, batch))
                for part in inference_stream(net, feed, dev, depth=3):    # GPU on batch i+1, i+2 while the host cores fit batch i
                    out += part
                barrier()
                if rep > 0 or timing_repeats == 0:
                    times.append(time.perf_counter() - t0)
                layouts = out
            seconds = min(times)
            # raw signals against the reference's: the first panoramas in full, and the per-panorama sums of ALL of them
            if nfull > 0:
                x = lut[data[:nfull].long()].permute(0, 3, 1, 2).contiguous()
                bon, cor = net(x)
                sig_err = max(float(np.abs(bon.cpu().numpy() - z["bon"][lo:lo + nfull]).max()),
                              float(np.abs(cor.cpu().numpy() - z["cor"][lo:lo + nfull]).max()))
            sums = []
            for a in range(0, m, batch):
                bon, cor = net(lut[data[a:a + batch].long()].permute(0, 3, 1, 2).contiguous())
                sums.append(torch.stack([bon.double().sum(dim=(1, 2)), cor.double().sum(dim=(1, 2))], 1).cpu().numpy())
            sums = np.concatenate(sums) if sums else np.zeros((0, 2))
            if dist is not None and world > 1:
                parts = [None] * world
                dist.all_gather_object(parts, {"layouts": layouts, "seconds": seconds, "sig_err": sig_err, "sums": sums,
                                               "crc": res["render_crc_mismatches"]})
                if rank != 0:
                    continue
                layouts = [l for p_ in parts for l in p_["layouts"]]
                seconds = max(p_["seconds"] for p_ in parts)
                sig_err = max(p_["sig_err"] for p_ in parts)
                sums = np.concatenate([p_["sums"] for p_ in parts])
                res["render_crc_mismatches"] = int(sum(p_["crc"] for p_ in parts))
            mine = [l[0] * scale for l in layouts]
            theirs = [c * scale for c in ref_cor[:n]]
            ious = pool.map(_pair_iou, list(zip(mine, theirs)))
            ious_gt = pool.map(_pair_iou, [(m_, np.asarray(g, np.float64)) for m_, g in zip(mine, gt)])
            ok = np.array([v for v in ious if v is not None])
            okg = np.array([v for v in ious_gt if v is not None])
            z1_err = max(abs(float(l[2]) - float(z["z1"][i])) / abs(float(z["z1"][i])) for i, l in enumerate(layouts))
            res[prec] = {
                "iou3d_mean": round(float(ok.mean()), 6), "iou3d_min": round(float(ok.min()), 6),
                "iou3d_below_0.99": int((ok < 0.99).sum()), "iou3d_failed": int(len(ious) - len(ok)),
                "corner_count_mismatches": int(sum(1 for a, b in zip(mine, theirs) if len(a) != len(b))),
                "identical_cor_id": int(sum(1 for a, b in zip(mine, theirs) if a.shape == b.shape and np.array_equal(a, b))),
                "cor_id_max_abs_px_where_counts_agree": round(float(max([np.abs(a - b).max() for a, b in zip(mine, theirs) if a.shape == b.shape] or [0.0])), 4),
                # the same with the 0 / 1024 column seam handled (a corner crossing it re-orders the list: the raw figure above then reads ~1000 px)
                "cor_id_max_abs_px_seam_aware": round(float(max([seam_aware_max_abs(a, b) for a, b in zip(mine, theirs) if a.shape == b.shape] or [0.0])), 4),
                "cor_id_p99_px_seam_aware": round(float(np.percentile([seam_aware_max_abs(a, b) for a, b in zip(mine, theirs) if a.shape == b.shape] or [0.0], 99)), 4),
                "corner_count_mismatch_panoramas": [int(i) for i, (a, b) in enumerate(zip(mine, theirs)) if len(a) != len(b)][:16],
                "iou3d_min_panorama": int(np.argmin([v if v is not None else 2.0 for v in ious])),
                "z1_max_rel_err": float(z1_err), "signal_max_abs_vs_reference": sig_err,
                "iou3d_vs_ground_truth_mean": round(float(okg.mean()), 5),
                "panoramas_per_s_end_to_end": round(n / seconds, 1), "seconds": round(seconds, 3)}
            if "signal_sum" in z.files:      # per-panorama sums of bon / cor over ALL n panoramas (reference: float64 sums of its f32 outputs)
                d = np.abs(sums[:n] - np.asarray(z["signal_sum"], np.float64)[:n])
                res[prec]["signal_sum_max_abs_vs_reference"] = {"bon": float(d[:, 0].max()), "cor": float(d[:, 1].max()),
                                                                "panoramas": int(n), "elements_per_sum": {"bon": 2048, "cor": 1024}}
    net.precision = "f32"
    if own_pool is not None:
        own_pool.close()
    return res if rank == 0 else None


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, c5.ROOT)
    print(json.dumps(run_layout_eval(torch.device("cuda:0"), n=int(sys.argv[1]) if len(sys.argv) > 1 else 1000), indent=1))
