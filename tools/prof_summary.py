"""Summarise rocprofv3 CSV output (-f csv) for profiles/ (MEASUREMENT TOOL).

    python tools/prof_summary.py stats   <dir> <title>                 -> kernel table (calls, total, average, share)
    python tools/prof_summary.py counters <dir> <forwards> <title>     -> JSON: per kernel family sums of every counter / forwards

Counter conventions (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE
counts 128-byte read requests as 64 bytes -> doubled here; WRITE_SIZE is left as reported (uncalibrated); Infinity-Cache
hits are counted."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def family(name):
    n = name
    if "at::native" in n or "rocclr" in n:
        return "other:framework"
    if "multi_job" in n:
        return "pack_multi_job"                         # weight packing, once per engine
    for key in ("conv3x3_dwr64_bf16", "conv3x3_dwr_bf16", "conv_igemm_bf16_pp", "bn_bwd_reduce", "bn_bwd_apply", "affine_act_bn", "adam_flat", "labels_rasterise", "conv_igemm_bf16_w8", "conv1x1_dual_bf16", "conv1x1_chain_bf16", "conv1x1_dual_f32", "conv_igemm_bf16", "conv_igemm_kernel", "conv_wgrad",
                "splitk_reduce", "stem_pool_bf16", "stem_pool_f32", "mfma_rate", "lstm_layer_bf16", "lstm_layer",
                "maxpool", "prep_nhwc", "upsample_flatten", "linear_head", "f32_to_bf16", "pack_", "fold_bn", "find_peaks", "pano_stretch",
                "augment"):
        if key in n:
            return key
    return "other:" + n[-40:]


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return hits


def stats(d, title):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    print("# " + title)
    print("%-110s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows:
        name = r.get("Name") or r.get("Kernel_Name") or "?"
        tot = float(r.get("TotalDurationNs", 0)) / 1e3
        avg = float(r.get("AverageNs", 0)) / 1e3
        print("%-110s %8d %12.0f %10.1f %7.2f" % (name[:110], int(r.get("Calls", 0)), tot, avg, float(r.get("Percentage", 0))))


def counters(d, forwards, title):
    fam = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = family(r["Kernel_Name"])
            fam[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k].add(r["Dispatch_Id"])
    out = {"what": title, "forwards_in_run": forwards, "per_forward_by_kernel_family": {}, "per_forward_total": defaultdict(float)}
    for k, cs in sorted(fam.items()):
        if k.startswith("pack_") or k.startswith("fold_bn") or k == "other:framework":
            continue                                   # weight packing, torch fills/copies: once per process, not part of a forward
        row = {"dispatches_per_forward": len(ndisp[k]) / forwards}
        for c, v in cs.items():
            v = v / forwards
            if c == "FETCH_SIZE":
                row["fetch_bytes"] = v * 1024.0 * 2.0
                out["per_forward_total"]["fetch_bytes"] += row["fetch_bytes"]
            elif c == "WRITE_SIZE":
                row["write_bytes"] = v * 1024.0
                out["per_forward_total"]["write_bytes"] += row["write_bytes"]
            else:
                row[c] = v
                out["per_forward_total"][c] += v
        out["per_forward_by_kernel_family"][k] = row
    out["per_forward_total"] = dict(out["per_forward_total"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else sys.argv[2])
    else:
        counters(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else sys.argv[2])
