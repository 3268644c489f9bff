"""Merge the per-counter JSONs of tools/profile_train.sh into profiles/rN_pmc_train.json (MEASUREMENT TOOL): HBM-side bytes of ONE bf16
training step at B = 64 (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes), the ratio to the bytes of bench.py's
train_mixed_roofline model, the matrix-pipe busy share per kernel family -- the record bench.py quotes in train_bf16.roofline.traffic,
tied to the kernel sources it was measured on by their SHA-256 (horizonnet_amd._lib.source_fingerprint).
    python tools/merge_pmc_train.py gpurun_out r4 > profiles/r4_pmc_train.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    from bench import train_mixed_roofline
    t_mixed, fl, by, t_hbm = train_mixed_roofline(64)
    recs = {c: json.load(open(os.path.join(d, "%s_train_pmc_%s.json" % (tag, c)))) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")
            if os.path.exists(os.path.join(d, "%s_train_pmc_%s.json" % (tag, c)))}
    fam = {}
    for k, row in recs["FETCH_SIZE"]["per_forward_by_kernel_family"].items():
        fam[k] = {"dispatches_per_step": row["dispatches_per_forward"], "fetch_bytes": row.get("fetch_bytes", 0.0)}
    for k, row in recs["WRITE_SIZE"]["per_forward_by_kernel_family"].items():
        fam.setdefault(k, {"dispatches_per_step": row["dispatches_per_forward"]})["write_bytes"] = row.get("write_bytes", 0.0)
    for k, row in recs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("per_forward_by_kernel_family", {}).items():
        busy, tot = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), row.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
        f = fam.setdefault(k, {"dispatches_per_step": row["dispatches_per_forward"]})
        f["mfma_busy_pct"] = round(100.0 * busy / tot, 1) if tot else 0.0
        f["gpu_cycles_per_xcd"] = row.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    fetch = sum(f.get("fetch_bytes", 0.0) for f in fam.values())
    write = sum(f.get("write_bytes", 0.0) for f in fam.values())
    lib = os.path.join(ROOT, "horizonnet_amd", "libhorizonnet_hip.so")
    cyc = sum(f.get("gpu_cycles_per_xcd", 0.0) for f in fam.values())
    busy_all = sum(f.get("mfma_busy_pct", 0.0) * f.get("gpu_cycles_per_xcd", 0.0) for f in fam.values())
    print(json.dumps({
        "what": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} over tools/prof_train_target.py "
                "bf16 64 3 (4 identical steps: forward + backward + FusedAdam), three separate passes (tools/profile_train.sh); per step",
        "correction": "FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64 bytes), KB -> bytes; WRITE_SIZE as reported; Infinity-Cache hits included",
        "measured_on": {"csrc_sha256": __import__("horizonnet_amd._lib", fromlist=["x"]).source_fingerprint(),
                        "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
                        "git_head": os.environ.get("HN_GIT_HEAD", "unknown (no .git on the GPU box: pass HN_GIT_HEAD)")},
        "fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write,
        "model_bytes": by, "counter_over_model": round((fetch + write) / by, 3), "model_flop": fl, "mixed_roofline_ms": round(t_mixed * 1e3, 2),
        "mfma_busy_pct_whole_step": round(busy_all / cyc, 1) if cyc else None, "by_kernel_family": fam}, indent=1))


if __name__ == "__main__":
    main()
