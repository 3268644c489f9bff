"""Merge the per-counter JSONs of tools/profile_forward.sh into profiles/rN_pmc_forward.json (MEASUREMENT TOOL): per precision
the HBM-side bytes per forward (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes), the ratio to the algorithmic
minimum and the matrix-pipe busy share per kernel family -- the record bench.py quotes in `roofline.traffic`.
    python tools/merge_pmc.py gpurun_out r4 > profiles/r4_pmc_forward.json
The record carries the SHA-256 of the kernel SOURCES it was measured on (horizonnet_amd._lib.source_fingerprint; the built library's
hash is not reproducible from a fresh build) and the git HEAD: bench.py sets `traffic_stale` when the tree it runs from differs."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ALGORITHMIC_MIN = {"f32": 31.686e9, "bf16": 15.9e9}      # every conv reads in + W once, writes out once (B = 32)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {"what": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} over "
                   "tools/prof_target.py {f32p,bf16p} 32 4 (pipelined entry), three separate passes per precision (tools/profile_forward.sh)",
           "correction": "FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64 bytes), KB -> bytes; WRITE_SIZE as reported; Infinity-Cache hits included",
           "precisions": {}}
    # what these counters were measured ON: bench.py compares the hash with the library it loads and flags `traffic_stale` otherwise
    lib = os.path.join(ROOT, "horizonnet_amd", "libhorizonnet_hip.so")
    sys.path.insert(0, ROOT)
    from horizonnet_amd import _lib as _l
    out["measured_on"] = {"csrc_sha256": _l.source_fingerprint(),
                          "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
                          "git_head": (subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or
                                       os.environ.get("HN_GIT_HEAD", "unknown (no .git on the GPU box: pass HN_GIT_HEAD)"))}
    for prec in ("f32", "bf16"):
        recs = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            p = os.path.join(d, "%s_%s_pmc_%s.json" % (tag, prec, c))
            if os.path.exists(p):
                recs[c] = json.load(open(p))
        if "FETCH_SIZE" not in recs or "WRITE_SIZE" not in recs:
            continue
        fam = {}
        for k, row in recs["FETCH_SIZE"]["per_forward_by_kernel_family"].items():
            fam[k] = {"dispatches_per_forward": row["dispatches_per_forward"], "fetch_bytes": row.get("fetch_bytes", 0.0)}
        for k, row in recs["WRITE_SIZE"]["per_forward_by_kernel_family"].items():
            fam.setdefault(k, {"dispatches_per_forward": row["dispatches_per_forward"]})["write_bytes"] = row.get("write_bytes", 0.0)
        for k, row in recs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("per_forward_by_kernel_family", {}).items():
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the busy counter over the 1024 SIMDs
            busy, tot = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), row.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
            f = fam.setdefault(k, {"dispatches_per_forward": row["dispatches_per_forward"]})
            f["mfma_busy_pct"] = round(100.0 * busy / tot, 1) if tot else 0.0
            f["gpu_cycles_per_xcd"] = row.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        fetch = sum(f.get("fetch_bytes", 0.0) for f in fam.values())
        write = sum(f.get("write_bytes", 0.0) for f in fam.values())
        out["precisions"][prec] = {"fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write,
                                   "algorithmic_min_bytes": ALGORITHMIC_MIN[prec],
                                   "counter_over_algorithmic": round((fetch + write) / ALGORITHMIC_MIN[prec], 3), "by_kernel_family": fam}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
