"""Per-launch-group times of ONE panorama (B = 1) through the plain forward, float32 and bf16 (hn_set_profiling): where the 3.9 / 2.5 ms go.
    python tools/b1_profile.py [B=1] [all]        ("all": every launch group with its TF/s, e.g. B = 32 -> profiles/rN_{f32,bf16}_forward_B32_launch_groups.txt)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from horizonnet_amd import HorizonNet       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ALL = len(sys.argv) > 2 and sys.argv[2] == "all"
dev = torch.device("cuda:0")
net = HorizonNet("resnet50", True).to(dev).eval()
x = torch.rand(B, 3, 512, 1024, device=dev)
for prec in ("f32", "bf16"):
    net.precision = prec
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            _, _, entries = net.profile_forward(x)
            tot = sum(ms for _, ms, _ in entries)
            if best is None or tot < best[0]:
                best = (tot, entries)
    tot, entries = best
    print("== %s B=%d: sum of launch groups %.3f ms" % (prec, B, tot))
    fam = {}
    for name, ms, fl in entries:
        key = ("stem" if name.startswith("stem") else "recurrence" if "recurrence" in name else "input_gemm" if "input_gemm" in name else
               "ghc" if "ghc_lst" in name else name.split(".")[2] if "encoder.layer" in name else "other")
        fam[key] = fam.get(key, 0.0) + ms
    print("   " + ", ".join("%s %.3f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])))
    if ALL:
        for name, ms, fl in entries:
            print("   %-70s %8.3f ms %8.2f TF/s %6.2f%%" % (name[-70:], ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, 100.0 * ms / tot))
    else:
        for name, ms, fl in sorted(entries, key=lambda e: -e[1])[:12]:
            print("   %-70s %.3f ms" % (name[-70:], ms))
