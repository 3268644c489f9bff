"""Does a smaller batch keep the early stages' activations in the 256 MB Infinity Cache?  (MEASUREMENT TOOL)

Per-launch-group times of the profiled forward (hn_set_profiling: no chaining, no branch stream) at several batch sizes,
summed per stage and divided by the batch: microseconds per panorama.  A stage whose per-panorama time drops at B=8 (its
64-/256-channel activations = 33 / 134 MB) against B=32 (134 / 537 MB) is a candidate for sub-batched execution inside the
B=32 forward.      python tools/subbatch_probe.py [bf16|f32] [reps]
"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import seeded_net  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
net = seeded_net(0).to(dev).eval()
net.precision = prec


def stage(name):
    for k in ("layer1", "layer2", "layer3", "layer4"):
        if k in name:
            return k
    if "reduce_height" in name or "ghc" in name:
        return "height_compression"
    if "lstm" in name or "rnn" in name:
        return "lstm"
    return name.split(".")[0] if "." in name else name


rows = OrderedDict()
detail = {}
for B in (4, 8, 16, 32):
    x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(1000)).to(dev)
    with torch.no_grad():
        net.profile_forward(x)
        acc = OrderedDict()
        per = OrderedDict()
        for _ in range(reps):
            _, _, entries = net.profile_forward(x)
            for name, ms, _fl in entries:
                acc[stage(name)] = acc.get(stage(name), 0.0) + ms * 1e3 / reps / B
                per[name] = per.get(name, 0.0) + ms * 1e3 / reps / B
    rows[B] = acc
    detail[B] = per
    del x
keys = list(rows[32].keys())
print("# %s: microseconds per panorama per stage (profiled forward: one launch group at a time)" % prec)
print("%-26s" % "stage" + "".join("%10s" % ("B=%d" % b) for b in rows))
for k in keys:
    print("%-26s" % k + "".join("%10.1f" % rows[b].get(k, 0.0) for b in rows))
print("%-26s" % "total" + "".join("%10.1f" % sum(rows[b].values()) for b in rows))
print("# per launch group (layer1 and the stem only)")
for k in detail[32]:
    if "layer1" in k or stage(k) not in ("layer2", "layer3", "layer4", "height_compression", "lstm"):
        print("%-40s" % k[:40] + "".join("%10.1f" % detail[b].get(k, 0.0) for b in rows))
