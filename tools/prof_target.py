"""Profiling target for rocprofv3 (MEASUREMENT TOOL): N identical B=32 forwards of the seeded network in one precision.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/pmc_x -- python tools/prof_target.py bf16
Every forward launches the same kernels, so per-forward figures are the per-kernel totals / N (weight packing runs once
and has its own kernel names)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import seeded_net  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
net = seeded_net(0).to(dev).eval()
piped = prec.endswith("p")       # "bf16p" / "f32p": the pipelined entry (forward_async: head of batch i beside the trunk of batch i+1)
net.precision = prec[:-1] if piped else prec
if len(sys.argv) > 4:            # geometry of the wide recurrence kernel: rows,xcds
    r_, x_ = sys.argv[4].split(",")
    net.set_engine_option("lstm_wide_rows", int(r_))
    net.set_engine_option("lstm_wide_xcds", int(x_))
x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(1000)).to(dev)
with torch.no_grad():
    pend = None
    for _ in range(N):
        if piped:
            p_ = net.forward_async(x)
            if pend is not None:
                bon, cor = pend.result()
            pend = p_
        else:
            bon, cor = net(x)
    if piped:
        bon, cor = pend.result()
torch.cuda.synchronize()
assert net.hip_status(dev) == 0
print("PROF_TARGET precision=%s B=%d forwards=%d" % (prec, B, N))
