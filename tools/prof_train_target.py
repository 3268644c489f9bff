"""Profiling target for rocprofv3 (MEASUREMENT TOOL): N identical training steps (forward + backward + FusedAdam) of the
seeded network on one fixed batch, no data pipeline and no loss-curve check, so per-step figures = per-kernel totals / N.
    rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/tr -- python tools/prof_train_target.py bf16 32 4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from bench import seeded_net  # noqa: E402
from horizonnet_amd.optim import FusedAdam  # noqa: E402
from horizonnet_amd.train import objective  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
net = seeded_net(0).to(dev).train()
net.train_precision = prec
opt = FusedAdam(net, lr=1e-4)
g = torch.Generator().manual_seed(1000)
x = torch.rand(B, 3, 512, 1024, generator=g).to(dev)
yb = (torch.rand(B, 2, 1024, generator=g) - 0.5).to(dev)
yc = torch.rand(B, 1, 1024, generator=g).to(dev)
import time  # noqa: E402
for i in range(N + 1):
    if i == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    bon, cor = net(x)
    loss = objective(bon, yb, cor, yc)["total"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
assert net.hip_status(dev) == 0
print("PROF_TRAIN_TARGET precision=%s B=%d steps=%d(+1 warm) ms_per_step=%.2f panos/s=%.1f" % (prec, B, N, ms, B / ms * 1e3))
