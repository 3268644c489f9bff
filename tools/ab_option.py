"""MEASUREMENT TOOL: A/B of an engine option on the B = 32 bf16 forward (plain and pipelined).
    python tools/ab_option.py chain_layer1 [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import seeded_net  # noqa: E402

opt = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
net = seeded_net(0).to(dev).eval()
net.precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
x = torch.rand(32, 3, 512, 1024, generator=torch.Generator().manual_seed(1000)).to(dev)


def plain(n):
    for _ in range(n):
        net(x)


def piped(n):
    pend = None
    for _ in range(n):
        p = net.forward_async(x)
        if pend is not None:
            pend.result()
        pend = p
    pend.result()


def timed(fn):
    fn(3)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    return best


with torch.no_grad():
    for v in (0, 1, 0, 1):
        net.set_engine_option(opt, v)
        print("%s=%d: plain %.3f ms  pipelined %.3f ms" % (opt, v, timed(plain), timed(piped)), flush=True)
