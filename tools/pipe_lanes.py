"""MEASUREMENT TOOL: L engine lanes (own handle, workspace, trunk stream) driven round-robin through forward_async -- the
HBM-bound stem / layer1 of one batch beside the MFMA-bound layer3 / layer4 of another, on top of the head overlap.
    python tools/pipe_lanes.py [bf16|f32] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import seeded_net  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(os.environ.get("PIPE_B", "32"))
dev = torch.device("cuda:0")
x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(1000)).to(dev)
for lanes in (1, 2, 3):
    nets = [seeded_net(0).to(dev).eval() for _ in range(lanes)]
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
    for n in nets:
        n.precision = prec

    def run(n):
        pend = []
        for i in range(n):
            k = i % lanes
            with torch.cuda.stream(streams[k]):
                p = nets[k].forward_async(x)
            pend.append((k, p))
            if len(pend) > lanes:
                kk, q = pend.pop(0)
                with torch.cuda.stream(streams[kk]):
                    q.result()
        for kk, q in pend:
            with torch.cuda.stream(streams[kk]):
                q.result()

    with torch.no_grad():
        run(2 * lanes)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K)
    print("%s lanes=%d: %.3f ms per batch  %.1f panoramas/s  status %s" % (prec, lanes, best * 1e3, B / best, [n.hip_status(dev) for n in nets]))
    del nets
    torch.cuda.empty_cache()
