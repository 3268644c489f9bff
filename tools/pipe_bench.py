"""MEASUREMENT TOOL: plain bf16 forward vs the pipelined entry (forward_async: recurrent head of batch i beside the trunk of
batch i+1) at B = 32 for every geometry of the wide recurrence kernel, plus the wide kernel alone.
    python tools/pipe_bench.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import seeded_net  # noqa: E402
from horizonnet_amd import _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32
dev = torch.device("cuda:0")
net = seeded_net(0).to(dev).eval()
net.precision = "bf16"
x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(1000)).to(dev)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


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


with torch.no_grad():
    plain(3)
    ms = timed(plain, K)
    print("plain bf16 forward            : %.3f ms  %.1f panoramas/s" % (ms, B / ms * 1e3))
    for rows, xcds in ((16, 1), (16, 2), (8, 1), (8, 2)):
        net.set_engine_option("lstm_wide_rows", rows)
        net.set_engine_option("lstm_wide_xcds", xcds)
        piped(3)
        ms = min(timed(piped, K), timed(piped, K))
        print("pipelined rows=%2d xcds=%d       : %.3f ms  %.1f panoramas/s  status %d" % (rows, xcds, ms, B / ms * 1e3, net.hip_status(dev)))

net.precision = "f32"
with torch.no_grad():
    plain(2)
    ms = timed(plain, 8)
    print("plain f32 forward             : %.3f ms  %.1f panoramas/s" % (ms, B / ms * 1e3))
    piped(2)
    ms = min(timed(piped, 8), timed(piped, 8))
    print("pipelined f32 forward         : %.3f ms  %.1f panoramas/s  status %d" % (ms, B / ms * 1e3, net.hip_status(dev)))
net.precision = "bf16"

# the recurrence kernels alone (one layer, B = 32)
L = _lib.load()
T = 256
gx = ((torch.rand(T * B, 4096) - 0.5) * 2).to(dev)
w = [((torch.rand(2048, 512) - 0.5) * 0.12).to(torch.bfloat16).to(dev) for _ in range(2)]
y = torch.empty(T * B, 1024, device=dev)
yh = torch.empty(T * B, 1024, dtype=torch.bfloat16, device=dev)
sync = torch.zeros(4096, dtype=torch.uint8, device=dev)
xn = torch.empty(L.hn_lstm_bf16_exchange_bytes(), dtype=torch.uint8, device=dev)
P, sp = _lib.ptr, _lib.stream_ptr


def narrow(n):
    for _ in range(n):
        _lib.check(L.hn_lstm_layer_bf16(P(gx), P(w[0]), P(w[1]), P(y), P(yh), T, B, P(xn), P(sync), sp(dev)), "narrow")


narrow(2)
print("narrow recurrence (256 CUs)   : %.3f ms per layer" % timed(narrow, 10))
for rows, xcds in ((16, 1), (16, 2), (8, 1), (8, 2)):
    def wide(n):
        for _ in range(n):
            _lib.check(L.hn_lstm_layer_bf16_wide(P(gx), P(w[0]), P(w[1]), P(y), P(yh), T, B, P(sync), rows, xcds, sp(dev)), "wide")
    wide(2)
    print("wide rows=%2d xcds=%d (%3d CUs)  : %.3f ms per layer  status %d" % (rows, xcds, 2 * (B // rows) * 8, timed(wide, 10),
                                                                             int(sync.view(torch.int32)[512])))

wf = [((torch.rand(2048, 512) - 0.5) * 0.12).to(dev) for _ in range(2)]


def n32(n):
    for _ in range(n):
        _lib.check(L.hn_lstm_layer(P(gx), P(wf[0]), P(wf[1]), P(y), T, B, P(sync), sp(dev)), "f32")


def w32(n):
    for _ in range(n):
        _lib.check(L.hn_lstm_layer_wide(P(gx), P(wf[0]), P(wf[1]), P(y), T, B, P(sync), sp(dev)), "f32 wide")


n32(2)
print("f32 recurrence (256 CUs)      : %.3f ms per layer" % timed(n32, 10))
w32(2)
print("f32 wide recurrence (64 CUs)  : %.3f ms per layer  status %d" % (timed(w32, 10), int(sync.view(torch.int32)[512])))
