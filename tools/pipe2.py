"""Throughput with two forwards in flight (two HIP streams, two workspaces): the latency-bound persistent LSTM of
one batch overlaps the convolutions of the next."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from horizonnet_amd import HorizonNet
from oracle.weights import make_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = 32
dev = torch.device("cuda:0")
sd = make_state_dict(0, "random")
nets = []
for i in range(2):
    n = HorizonNet("resnet50", True); n.load_state_dict(sd); n = n.to(dev).eval(); n.precision = prec; nets.append(n)
x = torch.rand(B, 3, 512, 1024).to(dev)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
with torch.no_grad():
    for i in range(2):
        with torch.cuda.stream(streams[i]): nets[i](x)
    torch.cuda.synchronize()
    for mode in ("1 stream", "2 streams"):
        K = 20
        t0 = time.perf_counter()
        for i in range(K):
            j = i % 2 if mode == "2 streams" else 0
            with torch.cuda.stream(streams[j]): out = nets[j](x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s %s: %.3f ms/step, %.1f panos/s, status %d %d" % (prec, mode, dt / K * 1e3, B * K / dt, nets[0].hip_status(dev), nets[1].hip_status(dev)))
