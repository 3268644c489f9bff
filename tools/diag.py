"""GPU diagnostic: wall-clock of every host step and HIP-event time of every launch group."""
import os, sys, time
t00 = time.time()
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def lap(msg, t=[time.time()]):
    now = time.time(); print("[%7.2fs +%6.2fs] %s" % (now - t00, now - t[0], msg), flush=True); t[0] = now
lap("import torch")
from horizonnet_amd import HorizonNet
from oracle.weights import make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
sd = make_state_dict(0, "random"); lap("make_state_dict")
net = HorizonNet("resnet50", True); lap("construct module")
net.load_state_dict(sd); net = net.to("cuda:0").eval(); net.precision = PREC; torch.cuda.synchronize(); lap("to(cuda)")
x = torch.rand(B, 3, 512, 1024).to("cuda:0"); torch.cuda.synchronize(); lap("input B=%d" % B)
with torch.no_grad():
    net(x); torch.cuda.synchronize(); lap("first forward (pack + load code objects)")
    net(x); torch.cuda.synchronize(); lap("second forward")
    _, _, ent = net.profile_forward(x); lap("profiled forward")
print("status", net.hip_status("cuda:0"))
tot = sum(e[1] for e in ent)
print("sum of launch groups: %.3f ms" % tot)
for name, ms, fl in sorted(ent, key=lambda e: -e[1])[:25]:
    print("  %-70s %9.3f ms  %7.2f TF/s" % (name, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/profile_B%d%s.txt" % (B, "" if PREC == "f32" else "_" + PREC), "w") as f:
    f.write("# hn_forward launch groups, B=%d, HIP events; total %.3f ms\n" % (B, tot))
    for name, ms, fl in ent:
        f.write("%-70s %9.3f ms %8.2f TF/s %6.2f%%\n" % (name, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, 100 * ms / tot))
