import sys, time, torch
sys.path.insert(0, '.')
from horizonnet_amd import HorizonNet
from oracle.weights import make_state_dict
net = HorizonNet("resnet50", True); net.load_state_dict(make_state_dict(0, "random")); net = net.to("cuda:0").eval()
for prec in ("f32", "bf16"):
    net.precision = prec
    for B in (1, 2, 4):
        x = torch.rand(B, 3, 512, 1024).to("cuda:0")
        with torch.no_grad():
            for _ in range(5): net(x)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): net(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
            _, _, ent = net.profile_forward(x) if prec == "f32" else (None, None, [])
        print(prec, "B=%d wall %.2f ms/forward (%.0f panos/s)" % (B, dt * 1e3, B / dt), "gpu-sum %.2f ms" % sum(e[1] for e in ent) if ent else "")
