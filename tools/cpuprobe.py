import os, time, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; free -g | head -2")
from oracle import horizonnet_ref
from oracle.weights import make_state_dict
sd = make_state_dict(0, "random")
x = torch.rand(1, 3, 512, 1024)
for nt in (torch.get_num_threads(), 8, 16, 32, 64):
    torch.set_num_threads(nt)
    horizonnet_ref.forward(x, sd)
    t = time.time(); horizonnet_ref.forward(x, sd); dt = time.time() - t
    print("threads %3d: oracle forward B=1 %.3f s" % (nt, dt), flush=True)
    if dt > 20: break
