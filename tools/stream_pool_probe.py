"""Does torch's stream pool cost the pipelined forward its overlap?  

The engine's trunk / branch / head streams are HIP streams the library creates itself; the HIP runtime multiplexes ALL of a process's streams onto a
small set of hardware queues (GPU_MAX_HW_QUEUES, default 4).  torch's first side stream creates its pool of 32 streams per priority, and
ProcessGroupNCCL takes its collective streams from that pool, so a process that merely initialised RCCL can end up with two engine streams
sharing one hardware queue -- where their kernels serialise.

    python tools/stream_pool_probe.py MODE [dtype=bf16] [steps=40]        MODE: clean | pool_first | pool_after | nccl_first | nccl_init_only | nccl_destroyed | nccl_after | gloo_first | used1 | used4 | usedhi |
          nccl_big | nccl_big_f32first | f32first | nccl_bench | nccl_bench_nodev | nccl_bench_initonly   (nccl_bench* = init_process_group with / without device_id)

prints TWO lines: plain-forward panoramas/s + B = 1 latency; mode, GPU_MAX_HW_QUEUES, pipelined panoramas/s at B = 32.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "clean"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    keep = []
    if mode == "pool_first":
        keep = [torch.cuda.Stream(device=dev) for _ in range(32)] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(4)]
    def init_pg(backend, collective=True):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend, rank=0, world_size=1)
        if collective:
            t = torch.ones(1 << 20, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            torch.cuda.synchronize()
        return dist

    if mode == "nccl_first":
        init_pg("nccl")
    if mode.startswith("nccl_bench"):    # bench.py --force-rccl's exact preamble: eager init bound to the device, all_gather_object, barrier
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {} if mode.endswith("_nodev") else {"device_id": dev}
        dist.init_process_group(backend="nccl", rank=0, world_size=1, timeout=datetime.timedelta(minutes=30), **kw)
        if not mode.endswith("_initonly"):
            names = [None]
            dist.all_gather_object(names, "rank 0")
            dist.barrier()
        torch.cuda.synchronize()
    if mode in ("nccl_big", "nccl_big_f32first", "f32first"):               # what bench.py --force-rccl does before the first forward: large all-reduces, barriers, a MAX reduce
        dist_ = init_pg("nccl") if mode != "f32first" else None
        big = torch.ones(81_500_000, device=dev) if dist_ else None
        if dist_:
            for _ in range(4):
                dist_.all_reduce(big)
            dist_.barrier()
            t_ = torch.tensor([1.0], dtype=torch.float64, device=dev)
            dist_.all_reduce(t_, op=dist_.ReduceOp.MAX)
            torch.cuda.synchronize()
            del big
    if mode == "nccl_init_only":
        init_pg("nccl", collective=False)
    if mode == "nccl_destroyed":
        init_pg("nccl").destroy_process_group()
        torch.cuda.synchronize()
    if mode == "gloo_first":
        init_pg("gloo")
    if mode.startswith("used"):          # used1 / used4 / usedhi: side streams that actually RAN something before the engine exists
        n = 4 if mode == "used4" else 1
        keep = [torch.cuda.Stream(device=dev, priority=(-1 if mode == "usedhi" else 0)) for _ in range(n)]
        for st_ in keep:
            with torch.cuda.stream(st_):
                torch.ones(1 << 20, device=dev).mul_(2.0)
        torch.cuda.synchronize()
    from horizonnet_amd import HorizonNet
    net = HorizonNet("resnet50", True).to(dev).eval()
    net.precision = dtype
    x = torch.rand(32, 3, 512, 1024, device=dev)

    def run(n):
        pend = None
        for _ in range(n):
            nxt = net.forward_async(x)
            if pend is not None:
                pend.result()
            pend = nxt
        return pend.result()

    with torch.no_grad():
        if mode.endswith("f32first"):    # bench.py's order: the float32 pipelined forward (head stream, no branch stream) runs BEFORE the bf16 leg
            net.precision = "f32"
            run(3)
            for _ in range(2):
                net(x)
            torch.cuda.synchronize()
            net.precision = dtype
        run(3)
        torch.cuda.synchronize()
        if mode == "pool_after":
            keep = [torch.cuda.Stream(device=dev) for _ in range(32)]
            run(2)
            torch.cuda.synchronize()
        if mode == "nccl_after":         # the engine's streams exist and have run; THEN the process group + one collective
            init_pg("nccl")
            run(2)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the plain (one call per batch) forward and the B = 1 latency: the paths that use the branch stream without the head stream
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
        torch.cuda.synchronize()
        plain = 32 * steps / (time.perf_counter() - t0)
        x1 = x[:1].contiguous()
        for _ in range(5):
            net(x1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            net(x1)
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 50 * 1e3
    assert net.hip_status(dev) == 0
    print("[streams] mode %-10s dtype %s: plain forward %.1f panoramas/s, B = 1 latency %.3f ms" % (mode, dtype, plain, lat), flush=True)
    print("[streams] mode %-10s dtype %s GPU_MAX_HW_QUEUES=%s HN_HW_QUEUES=%s: %.1f panoramas/s (%.3f ms per batch; host submit %.3f ms per batch)"
          % (mode, dtype, os.environ.get("GPU_MAX_HW_QUEUES", "-"), os.environ.get("HN_HW_QUEUES", "-"), 32 * steps / dt, dt / steps * 1e3,
             th / steps * 1e3), flush=True)
    del keep


if __name__ == "__main__":
    main()
