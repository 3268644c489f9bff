"""What does HBM deliver to simple streaming kernels on this box?  (MEASUREMENT TOOL; torch's own kernels as the yardstick)
read-only (sum), copy (read + write), and a + b -> c (2 reads + 1 write) on 1 GiB bf16 / f32 tensors; GB/s of bytes moved."""
import torch
dev = torch.device("cuda:0")
def t(fn, nbytes, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return nbytes / ms / 1e6
for dt, nm in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
    n = (1 << 30) // torch.empty(0, dtype=dt).element_size()
    a = torch.randn(n, device=dev).to(dt); b = torch.randn(n, device=dev).to(dt); c = torch.empty_like(a)
    nb = a.numel() * a.element_size()
    print("%s 1 GiB: sum %.0f GB/s | copy %.0f GB/s | add %.0f GB/s | relu_ %.0f GB/s" % (
        nm, t(lambda: a.sum(), nb), t(lambda: c.copy_(a), 2 * nb), t(lambda: torch.add(a, b, out=c), 3 * nb), t(lambda: a.relu_(), 2 * nb)))
