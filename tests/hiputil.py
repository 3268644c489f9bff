"""Helpers for the GPU parity tests: call the C ABI (libhorizonnet_hip.so) on torch device tensors."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

from horizonnet_amd import _lib

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib():
    return _lib.load()


def sp():
    return _lib.stream_ptr(torch.device(DEV))


def P(t):
    return _lib.ptr(t)


def conv_hip(x_nhwc, w_oihw, bias, bn, stride, relu, residual=None):
    """x_nhwc: host float32 [B,H,W,Cin]; returns host [B,Ho,Wo,Cout] through hn_pack_conv_weight/hn_fold_bn/hn_conv2d_nhwc."""
    L = lib()
    x = x_nhwc.to(DEV).contiguous()
    w = w_oihw.to(DEV).contiguous()
    cout, cin, kh, kw = w.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    B, Hi, Wi, _ = x.shape
    Ho = (Hi + 2 * (kh // 2) - kh) // sh + 1
    Wo = (Wi + 2 * (kw // 2) - kw) // sw + 1
    wp = torch.empty(L.hn_packed_conv_weight_floats(cout, cin, kh, kw), dtype=torch.float32, device=DEV)
    _lib.check(L.hn_pack_conv_weight(P(w), P(wp), cout, cin, kh, kw, sp()), "pack")
    scale = torch.empty(cout, device=DEV)
    shift = torch.empty(cout, device=DEV)
    bd = None if bias is None else bias.to(DEV)
    if bn is None:
        _lib.check(L.hn_fold_bn(None, None, None, None, P(bd), P(scale), P(shift), cout, sp()), "fold")
    else:
        g, b, m, v = (t.to(DEV) for t in bn)
        _lib.check(L.hn_fold_bn(P(g), P(b), P(m), P(v), P(bd), P(scale), P(shift), cout, sp()), "fold")
    res = None if residual is None else residual.to(DEV).contiguous()
    y = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(L.hn_conv2d_nhwc(P(x), P(wp), P(scale), P(shift), P(res), P(y), B, Hi, Wi, cin, cout, kh, kw, sh, sw,
                                int(relu), sp()), "conv")
    torch.cuda.synchronize()
    return y.cpu()


def report(name, got, want, tol):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    d = np.abs(got - want)
    nanc = int(np.isnan(got).sum())
    mx = float(np.nanmax(d)) if d.size else 0.0
    scale = float(np.abs(want).max()) if want.size else 0.0
    print("[parity] %-40s max-abs %.3e (|ref|max %.3e, tol %.1e, nan %d)" % (name, mx, scale, tol, nanc))
    return nanc == 0 and mx <= tol


def bench_ranks(extra, timeout=1200, ranks=2):
    """`python bench.py --gpus 2 ...` WITHOUT torchrun's environment: bench.py becomes the launcher (the re-exec branch the
    driver's N = 2, 4, 8 runs go through), both ranks share the box's one GPU, gloo stands in for RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # the child processes need the device memory this process's allocator is sitting on (and the parent's later allocations then come from
    # fresh mappings the children left their data in: tests/conftest.py poisons them before the next test)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    # no retry: a rank that aborts is a failure of the code under test; its stderr is the evidence
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--share-gpu"] + extra,
                         capture_output=True, text=True, timeout=timeout, env=env)
    if out.returncode != 0:
        print("[rig] %d-rank launch failed (rc %d); stderr tail:\n%s" % (ranks, out.returncode, out.stderr[-6000:]))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-6000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line (rank 0 only): %d" % len(lines)
    return json.loads(lines[0])
