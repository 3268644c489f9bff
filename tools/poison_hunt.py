"""Hunt for reads of memory an engine entry never wrote (the float32 training forward that came out 0.12 off after a bf16 step
in one full-suite run).

A sequence -- bf16 training step (forward + backward), then two float32 training forwards on the same net / workspace -- is repeated with the
memory the caching allocator hands out pre-filled with different byte patterns (0x00, 0xFF = NaN, 0x7F = 3.4e38) and with the engine's own poison
instrument (option "poison_ws": every workspace / scratch / packed-weight / gradient range filled before each entry) off / NaN / 0x7F.  Every
deterministic result must be the SAME BITS in all runs; the per-unit checksums of the float32 forward name the first tensor that is not.

    python tools/poison_hunt.py [B=1] [fold=1]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from horizonnet_amd import HorizonNet, _lib                     # noqa: E402
from oracle.weights import make_state_dict                       # noqa: E402

DEV = torch.device("cuda:0")


def checksum(t):
    """order-independent exact checksum of a byte range (int64 sum of its int32 words) + NaN count when read as f32"""
    w = t.view(torch.int32)
    return int(w.to(torch.int64).sum().item())


def unit_info(B):
    L = _lib.load()
    out8 = (ctypes.c_int64 * 8)()
    L.hn_train_debug_unit(B, 0, out8)
    info = []
    for u in range(out8[7]):
        L.hn_train_debug_unit(B, u, out8)
        info.append(tuple(out8) + (int(L.hn_train_debug_unit_yh(B, u)),))
    return info


def sequence(B, fold, alloc_byte, engine_poison, sd, x, wb, wc, info):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    if alloc_byte is not None:       # what torch.empty() returns next comes out of this block
        blk = torch.empty(12 << 30, dtype=torch.uint8, device=DEV)
        blk.fill_(alloc_byte)
        torch.cuda.synchronize()
        del blk
    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    net.set_engine_option("fuse_bn_fold", fold)
    net.set_engine_option("poison_ws", engine_poison)
    out = {}
    xd = x.to(DEV)
    net.train_precision = "bf16"
    bon, cor = net(xd)
    ((bon * wb.to(DEV)).sum() + (cor * wc.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    out["bf16.bon"] = checksum(bon.detach())
    out["bf16.cor"] = checksum(cor.detach())
    out["bf16.nan"] = int(torch.isnan(bon).sum() + torch.isnan(cor).sum())
    st = net._hip_states[0]
    ws8 = st.train_workspace(B)
    for u, t in enumerate(info):
        if u == 0:
            continue                   # the stem's bf16 activation is consumed inside the fused pool pass, never stored
        n = t[1] * t[2]
        out["bf16.y%03d" % u] = checksum(ws8[t[8] * 4:t[8] * 4 + 2 * n])
    gn = {}
    nan_g = 0
    for k, p in net.named_parameters():
        g = p.grad
        nan_g += int(torch.isnan(g).sum())
        gn[k] = float(g.double().norm())
    out["bf16.grad_nan"] = nan_g
    out["_gradnorm"] = gn
    net.zero_grad(set_to_none=True)
    net.train_precision = "f32"
    for rep in ("a", "b"):
        b32, c32 = net(xd)
        torch.cuda.synchronize()
        assert net.hip_status(DEV) == 0
        out["f32%s.bon" % rep] = checksum(b32.detach())
        out["f32%s.cor" % rep] = checksum(c32.detach())
        out["f32%s.nan" % rep] = int(torch.isnan(b32).sum() + torch.isnan(c32).sum())
        ws = ws8.view(torch.float32)
        for u, t in enumerate(info):
            n = t[1] * t[2]
            out["f32%s.z%03d" % (rep, u)] = checksum(ws[t[4]:t[4] + n])
            out["f32%s.y%03d" % (rep, u)] = checksum(ws[t[5]:t[5] + n])
        out["_f32%s" % rep] = b32.detach().cpu()
    del net, st, ws8, ws
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    fold = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sd = make_state_dict(11, "random")
    g = torch.Generator().manual_seed(12)
    x = torch.rand(B, 3, 512, 1024, generator=g)
    wb = torch.randn(B, 2, 1024, generator=torch.Generator().manual_seed(13))
    wc = torch.randn(B, 1, 1024, generator=torch.Generator().manual_seed(14))
    info = unit_info(B)
    runs = [("alloc00", 0x00, 0), ("allocFF", 0xFF, 0), ("alloc7F", 0x7F, 0), ("alloc00+engFF", 0x00, 1), ("allocFF+eng7F", 0xFF, 2),
            ("alloc00 again", 0x00, 0)]
    res = []
    for name, ab, ep in runs:
        r = sequence(B, fold, ab, ep, sd, x, wb, wc, info)
        res.append((name, r))
        print("[hunt] B=%d fold=%d %-16s bf16 nan %d grad-nan %d | f32 nan %d / %d" % (B, fold, name, r["bf16.nan"], r["bf16.grad_nan"], r["f32a.nan"],
                                                                                    r["f32b.nan"]), flush=True)
    base_name, base = res[0]
    bad = False
    for name, r in res[1:]:
        diff = sorted(k for k in base if not k.startswith("_") and base[k] != r[k])
        d32 = float((r["_f32a"] - base["_f32a"]).abs().max())
        d32b = float((r["_f32b"] - r["_f32a"]).abs().max())
        gd = max(abs(r["_gradnorm"][k] - base["_gradnorm"][k]) / (base["_gradnorm"][k] + 1e-30) for k in base["_gradnorm"])
        print("[hunt] %-16s vs %s: %d checksums differ; f32 bon max-abs diff %.3e; f32 repeat diff %.3e; worst gradient-norm rel diff %.3e"
              % (name, base_name, len(diff), d32, d32b, gd))
        if diff:
            print("[hunt]    first differing: %s" % ", ".join(diff[:12]))
        rels = sorted(((abs(r["_gradnorm"][k] - base["_gradnorm"][k]) / (base["_gradnorm"][k] + 1e-30), k) for k in base["_gradnorm"]), reverse=True)
        names = list(base["_gradnorm"].keys())
        last = [k for k in names if abs(r["_gradnorm"][k] - base["_gradnorm"][k]) / (base["_gradnorm"][k] + 1e-30) > 1e-6]
        print("[hunt]    gradient norms that moved by > 1e-6: %d of %d; the LAST such tensor in state_dict order (= the first the backward produces): %s; top: %s"
              % (len(last), len(names), last[-1] if last else "-", ", ".join("%s %.1e" % (k[-40:], v) for v, k in rels[:3])))
        det = [k for k in diff if k.startswith("f32") or k.startswith("bf16.y") or k in ("bf16.bon", "bf16.cor")]
        bad = bad or bool(det)
    print("[hunt] RESULT: %s" % ("DIFFERENCES in deterministic tensors" if bad else "all deterministic tensors bit-identical across poison patterns"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
