"""Where do the phases of the ping-pong conv kernel spend their time?  (MEASUREMENT TOOL; needs tools/probe/pp_abl_stamp.so:
    tools/pp_ablate.sh build stamp)
Every wave of the first 64 workgroups stamps s_memtime at the start / end of each load block and each MFMA block of its first 24
chunks.  Printed per phase and wave group (cycles, mean over chunks 6..21 and workgroups): load block, wait at the barrier in front
of the MFMA block, MFMA block, wait at the barrier behind it; plus the chunk period.
    STAMP_ONLY=ghc1.0,layer3.x.conv1 python tools/pp_stamps.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from horizonnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("TRACE_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "pp_abl_stamp.so"))
from tools.conv_sweep import SHAPES, B, DEV  # noqa: E402

L = _lib.load()
NCH = 24
trace = torch.zeros(64 * 8 * NCH * 16, dtype=torch.int64, device=DEV)
L.hn_debug_conv_trace.argtypes = [ctypes.c_void_p]
L.hn_debug_conv_trace(_lib.ptr(trace))
os.environ["HN_BF16_W8"] = "4"


def run(shape):
    name, Hi, Wi, cin, cout, k, sh, sw, res = shape
    out_f32 = name.startswith("lstm")
    Bx = 1 if out_f32 else B
    Ho = (Hi + 2 * (k // 2) - k) // sh + 1
    Wo = (Wi + 2 * (k // 2) - k) // sw + 1
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.rand(Bx, Hi, Wi, cin, generator=g) - 0.5).to(DEV).to(torch.bfloat16)
    w = ((torch.rand(cout, cin, k, k, generator=g) - 0.5) / (cin * k * k) ** 0.5).to(DEV)
    wp = torch.empty(cout * k * k * cin, dtype=torch.bfloat16, device=DEV)
    scale = torch.ones(cout, device=DEV)
    shift = torch.zeros(cout, device=DEV)
    r = (torch.rand(Bx, Ho, Wo, cout, generator=g) - 0.5).to(DEV).to(torch.bfloat16) if res else None
    y = torch.empty(Bx, Ho, Wo, cout, dtype=torch.float32 if out_f32 else torch.bfloat16, device=DEV)
    sp = _lib.stream_ptr(DEV)

    def call(wsrc):
        _lib.check(L.hn_conv2d_nhwc_bf16(_lib.ptr(x), _lib.ptr(wsrc) if wsrc is not None else None, _lib.ptr(wp), _lib.ptr(scale),
                                         _lib.ptr(shift), _lib.ptr(r), _lib.ptr(y), Bx, Hi, Wi, cin, cout, k, k, sh, sw, 1, int(out_f32), sp), "conv")
    call(w)
    for _ in range(3):
        call(None)
    torch.cuda.synchronize()
    trace.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(None)
    e1.record()
    torch.cuda.synchronize()
    nk = cin * k * k // 64
    return e0.elapsed_time(e1) * 1e3, trace.cpu().numpy().reshape(64, 8, NCH, 4, 4).astype(np.float64), nk


def analyse(name, us, t, nk):
    # t[wg, wave, chunk, phase, (c: load start, d: load end, a: mfma start, b: mfma end)]
    lo, hi = int(os.environ.get("STAMP_LO", "6")), int(os.environ.get("STAMP_HI", str(NCH - 2)))
    ok = (t[:, :, :hi + 1, :, 0] != 0).all(axis=(1, 2, 3))
    t = t[ok]
    if len(t) == 0:
        print(name, "no stamps")
        return
    print("%-22s event %.1f us, %d chunks per tile, %d workgroups stamped" % (name[:22], us, nk, len(t)))
    for grp, waves in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
        x = t[:, waves, lo:hi]                                # [wg, wave, chunk, phase, 4]
        nxt_c = np.concatenate([x[:, :, :, 1:, 0], t[:, waves, lo + 1:hi + 1, :1, 0]], axis=3)   # start of the following load block
        load = x[..., 1] - x[..., 0]
        w1 = x[..., 2] - x[..., 1]
        mf = x[..., 3] - x[..., 2]
        w2 = nxt_c - x[..., 3]
        period = (t[:, waves, lo + 1:hi + 1, 0, 0] - t[:, waves, lo:hi, 0, 0]).mean()
        print("  %s: chunk period %.0f cycles" % (grp, period))
        for ph in range(4):
            print("    phase %d: load %5.0f | barrier %5.0f | mfma %5.0f | barrier %5.0f   (p10-p90 of load %4.0f-%4.0f, mfma %4.0f-%4.0f)" % (
                ph + 1, load[..., ph].mean(), w1[..., ph].mean(), mf[..., ph].mean(), w2[..., ph].mean(),
                np.percentile(load[..., ph], 10), np.percentile(load[..., ph], 90), np.percentile(mf[..., ph], 10), np.percentile(mf[..., ph], 90)))
    # alignment of the two groups: offset between group 0's and group 1's MFMA starts inside one workgroup
    d = (t[:, 4:8, lo:hi, :, 2].mean(axis=1) - t[:, 0:4, lo:hi, :, 2].mean(axis=1))
    print("  group 1's MFMA blocks start %.0f cycles after group 0's (same phase)" % d.mean())


def main():
    only = [s for s in os.environ.get("STAMP_ONLY", "ghc1.0,layer3.x.conv1,layer3.x.conv3").split(",") if s]
    for shp in SHAPES:
        if only and not any(s in shp[0] for s in only):
            continue
        us, t, nk = run(shp)
        analyse(shp[0], us, t, nk)


if __name__ == "__main__":
    main()
