"""Where does a conv workgroup's time go?  (MEASUREMENT TOOL, needs a -DHN_CONV_TRACE build of the library:
    HN_BUILD_OUT=$PWD/tools/probe/ab_b.so HN_BUILD_OBJ=$PWD/build/obj_trace HN_BUILD_FLAGS=-DHN_CONV_TRACE bash horizonnet_amd/csrc/build.sh)

The traced kernels stamp s_memrealtime (100 MHz) at workgroup entry, after the first K chunk has landed, after the k loop and
after every epilogue band, plus HW_ID / XCC_ID.  Per shape: kernel span, mean phase durations, workgroups per CU, the idle
gap between consecutive workgroups of one CU, and how long the first wave of workgroups takes to start.
    TRACE_ONLY=layer3.x HN_BF16_W8=-1 python tools/conv_trace.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from horizonnet_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("TRACE_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "ab_b.so"))
from tools.conv_sweep import SHAPES, B, DEV  # noqa: E402

L = _lib.load()
NWG = 1 << 16
trace = torch.zeros(NWG * 8, dtype=torch.int64, device=DEV)
L.hn_debug_conv_trace.argtypes = [__import__("ctypes").c_void_p]
L.hn_debug_conv_trace(_lib.ptr(trace))


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
    t = trace.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    return e0.elapsed_time(e1) * 1e3, t


def analyse(name, us_event, t):
    n = len(t)
    hw = t[:, 7]
    lo = hw & 0xFFFFFFFF
    xcc = (hw >> 32) & 0xF
    cu = (xcc << 8) | (((lo >> 13) & 3) << 5) | (((lo >> 12) & 1) << 4) | ((lo >> 8) & 0xF)
    stamps = t[:, :7].astype(np.float64) * 0.01          # microseconds
    nb = int((t[0, 3:7] != 0).sum())                     # epilogue bands stamped
    t_end = stamps[:, 2 + nb]
    t0 = stamps[:, 0].min()
    span = t_end.max() - t0
    ph = [stamps[:, 1] - stamps[:, 0], stamps[:, 2] - stamps[:, 1]] + [stamps[:, 3 + i] - stamps[:, 2 + i] for i in range(nb)]
    tot = t_end - stamps[:, 0]
    cus = np.unique(cu)
    gaps, conc = [], []
    for c in cus:
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(stamps[idx, 0])]
        s_, e_ = stamps[o, 0], t_end[o]
        # concurrency on this CU: average number of resident workgroups over its busy interval
        conc.append((e_ - s_).sum() / max(e_.max() - s_.min(), 1e-9))
        for a in range(1, len(o)):
            gaps.append(s_[a] - e_[:a].max())
    starts = np.sort(stamps[:, 0] - t0)
    first_wave = starts[min(len(cus), n) - 1]
    print("%-24s wgs %5d on %3d CUs | event %7.1f us, span %7.1f | per wg: total %6.2f = first chunk %5.2f + k loop %5.2f + bands %s | resident/CU %.2f | "
          "gap between wgs on a CU %5.2f (max %5.2f) | first %d wgs started within %5.2f us | tail: last wg ends %.1f after the median end" % (
              name[:24], n, len(cus), us_event, span, tot.mean(), ph[0].mean(), ph[1].mean(), "+".join("%.2f" % p.mean() for p in ph[2:]),
              float(np.mean(conc)), float(np.mean(gaps)) if gaps else 0.0, float(np.max(gaps)) if gaps else 0.0, min(len(cus), n), first_wave,
              t_end.max() - np.median(t_end)))


def main():
    only = [s for s in os.environ.get("TRACE_ONLY", "layer2.x,layer3.x,layer4.x,ghc1.0,ghc2.0").split(",") if s]
    for shp in SHAPES:
        if only and not any(s in shp[0] for s in only):
            continue
        us, t = run(shp)
        analyse(shp[0], us, t)


if __name__ == "__main__":
    main()
