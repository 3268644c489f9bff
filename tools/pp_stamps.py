"""Where do the phases of the ping-pong conv kernel spend their time?  (MEASUREMENT TOOL; needs tools/probe/pp_abl_stamp1.so and
pp_abl_stamp2.so: tools/pp_ablate.sh build "stamp1 stamp2")
Every wave of the first 64 workgroups stamps s_memtime at every barrier RELEASE (build 1: the starts of a chunk's four load blocks
and four MFMA blocks) or at every barrier ARRIVAL (build 2: their ends) for its first 24 chunks; the eight values of a chunk stay in
SGPRs and leave once per chunk.  Printed (cycles, mean over workgroups and chunks STAMP_LO..STAMP_HI): the eight barrier-to-barrier
intervals of a chunk with what each wave group does in them, and -- from build 2 against build 1 of the same wave -- how long each
block runs and how long the wave then waits at the barrier.
    STAMP_ONLY=ghc1.0,layer3.x.conv1 python tools/pp_stamps.py
"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

NCH = 24
HERE = os.path.dirname(os.path.abspath(__file__))


def child(kind):
    import torch
    from horizonnet_amd import _lib
    _lib.LIB_PATH = os.path.join(HERE, "probe", "pp_abl_stamp%d.so" % kind)
    from tools.conv_sweep import SHAPES, B, DEV
    L = _lib.load()
    trace = torch.zeros(64 * 8 * NCH * 8, dtype=torch.int64, device=DEV)
    L.hn_debug_conv_trace.argtypes = [ctypes.c_void_p]
    L.hn_debug_conv_trace(_lib.ptr(trace))
    os.environ["HN_BF16_W8"] = os.environ.get("STAMP_VARIANT", "4")
    only = [s for s in os.environ.get("STAMP_ONLY", "ghc1.0,layer3.x.conv1").split(",") if s]
    out = {}
    for shape in SHAPES:
        name, Hi, Wi, cin, cout, k, sh, sw, res = shape
        if only and not any(s in name for s in only):
            continue
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
        out[name] = (e0.elapsed_time(e1) * 1e3, trace.cpu().numpy().reshape(64, 8, NCH, 8).copy(), cin * k * k // 64)
    np.save(os.path.join(os.environ.get("TMPDIR", "/tmp"), "pp_stamps_%d.npy" % kind), out, allow_pickle=True)


BLOCKS = ["load 1", "mfma 1", "load 2", "mfma 2", "load 3", "mfma 3", "load 4", "mfma 4"]


def analyse(name, d1, d2):
    us1, t1, nk = d1
    us2, t2, _ = d2
    lo, hi = int(os.environ.get("STAMP_LO", "6")), int(os.environ.get("STAMP_HI", str(NCH - 3)))
    t1 = t1.astype(np.float64)
    t2 = t2.astype(np.float64)
    ok = (t1[:, :, :hi + 2] != 0).all(axis=(1, 2, 3)) & (t2[:, :, :hi + 2] != 0).all(axis=(1, 2, 3))
    t1, t2 = t1[ok], t2[ok]
    if len(t1) == 0:
        print(name, "no complete stamp sets (fewer than %d chunks per workgroup?)" % (hi + 2))
        return
    print("%-22s event %.1f / %.1f us (release / arrival build), %d chunks per tile, %d workgroups" % (name[:22], us1, us2, nk, len(t1)))
    # build 1: start of block q of chunk c = release of a barrier; the next start is the next release
    st = t1[:, :, lo:hi + 1]                                           # [wg, wave, chunk, 8]
    nxt = np.concatenate([st[..., 1:], t1[:, :, lo + 1:hi + 2, :1]], axis=3)
    interval = nxt - st                                                # barrier release -> next release, per wave
    print("  chunk period %.0f cycles" % (t1[:, :, lo + 1:hi + 2, 0] - t1[:, :, lo:hi + 1, 0]).mean())
    # build 2 (another run of the same code): block length = arrival - release is not available across builds per wave and chunk;
    # use means: mean arrival offset inside the chunk against mean release offset inside the chunk
    rel1 = (t1[:, :, lo:hi + 1] - t1[:, :, lo:hi + 1, :1])              # release offsets inside a chunk (from the chunk's first release)
    arr2 = (t2[:, :, lo:hi + 1] - t2[:, :, lo:hi + 1, :1])              # arrival offsets from the chunk's first ARRIVAL
    for grp, waves in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
        iv = interval[:, waves].mean(axis=(0, 1, 2))
        # per-wave spread of the block lengths: arrival(q) - arrival(0) - (release(q) - release(0)) + len(block 0) is unknown without a
        # common clock, so print the arrival-to-arrival distances too: they differ from release-to-release by the change of the wait
        av = np.concatenate([arr2[:, waves, :, 1:], (t2[:, waves, lo + 1:hi + 2, :1] - t2[:, waves, lo:hi + 1, :1])], axis=3) - arr2[:, waves]
        print("  %s: release->release per block: %s" % (grp, "  ".join("%s %4.0f" % (BLOCKS[q], iv[q]) for q in range(8))))
        print("           arrival->arrival        : %s" % "  ".join("%s %4.0f" % (BLOCKS[q], av.mean(axis=(0, 1, 2))[q]) for q in range(8)))
        # spread between the four waves of the group at each arrival (who is last?)
        a = t2[:, waves, lo:hi + 1]
        spread = (a.max(axis=1) - a.min(axis=1)).mean(axis=(0, 1))
        print("           arrival spread inside the group (max - min over its 4 waves): %s" % "  ".join("%4.0f" % v for v in spread))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        return
    for kind in (1, 2):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(kind)], check=True)
    tmp = os.environ.get("TMPDIR", "/tmp")
    d1 = np.load(os.path.join(tmp, "pp_stamps_1.npy"), allow_pickle=True).item()
    d2 = np.load(os.path.join(tmp, "pp_stamps_2.npy"), allow_pickle=True).item()
    for name in d1:
        analyse(name, d1[name], d2[name])


if __name__ == "__main__":
    main()
