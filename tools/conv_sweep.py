"""Per-layer tile sweep of the bf16 implicit-GEMM convolution on the network's real shapes (MEASUREMENT TOOL).

For every distinct conv shape of the B=32 forward it times hn_conv2d_nhwc_bf16 with the 4-wave kernels (HN_BF16_W8=0)
and each legal 8-wave tile (1: 256x256, 2: 256x128, 3: 128x256; 4 / 5: the ping-pong persistent 256x256 kernel), HIP events around `iters` back-to-back launches, and prints
TF/s + the winner -- the data behind the dispatch heuristic in conv_igemm_bf16.hip.  SWEEP_VARIANTS / SWEEP_ONLY select
columns / layers.  (Variants 4-7 of profiles/r2_conv_tile_sweep.txt were the phased kernels of commit 053652b, since removed.)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from horizonnet_amd import _lib  # noqa: E402

if os.environ.get("SWEEP_LIB"):      # a measurement build of the library (tools/pp_ablate.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["SWEEP_LIB"])
DEV = torch.device("cuda:0")
B = int(os.environ.get("SWEEP_B", "32"))

SHAPES = []   # name, Hi, Wi, Cin, Cout, k, sh, sw, residual


def add(name, Hi, Wi, cin, cout, k, sh, sw, res=False):
    SHAPES.append((name, Hi, Wi, cin, cout, k, sh, sw, res))


H, W, cin = 128, 256, 64
for li, (width, n) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
    s = 1 if li == 0 else 2
    add("layer%d.0.conv1" % (li + 1), H, W, cin, width, 1, 1, 1)
    add("layer%d.0.conv2" % (li + 1), H, W, width, width, 3, s, s)
    add("layer%d.0.downsample" % (li + 1), H, W, cin, width * 4, 1, s, s)
    H, W = H // s, W // s
    add("layer%d.x.conv3" % (li + 1), H, W, width, width * 4, 1, 1, 1, True)
    add("layer%d.x.conv1" % (li + 1), H, W, width * 4, width, 1, 1, 1)
    add("layer%d.x.conv2" % (li + 1), H, W, width, width, 3, 1, 1)
    c = width * 4
    gh = H
    for k_, co in enumerate((c // 2, c // 2, c // 4, c // 8)):
        add("ghc%d.%d" % (li, k_), gh, W, c if k_ == 0 else prev, co, 3, 2, 1)
        prev = co
        gh //= 2
    cin = width * 4
add("lstm.input_gemm(f32 out)", 1, 256 * B, 1024, 4096, 1, 1, 1)


def run(shape, variant, iters=10):
    name, Hi, Wi, cin, cout, k, sh, sw, res = shape
    L = _lib.load()
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
    os.environ["HN_BF16_W8"] = str(variant)
    sp = _lib.stream_ptr(DEV)

    def call(wsrc):
        _lib.check(L.hn_conv2d_nhwc_bf16(_lib.ptr(x), _lib.ptr(wsrc) if wsrc is not None else None, _lib.ptr(wp), _lib.ptr(scale),
                                         _lib.ptr(shift), _lib.ptr(r), _lib.ptr(y), Bx, Hi, Wi, cin, cout, k, k, sh, sw, 1, int(out_f32), sp), "conv")
    call(w)
    call(None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call(None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * Bx * Ho * Wo * cout * cin * k * k
    return ms, flops / (ms * 1e-3) / 1e12, y


def main():
    print("# bf16 conv tile sweep, B=%d; columns = HN_BF16_W8 variants %s (0: 4-wave kernels, 1: 256x256, 2: 256x128, 3: 128x256, -1: dispatch heuristic); ms, TF/s" % (B, os.environ.get("SWEEP_VARIANTS", "0,1,2,3,-1")))
    tot = {0: 0.0, "best": 0.0, "auto": 0.0}
    only = [t for t in os.environ.get("SWEEP_ONLY", "").split(",") if t]
    variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "0,1,2,3,-1").split(",")]
    for shp in SHAPES:
        name, Hi, Wi, cin, cout, k, sh, sw, res = shp
        if only and not any(t in name for t in only):
            continue
        row, best, ref = [], None, None
        for v in variants:
            legal = v in (0, -1) or (v in (1, 3) and cout % 256 == 0) or (v == 2 and cout % 128 == 0) or \
                (v in (4, 5) and cout % 256 == 0 and cin * k * k >= 128) or \
                (v == 6 and k == 3 and sw == 1 and cout % 256 == 0 and 32 <= Wi <= 128) or \
                (v == 7 and k == 3 and sw == 1 and cout % 128 == 0 and 32 <= Wi <= 256) or \
                (v == 8 and k == 3 and sw == 1 and cout % 64 == 0 and 32 <= Wi <= 256 and not res)      # 8: dw-reuse 512x64; 4 / 5: ping-pong kernel; 6 / 7: dw-reuse 3x3 kernel, 256x256 / 512x128 tiles
            if not legal:
                row.append("      -      ")
                continue
            ms, tf, y = run(shp, v)
            if v == 0:
                ref = y.float().clone()
                tot[0] += ms
            elif ref is not None:        # every other variant, the dispatch heuristic (-1: persistent 8-wave form) included
                err = float((y.float() - ref).abs().max())
                if os.environ.get("SWEEP_NOASSERT") and err != 0.0:
                    print("MISMATCH %s variant %d max-abs %.4g" % (name, v, err))
                else:
                    assert err == 0.0 or os.environ.get("HN_W8_ABL", "0") != "0", (name, v, err)   # same k order -> bit-identical to the 4-wave kernel
            if v == -1:
                tot["auto"] += ms
            elif best is None or ms < best[0]:
                best = (ms, v)
            row.append("%6.3f %6.0f" % (ms, tf))
        tot["best"] += best[0]
        print("%-28s Hi=%3d Wi=%3d %4d->%4d k%d s%d%d | %s | best %d" % (name, Hi, Wi, cin, cout, k, sh, sw, " | ".join(row), best[1]))
    print("# sum over distinct shapes (not weighted by repeats): 4-wave %.3f ms, best-per-layer %.3f ms, auto heuristic %.3f ms" % (tot[0], tot["best"], tot["auto"]))


if __name__ == "__main__":
    main()
