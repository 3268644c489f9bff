"""GPU parity tests of the training step (-m gpu): data/weight gradient kernels against torch autograd on
the host, and the whole train-mode forward + backward of HorizonNet against the oracle's autograd."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from horizonnet_amd import HorizonNet, _lib  # noqa: E402
from oracle import horizonnet_ref  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402
from oracle.hostinfo import usable_cores  # noqa: E402

from hiputil import DEV, P, lib, report, sp  # noqa: E402

torch.set_num_threads(usable_cores())


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _conv_ref(x_nhwc, w, stride):
    k = w.shape[2]
    x = x_nhwc.permute(0, 3, 1, 2)
    if k // 2:
        x = horizonnet_ref.lr_pad(x, k // 2)
    return F.conv2d(x, w, None, stride=stride, padding=(k // 2, 0)).permute(0, 2, 3, 1)


GRAD_CASES = [
    # name, B, H, W, Cin, Cout, k, stride
    ("1x1 s1", 2, 8, 16, 64, 128, 1, 1),
    ("3x3 s1", 2, 8, 16, 64, 64, 3, 1),
    ("3x3 s2", 1, 16, 32, 64, 128, 3, 2),
    ("3x3 s(2,1)", 2, 8, 16, 128, 64, 3, (2, 1)),
    ("1x1 s2", 1, 16, 32, 128, 256, 1, 2),
    ("3x3 s(2,1) H=2", 2, 2, 32, 64, 128, 3, (2, 1)),
    ("3x3 s(2,1) Cout=32 (ghc0.3)", 2, 16, 32, 64, 32, 3, (2, 1)),
    ("ghc3.3 shape 512->256", 2, 2, 32, 512, 256, 3, (2, 1)),
    ("ghc2.2 shape 512->256", 2, 8, 64, 512, 256, 3, (2, 1)),
    ("ghc1.2 shape 256->128", 2, 16, 128, 256, 128, 3, (2, 1)),
    ("layer4 conv2 512->512 s1", 2, 16, 32, 512, 512, 3, 1),
]


@pytest.mark.parametrize("case", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_conv_dgrad_wgrad_stage(case):
    name, B, H, W, cin, cout, k, stride = case
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    x = _rand((B, H, W, cin), 1).requires_grad_(True)
    w = _rand((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k)).requires_grad_(True)
    y = _conv_ref(x, w, (sh, sw))
    dz = _rand(tuple(y.shape), 3)
    add = _rand((B, H, W, cin), 4)
    y.backward(dz)
    L = lib()
    dzd, wd, addd = dz.contiguous().to(DEV), w.detach().to(DEV), add.to(DEV)
    dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
    scr = torch.empty(cout * cin * k * k + 8192, device=DEV)
    _lib.check(L.hn_conv2d_dgrad_nhwc(P(dzd), P(wd), P(addd), P(dx), P(scr), B, H, W, cin, cout, k, k, sh, sw, sp()), "dgrad")
    dw = torch.full((cout, cin, k, k), float("nan"), device=DEV)
    scr2 = torch.empty(cout * max(k * k * cin, 256), device=DEV)
    xd = x.detach().to(DEV)
    _lib.check(L.hn_conv2d_wgrad_nhwc(P(xd), P(dzd), P(dw), P(scr2), B, H, W, cin, cout, k, k, sh, sw, 0, sp()), "wgrad")
    torch.cuda.synchronize()
    ok = report("dgrad " + name, dx.cpu().numpy(), (x.grad + add).numpy(), 2e-5 * max(1.0, float(x.grad.abs().max())))
    ok &= report("wgrad " + name, dw.cpu().numpy(), w.grad.numpy(), 2e-5 * max(1.0, float(w.grad.abs().max())))
    assert ok


@pytest.mark.parametrize("case", [c for c in GRAD_CASES if c[5] % 64 == 0], ids=[c[0] for c in GRAD_CASES if c[5] % 64 == 0])
def test_conv_dgrad_bf16_stage(case):
    """Data gradient on the bf16 matrix cores == torch autograd on the bf16-rounded operands (products of bf16 values are
    exact in f32, so only the f32 summation order differs)."""
    name, B, H, W, cin, cout, k, stride = case
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    x = _rand((B, H, W, cin), 1).requires_grad_(True)
    w = _rand((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
    w16 = r16(w).requires_grad_(True)
    y = _conv_ref(x, w16, (sh, sw))
    dz = _rand(tuple(y.shape), 3)
    add = _rand((B, H, W, cin), 4)
    y.backward(r16(dz))
    dzd, wd, addd = dz.contiguous().to(DEV), w.to(DEV), add.to(DEV)
    dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
    scr = torch.empty(cout * cin * k * k + 8192 + dz.numel() // 2 + 64, device=DEV)
    _lib.check(lib().hn_conv2d_dgrad_nhwc_bf16(P(dzd), P(wd), P(addd), P(dx), P(scr), B, H, W, cin, cout, k, k, sh, sw, sp()), "dgrad bf16")
    torch.cuda.synchronize()
    assert report("dgrad bf16 " + name, dx.cpu().numpy(), (x.grad + add).numpy(), 3e-5 * max(1.0, float(x.grad.abs().max())))


WGRAD_BF16_CASES = [c for c in GRAD_CASES if c[4] % 64 == 0 and c[5] % 64 == 0] + [
    ("layer1 conv2 64->64 big M", 2, 64, 128, 64, 64, 3, 1),
    ("layer2.0 conv2 128->128 s2", 2, 32, 64, 128, 128, 3, 2),
    ("1x1 256->64", 2, 16, 64, 256, 64, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD_BF16_CASES, ids=[c[0] for c in WGRAD_BF16_CASES])
def test_conv_wgrad_bf16_stage(case):
    """Weight gradient on the bf16 matrix cores (transpose-read operands) == torch autograd on the bf16-rounded x and
    dz: products of bf16 values are exact in f32, only the summation order differs."""
    name, B, H, W, cin, cout, k, stride = case
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    x = _rand((B, H, W, cin), 1)
    w = _rand((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k)).requires_grad_(True)
    y = _conv_ref(r16(x), w, (sh, sw))
    dz = _rand(tuple(y.shape), 3)
    y.backward(r16(dz))
    xd, dzd = x.to(DEV), dz.contiguous().to(DEV)
    dw = torch.full((cout, cin, k, k), float("nan"), device=DEV)
    scr = torch.empty(cout * k * k * cin + (x.numel() + dz.numel()) // 2 + 256, device=DEV)
    _lib.check(lib().hn_conv2d_wgrad_nhwc_bf16(P(xd), P(dzd), P(dw), P(scr), B, H, W, cin, cout, k, k, sh, sw, sp()), "wgrad bf16")
    torch.cuda.synchronize()
    assert report("wgrad bf16 " + name, dw.cpu().numpy(), w.grad.numpy(), 3e-5 * max(1.0, float(w.grad.abs().max())))


@pytest.mark.parametrize("tile", ["0", "1", "2"])
def test_conv_wgrad_bf16_tap_reuse_kernel(tile, monkeypatch):
    """conv_wgrad_row3_bf16_kernel (3x3, stride 1 along W: the three taps of a filter row on one pass over the operands) forced on every
    eligible shape and every tile variant (the dispatcher only takes it where it wins on the B = 64 step): == torch autograd on the
    bf16-rounded operands, as the per-tap kernel in test_conv_wgrad_bf16_stage -- stride (2,1) with H = 2 (a filter row entirely in the
    zero padding), circular columns at both ends of an image row, several chunks per row, ragged m split."""
    monkeypatch.setenv("HN_WGRAD_ROW3", "1")
    monkeypatch.setenv("HN_WGRAD_ROW3_TILE", tile)
    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    cases = [("64->64 W=128", 2, 64, 128, 64, 64, (1, 1)), ("128->128 s(2,1) W=32 H=2", 2, 2, 32, 128, 128, (2, 1)), ("256->128 s(2,1) W=128", 2, 16, 128, 256, 128, (2, 1)),
             ("128->256 W=64", 3, 8, 64, 128, 256, (1, 1)), ("512->512 W=32", 2, 16, 32, 512, 512, (1, 1)), ("64->128 s(2,1) W=256", 1, 6, 256, 64, 128, (2, 1))]
    for name, B, H, W, cin, cout, (sh, sw) in cases:
        x = _rand((B, H, W, cin), 1)
        w = _rand((cout, cin, 3, 3), 2, 1.0 / np.sqrt(cin * 9)).requires_grad_(True)
        y = _conv_ref(r16(x), w, (sh, sw))
        dz = _rand(tuple(y.shape), 3)
        y.backward(r16(dz))
        xd, dzd = x.to(DEV), dz.contiguous().to(DEV)
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        scr = torch.empty(cout * 9 * cin + (x.numel() + dz.numel()) // 2 + 256, device=DEV)
        _lib.check(lib().hn_conv2d_wgrad_nhwc_bf16(P(xd), P(dzd), P(dw), P(scr), B, H, W, cin, cout, 3, 3, sh, sw, sp()), "wgrad bf16 row3")
        torch.cuda.synchronize()
        assert report("wgrad bf16 tap reuse (tile %s) %s" % (tile, name), dw.cpu().numpy(), w.grad.numpy(), 3e-5 * max(1.0, float(w.grad.abs().max())))


def test_stem_wgrad_stage():
    B, H, W = 2, 32, 64
    x = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(5))
    xn = horizonnet_ref.prepare_x(x)
    w = _rand((64, 3, 7, 7), 6, 0.1).requires_grad_(True)
    y = F.conv2d(horizonnet_ref.lr_pad(xn, 3), w, None, stride=2, padding=(3, 0)).permute(0, 2, 3, 1)
    dz = _rand(tuple(y.shape), 7)
    y.backward(dz)
    x4 = torch.zeros((B, H, W, 4))
    x4[..., :3] = xn.permute(0, 2, 3, 1)
    dw = torch.full((64, 3, 7, 7), float("nan"), device=DEV)
    scr = torch.empty(64 * 256, device=DEV)
    x4d, dzd = x4.to(DEV), dz.contiguous().to(DEV)          # keep alive: the ABI takes raw pointers
    _lib.check(lib().hn_conv2d_wgrad_nhwc(P(x4d), P(dzd), P(dw), P(scr), B, H, W, 3, 64, 7, 7, 2, 2, 1, sp()), "wgrad")
    torch.cuda.synchronize()
    assert report("wgrad stem 7x7", dw.cpu().numpy(), w.grad.numpy(), 2e-5 * float(w.grad.abs().max()))


def test_stem_wgrad_bf16_stage():
    """The stem's weight gradient on the bf16 matrix cores (train_precision bf16): bf16-rounded NHWC4 input and dz, exact products,
    float32 accumulation == autograd on the rounded operands (only the summation order differs); two sizes so that the m split
    and the row stepping of the loader are both exercised."""
    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    for B, H, W in ((2, 32, 64), (3, 128, 256)):
        x = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(5))
        xn = r16(horizonnet_ref.prepare_x(x))
        w = _rand((64, 3, 7, 7), 6, 0.1).requires_grad_(True)
        y = F.conv2d(horizonnet_ref.lr_pad(xn, 3), w, None, stride=2, padding=(3, 0)).permute(0, 2, 3, 1)
        dz = _rand(tuple(y.shape), 7)
        y.backward(r16(dz))
        x4 = torch.zeros((B, H, W, 4))
        x4[..., :3] = xn.permute(0, 2, 3, 1)
        dw = torch.full((64, 3, 7, 7), float("nan"), device=DEV)
        scr = torch.empty(64 * 256 + (B * H * W * 4 + B * H * W * 16) // 2 + 256, device=DEV)
        x4d, dzd = x4.to(DEV), dz.contiguous().to(DEV)
        _lib.check(lib().hn_conv2d_wgrad_nhwc_bf16(P(x4d), P(dzd), P(dw), P(scr), B, H, W, 3, 64, 7, 7, 2, 2, sp()), "stem wgrad bf16")
        torch.cuda.synchronize()
        assert report("wgrad stem 7x7 bf16 B=%d" % B, dw.cpu().numpy(), w.grad.numpy(), 3e-5 * float(w.grad.abs().max()))


def _oracle_grads(sd, x, wb, wc, dtype):
    ref = {k: (v.clone().to(dtype) if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}   # never alias sd
    for k, v in ref.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    rb, rc_ = horizonnet_ref.forward_train(x.to(dtype), ref, 0.1)
    ((rb * wb.to(dtype)).sum() + (rc_ * wc.to(dtype)).sum()).backward()
    return ref, rb.detach(), rc_.detach()


def test_train_step_matches_oracle_autograd():
    """Whole train-mode forward + backward (B=1, dropout off): outputs, all 241 parameter gradients and the
    updated BatchNorm running statistics.

    Gradients of a ReLU network are discontinuous in the pre-activations, so two float32 implementations
    legitimately differ wherever a pre-activation sits within rounding noise of zero (a handful of mask flips
    per layer; they weigh 1/M in a parameter gradient and M is as small as 32 here).  The yardstick is
    therefore the float64 oracle: the engine must be as close to it as the float32 oracle (= the reference's
    own arithmetic) is, up to a small factor, and point in the same direction."""
    B = 1
    sd = make_state_dict(11, "random")
    x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12))
    wb = _rand((B, 2, 1024), 13)
    wc = _rand((B, 1, 1024), 14)
    ref64, rb64, rc64 = _oracle_grads(sd, x, wb, wc, torch.float64)
    ref32, rb32, rc32 = _oracle_grads(sd, x, wb, wc, torch.float32)

    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    bon, cor = net(x.to(DEV))
    ((bon * wb.to(DEV)).sum() + (cor * wc.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    ok = report("train fwd bon vs f32 oracle", bon.detach().cpu().numpy(), rb32.numpy(), 1e-3)
    ok &= report("train fwd cor vs f32 oracle", cor.detach().cpu().numpy(), rc32.numpy(), 1e-3)
    rows = []
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        t = ref64[k].grad
        nt = float(t.norm()) + 1e-30
        e_eng = float((p.grad.cpu().double() - t).norm()) / nt
        e_o32 = float((ref32[k].grad.double() - t).norm()) / nt
        cos = float((p.grad.cpu().double() * t).sum()) / (float(p.grad.cpu().double().norm()) * nt + 1e-30)
        zero_grad = float(t.abs().max()) < 1e-6            # conv biases in front of a batch-stat BN: true gradient is 0
        rows.append((e_eng, e_o32, cos, k, zero_grad, float(p.grad.abs().max())))
    rows.sort(reverse=True)
    for e_eng, e_o32, cos, k, zg, gm in [r for r in rows if not r[4]][:10]:
        print("[parity] grad %-60s L2-rel vs f64: engine %.2e, f32 oracle %.2e, cos %.6f" % (k[-60:], e_eng, e_o32, cos))
    bad = [r for r in rows if not r[4] and not (r[0] <= 10.0 * max(r[1], 1e-5) and r[2] > 0.999)]
    badz = [r for r in rows if r[4] and not (r[5] < 1e-4)]
    med_ratio = float(np.median([r[0] / max(r[1], 1e-12) for r in rows if not r[4]]))
    print("[parity] %d / %d gradients worse than 10x the f32 oracle's own error; median engine/oracle error ratio %.2f; "
          "%d zero-gradient biases not ~0" % (len(bad), len(rows), med_ratio, len(badz)))
    ok &= not bad and not badz and med_ratio < 3.0
    # running statistics (momentum 0.1, unbiased variance) and num_batches_tracked
    sd_after = net.state_dict()
    for k in ("feature_extractor.encoder.bn1", "feature_extractor.encoder.layer3.2.bn2",
              "reduce_height_module.ghc_lst.3.layer.3.layers.1"):
        ok &= report("running_mean " + k, sd_after[k + ".running_mean"].cpu().numpy(), ref32[k + ".running_mean"].numpy(), 1e-4)
        ok &= report("running_var " + k, sd_after[k + ".running_var"].cpu().numpy(), ref32[k + ".running_var"].numpy(), 1e-3)
        assert int(sd_after[k + ".num_batches_tracked"]) == 1
    assert ok


def test_train_step_matches_reference_golden(golden_dir):
    """The engine's float32 training step against the UNMODIFIED reference module (train mode, train.py's loss,
    tests/golden/train_step_seed31.npz): loss, outputs, the norm and sum of every parameter gradient, sampled entries."""
    import json
    g = np.load(os.path.join(golden_dir, "train_step_seed31.npz"))
    names = json.load(open(os.path.join(golden_dir, "train_step_seed31.json")))["names"]
    sd = make_state_dict(31, "random")
    gen = torch.Generator().manual_seed(32)
    x = torch.rand(2, 3, 512, 1024, generator=gen)
    y_bon = (torch.rand(2, 2, 1024, generator=gen) - 0.5) * 1.2
    y_cor = (torch.rand(2, 1, 1024, generator=gen) < 0.05).float()
    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    bon, cor = net(x.to(DEV))
    loss = F.l1_loss(bon, y_bon.to(DEV)) + F.binary_cross_entropy_with_logits(cor, y_cor.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    ok = report("train step bon vs reference golden", bon.detach().cpu().numpy(), g["bon"], 1e-3)
    ok &= report("train step cor vs reference golden", cor.detach().cpu().numpy(), g["cor"], 1e-3)
    params = dict(net.named_parameters())
    worst = (0.0, "")
    for i, k in enumerate(names):
        if k.endswith("layers.0.1.bias"):
            continue
        n = float(params[k].grad.double().norm())
        e = abs(n - g["grad_norm"][i]) / g["grad_norm"][i]
        worst = max(worst, (e, k))
    print("[parity] gradient norms vs the reference: worst relative difference %.2e (%s)" % worst)
    ok &= worst[0] < 2e-2                      # ReLU-mask flips of a few pixels weigh most on BN parameters of small layers
    for k in ("linear.weight", "bi_rnn.bias_ih_l1", "reduce_height_module.ghc_lst.3.layer.3.layers.0.1.weight",
              "feature_extractor.encoder.conv1.1.weight"):
        t = params[k].grad.flatten().cpu()
        got = t[:: max(1, t.numel() // 4096)].numpy()
        want = g["grad:" + k]
        # (the stem's gradient sums over 131072 pixels per image: a handful of ReLU-mask flips moves single entries by ~2 %)
        ok &= report("train step grad sample " + k[-40:], got, want, (6e-2 if "conv1.1" in k else 2e-2) * float(np.abs(want).max()))
    assert ok


def test_loss_curve_first_steps_matches_oracle():
    """BASELINE configs[2] agreement check: the first K optimiser steps of train.py:272-281 (L1 + BCE-with-logits,
    Adam lr 1e-4) on the engine against the same steps on the CPU float32 oracle of the reference -- same weights,
    same batch, dropout off.  The first Adam step moves every weight by +-lr (g / sqrt(g^2)), so sign noise in
    near-zero gradient components makes the two weight sets differ slightly; the losses must still track."""
    K, B = 3, 2
    sd = make_state_dict(31, "random")
    g = torch.Generator().manual_seed(32)
    x = torch.rand(B, 3, 512, 1024, generator=g)
    y_bon = (torch.rand(B, 2, 1024, generator=g) - 0.5) * 1.2
    y_cor = (torch.rand(B, 1, 1024, generator=g) < 0.05).float()

    ref = {k: v.clone() for k, v in sd.items()}
    params = [v.requires_grad_(True) for k, v in ref.items() if v.is_floating_point() and "running_" not in k]
    opt_ref = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    want = []
    for _ in range(K):
        rb, rc_ = horizonnet_ref.forward_train(x, ref, 0.1)
        loss = F.l1_loss(rb, y_bon) + F.binary_cross_entropy_with_logits(rc_, y_cor)
        opt_ref.zero_grad(set_to_none=True)
        loss.backward()
        opt_ref.step()
        want.append(float(loss))

    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    xd, yb, yc = x.to(DEV), y_bon.to(DEV), y_cor.to(DEV)
    got = []
    for _ in range(K):
        bon, cor = net(xd)
        loss = F.l1_loss(bon, yb) + F.binary_cross_entropy_with_logits(cor, yc)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        got.append(float(loss))
    assert net.hip_status(DEV) == 0
    print("[parity] loss curve, engine vs f32 oracle: " + ", ".join("%.6f / %.6f" % (a, b) for a, b in zip(got, want)))
    assert abs(got[0] - want[0]) < 1e-5 * abs(want[0]) + 1e-6          # same weights: forward parity
    for a, b in zip(got[1:], want[1:]):
        assert abs(a - b) < 2e-2 * abs(b)
    assert want[-1] < want[0] and got[-1] < got[0]                       # and both actually descend


def test_segmented_backward_equals_monolithic():
    """hn_train_backward_segment 0..4 (the data-parallel overlap path) == hn_train_backward: same gradients up to the
    summation-order noise of the float atomics in the weight-gradient kernels, and the segment ranges tile the buffer."""
    import ctypes
    L = lib()
    nseg = L.hn_grad_segments()
    lo, cnt = ctypes.c_int64(), ctypes.c_int64()
    ranges = []
    for sgm in range(nseg):
        _lib.check(L.hn_grad_segment_range(sgm, ctypes.byref(lo), ctypes.byref(cnt)), "range")
        ranges.append((lo.value, cnt.value))
    assert nseg == 5 and sorted(ranges)[0][0] == 0
    srt = sorted(ranges)
    assert all(srt[i][0] + srt[i][1] == srt[i + 1][0] for i in range(nseg - 1)) and srt[-1][0] + srt[-1][1] == L.hn_grad_floats()
    assert ranges[0][0] == L.hn_grad_offset(b"bi_rnn.weight_ih_l0")            # finished first: the tail of the buffer

    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(41, "random"))
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(42)).to(DEV)
    w = _rand((2, 2, 1024), 43).to(DEV)

    def grads(segmented):
        net.segmented_backward = segmented
        for p in net.parameters():
            p.grad = None
        bon, cor = net(x)
        ((bon * w).sum() + cor.sum()).backward()
        return {k: p.grad.clone() for k, p in net.named_parameters()}

    a, b = grads(False), grads(True)
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    # conv biases in front of a batch-statistics BN have a zero true gradient: both runs hold rounding noise there
    rel = {k: float((a[k] - b[k]).norm() / (a[k].norm() + 1e-20)) for k in a if not k.endswith("layers.0.1.bias")}
    worst = max(rel, key=rel.get)
    print("[parity] segmented vs monolithic backward: worst relative L2 difference %.2e (%s)" % (rel[worst], worst))
    assert rel[worst] < 1e-4
    assert all(float(b[k].abs().max()) < 1e-4 for k in b if k.endswith("layers.0.1.bias"))


def _oracle_grads_bf16(sd, x, wb, wc, dtype):
    ref = {k: (v.clone().to(dtype) if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
    for k, v in ref.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    rb, rc_ = horizonnet_ref.forward_train(x.to(dtype), ref, 0.1, bf16_convs=True)
    ((rb * wb.to(dtype)).sum() + (rc_ * wc.to(dtype)).sum()).backward()
    return ref, rb.detach(), rc_.detach()


def test_train_step_bf16_matrix_cores():
    """train_precision = "bf16": forward + data-gradient convolutions with bf16 operands (f32 accumulation), everything
    else float32.  Checked against a torch restatement of exactly that arithmetic (oracle ..bf16_convs: operands
    rounded to bf16 at the same points, weight gradients from the unrounded tensors) evaluated in float64.

    Rounding to bf16 is discontinuous: a 1e-7 difference in front of a rounding point becomes a 4e-3 one behind it,
    and this randomly initialised batch-statistics network amplifies perturbations ~100x from stem to head, so two
    correct implementations legitimately differ at the 1e-2..1e-1 level.  The yardstick is therefore the SAME
    restatement evaluated in float32: the engine must be as close to the float64 result as that is."""
    B = 1
    sd = make_state_dict(11, "random")
    x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12))
    wb = _rand((B, 2, 1024), 13)
    wc = _rand((B, 1, 1024), 14)
    ref64, rb64, rc64 = _oracle_grads_bf16(sd, x, wb, wc, torch.float64)
    ref32, rb32, rc32 = _oracle_grads_bf16(sd, x, wb, wc, torch.float32)

    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    net.train_precision = "bf16"
    bon, cor = net(x.to(DEV))
    ((bon * wb.to(DEV)).sum() + (cor * wc.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    f_eng = max(float((bon.detach().cpu() - rb64).abs().max()), float((cor.detach().cpu() - rc64).abs().max()))
    f_o32 = max(float((rb32 - rb64).abs().max()), float((rc32 - rc64).abs().max()))
    print("[parity] bf16 train forward, max-abs vs f64 restatement: engine %.3e, f32 restatement %.3e" % (f_eng, f_o32))
    rows = []
    for k, p in net.named_parameters():
        t = ref64[k].grad
        if float(t.abs().max()) < 1e-6:
            continue
        g = p.grad.cpu().double()
        nt = float(t.norm())
        e_eng = float((g - t).norm()) / nt
        e_o32 = float((ref32[k].grad.double() - t).norm()) / nt
        rows.append((e_eng, e_o32, float((g * t).sum()) / (float(g.norm()) * nt + 1e-30), k))
    rows.sort(reverse=True)
    for e, eo, cs, k in rows[:6] + rows[-3:]:
        print("[parity] bf16 grad %-56s L2-rel vs f64: engine %.3e, f32 restatement %.3e, cos %.5f" % (k[-56:], e, eo, cs))
    ratio = float(np.median([r[0] / max(r[1], 1e-6) for r in rows]))
    med_e, med_o = float(np.median([r[0] for r in rows])), float(np.median([r[1] for r in rows]))
    print("[parity] bf16 training: median relative gradient error engine %.3e / f32 restatement %.3e (ratio of medians %.2f, median "
          "ratio %.2f), min cosine %.4f" % (med_e, med_o, med_e / max(med_o, 1e-12), ratio, min(r[2] for r in rows)))
    # (the discriminating check of the bf16 path is test_bf16_train_units_locally_consistent below: on this network even
    # the restatement's own f32 and f64 evaluations disagree at the 100 % level in the early layers)
    assert f_eng <= 5.0 * max(f_o32, 1e-3)
    assert med_e <= 1.5 * max(med_o, 1e-3)
    late = [r for r in rows if r[3].startswith(("bi_rnn", "linear"))]
    assert late and all(r[0] <= 2.0 * max(r[1], 1e-3) for r in late)        # the LSTM / head see a forward that is only ~3e-2 off
    # switching back restores the float32 path
    net.train_precision = "f32"
    b32, _ = net(x.to(DEV))
    torch.cuda.synchronize()
    assert net.hip_status(DEV) == 0
    # One full-suite run in round 5 saw this forward 0.12 off.  Round 6 looked for the cause with the poison instrument (tools/poison_hunt.py, the
    # autouse fixture in conftest.py, HN_POISON_WS=1 over the whole suite): every deterministic tensor of this sequence is the same bits whatever the
    # allocator hands out, so it was not a read of unwritten memory; 240 repetitions (tools/soak_determinism.py) were bit-identical too.  The
    # float32 training forward is deterministic (its double-precision statistics atomics round to the same float), so a repeat must be EQUAL:
    b32_again, _ = net(x.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(b32_again, b32), "f32 train fwd after bf16: repeat differs by %.3e" % float((b32_again - b32).abs().max())
    _, rbf, _ = _oracle_grads(sd, x, wb, wc, torch.float32)
    assert report("f32 train fwd after bf16", b32.detach().cpu().numpy(), rbf.numpy(), 1e-3)


def test_bf16_train_units_locally_consistent():
    """train_precision = "bf16", layer by layer, using the engine's OWN tensors as inputs (debug taps + workspace), so
    that the chaos of the whole network plays no role: for producer -> consumer pairs of units
      * the consumer's z        == bf16(conv(bf16(producer y), bf16(w)) (+ bias))    (bf16 MFMA forward, z stored as bf16)
      * the saved mean / invstd == batch statistics of that z                        (statistics fused in the epilogue)
      * the producer's dy       == conv_transpose(bf16(consumer dz), bf16(w))        (bf16 MFMA data gradient)
      * the consumer's dW       == wgrad(bf16(producer y), bf16(consumer dz))        (bf16 MFMA weight gradient, f32 accumulation)"""
    import ctypes
    from oracle.weights import conv_specs
    B = 1
    sd = make_state_dict(11, "random")
    x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12)).to(DEV)
    wb, wc = _rand((B, 2, 1024), 13).to(DEV), _rand((B, 1, 1024), 14).to(DEV)
    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    net.train_precision = "bf16"
    net.set_engine_option("fuse_bn_fold", 0)      # this test reads z / dz of conv3 units, which the folded form never stores (its own test: test_bn_folded_forward_unit)
    L = lib()
    names = [c[0] for c in conv_specs()]
    out8 = (ctypes.c_int64 * 8)()
    L.hn_train_debug_unit(B, 0, out8)
    info = []
    for u in range(out8[7]):
        L.hn_train_debug_unit(B, u, out8)
        info.append(tuple(out8))

    def unit_of(name):
        return [i for i, t in enumerate(info) if names[t[0]].endswith(name)][0]

    def run(pname, pshape, cname, cshape):
        """ONE forward + backward with both taps set -> (producer dict, consumer dict) of NHWC host tensors."""
        up, uc = unit_of(pname), unit_of(cname)
        bufs = {}
        for u, shape in ((up, pshape), (uc, cshape)):
            assert info[u][1] * info[u][2] == int(np.prod(shape))
            bufs[u] = (torch.zeros(int(np.prod(shape)), device=DEV), torch.zeros(int(np.prod(shape)), device=DEV))
        bon, cor = net(x)
        st = net._hip_states[0]
        _lib.check(L.hn_train_debug_set(st.handle, up, P(bufs[up][0]), P(bufs[up][1])), "tap")
        _lib.check(L.hn_train_debug_set2(st.handle, uc, P(bufs[uc][0]), P(bufs[uc][1])), "tap2")
        for p in net.parameters():
            p.grad = None
        ((bon * wb).sum() + (cor * wc).sum()).backward()
        torch.cuda.synchronize()
        _lib.check(L.hn_train_debug_set(st.handle, -1, None, None), "untap")
        _lib.check(L.hn_train_debug_set2(st.handle, -1, None, None), "untap2")
        ws = st.train_workspace(B).view(torch.float32)
        grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
        out = []
        for u, shape in ((up, pshape), (uc, cshape)):
            n, C = int(np.prod(shape)), info[u][2]
            stt = ws[info[u][6]:info[u][6] + 2 * C].cpu()
            zraw = st.train_workspace(B)[info[u][4] * 4:info[u][4] * 4 + 2 * n]        # bf16 mode stores z as bf16
            yh_off = L.hn_train_debug_unit_yh(B, u)                                           # bf16 mode keeps y as bf16 only
            yraw = st.train_workspace(B)[yh_off * 4:yh_off * 4 + 2 * n]
            out.append(dict(dy=bufs[u][0].cpu().view(shape), dz=bufs[u][1].cpu().view(shape),
                            z=zraw.view(torch.bfloat16).float().cpu().view(shape), y=yraw.view(torch.bfloat16).float().cpu().view(shape),
                            mean=stt[:C], invstd=stt[C:], grads=grads))
        return out

    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    enc = "feature_extractor.encoder."
    ghc = "reduce_height_module.ghc_lst.2.layer."
    pairs = [  # producer unit, its output shape NHWC, consumer unit, its output shape, stride, kernel
        (enc + "layer3.2.conv1", (B, 32, 64, 256), enc + "layer3.2.conv2.1", (B, 32, 64, 256), (1, 1), 3),
        (enc + "layer3.2.conv2.1", (B, 32, 64, 256), enc + "layer3.2.conv3", (B, 32, 64, 1024), (1, 1), 1),
        (enc + "layer2.0.conv1", (B, 128, 256, 128), enc + "layer2.0.conv2.1", (B, 64, 128, 128), (2, 2), 3),
        (ghc + "1.layers.0.1", (B, 8, 64, 512), ghc + "2.layers.0.1", (B, 4, 64, 256), (2, 1), 3),
    ]
    ok = True
    for pn, pshape, cn, cshape, stride, k in pairs:
        prod, cons = run(pn, pshape, cn, cshape)
        w = sd[cn + ".weight"]
        bias = sd.get(cn + ".bias")
        yin = r16(prod["y"]).requires_grad_(True)
        w16 = r16(w)
        z_want = _conv_ref(yin, w16, stride)
        if bias is not None:
            z_want = z_want + bias
        # z is stored rounded to bf16 (one bf16 ulp = 2^-8 relative); its statistics come from the f32 accumulators
        ok &= report("bf16 train z   %s" % cn[-34:], cons["z"].numpy(), z_want.detach().numpy(), 2.0 ** -8 * float(z_want.abs().max()))
        zz = z_want.detach().reshape(-1, cshape[3]).double()
        ok &= report("bf16 train mean %s" % cn[-33:], cons["mean"].numpy(), zz.mean(0).numpy(), 1e-4 * float(zz.abs().max()))
        inv = 1.0 / torch.sqrt(zz.var(0, unbiased=False) + 1e-5)
        ok &= report("bf16 train invstd %s" % cn[-31:], cons["invstd"].numpy(), inv.numpy(), 1e-3 * float(inv.abs().max()))
        z_want.backward(r16(cons["dz"]))
        # gradients between units are stored rounded to bf16 (f32 accumulators, one rounding at the store)
        ok &= report("bf16 train dy  %s" % pn[-34:], prod["dy"].numpy(), yin.grad.numpy(), 2.0 ** -8 * float(yin.grad.abs().max()))
        w32 = w.clone().requires_grad_(True)                               # weight gradient: bf16 operands, exact products
        _conv_ref(r16(prod["y"]), w32, stride).backward(r16(cons["dz"]))
        ok &= report("bf16 train dW  %s" % cn[-34:], cons["grads"][cn + ".weight"].numpy(), w32.grad.numpy(), 2e-4 * float(w32.grad.abs().max()))
    assert ok


def test_dropout_statistics_and_eval_after_train():
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(0, "random"))
    net = net.to(DEV).train()
    x = torch.rand(1, 3, 512, 1024, generator=torch.Generator().manual_seed(3)).to(DEV)
    torch.manual_seed(1)
    b1, c1 = net(x)
    torch.manual_seed(2)
    b2, c2 = net(x)
    torch.cuda.synchronize()
    assert float((b1 - b2).abs().max()) > 1e-4          # different dropout masks
    assert bool(torch.isfinite(b1).all()) and bool(torch.isfinite(c2).all())
    net.eval()
    with torch.no_grad():
        e1, _ = net(x)
        e2, _ = net(x)
    assert torch.equal(e1, e2)                            # eval path re-packs with the updated running stats


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from horizonnet_amd import HorizonNet, broadcast_module_
from oracle.weights import make_state_dict
dist.init_process_group(backend="gloo")          # two ranks share the one GPU of the test box (RCCL refuses that)
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = HorizonNet("resnet50", True)
net.load_state_dict(make_state_dict(21 + rank, "random"))   # different weights per rank: broadcast must fix that
net = net.to(dev).train()
net.bi_rnn.dropout = 0.0
net.drop_out.p = 0.0
broadcast_module_(net, src=0)
xs = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(5)).to(dev)
wb = torch.rand(2, 2, 1024, generator=torch.Generator().manual_seed(6)).to(dev) - 0.5
def grads(x, w, sync):
    net.sync_gradients = sync
    for p in net.parameters():
        p.grad = None
    bon, cor = net(x)
    ((bon * w).sum() + cor.sum()).backward()
    return torch.cat([p.grad.flatten() for p in net.parameters()])
g_dp = grads(xs[rank:rank + 1], wb[rank:rank + 1], True)          # data parallel: own shard, all-reduce(mean) inside backward
g0 = grads(xs[0:1], wb[0:1], False)
g1 = grads(xs[1:2], wb[1:2], False)
want = 0.5 * (g0 + g1)
err = float((g_dp - want).norm() / want.norm())
# FusedAdam's form: backward leaves the SUM over the ranks, the 1 / world goes into hn_adam_step's grad_scale
from horizonnet_amd.optim import FusedAdam
opt = FusedAdam(net, lr=1e-4)                                   # sets net.defer_grad_mean; only a LIVE FusedAdam defers the mean
g_sum = grads(xs[rank:rank + 1], wb[rank:rank + 1], True)
err_sum = float((g_sum - 2.0 * want).norm() / want.norm()) + abs(net._grad_mean_scale - 0.5)
opt.mean_gradients_()                                           # gradient clipping / logging in front of step(): the mean, now
g_mean = torch.cat([p.grad.flatten() for p in net.parameters()])
err_sum += float((g_mean - want).norm() / want.norm()) + abs(net._grad_mean_scale - 1.0)
del opt                                                         # the optimiser is gone: backward stores the mean again (ADVICE r3)
import gc; gc.collect()
g_back = grads(xs[rank:rank + 1], wb[rank:rank + 1], True)
err_sum += float((g_back - want).norm() / want.norm())
net.defer_grad_mean = False
# bf16 on the wire (half the xGMI bytes): the mean of bf16-rounded sums
net.allreduce_dtype = "bf16"
try:
    g_h = grads(xs[rank:rank + 1], wb[rank:rank + 1], True)
    err_h = float((g_h - want).norm() / want.norm())
except RuntimeError as e:          # gloo builds without bf16 reductions on device tensors: RCCL has them
    err_h = -1.0
net.allreduce_dtype = "f32"
allr = [None] * world
dist.all_gather_object(allr, (err, err_sum, err_h))
if rank == 0:
    print("DPERR %%.3e %%.3e status %%d" %% (allr[0][0], allr[1][0], net.hip_status(dev)))
    print("DPSUM %%.3e %%.3e" %% (allr[0][1], allr[1][1]))
    print("DPBF16 %%.3e %%.3e" %% (allr[0][2], allr[1][2]))
dist.destroy_process_group()
"""


def test_two_rank_data_parallel_gradients(tmp_path):
    """The exchange step of training end to end on the GPU: two ranks, each its own panorama; the gradient every
    rank ends up with must be the mean of the two per-shard gradients.  (BatchNorm statistics stay per replica, as in the
    reference's DataParallel; the running-stat updates of the extra local passes are irrelevant to the gradient.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("DPERR")][0]
    print("[parity] data-parallel gradient vs mean of shard gradients:", line)
    e0, e1 = float(line.split()[1]), float(line.split()[2])
    assert e0 < 1e-5 and e1 < 1e-5 and line.endswith("status 0")
    sums = [float(v) for v in [l for l in out.stdout.splitlines() if l.startswith("DPSUM")][0].split()[1:]]
    halves = [float(v) for v in [l for l in out.stdout.splitlines() if l.startswith("DPBF16")][0].split()[1:]]
    print("[parity] deferred mean (sum left for FusedAdam):", sums, " bf16 wire:", halves)
    assert max(sums) < 3e-5                                      # three terms: sum left for FusedAdam, mean_gradients_(), mean again once the optimiser is gone
    assert all(h < 0 or 1e-5 < h < 1e-2 for h in halves)          # bf16 rounding of the summed ranges: ~2^-9 relative; -1 = backend lacks bf16


def _dgrad_bf16g(dz, w, add, B, H, W, cin, cout, k, sh, sw):
    dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
    scr = torch.empty(cout * cin * k * k + 8192 + dz.numel() // 2 + 64 + 2 * (B * H * W * cin // 2 + 64), device=DEV)
    _lib.check(lib().hn_conv2d_dgrad_nhwc_bf16g(P(dz), P(w), P(add), P(dx), P(scr), B, H, W, cin, cout, k, k, sh, sw, sp()), "dgrad bf16g")
    torch.cuda.synchronize()
    return dx


@pytest.mark.parametrize("case", [c for c in GRAD_CASES if c[5] % 64 == 0], ids=[c[0] for c in GRAD_CASES if c[5] % 64 == 0])
def test_conv_dgrad_bf16_gradients_stage(case):
    """The data gradient in the form the bf16 training step runs (bf16 dz / identity gradient / dx; stride 1 = a forward conv with
    flipped taps, strided = per-class kernels) == torch autograd on the bf16-rounded operands, up to the bf16 rounding of the
    identity input and of the result."""
    name, B, H, W, cin, cout, k, stride = case
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    r16 = lambda t: t.bfloat16().float()                                                    # noqa: E731
    x = _rand((B, H, W, cin), 1).requires_grad_(True)
    w = _rand((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
    w16 = r16(w).requires_grad_(True)
    y = _conv_ref(x, w16, (sh, sw))
    dz = _rand(tuple(y.shape), 3)
    add = _rand((B, H, W, cin), 4)
    y.backward(r16(dz))
    dx = _dgrad_bf16g(dz.contiguous().to(DEV), w.to(DEV), add.to(DEV), B, H, W, cin, cout, k, sh, sw)
    want = x.grad + r16(add)
    assert report("dgrad bf16 gradients " + name, dx.cpu().numpy(), want.numpy(), 2.0 ** -8 * max(1.0, float(want.abs().max())))


def test_strided_dgrad_bf16_8wave_equals_4wave():
    """Data gradient of a strided conv on the 256x256 8-wave kernel (plain at 256 tiles, persistent above) against the 128x128
    data-gradient kernel: same per-class packing and k order -> bit-identical; the 128x128 form is checked against autograd in
    test_conv_dgrad_bf16_gradients_stage."""
    import os
    gen = torch.Generator().manual_seed(91)
    for (B, H, W, cin, cout, sh, sw) in ((4, 128, 256, 256, 128, 2, 1), (6, 128, 256, 256, 128, 2, 1), (8, 128, 256, 256, 64, 2, 2)):
        k = 3
        Ho, Wo = H // sh, W // sw
        dz = (torch.rand(B, Ho, Wo, cout, generator=gen) - 0.5).to(DEV)
        w = ((torch.rand(cout, cin, k, k, generator=gen) - 0.5) / np.sqrt(cin * k * k)).to(DEV)
        add = (torch.rand(B, H, W, cin, generator=gen) - 0.5).to(DEV)
        out = {}
        for mode in ("0", "1"):
            os.environ["HN_DGRAD_W8"] = mode
            out[mode] = _dgrad_bf16g(dz, w, add, B, H, W, cin, cout, k, sh, sw)
        os.environ.pop("HN_DGRAD_W8", None)
        assert torch.isfinite(out["1"]).all()
        assert torch.equal(out["0"], out["1"]), "B=%d: 8-wave data gradient differs, max %g" % (B, float((out["0"] - out["1"]).abs().max()))


def test_train_stem_conv_direct_matches_implicit_gemm():
    """train_precision bf16: the stem conv of the training forward as the direct kernel (LDS ring of input rows, z + batch statistics)
    against the implicit-GEMM stem: same k order -> the bf16 z is bit-identical, so a whole training forward differs only through
    the summation order of the statistics (float atomics either way): outputs within 1e-3 of each other's scale."""
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(11, "random"))
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    net.train_precision = "bf16"
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(21)).to(DEV)
    outs = []
    for opt in (0, 1):
        net.set_engine_option("fuse_stem_pool", opt)
        bon, cor = net(x)
        torch.cuda.synchronize()
        outs.append((bon.detach().clone(), cor.detach().clone()))
    net.set_engine_option("fuse_stem_pool", 1)
    assert net.hip_status(DEV) == 0
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(a.abs().max())), float((a - b).abs().max())


def _train_probe_in_subprocess(env, B, precs=("f32", "bf16")):
    """One train-mode forward + backward per precision in a fresh process (the switches below are read once per process): output sums,
    running statistics and a few gradient norms."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, sys, torch
sys.path.insert(0, %r)
from horizonnet_amd import HorizonNet
from oracle.weights import make_state_dict
out = {}
for prec in %r:
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(51, "random"))
    net = net.to("cuda:0").train()
    net.train_precision = prec
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    x = torch.rand(%d, 3, 512, 1024, generator=torch.Generator().manual_seed(52)).to("cuda:0")
    bon, cor = net(x)
    (bon.abs().sum() + cor.sum()).backward()
    torch.cuda.synchronize()
    sd = net.state_dict()
    out[prec] = {"bon": float(bon.double().sum()), "cor": float(cor.double().sum()),
                 "rm": [float(sd[k].double().norm()) for k in sorted(sd) if k.endswith("running_mean")][:12],
                 "rv": [float(sd[k].double().norm()) for k in sorted(sd) if k.endswith("running_var")][:12],
                 "g": [float(p.grad.double().norm()) for n_, p in sorted(net.named_parameters()) if n_.endswith("conv1.weight") or n_.endswith("conv2.weight")][:16],
                 "stem": [float(p.grad.double().norm()) for n_, p in sorted(net.named_parameters()) if n_.startswith("feature_extractor.encoder.conv1.") or n_.startswith("feature_extractor.encoder.bn1.")],
                 "status": int(net.hip_status(torch.device("cuda:0")))}
print("RESULT " + json.dumps(out))
''' % (ROOT, tuple(precs), B)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def _assert_probes_agree(a, b, prec, tol):
    """Forward quantities (output sums, running statistics) in both precisions; gradient norms in float32 only: in bf16 mode two runs of
    the SAME configuration already differ by percents in some layers' gradient norms (atomics order -> a last-bit change of a mean ->
    other bf16 rounding points and ReLU masks downstream; measured 2.2 %)."""
    assert a["status"] == 0 and b["status"] == 0
    for key in ("rm", "rv") + (("g",) if prec == "f32" else ()):
        worst = max(abs(u - v) / (abs(u) + 1e-12) for u, v in zip(a[key], b[key]))
        assert worst < tol, (prec, key, worst)
    assert abs(a["bon"] - b["bon"]) < tol * (abs(a["bon"]) + 1.0) * 10 and abs(a["cor"] - b["cor"]) < tol * (abs(a["cor"]) + 1.0) * 10, (prec, a["bon"], b["bon"], a["cor"], b["cor"])


def test_replicated_statistics_slots_equal_the_single_slot():
    """ConvDesc::stat_rep (csrc/stat_commit.h): the conv epilogues of the few-channel layers add their batch statistics into 8-16
    replicas that hn_launch_stat_replica_sum adds up.  The rule only switches it on for M >= 400000 (B >= 13); HN_STAT_REPLICAS=all
    forces it at B = 2, HN_STAT_REPLICAS=0 never uses it: same outputs, running statistics and gradients up to summation order, in
    both precisions."""
    res = {mode: _train_probe_in_subprocess({"HN_STAT_REPLICAS": mode}, 2) for mode in ("0", "all")}
    # norm-level agreement: the runs differ by the order of the double atomics (statistics) and float atomics (weight gradients) anyway;
    # bf16: z is rounded to bf16 after the statistics, a last-bit change of a mean moves rounding points downstream.  A lost replica
    # would change a mean by 1/16 and every figure below by percents.
    for prec, tol in (("f32", 3e-4), ("bf16", 1e-2)):
        _assert_probes_agree(res["0"][prec], res["all"][prec], prec, tol)


def test_dw_reuse_statistics_epilogue_equals_the_four_wave_kernel():
    """conv3x3_dwr64_bf16_kernel<STATS> (layer1's conv2 in the bf16 training forward from B = 4 on: bf16 z + batch statistics summed per
    lane across the persistent workgroup's tiles, stat_wave.h) against the 4-wave kernel's statistics epilogue (HN_BF16_DWR64=0): z is
    bit-identical (same k order), the statistics differ by summation order only.  With the replicas forced on top."""
    base = _train_probe_in_subprocess({"HN_BF16_DWR64": "0"}, 4, ("bf16",))["bf16"]
    for env in ({"HN_BF16_DWR64": "1"}, {"HN_BF16_DWR64": "1", "HN_STAT_REPLICAS": "all"}):
        _assert_probes_agree(base, _train_probe_in_subprocess(env, 4, ("bf16",))["bf16"], "bf16", 1e-2)


def test_fused_stem_batchnorm_pool_equals_the_two_passes():
    """affine_act_bn_pool_kernel (BatchNorm + ReLU + 3x3 / 2 max-pool + position words + ReLU masks in one pass over the stem's z, bf16
    training forward) against affine_act_bn_kernel + maxpool_fwd_idx_kernel (HN_FUSE_STEM_BNPOOL=0): the pooled tensor, positions and
    masks are the same bits, so everything downstream agrees to the atomics' noise."""
    base = _train_probe_in_subprocess({"HN_FUSE_STEM_BNPOOL": "0"}, 2, ("bf16",))["bf16"]
    fused = _train_probe_in_subprocess({"HN_FUSE_STEM_BNPOOL": "1"}, 2, ("bf16",))["bf16"]
    _assert_probes_agree(base, fused, "bf16", 1e-2)
    assert abs(base["rm"][0] - fused["rm"][0]) <= 1e-6 * abs(base["rm"][0]) and abs(base["rv"][0] - fused["rv"][0]) <= 1e-6 * abs(base["rv"][0])   # the stem's own running statistics


def test_stem_batchnorm_adjoint_from_the_pooled_gradient_equals_the_separate_pass():
    """hn_launch_bn_bwd_pool (the stem's BatchNorm adjoint gathers the max-pool adjoint from the pooled gradient: bn_bwd_*_h8_kernel<.., POOL>)
    against maxpool_bwd_idx_kernel + the plain adjoint (engine option "fuse_stem_poolbwd" = 0): TWO backward passes over the activations of ONE
    bf16 training forward, so everything upstream of the stem is the same up to the atomics' order; the stem's dz is the same bits, its
    conv / BatchNorm gradients agree to that noise."""
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(61, "random"))
    net = net.to(DEV).train()
    net.train_precision = "bf16"
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(62)).to(DEV)
    bon, cor = net(x)
    loss = bon.abs().sum() + cor.sum()
    names = ["feature_extractor.encoder.conv1.1.weight", "feature_extractor.encoder.bn1.weight", "feature_extractor.encoder.bn1.bias",
             "feature_extractor.encoder.layer1.0.conv1.weight"]
    params = dict(net.named_parameters())
    got = {}
    for mode in (0, 1, 0):
        net.set_engine_option("fuse_stem_poolbwd", mode)
        for p in net.parameters():
            p.grad = None
        loss.backward(retain_graph=True)
        torch.cuda.synchronize()
        got.setdefault(mode, []).append({k: params[k].grad.detach().double().clone() for k in names})
    assert net.hip_status(DEV) == 0
    noise = {k: float((got[0][0][k] - got[0][1][k]).norm() / got[0][0][k].norm()) for k in names}       # the separate pass against itself
    diff = {k: float((got[0][0][k] - got[1][0][k]).norm() / got[0][0][k].norm()) for k in names}
    print("[parity] stem adjoint from the pooled gradient: relative L2 difference", diff, "| run-to-run noise of the separate pass", noise)
    for k in names:
        assert diff[k] <= max(3.0 * noise[k], 1e-5), (k, diff[k], noise[k])


def test_dual_batchnorm_adjoint_equals_two_adjoints():
    """hn_launch_bn_bwd_dual (block 0 of every ResNet stage: conv3's and the downsample branch's BatchNorm adjoints in one reduce + one
    apply pass, the masked gradient never written) against the two separate adjoints (engine option "fuse_bn_dual" = 0): two backward
    passes over ONE bf16 training forward; the affected parameter gradients agree to the run-to-run noise of the separate form."""
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(71, "random"))
    net = net.to(DEV).train()
    net.train_precision = "bf16"
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(72)).to(DEV)
    bon, cor = net(x)
    loss = bon.abs().sum() + cor.sum()
    enc = "feature_extractor.encoder."
    names = []
    for li in (1, 2, 3, 4):
        names += [enc + "layer%d.0.conv3.weight" % li, enc + "layer%d.0.bn3.weight" % li, enc + "layer%d.0.bn3.bias" % li,
                  enc + "layer%d.0.downsample.0.weight" % li, enc + "layer%d.0.downsample.1.weight" % li, enc + "layer%d.0.downsample.1.bias" % li,
                  enc + "layer%d.0.conv1.weight" % li]
    names.append(enc + "conv1.1.weight")
    params = dict(net.named_parameters())
    got = {}
    for mode in (0, 1, 0):
        net.set_engine_option("fuse_bn_dual", mode)
        for p in net.parameters():
            p.grad = None
        loss.backward(retain_graph=True)
        torch.cuda.synchronize()
        got.setdefault(mode, []).append({k: params[k].grad.detach().double().clone() for k in names})
    assert net.hip_status(DEV) == 0
    noise = {k: float((got[0][0][k] - got[0][1][k]).norm() / got[0][0][k].norm()) for k in names}
    diff = {k: float((got[0][0][k] - got[1][0][k]).norm() / got[0][0][k].norm()) for k in names}
    worst = max(names, key=lambda k: diff[k] / max(noise[k], 1e-7))
    print("[parity] dual BatchNorm adjoint: worst relative L2 difference %.2e (%s; run-to-run noise there %.2e)" % (diff[worst], worst, noise[worst]))
    for k in names:
        assert diff[k] <= max(4.0 * noise[k], 2e-5), (k, diff[k], noise[k])


@pytest.mark.gpu
def test_bn_folded_adjoint_equals_classical_adjoint():
    """bn_fold.hip (the conv3 / stride-1 downsample units of the bf16 step: BatchNorm adjoint folded algebraically into the weight-
    gradient GEMM and two 1x1 data-gradient convs -- no reduce pass, no apply pass, no dz tensor) against the classical two-pass adjoint
    (engine option "fuse_bn_fold" = 0): two backward passes over ONE bf16 training forward.  The two forms round at different places
    (the classical one rounds z and dz to bf16 per element, the folded one rounds c1 * W and Q), so equality is checked where both see
    IDENTICAL inputs -- the first folded unit of the backward order, layer4.2.conv3 + bn3 (its incoming gradient comes from the
    height-compression branch, which the fold does not touch) and the unit that consumes its data gradient (layer4.2.conv2 / bn2) --
    at the level of bf16's own rounding; tensors behind the fold in the backward order (height compression, LSTM, head) must
    agree to the run-to-run noise; everything downstream is bounded loosely (a batch of 2 through 50 batch-statistics BatchNorms
    amplifies any rounding difference: the classical form differs from the float32 step by O(1) there)."""
    net = HorizonNet("resnet50", True)
    net.load_state_dict(make_state_dict(71, "random"))
    net = net.to(DEV).train()
    net.train_precision = "bf16"
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    x = torch.rand(2, 3, 512, 1024, generator=torch.Generator().manual_seed(72)).to(DEV)
    net.set_engine_option("fuse_bn_fold", 2)       # classical forward (z is stored, so both adjoints can run over it); 1 would fold the forward too
    bon, cor = net(x)
    loss = bon.abs().sum() + cor.sum()
    params = dict(net.named_parameters())
    names = list(params.keys())
    got = {}
    for mode in (0, 1, 0):
        net.set_engine_option("fuse_bn_fold", 2 * mode)
        for p in net.parameters():
            p.grad = None
        loss.backward(retain_graph=True)
        torch.cuda.synchronize()
        got.setdefault(mode, []).append({k: params[k].grad.detach().double().clone() for k in names})
    assert net.hip_status(DEV) == 0
    enc = "feature_extractor.encoder."
    rel = lambda a, b: float((a - b).norm() / max(float(b.norm()), 1e-30))
    noise = {k: rel(got[0][1][k], got[0][0][k]) for k in names}
    diff = {k: rel(got[1][0][k], got[0][0][k]) for k in names}
    behind = [k for k in names if not k.startswith(enc)]
    for k in behind:
        assert diff[k] <= max(4.0 * noise[k], 2e-5), (k, diff[k], noise[k])
    first = [enc + "layer4.2." + t for t in ("conv3.weight", "bn3.weight", "bn3.bias", "conv2.1.weight", "bn2.weight", "bn2.bias")]
    print("[parity] folded BatchNorm adjoint vs classical: first folded unit " + ", ".join("%s %.1e" % (k.split("layer4.2.")[1], diff[k]) for k in first) +
          "; all encoder tensors: median %.1e, max %.1e (%s)" % (sorted(diff[k] for k in names if k.startswith(enc))[len([k for k in names if k.startswith(enc)]) // 2],
                                                                 max(diff[k] for k in names if k.startswith(enc)), max((k for k in names if k.startswith(enc)), key=lambda k: diff[k])))
    for k in first[:3]:
        assert diff[k] <= 4e-3, (k, diff[k])
    for k in first[3:]:
        assert diff[k] <= 8e-3, (k, diff[k])
    encd = sorted(diff[k] for k in names if k.startswith(enc))
    assert encd[len(encd) // 2] <= 0.03 and encd[-1] <= 0.3, (encd[len(encd) // 2], encd[-1])


@pytest.mark.gpu
def test_bn_folded_forward_unit():
    """bn_fold_forward (bf16 training forward of the conv3 units: batch statistics from the Gram matrix of the input, BatchNorm + residual
    + ReLU + mask in the conv epilogue, z never stored) on the engine's OWN tensors: for layer3.2 and layer1.1, from the bf16 input of
    conv3 and the bf16 block input read out of the workspace,
      * saved mean / invstd == batch statistics of conv(bf16 input, bf16 w) in float32
      * the stored block output == bf16(relu(bn(z) + residual)) to one bf16 ulp
      * the stored ReLU bit mask == (block output > 0)."""
    import ctypes
    from oracle.weights import conv_specs
    B = 2
    sd = make_state_dict(11, "random")
    x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12)).to(DEV)
    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    net.train_precision = "bf16"
    L = lib()
    names = [c[0] for c in conv_specs()]
    out8 = (ctypes.c_int64 * 8)()
    L.hn_train_debug_unit(B, 0, out8)
    info = []
    for u in range(out8[7]):
        L.hn_train_debug_unit(B, u, out8)
        info.append(tuple(out8))
    unit_of = lambda name: [i for i, t in enumerate(info) if names[t[0]].endswith(name)][0]      # noqa: E731
    bon, cor = net(x)
    torch.cuda.synchronize()
    st = net._hip_states[0]
    ws8 = st.train_workspace(B)
    ws = ws8.view(torch.float32)
    enc = "feature_extractor.encoder."

    def yh(u, shape):
        off = L.hn_train_debug_unit_yh(B, u)
        n = int(np.prod(shape))
        return ws8[off * 4:off * 4 + 2 * n].view(torch.bfloat16).float().cpu().view(shape)

    ok = True
    for blk, prev, hw, kc, nc in (("layer3.2", "layer3.1", (32, 64), 256, 1024), ("layer1.1", "layer1.0", (128, 256), 64, 256)):
        u3, u2, up = unit_of(enc + blk + ".conv3"), unit_of(enc + blk + ".conv2.1"), unit_of(enc + prev + ".conv3")
        a2 = yh(u2, (B,) + hw + (kc,))
        res = yh(up, (B,) + hw + (nc,))
        got = yh(u3, (B,) + hw + (nc,))
        w = sd[enc + blk + ".conv3.weight"].bfloat16().float().view(nc, kc)
        z = a2.view(-1, kc).double() @ w.double().t()
        mean, var = z.mean(0), z.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + 1e-5)
        stt = ws[info[u3][6]:info[u3][6] + 2 * nc].cpu().double()
        ok &= report("fold fwd mean   %s" % blk, stt[:nc].numpy(), mean.numpy(), 1e-4 * float(z.abs().max()))
        ok &= report("fold fwd invstd %s" % blk, stt[nc:].numpy(), inv.numpy(), 1e-3 * float(inv.abs().max()))
        gam, bet = sd[enc + blk + ".bn3.weight"].double(), sd[enc + blk + ".bn3.bias"].double()
        want = torch.relu((z - mean) * inv * gam + bet + res.view(-1, nc).double()).float()
        ok &= report("fold fwd y      %s" % blk, got.view(-1, nc).numpy(), want.numpy(), 2.0 ** -7 * float(want.abs().max()))
        C4 = info[u3][1] * info[u3][2] // 4
        mk_off = None
        # the mask sits behind the unit's [mean|invstd|a|b] block (train.hip: add_unit takes st = 4 C floats, rounded up to 64, then mk)
        mk_off = info[u3][6] + (4 * nc + 63) // 64 * 64
        mk = ws8[mk_off * 4:mk_off * 4 + C4].cpu().numpy()
        bits = np.stack([(mk >> k) & 1 for k in range(4)], axis=1).reshape(-1)
        pos = (got.view(-1).numpy() > 0).astype(np.uint8)
        # (a value that is positive in float32 but rounds to a bf16 zero cannot occur: bf16 keeps float32's exponent range)
        mism = int((bits != pos).sum())
        print("[parity] fold fwd mask %s: %d of %d bits differ from (stored y > 0)" % (blk, mism, bits.size))
        ok &= mism == 0
    assert net.hip_status(DEV) == 0
    assert ok


@pytest.mark.gpu
def test_train_step_bf16_b16_vs_reference(golden_dir):
    """The bf16 training step at B = 16 on the TRAINED config-5 checkpoint against the UNMODIFIED reference's float32 step on the same
    sixteen panoramas and targets (tests/golden/train_step_trained_b16.npz, oracle/gen_golden.py train_big; B = 16 is the largest batch
    the 62 GB build container holds -- the reference's autograd keeps ~2 GB per panorama, configs[2]'s B = 64 does not fit): loss,
    outputs, and the norm of every parameter gradient.  A well-conditioned network at a real batch size, unlike the B = 2 seeded-random
    fixtures: here bf16's distance from float32 is a property of the kernels, not of fifty BatchNorms over two samples.  Both training
    forms run: the BatchNorm-folded units (default) and the classical passes."""
    import json
    import torch.nn.functional as F
    from tools import c5_common as c5
    path = os.path.join(golden_dir, "train_step_trained_b16.npz")
    g = np.load(path)
    names = json.load(open(path[:-4] + ".json"))["names"]
    B = int(g["bon"].shape[0])
    imgs = np.stack([c5.make_room(c5.room_jobs(1, c5.VAL_SEED0, int(i))[0])[0] for i in g["rooms"]])
    assert [c5.image_crc(im) for im in imgs] == [int(v) for v in g["crc"]]
    x = torch.FloatTensor(imgs.transpose(0, 3, 1, 2) / 255).to(DEV)
    gen = torch.Generator().manual_seed(52)
    y_bon = ((torch.rand(B, 2, 1024, generator=gen) - 0.5) * 1.2).to(DEV)
    y_cor = (torch.rand(B, 1, 1024, generator=gen) < 0.05).float().to(DEV)
    ok = True
    for fold in (1, 0):
        net = HorizonNet("resnet50", True)
        net.load_state_dict(c5.decode_state_dict(), strict=True)
        net = net.to(DEV).train()
        net.bi_rnn.dropout = 0.0
        net.drop_out.p = 0.0
        net.train_precision = "bf16"
        net.set_engine_option("fuse_bn_fold", fold)
        bon, cor = net(x)
        loss = F.l1_loss(bon, y_bon) + F.binary_cross_entropy_with_logits(cor, y_cor)
        loss.backward()
        torch.cuda.synchronize()
        assert net.hip_status(DEV) == 0
        params = dict(net.named_parameters())
        gn = np.array([float(params[k].grad.double().norm()) for k in names])
        rel = np.abs(gn - g["grad_norm"]) / np.maximum(g["grad_norm"], 1e-12)
        big = g["grad_norm"] > 1e-3 * g["grad_norm"].max()             # (tensors whose gradient is numerically nothing have no relative error)
        tot = float(np.sqrt((gn ** 2).sum()) / np.sqrt((g["grad_norm"] ** 2).sum()))
        d_out = max(float(np.abs(bon.detach().cpu().numpy() - g["bon"]).max()), float(np.abs(cor.detach().cpu().numpy() - g["cor"]).max()))
        samp = 0.0
        for k in g.files:
            if k.startswith("grad:"):
                t = params[k[5:]].grad.flatten()
                got = t[:: max(1, t.numel() // 4096)].cpu().numpy()
                samp = max(samp, float(np.abs(got - g[k]).max() / max(np.abs(g[k]).max(), 1e-30)))
        print("[parity] bf16 train step B=%d trained ckpt (fold=%d) vs reference f32: loss %.6f vs %.6f, outputs max-abs %.3e, gradient norms: total ratio %.4f, "
              "worst rel %.3e (%s), median %.3e; sampled gradient entries worst %.3e of the tensor's max" % (
                  B, fold, float(loss), float(g["loss"]), d_out, tot, float(rel[big].max()), names[int(np.argmax(np.where(big, rel, 0)))],
                  float(np.median(rel[big])), samp))
        # Round 6: the step is run-to-run REPRODUCIBLE (the stem's statistics meet in wave order, the folded units' P-GEMM goes through the ordered
        # slab reduce, their Q GEMM has at most two commuting partials; what is left are float atomics that END in leaf weight gradients, 1e-8 of a
        # norm): a second forward + backward on the same net must give the same output bits and the same gradient norms to 1e-6.  Up to round 5 the
        # worst tensor moved 6.0e-2 .. 7.4e-2 (folded) / 9.7e-2 .. 1.4e-1 (classical) between runs and the bounds below were sized to that noise.
        net.zero_grad(set_to_none=True)
        bon2, cor2 = net(x)
        (F.l1_loss(bon2, y_bon) + F.binary_cross_entropy_with_logits(cor2, y_cor)).backward()
        torch.cuda.synchronize()
        gn2 = np.array([float(params[k].grad.double().norm()) for k in names])
        rep = float((np.abs(gn2 - gn) / np.maximum(gn, 1e-30))[big].max())
        same_bits = bool(torch.equal(bon2, bon) and torch.equal(cor2, cor))
        print("[parity] bf16 train step B=%d (fold=%d) repeated: outputs bit-identical %s, worst gradient-norm change %.3e" % (B, fold, same_bits, rep))
        ok &= same_bits and rep <= 1e-6
        # measured (round 6, deterministic): folded form loss 3e-5 rel, outputs 1.10e-2, norms: total 0.9970, median 1.18e-3, worst 6.1e-2 (the stem's
        # weight: the end of the backward chain); classical form outputs 8.8e-3, total 0.9943, median 1.19e-3, worst 1.40e-1 -- the folded adjoint is
        # the MORE accurate of the two (it never rounds z or dz to bf16).  Bounds = ~1.1 x measured: the arithmetic, no longer the atomics.
        worst_b, tot_b = (0.068, 0.0035) if fold else (0.155, 0.0064)
        ok &= abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])) and d_out <= 0.0122
        ok &= abs(tot - 1.0) <= tot_b and float(rel[big].max()) <= worst_b and float(np.median(rel[big])) <= 1.32e-3
    assert ok


@pytest.mark.gpu
def test_fused_objective_equals_torch_losses():
    """hn_loss_l1_bce (train.py:53-56 in one launch: both means, their sum, both gradients) against F.l1_loss +
    F.binary_cross_entropy_with_logits and their autograd, incl. exact zeros in the L1 difference (sign(0) = 0) and a scaled adjoint."""
    from horizonnet_amd.train import objective
    g = torch.Generator().manual_seed(5)
    for B in (1, 7, 64):
        bon = ((torch.rand(B, 2, 1024, generator=g) - 0.5) * 2).to(DEV).requires_grad_(True)
        y_bon = ((torch.rand(B, 2, 1024, generator=g) - 0.5) * 2).to(DEV)
        with torch.no_grad():
            y_bon[:, :, ::17] = bon[:, :, ::17]                    # exact ties
        cor = ((torch.rand(B, 1, 1024, generator=g) - 0.5) * 30).to(DEV).requires_grad_(True)
        y_cor = torch.rand(B, 1, 1024, generator=g).to(DEV)
        got = objective(bon, y_bon, cor, y_cor)
        (got["total"] * 3.0).backward()
        gb, gc = bon.grad.clone(), cor.grad.clone()
        bon.grad = cor.grad = None
        wb, wc = F.l1_loss(bon, y_bon), F.binary_cross_entropy_with_logits(cor, y_cor)
        ((wb + wc) * 3.0).backward()
        assert abs(float(got["bon"]) - float(wb)) <= 2e-6 * abs(float(wb)) and abs(float(got["cor"]) - float(wc)) <= 2e-6 * abs(float(wc))
        assert abs(float(got["total"]) - float(wb + wc)) <= 2e-6 * abs(float(wb + wc))
        assert torch.equal(gb, bon.grad), float((gb - bon.grad).abs().max())
        assert float((gc - cor.grad).abs().max()) <= 1e-6 * float(cor.grad.abs().max())


@pytest.mark.gpu
def test_soak_deterministic_entries_repeat_bit_for_bit():
    """tools/soak_determinism.py in short form (12 iterations x B in {1, 3}, a second stream keeping HBM and CUs busy): the bf16 training forward, the
    float32 training forward after a bf16 step (the sequence of round 5's one-off failure) and the three eval entries give the same bits every
    iteration.  Before round 6 about 1 % of the B = 3 bf16 training forwards took another rounding path (float LDS atomics in the stem's statistics)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_determinism.py"), "12", "1"], capture_output=True, text=True, timeout=900)
    print(out.stdout[-1500:])
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "bit-identical every iteration" in out.stdout
