"""CPU restatement of ``HorizonNet.forward`` (resnet50 backbone, use_rnn=True, eval mode).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Functional torch fp32 on
the host cores, driven directly by a reference-format state_dict; it does not
need the reference tree or torchvision, so it travels to the GPU box.  Each step
cites the reference lines it follows.  Validated against the unmodified
reference module by ``oracle/gen_golden.py`` (max-abs printed there).
"""
import torch
import torch.nn.functional as F

from .weights import BN_EPS

X_MEAN = (0.485, 0.456, 0.406)   # reference model.py:186
X_STD = (0.229, 0.224, 0.225)    # reference model.py:187


def lr_pad(x, p):
    """Circular left/right padding -- reference model.py:27-29."""
    return torch.cat([x[..., -p:], x, x[..., :p]], dim=3)


_TRAIN = {"on": False, "momentum": 0.1, "bf16_convs": False}     # set by forward_train(): batch statistics + running-stat update


def _r16(t):
    """Round to bfloat16 (nearest even) and back: the storage precision of the engine's bf16 operands."""
    return t.to(torch.bfloat16).to(t.dtype)


class _ConvBf16Operands(torch.autograd.Function):
    """What the engine's train_precision "bf16" computes for one convolution, restated with torch ops: forward and data
    gradient see bf16-ROUNDED operands (x, w, dz) with exact products and wide accumulation, and so does the weight
    gradient (bf16 x and dz) -- except for the 7x7 stem (3 input channels) whose weight gradient, and ghc0.3 (Cout = 32,
    round_bwd False) whose data and weight gradients, stay on the float32 kernels."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, round_bwd):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, round_bwd, b is not None)
        return _r16(F.conv2d(_r16(x), _r16(w), b, stride=stride, padding=padding))     # z is stored as bf16 too

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        stride, padding, round_bwd, has_b = ctx.cfg
        wq, dq = (_r16(w), _r16(dz)) if round_bwd else (w, dz)
        dx = torch.nn.grad.conv2d_input(x.shape, wq, dq, stride=stride, padding=padding)
        dw = torch.nn.grad.conv2d_weight(_r16(x), w.shape, dq, stride=stride, padding=padding) if round_bwd and x.shape[1] % 64 == 0 \
            else torch.nn.grad.conv2d_weight(x, w.shape, dz, stride=stride, padding=padding)
        return dx, dw, (dz.sum((0, 2, 3)) if has_b else None), None, None, None


def _bn(x, sd, k):
    return F.batch_norm(x, sd[k + ".running_mean"], sd[k + ".running_var"],
                        sd[k + ".weight"], sd[k + ".bias"], _TRAIN["on"], _TRAIN["momentum"], BN_EPS)


def _conv(x, sd, k, stride, ks):
    """Conv2d with H zero-pad ks//2 and W circular-pad ks//2 (reference model.py:42-55)."""
    w = sd[k + ".weight"]
    b = sd.get(k + ".bias")
    p = ks // 2
    if p:
        x = lr_pad(x, p)
    if _TRAIN["on"] and _TRAIN["bf16_convs"]:
        return _ConvBf16Operands.apply(x, w, b, stride, (p, 0), w.shape[0] % 64 == 0)
    return F.conv2d(x, w, b, stride=stride, padding=(p, 0))


def _bottleneck(x, sd, p, stride, down):
    """torchvision Bottleneck v1.5 as called from reference model.py:78-81."""
    y = F.relu(_bn(_conv(x, sd, p + "conv1", 1, 1), sd, p + "bn1"))
    y = F.relu(_bn(_conv(y, sd, p + "conv2.1", stride, 3), sd, p + "bn2"))
    y = _bn(_conv(y, sd, p + "conv3", 1, 1), sd, p + "bn3")
    if down:
        x = _bn(_conv(x, sd, p + "downsample.0", stride, 1), sd, p + "downsample.1")
    return F.relu(y + x)


def prepare_x(x):
    """reference model.py:248-252."""
    mean = torch.tensor(X_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(X_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x[:, :3] - mean) / std


def backbone(x, sd, taps=None):
    """reference model.py:71-82: stem + maxpool + layer1..4 -> [C1..C4]."""
    enc = "feature_extractor.encoder."
    y = F.relu(_bn(_conv(x, sd, enc + "conv1.1", 2, 7), sd, enc + "bn1"))
    if taps is not None:
        taps["stem"] = y
    y = F.max_pool2d(y, 3, 2, 1)          # ordinary padding, NOT circular (model.py:44 skips it)
    if taps is not None:
        taps["pool"] = y
    feats = []
    for li, n in enumerate((3, 4, 6, 3)):
        for j in range(n):
            p = "%slayer%d.%d." % (enc, li + 1, j)
            y = _bottleneck(y, sd, p, 2 if (j == 0 and li > 0) else 1, j == 0)
        feats.append(y)
    return feats


def global_height_conv(x, sd, s, out_w):
    """reference model.py:138-156."""
    for k in range(4):
        p = "reduce_height_module.ghc_lst.%d.layer.%d.layers." % (s, k)
        x = F.relu(_bn(_conv(x, sd, p + "0.1", (2, 1), 3), sd, p + "1"))
    factor = out_w // x.shape[3]
    x = torch.cat([x[..., -1:], x, x[..., :1]], 3)
    x = F.interpolate(x, size=(x.shape[2], out_w + 2 * factor), mode="bilinear", align_corners=False)
    return x[..., factor:-factor]


def lstm_ref(feature_tbc, sd):
    """2-layer bi-LSTM, gates i,f,g,o -- explicit loops (reference model.py:222-227,263-264).

    Written out instead of calling nn.LSTM so the arithmetic being matched is visible.
    """
    T, B, _ = feature_tbc.shape
    x = feature_tbc
    for layer in range(2):
        outs = []
        for suf in ("", "_reverse"):
            wih = sd["bi_rnn.weight_ih_l%d%s" % (layer, suf)]
            whh = sd["bi_rnn.weight_hh_l%d%s" % (layer, suf)]
            bias = sd["bi_rnn.bias_ih_l%d%s" % (layer, suf)] + sd["bi_rnn.bias_hh_l%d%s" % (layer, suf)]
            gx = x @ wih.t() + bias
            h = x.new_zeros(B, 512)
            c = x.new_zeros(B, 512)
            ys = [None] * T
            order = range(T) if suf == "" else range(T - 1, -1, -1)
            for t in order:
                g = gx[t] + h @ whh.t()
                i, f, gg, o = g.chunk(4, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                ys[t] = h
            outs.append(torch.stack(ys, 0))
        x = torch.cat(outs, dim=2)
    return x


def forward(x, sd, taps=None):
    """Full eval-mode forward -- reference model.py:254-281.  x: [B,>=3,512,1024] f32 in [0,1]."""
    if x.shape[2] != 512 or x.shape[3] != 1024:
        raise NotImplementedError()
    with torch.no_grad():
        xn = prepare_x(x.float())
        feats = backbone(xn, sd, taps)
        B = x.shape[0]
        feature = torch.cat([global_height_conv(f, sd, s, 256).reshape(B, -1, 256)
                             for s, f in enumerate(feats)], dim=1)          # model.py:172-179
        seq = feature.permute(2, 0, 1).contiguous()                          # [256,B,1024]
        out = lstm_ref(seq, sd)                                              # [256,B,1024]
        lin = out @ sd["linear.weight"].t() + sd["linear.bias"]             # [256,B,12]
        lin = lin.view(256, B, 3, 4).permute(1, 2, 0, 3).contiguous().view(B, 3, 1024)
        if taps is not None:
            taps.update(c1=feats[0], c2=feats[1], c3=feats[2], c4=feats[3],
                        feature=feature, lstm=out)
        return lin[:, 1:], lin[:, :1]                                        # bon, cor


def forward_train(x, sd, momentum=0.1, bf16_convs=False):
    """Train-mode forward WITH autograd (reference train.py:44-58 calls net(x) with net.train()): batch-statistics
    BatchNorm (running stats in `sd` are updated in place), dropout disabled (p = 0: the parity tests compare
    deterministic arithmetic; dropout is checked statistically).  Tensors in `sd` that require grad get .grad from
    a backward() on the returned (bon, cor)."""
    if x.shape[2] != 512 or x.shape[3] != 1024:
        raise NotImplementedError()
    _TRAIN["on"], _TRAIN["momentum"], _TRAIN["bf16_convs"] = True, momentum, bool(bf16_convs)
    try:
        xn = prepare_x(x)                      # dtype follows x: float32 = the reference, float64 = ground truth for tests
        feats = backbone(xn, sd)
        B = x.shape[0]
        feature = torch.cat([global_height_conv(f, sd, s, 256).reshape(B, -1, 256) for s, f in enumerate(feats)], dim=1)
        seq = feature.permute(2, 0, 1).contiguous()
        out = lstm_ref(seq, sd)
        lin = out @ sd["linear.weight"].t() + sd["linear.bias"]
        lin = lin.view(256, B, 3, 4).permute(1, 2, 0, 3).contiguous().view(B, 3, 1024)
        return lin[:, 1:], lin[:, :1]
    finally:
        _TRAIN["on"] = False
        _TRAIN["bf16_convs"] = False


def forward_bf16_emulated(x, sd, bf16_lstm=True):
    """What the engine's bf16 mode computes, restated on the CPU: every tensor the engine stores as bf16 (normalised
    input, conv weights, conv outputs after the folded-BN / residual / ReLU epilogue, the sequence matrix, the LSTM
    layer-0 output fed to layer 1) is rounded to bf16; products are exact and accumulation is f32; BatchNorm is the
    f32 per-channel scale/shift fold; the gate pre-activations, gates, cell state and the head are f32.  bf16_lstm (the
    engine's default, lstm_bf16.hip): h_{t-1} and W_hh enter the recurrent matmul rounded to bf16 (f32 accumulation);
    False = the float32 recurrence (engine option "bf16_lstm" = 0).
    This is NOT the reference's arithmetic (that is forward()); it exists to check the bf16 kernels tightly."""
    def fold(k_conv, k_bn):
        s_ = sd[k_bn + ".weight"] / torch.sqrt(sd[k_bn + ".running_var"] + BN_EPS)
        b_ = sd.get(k_conv + ".bias")
        b_ = torch.zeros_like(s_) if b_ is None else b_
        return s_, (b_ - sd[k_bn + ".running_mean"]) * s_ + sd[k_bn + ".bias"]

    def unit(t, k_conv, k_bn, stride, ks, relu, res=None):
        p_ = ks // 2
        if p_:
            t = lr_pad(t, p_)
        y = F.conv2d(t, _r16(sd[k_conv + ".weight"]), None, stride=stride, padding=(p_, 0))
        s_, t_ = fold(k_conv, k_bn)
        y = y * s_.view(1, -1, 1, 1) + t_.view(1, -1, 1, 1)
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return _r16(y)

    with torch.no_grad():
        enc = "feature_extractor.encoder."
        y = _r16(prepare_x(x.float()))
        y = unit(y, enc + "conv1.1", enc + "bn1", 2, 7, True)
        y = F.max_pool2d(y, 3, 2, 1)
        feats = []
        for li, n in enumerate((3, 4, 6, 3)):
            for j in range(n):
                p = "%slayer%d.%d." % (enc, li + 1, j)
                st = 2 if (j == 0 and li > 0) else 1
                t1 = unit(y, p + "conv1", p + "bn1", 1, 1, True)
                t2 = unit(t1, p + "conv2.1", p + "bn2", st, 3, True)
                idt = unit(y, p + "downsample.0", p + "downsample.1", st, 1, False) if j == 0 else y
                y = unit(t2, p + "conv3", p + "bn3", 1, 1, True, idt)
            feats.append(y)
        B = x.shape[0]
        cols = []
        for s_i, f in enumerate(feats):
            g = f
            for k in range(4):
                p = "reduce_height_module.ghc_lst.%d.layer.%d.layers." % (s_i, k)
                g = unit(g, p + "0.1", p + "1", (2, 1), 3, True)
            fac = 256 // g.shape[3]
            gp = torch.cat([g[..., -1:], g, g[..., :1]], 3)
            up = F.interpolate(gp, size=(g.shape[2], 256 + 2 * fac), mode="bilinear", align_corners=False)[..., fac:-fac]
            cols.append(_r16(up).reshape(B, -1, 256))
        seq = torch.cat(cols, dim=1).permute(2, 0, 1).contiguous()
        xin = seq
        for layer in range(2):
            outs = []
            for suf in ("", "_reverse"):
                wih = _r16(sd["bi_rnn.weight_ih_l%d%s" % (layer, suf)])
                whh = sd["bi_rnn.weight_hh_l%d%s" % (layer, suf)]
                if bf16_lstm:
                    whh = _r16(whh)
                bias = sd["bi_rnn.bias_ih_l%d%s" % (layer, suf)] + sd["bi_rnn.bias_hh_l%d%s" % (layer, suf)]
                gx = xin @ wih.t() + bias
                h = xin.new_zeros(B, 512)
                c = xin.new_zeros(B, 512)
                ys = [None] * 256
                for t in (range(256) if suf == "" else range(255, -1, -1)):
                    g4 = gx[t] + (_r16(h) if bf16_lstm else h) @ whh.t()
                    i_, f_, gg, o_ = g4.chunk(4, dim=1)
                    c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(gg)
                    h = torch.sigmoid(o_) * torch.tanh(c)
                    ys[t] = h
                outs.append(torch.stack(ys, 0))
            out = torch.cat(outs, dim=2)
            xin = _r16(out)
        lin = out @ sd["linear.weight"].t() + sd["linear.bias"]
        lin = lin.view(256, B, 3, 4).permute(1, 2, 0, 3).contiguous().view(B, 3, 1024)
        return lin[:, 1:], lin[:, :1]


def conv_bn_act_nhwc(x_nhwc, w_oihw, bias, bn, stride, relu, residual=None):
    """Single fused conv step in NHWC, used by the per-kernel GPU parity tests.

    x_nhwc [B,H,W,Cin]; bn = (gamma, beta, mean, var) or None; W circular pad, H zero pad.
    """
    x = x_nhwc.permute(0, 3, 1, 2)
    ks = w_oihw.shape[2]
    p = ks // 2
    if p:
        x = lr_pad(x, p)
    y = F.conv2d(x, w_oihw, bias, stride=stride, padding=(p, 0))
    if bn is not None:
        y = F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, BN_EPS)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y.contiguous()
