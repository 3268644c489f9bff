"""Seeded weight generator for the 448-tensor HorizonNet(resnet50, use_rnn=True) state_dict.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

There is no checkpoint on disk and no network, so every
parity test runs on weights regenerated from a seed.  The key names and shapes
restate what the reference module tree produces (reference ``model.py:42-55``
renames wrapped convs to ``...conv1.1.weight`` / ``...conv2.1.weight`` /
``...layers.0.1.{weight,bias}``; ``model.py:123-146,222-233``); the exact list is
pinned by ``tests/golden/state_dict_spec.json`` generated from the real reference.
"""
import math
from collections import OrderedDict

import torch

BN_EPS = 1e-5


def conv_specs():
    """[(conv_key_prefix, bn_key_prefix, cin, cout, k, has_bias)] in state_dict order."""
    out = []
    enc = "feature_extractor.encoder."
    out.append((enc + "conv1.1", enc + "bn1", 3, 64, 7, False))
    cin = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for j in range(n):
            p = "%slayer%d.%d." % (enc, li + 1, j)
            out.append((p + "conv1", p + "bn1", cin, planes, 1, False))
            out.append((p + "conv2.1", p + "bn2", planes, planes, 3, False))
            out.append((p + "conv3", p + "bn3", planes, planes * 4, 1, False))
            if j == 0:
                out.append((p + "downsample.0", p + "downsample.1", cin, planes * 4, 1, False))
            cin = planes * 4
    for s, c in enumerate((256, 512, 1024, 2048)):
        chans = (c, c // 2, c // 2, c // 4, c // 8)
        for k in range(4):
            p = "reduce_height_module.ghc_lst.%d.layer.%d.layers." % (s, k)
            out.append((p + "0.1", p + "1", chans[k], chans[k + 1], 3, True))
    return out


def state_dict_spec():
    """OrderedDict key -> (shape tuple, dtype) for all 448 tensors."""
    spec = OrderedDict()
    for ck, bk, cin, cout, k, has_bias in conv_specs():
        spec[ck + ".weight"] = ((cout, cin, k, k), torch.float32)
        if has_bias:
            spec[ck + ".bias"] = ((cout,), torch.float32)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            spec[bk + "." + nm] = ((cout,), torch.float32)
        spec[bk + ".num_batches_tracked"] = ((), torch.int64)
    for layer in range(2):
        for suf in ("", "_reverse"):
            spec["bi_rnn.weight_ih_l%d%s" % (layer, suf)] = ((2048, 1024), torch.float32)
            spec["bi_rnn.weight_hh_l%d%s" % (layer, suf)] = ((2048, 512), torch.float32)
            spec["bi_rnn.bias_ih_l%d%s" % (layer, suf)] = ((2048,), torch.float32)
            spec["bi_rnn.bias_hh_l%d%s" % (layer, suf)] = ((2048,), torch.float32)
    spec["linear.weight"] = ((12, 1024), torch.float32)
    spec["linear.bias"] = ((12,), torch.float32)
    return spec


def make_state_dict(seed=0, bn="random"):
    """Deterministic weights.

    bn="identity": gamma=1, beta=0, mean=0, var=1 (what a fresh module holds).
    bn="random":   non-trivial running stats / affine so that the eval-mode BN
                   fold is actually exercised; the last BN of every bottleneck
                   gets a smaller gamma so activations stay O(1..10).
    Conv init: backbone Kaiming-normal fan_out (published ResNet recipe), the
    height-compression convs / LSTM / Linear torch's default uniform bounds,
    head bias fill as reference ``model.py:231-233``.
    """
    g = torch.Generator().manual_seed(int(seed))
    sd = OrderedDict()

    def uni(shape, lo, hi):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    def nrm(shape, std):
        return torch.randn(shape, generator=g) * std

    for ck, bk, cin, cout, k, has_bias in conv_specs():
        if has_bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            sd[ck + ".weight"] = uni((cout, cin, k, k), -bound, bound)
            sd[ck + ".bias"] = uni((cout,), -bound, bound)
        else:
            sd[ck + ".weight"] = nrm((cout, cin, k, k), math.sqrt(2.0 / (cout * k * k)))
        if bn == "identity":
            sd[bk + ".weight"] = torch.ones(cout)
            sd[bk + ".bias"] = torch.zeros(cout)
            sd[bk + ".running_mean"] = torch.zeros(cout)
            sd[bk + ".running_var"] = torch.ones(cout)
        elif bn == "random":
            last = bk.endswith("bn3") or bk.endswith("downsample.1")
            gs = 0.5 if last else 1.0
            sd[bk + ".weight"] = uni((cout,), 0.5, 1.5) * gs
            sd[bk + ".bias"] = nrm((cout,), 0.2)
            sd[bk + ".running_mean"] = nrm((cout,), 0.2)
            sd[bk + ".running_var"] = uni((cout,), 0.5, 2.0)
        else:
            raise ValueError(bn)
        sd[bk + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    bound = 1.0 / math.sqrt(512)
    for layer in range(2):
        for suf in ("", "_reverse"):
            sd["bi_rnn.weight_ih_l%d%s" % (layer, suf)] = uni((2048, 1024), -bound, bound)
            sd["bi_rnn.weight_hh_l%d%s" % (layer, suf)] = uni((2048, 512), -bound, bound)
            sd["bi_rnn.bias_ih_l%d%s" % (layer, suf)] = uni((2048,), -bound, bound)
            sd["bi_rnn.bias_hh_l%d%s" % (layer, suf)] = uni((2048,), -bound, bound)
    bound = 1.0 / math.sqrt(1024)
    sd["linear.weight"] = uni((12, 1024), -bound, bound)
    b = uni((12,), -bound, bound)
    b[0:4] = -1.0
    b[4:8] = -0.478
    b[8:12] = 0.425
    sd["linear.bias"] = b
    assert list(sd.keys()) == list(state_dict_spec().keys())
    return sd
