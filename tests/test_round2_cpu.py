"""CPU tests of the round-2 host pieces: reference-format checkpoints + resume bookkeeping, the learning-rate schedule
against values produced by the reference's own ``adjust_learning_rate``, ImageNet-weight loading into the LR_PAD-wrapped
trunk, the config-5 checkpoint codec and fixtures, the training driver's flag surface."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from horizonnet_amd import HorizonNet
from horizonnet_amd import utils as hutils
from horizonnet_amd import model as hmodel
from horizonnet_amd import evaluation as ev_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lr_schedule_matches_reference_values(golden_dir):
    """tests/golden/traincurve_seed41.npz column 2 = args.running_lr written by the reference's misc.utils.adjust_learning_rate."""
    g = np.load(os.path.join(golden_dir, "traincurve_seed41.npz"))
    a = argparse.Namespace(lr=1e-4, warmup_lr=1e-6, warmup_iters=0, max_iters=int(g["max_iters"]), lr_pow=0.9, cur_iter=0, running_lr=1e-4)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    got = []
    for _ in range(len(g["curve"])):
        hutils.adjust_learning_rate(opt, a)
        a.cur_iter += 1
        got.append(opt.param_groups[0]["lr"])
    assert np.array_equal(np.array(got), g["curve"][:, 2])
    # warm-up branch (misc/utils.py:36-39)
    a = argparse.Namespace(lr=1e-4, warmup_lr=1e-6, warmup_iters=10, max_iters=100, lr_pow=0.9, cur_iter=5, running_lr=0)
    hutils.adjust_learning_rate(opt, a)
    assert a.running_lr == pytest.approx(1e-6 + (1e-4 - 1e-6) * 0.5)
    a.cur_iter = 100
    hutils.adjust_learning_rate(opt, a)
    assert a.running_lr == 0.0


def test_save_model_format_and_resume_bookkeeping(tmp_path):
    torch.manual_seed(3)
    net = HorizonNet("resnet50", True)
    args = argparse.Namespace(id="t", lr=1e-4, best_valid_score=0.5, cur_iter=12)
    pth = str(tmp_path / "m.pth")
    hutils.save_model(net, pth, args)
    blob = torch.load(pth, map_location="cpu", weights_only=False)
    assert list(blob.keys()) == ["args", "kwargs", "state_dict"] and blob["kwargs"] == {"backbone": "resnet50", "use_rnn": True}
    back = hutils.load_trained_model(HorizonNet, pth)
    for (k, a), (k2, b) in zip(net.state_dict().items(), back.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    # rolling checkpoint: optimiser state, counters and RNG streams survive
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    for p in net.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    np.random.seed(5)
    torch.manual_seed(5)
    hutils.save_checkpoint(hutils.make_checkpoint(net, opt, 3, args), True, str(tmp_path), 3)
    nxt_np, nxt_t = np.random.rand(), float(torch.rand(1))
    assert os.path.isfile(tmp_path / "checkpoint.pth.tar") and os.path.isfile(tmp_path / "best_model_3.pth.tar")
    net2 = HorizonNet("resnet50", True)
    opt2 = torch.optim.Adam(net2.parameters(), lr=1e-4)
    args2 = argparse.Namespace(best_valid_score=0, cur_iter=0, iters_per_epoch=4)
    np.random.seed(99)
    first = hutils.resume_checkpoint(str(tmp_path / "checkpoint.pth.tar"), net2, opt2, args2, torch.device("cpu"))
    assert first == 4 and args2.cur_iter == 12 and args2.best_valid_score == 0.5
    assert np.random.rand() == nxt_np and float(torch.rand(1)) == nxt_t
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)
    # a checkpoint written by the reference itself (no cur_iter / rng) resumes from the epoch count
    ref_style = {"epoch": 2, "state_dict": net.state_dict(), "optimizer": opt.state_dict(), "best_valid_score": 0.25, "backbone": "resnet50"}
    torch.save(ref_style, tmp_path / "ref.pth.tar")
    assert hutils.resume_checkpoint(str(tmp_path / "ref.pth.tar"), net2, opt2, args2, torch.device("cpu")) == 3
    assert args2.cur_iter == 8 and args2.best_valid_score == 0.25
    assert hutils.unwrap(torch.nn.DataParallel(net)) is net


def test_imagenet_weights_are_loaded_when_available(tmp_path, monkeypatch):
    """reference model.py:66-69,204-207 builds the encoder from torchvision's IMAGENET1K_V1 weights.  A file in
    torchvision's key layout must land in the LR_PAD-wrapped trunk (conv1 -> conv1.1, convN.conv2 -> conv2.1)."""
    torch.manual_seed(11)
    trunk = hmodel._ResNet50Trunk()
    tv = {}
    for k, v in trunk.state_dict().items():
        tk = k.replace("conv1.1.weight", "conv1.weight") if k == "conv1.1.weight" else k.replace(".conv2.1.weight", ".conv2.weight")
        tv[tk] = torch.randn_like(v) if v.is_floating_point() else v.clone()
    tv["fc.weight"], tv["fc.bias"] = torch.randn(1000, 2048), torch.randn(1000)
    path = str(tmp_path / "resnet50-0676ba61.pth")
    torch.save(tv, path)
    monkeypatch.setenv("HORIZONNET_RESNET50_WEIGHTS", path)
    net = HorizonNet("resnet50", True)
    assert net.feature_extractor.pretrained_loaded
    sd = net.state_dict()
    assert torch.equal(sd["feature_extractor.encoder.conv1.1.weight"], tv["conv1.weight"])
    assert torch.equal(sd["feature_extractor.encoder.layer3.4.conv2.1.weight"], tv["layer3.4.conv2.weight"])
    assert torch.equal(sd["feature_extractor.encoder.layer4.0.downsample.1.running_var"], tv["layer4.0.downsample.1.running_var"])
    # without any source: loud warning, random init (documented difference)
    monkeypatch.delenv("HORIZONNET_RESNET50_WEIGHTS")
    monkeypatch.setattr(hmodel, "_torchvision_resnet50_state_dict", lambda: None)
    with pytest.warns(RuntimeWarning, match="ImageNet"):
        net = HorizonNet("resnet50", True)
    assert not net.feature_extractor.pretrained_loaded


def test_config5_checkpoint_codec_and_fixture(golden_dir):
    from tools import c5_common as c5
    w = (np.random.RandomState(0).randn(8, 16, 3, 3) * 0.05).astype(np.float32)
    for bits in (4, 6):
        q, s = c5.quantize_tensor(w, bits)
        d = c5.dequantize_tensor(q, s)
        q2, s2 = c5.quantize_tensor(d, bits)
        assert np.array_equal(q, q2) and np.abs(np.abs(q).max(1).max() - ((1 << (bits - 1)) - 1)) == 0     # on-grid values are fixed points
        assert np.abs(d - w).max() <= 0.5 * s.max() + 1e-9
    sd = c5.decode_state_dict()
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    assert list(sd.keys()) == [k[0] if isinstance(k, (list, tuple)) else k for k in spec["keys"]]
    net = HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    big = sd["reduce_height_module.ghc_lst.3.layer.0.layers.0.1.weight"]
    assert c5.bits_for(big.shape) == 4 and len(torch.unique(big[0] / big[0].abs().max() * 7).round().unique()) <= 15
    z = np.load(os.path.join(golden_dir, "config5", "reference_layouts.npz"))
    n = int(z["n"])
    assert n == 1000 and z["count"].sum() == len(z["cor_id"]) and len(z["crc"]) == n
    mix = dict(zip(*np.unique(z["count"] // 2, return_counts=True)))
    assert mix.get(4, 0) > 300 and sum(v for k, v in mix.items() if k > 4) > 150       # general layouts, not only cuboids
    img, _ = c5.make_room(c5.room_jobs(1, int(z["seed0"]), 3)[0])
    assert c5.image_crc(img) == int(z["crc"][3])                                       # the renderer is deterministic


def test_train_driver_keeps_reference_flags():
    from horizonnet_amd import train as drv
    flags = {a.option_strings[0] for a in drv.build_parser()._actions if a.option_strings}
    reference_flags = {"--id", "--ckpt", "--logs", "--pth", "--backbone", "--no_rnn", "--train_root_dir", "--valid_root_dir", "--no_flip",
                       "--no_rotate", "--no_gamma", "--no_pano_stretch", "--num_workers", "--freeze_earlier_blocks", "--batch_size_train",
                       "--batch_size_valid", "--epochs", "--optim", "--lr", "--lr_pow", "--warmup_lr", "--warmup_epochs", "--beta1",
                       "--weight_decay", "--bn_momentum", "--no_cuda", "--multi_gpu", "--device", "--seed", "--disp_iter", "--save_every"}
    assert reference_flags <= flags, reference_flags - flags                             # train.py:63-135
    a = drv.build_parser().parse_args(["--id", "x"])
    assert (a.lr, a.lr_pow, a.epochs, a.batch_size_train, a.optim, a.seed, a.save_every) == (1e-4, 0.9, 300, 8, "Adam", 594277, 25)


def test_frozen_block_fixture_is_consistent(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "train_step_frozen_seed33.json")))
    net = HorizonNet("resnet50", True)
    names = [k for k, _ in net.named_parameters()]
    assert names == meta["names"]
    blocks = net.feature_extractor.list_blocks()
    for i in range(meta["freeze_earlier_blocks"] + 1):
        for m in blocks[i]:
            for p in m.parameters():
                p.requires_grad = False
    assert [k for k, p in net.named_parameters() if p.requires_grad] == meta["live"]
    bn_names = [k for k, m in net.named_modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(bn_names) == 69 and all(any(b.startswith(n) for n in bn_names) for b in meta["frozen_bn_buffers"])


# ---- general polygon intersection (eval_general.py:69-79: shapely / GEOS in the reference) --------------------------------
def test_exact_polygon_oracle_closed_forms():
    """oracle/polygon_ref.py (rational arithmetic, boundary-integral form) against areas known in closed form, including
    every degenerate contact two Manhattan floor plans produce: shared walls, identical rings, touching corners."""
    from fractions import Fraction as Fr
    from oracle.polygon_ref import area_exact, intersection_area_exact as ix
    sq = [(0, 0), (2, 0), (2, 2), (0, 2)]
    ell = [(0, 0), (4, 0), (4, 1), (1, 1), (1, 4), (0, 4)]
    assert ix(sq, [(1, 1), (3, 1), (3, 3), (1, 3)]) == 1
    assert ix(sq, sq) == 4 and ix(sq, sq[::-1]) == 4 and ix(sq, sq + [sq[0]]) == 4           # identical, reversed, closed ring
    assert ix(ell, ell[2:] + ell[:2]) == area_exact(ell) == 7
    assert ix(sq, [(2, 0), (4, 0), (4, 2), (2, 2)]) == 0                                     # shared wall, interiors apart
    assert ix(sq, [(2, 2), (3, 2), (3, 3), (2, 3)]) == 0                                     # touching corner
    assert ix(sq, [(1, 0), (3, 0), (3, 2), (1, 2)]) == 2                                     # two shared walls, overlapping
    assert ix(sq, [(0.5, 0.5), (1.5, 0.5), (1.5, 1.5), (0.5, 1.5)]) == 1                     # nested
    assert ix(sq, [(0, 0), (1, 0), (1, 1), (0, 1)]) == 1                                     # nested, sharing a corner and two half walls
    assert ix(sq, [(5, 5), (6, 5), (6, 6)]) == 0
    assert ix(ell, [(0.5, 0.5), (3, 0.5), (3, 3), (0.5, 3)]) == Fr(9, 4)
    assert ix(ell, [(0, 0), (4, 0), (4, 4), (0, 4)]) == 7                                    # notch of the L lies on the square's inside
    assert ix([(0, 0), (4, 0), (0, 4)], [(0, 0), (4, 0), (4, 4)]) == 4                       # crossing diagonals: triangle (0,0),(4,0),(2,2)
    tri = [(0, 0), (3, 0), (0, 3)]
    assert ix(tri, [(1, -1), (2, -1), (2, 5), (1, 5)]) == Fr(3, 2)                           # strip through a triangle: x in [1,2], y in [0,3-x]


def test_polygon_intersection_area_matches_exact_oracle():
    """The product's float64 slab decomposition (horizonnet_amd/evaluation.py) against the exact oracle on random rooms:
    non-convex star polygons in general position, Manhattan rooms against shifted / re-notched copies (shared and
    collinear-overlapping walls), and identical rings.  |error| <= 1e-12 x area scale."""
    from oracle.polygon_ref import intersection_area_exact as ix
    from tools.synth_rooms import manhattan_polygon
    rng = np.random.RandomState(7)
    worst = 0.0

    def check(a, b):
        nonlocal worst
        want = float(ix(a, b))
        got = ev_mod.polygon_intersection_area(a, b)
        scale = max(ev_mod.polygon_area(a), ev_mod.polygon_area(b))
        worst = max(worst, abs(got - want) / scale)
        assert abs(got - want) <= 1e-12 * scale, (a, b, got, want)
        return want

    def star(n):
        from horizonnet_amd import postproc
        while True:                 # vertices sorted by angle: simple unless an angular gap exceeds pi -- reject those
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            rad = rng.uniform(0.6, 3.0, n)
            p = np.stack([rad * np.cos(ang) + rng.uniform(-0.5, 0.5), rad * np.sin(ang) + rng.uniform(-0.5, 0.5)], 1)
            if postproc.polygon_is_simple(p):
                return p

    nonzero = 0
    for _ in range(60):
        nonzero += check(star(rng.randint(3, 12)), star(rng.randint(3, 12))) > 0
    for _ in range(60):
        a = manhattan_polygon(rng, int(rng.choice([4, 6, 8, 10, 12])))
        b = manhattan_polygon(rng, int(rng.choice([4, 6, 8, 10, 12])))
        nonzero += check(a, b) > 0
        check(a, a.copy())                                                       # identical rings: every wall shared
        shift = np.array([rng.choice([0.0, 0.25, -0.5]), rng.choice([0.0, 0.125, 1.0])])
        check(a, a + shift)                                                      # collinear overlapping walls
        c = a.copy()
        c[:, 0] = np.where(c[:, 0] == c[:, 0].max(), c[:, 0] - 0.3, c[:, 0])     # one wall moved, the others shared
        check(a, c)
        check(a, star(7))                                                        # Manhattan against general position
    assert nonzero >= 100
    print("[parity] polygon intersection vs exact rational oracle: worst relative error %.2e" % worst)


def test_fused_adam_reads_torch_adam_state():
    """the default optimiser of horizonnet_amd.train is FusedAdam, the reference's checkpoint.pth.tar holds
    torch.optim.Adam's state (train.py:216-225,336-346: Adam over the parameters with requires_grad).  The conversion
    scatters exp_avg / exp_avg_sq to the engine's flat offsets; SGD state restarts the moments with a warning."""
    from horizonnet_amd import _lib
    from horizonnet_amd.optim import torch_adam_state_to_flat
    L = _lib.load()
    torch.manual_seed(11)
    net = HorizonNet("resnet50", True)
    for blk in net.feature_extractor.list_blocks()[:2]:            # --freeze_earlier_blocks 1
        for m in blk:
            for p in m.parameters():
                p.requires_grad = False
    live = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam(live, lr=3e-4, betas=(0.8, 0.99), eps=1e-7, weight_decay=1e-3)
    g = torch.Generator().manual_seed(12)
    for _ in range(2):
        for p in live:
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
        opt.step()
    sd = opt.state_dict()
    entries = [(int(L.hn_grad_offset(k.encode())), k, tuple(p.shape), bool(p.requires_grad)) for k, p in net.named_parameters()]
    total = int(L.hn_grad_floats())
    m, v, step, hyper = torch_adam_state_to_flat(sd, entries, total)
    assert step == 2 and hyper == {"lr": 3e-4, "betas": (0.8, 0.99), "eps": 1e-7, "weight_decay": 1e-3}
    idx, covered = 0, torch.zeros(total, dtype=torch.bool)
    for o, k, shape, req in entries:
        n = int(np.prod(shape))
        if req:
            assert torch.equal(m[o:o + n].view(shape), sd["state"][idx]["exp_avg"]), k
            assert torch.equal(v[o:o + n].view(shape), sd["state"][idx]["exp_avg_sq"]), k
            covered[o:o + n] = True
            idx += 1
    assert idx == len(live) and float(m[~covered].abs().sum()) == 0.0        # frozen parameters and alignment gaps stay zero
    # an optimiser over ALL parameters (no filter) maps one to one as well
    opt_all = torch.optim.Adam(net.parameters(), lr=1e-4)
    for p in net.parameters():
        p.grad = torch.full_like(p, 2e-3)
    opt_all.step()
    m2, _, step2, _ = torch_adam_state_to_flat(opt_all.state_dict(), entries, total)
    o, k, shape, _ = entries[0]
    assert step2 == 1 and torch.equal(m2[o:o + int(np.prod(shape))].view(shape), opt_all.state_dict()["state"][0]["exp_avg"])
    # SGD state: hyper-parameters only, moments restart
    sgd = torch.optim.SGD(live, lr=0.1, momentum=0.9)
    sgd.step()
    with pytest.warns(RuntimeWarning, match="not Adam"):
        m3, v3, step3, hyper3 = torch_adam_state_to_flat(sgd.state_dict(), entries, total)
    assert step3 == 0 and float(m3.abs().sum()) == 0.0 and hyper3["lr"] == 0.1
    with pytest.raises(KeyError):
        torch_adam_state_to_flat({"foo": 1}, entries, total)


def test_resume_restores_each_ranks_own_rng_streams(tmp_path):
    """only rank 0 writes the checkpoint; every rank must continue ITS OWN torch / numpy streams (dropout seeds,
    augmentation draws), not rank 0's.  Without per-rank states (older checkpoint, other world size) ranks > 0 are
    re-seeded deterministically and differently from rank 0."""
    torch.manual_seed(1)
    net = HorizonNet("resnet50", True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    args = argparse.Namespace(best_valid_score=0.0, cur_iter=7)
    states, nxt = [], []
    for r in range(2):
        np.random.seed(100 + r)
        torch.manual_seed(100 + r)
        states.append({"torch": torch.get_rng_state(), "numpy": np.random.get_state()})
        nxt.append((np.random.rand(), float(torch.rand(1))))
    hutils.save_checkpoint(hutils.make_checkpoint(net, opt, 1, args, rng_ranks=states), False, str(tmp_path), 1)
    path = str(tmp_path / "checkpoint.pth.tar")
    for r in range(2):
        a = argparse.Namespace(best_valid_score=0, cur_iter=0, iters_per_epoch=4)
        np.random.seed(5)
        torch.manual_seed(5)
        hutils.resume_checkpoint(path, net, opt, a, torch.device("cpu"), rank=r, world=2)
        assert a.rng_restore == "own" and (np.random.rand(), float(torch.rand(1))) == nxt[r]
    # world size changed: rank 0 continues its stream, the others get distinct deterministic seeds
    draws = {}
    for r in (0, 1, 2, 2):
        a = argparse.Namespace(best_valid_score=0, cur_iter=0, iters_per_epoch=4)
        hutils.resume_checkpoint(path, net, opt, a, torch.device("cpu"), rank=r, world=4)
        d = (np.random.rand(), float(torch.rand(1)))
        if r in draws:
            assert draws[r] == d                                   # deterministic
        draws[r] = d
        assert a.rng_restore == ("own" if r == 0 else "reseeded")
    assert draws[0] == nxt[0] and len({draws[0], draws[1], draws[2]}) == 3


def test_layout_comparison_handles_the_column_seam():
    """tools/c5_layout.seam_aware_max_abs: a corner that moves across the 0 / 1024 column seam re-orders the corner list; the raw
    |a - b| reads ~1000 px for a sub-pixel move, the seam-aware figure reads the move."""
    import numpy as np
    from tools.c5_layout import seam_aware_max_abs
    a = np.array([[0.2, 100], [0.2, 400], [300, 110], [300, 390], [700, 100], [700, 400], [900, 90], [900, 410]], float)
    b = np.array([[300.5, 110], [300.5, 390], [700, 100.4], [700, 400], [900, 90], [900, 410], [1023.9, 100], [1023.9, 400]], float)
    assert np.abs(a - b).max() > 300                      # what the round-3 bench line printed
    assert abs(seam_aware_max_abs(a, b) - 0.5) < 1e-9     # the real displacement: 0.5 px (x of the second corner)
    assert seam_aware_max_abs(a, a) == 0.0
    c = a.copy()
    c[3, 1] += 7.0
    assert seam_aware_max_abs(a, c) == 7.0


def test_data_parallel_over_several_devices_fails_at_wrap_time():
    """train.py:190-192 on a multi-GPU box: nn.DataParallel(net, device_ids=[0, 1]) must fail WHERE IT IS WRITTEN, with the
    torchrun command to use instead -- not at the first forward of a replica; other modules and the one-device wrap are untouched."""
    import pytest
    import torch.nn as nn
    import horizonnet_amd
    net = horizonnet_amd.HorizonNet("resnet50", True)
    with pytest.raises(RuntimeError, match="torch.distributed.run --nnodes=1 --nproc-per-node 2"):
        nn.DataParallel(net, device_ids=[0, 1])
    nn.DataParallel(nn.Linear(2, 2))                      # any other module: as before
    assert nn.DataParallel(net).module is net              # no visible devices here -> no replicas -> allowed (the one-device form runs on the GPU box)
