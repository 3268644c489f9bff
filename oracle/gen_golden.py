"""Generate ``tests/golden/*`` from the UNMODIFIED reference, and pin the oracle against it.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs
``/root/reference``); the GPU box never runs it.  Usage:

    python -m oracle.gen_golden            # from the repo root

It (1) imports reference ``model.py`` / ``misc/panostretch.py`` / ``inference.py``
with the stand-ins under ``oracle/standins`` on ``sys.path``, (2) loads seeded
weights (``oracle/weights.py``) into the reference ``HorizonNet``, (3) checks the
restatements in ``oracle/*.py`` against the reference outputs (prints max-abs and
asserts), and (4) writes small fixtures.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import horizonnet_ref, panostretch_ref, peaks_ref   # noqa: E402
from oracle.weights import make_state_dict, state_dict_spec      # noqa: E402


def load_demo_u8():
    from PIL import Image
    img = np.array(Image.open(os.path.join(REF, "assets", "demo.png")))[..., :3]
    assert img.shape == (512, 1024, 3) and img.dtype == np.uint8
    return img


def sample(t, step):
    """Strided sample of an NCHW tensor for a compact fixture."""
    t = t.detach().cpu().numpy()
    if t.ndim == 4:
        return t[:, ::step[0], ::step[1], ::step[2]].copy()
    return t


TAP_STEPS = {"stem": (8, 16, 32), "pool": (8, 8, 16), "c1": (16, 8, 16), "c2": (32, 4, 8),
             "c3": (64, 2, 4), "c4": (128, 1, 2)}


def gen_model():
    import model as ref_model                      # the reference's model.py, unmodified
    net = ref_model.HorizonNet("resnet50", True).eval()
    ref_sd = net.state_dict()
    spec = state_dict_spec()
    assert list(ref_sd.keys()) == list(spec.keys()), "state_dict key order mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]) and v.dtype == spec[k][1], k
    nparam = sum(p.numel() for p in net.parameters())
    assert len(ref_sd) == 448 and nparam == 81570348, (len(ref_sd), nparam)
    with open(os.path.join(GOLD, "state_dict_spec.json"), "w") as f:
        json.dump({"n_tensors": len(ref_sd), "n_params": nparam,
                   "keys": [[k, list(v.shape), str(v.dtype)] for k, v in ref_sd.items()]}, f)

    demo = load_demo_u8()
    np.savez_compressed(os.path.join(GOLD, "demo_input_u8.npz"), img=demo)
    x_demo = torch.from_numpy(demo.transpose(2, 0, 1)[None].astype(np.float32) / 255.0)  # inference.py:196-200
    g = torch.Generator().manual_seed(1234)
    x_rand = torch.rand(2, 3, 512, 1024, generator=g)

    cases = [("demo_seed0_bnrandom", 0, "random", x_demo),
             ("demo_seed1_bnidentity", 1, "identity", x_demo),
             ("rand2_seed2_bnrandom", 2, "random", x_rand)]
    for name, seed, bn, x in cases:
        sd = make_state_dict(seed, bn)
        net.load_state_dict(sd, strict=True)
        feats = {}
        with torch.no_grad():
            bon, cor = net(x)
            xn = net._prepare_x(x)
            c = net.feature_extractor(xn)
            feature = net.reduce_height_module(c, 256)
            lstm_out, _ = net.bi_rnn(feature.permute(2, 0, 1))
        taps = {}
        obon, ocor = horizonnet_ref.forward(x, sd, taps)
        d_bon = (obon - bon).abs().max().item()
        d_cor = (ocor - cor).abs().max().item()
        d_feat = (taps["feature"] - feature).abs().max().item()
        d_lstm = (taps["lstm"] - lstm_out).abs().max().item()
        rel_c = [((taps["c%d" % (i + 1)] - c[i]).abs().max() / c[i].abs().max()).item() for i in range(4)]
        print("%-24s oracle-vs-reference: bon %.2e cor %.2e feature %.2e lstm %.2e c1..4(rel) %s |bon|max %.3f |cor|max %.3f |feature|max %.2f cmax %s" % (
            name, d_bon, d_cor, d_feat, d_lstm, ["%.1e" % r for r in rel_c], bon.abs().max().item(),
            cor.abs().max().item(), feature.abs().max().item(), ["%.1f" % t.abs().max().item() for t in c]))
        assert d_bon < 2e-5 and d_cor < 2e-5, "oracle restatement deviates from the reference"
        sig = torch.sigmoid(cor)[:, 0].numpy()
        out = {"bon": bon.numpy(), "cor": cor.numpy(), "feature": feature.numpy()[:, ::8],
               "lstm": lstm_out.numpy()[::8], "seed": seed, "bn": bn,
               "w_checksum": np.float64(sd["feature_extractor.encoder.conv1.1.weight"].double().sum().item()),
               "w_checksum_lstm": np.float64(sd["bi_rnn.weight_hh_l1_reverse"].double().sum().item())}
        for k, st in TAP_STEPS.items():
            src = taps[k] if k in ("stem", "pool") else c[int(k[1]) - 1]
            out["tap_" + k] = sample(src, st)
        np.savez_compressed(os.path.join(GOLD, "forward_%s.npz" % name), **out)
        del sig


def gen_panostretch():
    from misc import panostretch as ref_ps
    demo = load_demo_u8().astype(np.float32) / 255.0                # dataset.py:51-54
    corners = np.array([[100.5, 150.0], [100.5, 400.0], [400.0, 160.0], [400.0, 380.0],
                        [700.0, 140.0], [700.0, 410.0], [950.0, 170.0], [950.0, 370.0]], np.float32)
    out = {"corners_in": corners}
    params = [(1.0, 1.0), (1.7, 1.2), (0.55, 1.9), (2.0, 0.5), (1.3, 1.3)]
    out["params"] = np.array(params, np.float64)
    for i, (kx, ky) in enumerate(params):
        ref_img, ref_cor = ref_ps.pano_stretch(demo, corners, kx, ky)
        o_img, o_cor = panostretch_ref.pano_stretch(demo, corners, kx, ky)
        d = np.abs(o_img - ref_img)
        print("pano_stretch kx=%.2f ky=%.2f oracle-vs-reference img max %.2e (n>1e-6: %d) corners max %.2e" % (
            kx, ky, d.max(), int((d > 1e-6).sum()), np.abs(o_cor - ref_cor).max()))
        assert d.max() < 1e-6 or kx == ky
        assert np.abs(o_cor - ref_cor).max() < 1e-9
        out["rows_%d" % i] = np.concatenate([ref_img[0:4], ref_img[254:258], ref_img[508:512]], 0)
        out["cols_%d" % i] = np.concatenate([ref_img[:, 0:4], ref_img[:, 1020:1024]], 1)
        out["grid_%d" % i] = ref_img[::8, ::8].copy()
        out["corners_%d" % i] = ref_cor
    # small full-size-independent cases: random images, odd sizes, C=1 and C=4
    rng = np.random.RandomState(7)
    small = []
    for j, (h, w, c, kx, ky) in enumerate([(64, 128, 3, 1.5, 0.8), (32, 64, 1, 0.6, 1.4),
                                            (48, 96, 4, 1.0, 1.9), (17, 40, 3, 1.25, 1.1)]):
        img = rng.rand(h, w, c).astype(np.float32)
        ref_img, _ = ref_ps.pano_stretch(img, corners[:2], kx, ky)
        o_img, _ = panostretch_ref.pano_stretch(img, corners[:2], kx, ky)
        print("pano_stretch small %dx%dx%d max %.2e" % (h, w, c, np.abs(o_img - ref_img).max()))
        assert np.abs(o_img - ref_img).max() < 1e-6
        out["small_in_%d" % j] = img
        out["small_out_%d" % j] = ref_img
        small.append((kx, ky))
    out["small_params"] = np.array(small, np.float64)
    np.savez_compressed(os.path.join(GOLD, "panostretch.npz"), **out)


def gen_peaks():
    import inference as ref_inf                    # reference inference.py (shapely stubbed)
    rng = np.random.RandomState(11)
    sigs, outs = [], {}
    for j in range(12):
        base = rng.rand(1024).astype(np.float32)
        k = np.exp(-0.5 * (np.arange(-20, 21) / (2.0 + j)) ** 2).astype(np.float32)
        s = np.convolve(np.tile(base, 3), k / k.sum(), mode="same")[1024:2048].astype(np.float32)
        s = (s - s.min()) / (s.max() - s.min())
        if j % 3 == 0:                              # plateaus / exact ties
            s = np.round(s * 16) / 16
        if j % 4 == 1:                              # sparse peaky signal like a trained cor head
            s = (s ** 8).astype(np.float32)
        sigs.append(s.astype(np.float32))
    sigs = np.stack(sigs)
    outs["signals"] = sigs
    cfgs = [(26, 0.05, None), (26, 0.0, 4), (29, 0.05, None), (5, 0.3, None), (26, 0.5, 4)]
    outs["cfgs"] = np.array([[r, mv, -1 if n is None else n] for r, mv, n in cfgs], np.float64)
    for ci, (r, mv, n) in enumerate(cfgs):
        for j in range(sigs.shape[0]):
            loc, val = ref_inf.find_N_peaks(sigs[j], r=r, min_v=mv, N=n)
            oloc, oval = peaks_ref.find_N_peaks(sigs[j], r=r, min_v=mv, N=n)
            assert np.array_equal(loc, oloc) and np.array_equal(val, oval), (ci, j)
            outs["loc_%d_%d" % (ci, j)] = loc.astype(np.int64)
    print("find_N_peaks: oracle == reference on %d signals x %d configs" % (sigs.shape[0], len(cfgs)))
    # sigmoid restatement vs torch (inference.py:80)
    x = torch.linspace(-12, 12, 4097)
    d = np.abs(peaks_ref.sigmoid_f32(x.numpy()) - torch.sigmoid(x).numpy()).max()
    print("sigmoid_f32 vs torch.sigmoid max %.2e" % d)
    np.savez_compressed(os.path.join(GOLD, "peaks.npz"), **outs)


def gen_stretch_params():
    """dataset.py:70-81,189-208 -- reference cor2xybound + the clamped (kx, ky) sampling sequence."""
    import dataset as ref_ds                        # reference dataset.py (shapely stubbed, never called here)
    rng = np.random.RandomState(21)
    out = {}
    cors = []
    for j in range(8):                              # synthetic cuboid rooms, label_cor layout (ceil, floor per wall corner)
        n = 4 if j < 6 else 6
        xs = np.sort(rng.uniform(20, 1000, n))
        yc = rng.uniform(100, 200, n)
        yf = rng.uniform(320, 420, n)
        cor = np.stack([np.repeat(xs, 2), np.stack([yc, yf], 1).reshape(-1)], 1).astype(np.float32)
        cors.append(cor)
        out["cor_%d" % j] = cor
        out["bound_%d" % j] = np.array(ref_ds.cor2xybound(cor), np.float64)
        ks = []
        for seed in range(6):
            np.random.seed(seed * 7 + j)
            xmin, ymin, xmax, ymax = ref_ds.cor2xybound(cor)
            kx = np.random.uniform(1.0, 2.0)
            ky = np.random.uniform(1.0, 2.0)
            if np.random.randint(2) == 0:
                kx = max(1 / kx, min(0.5 / xmin, 1.0))
            else:
                kx = min(kx, max(10.0 / xmax, 1.0))
            if np.random.randint(2) == 0:
                ky = max(1 / ky, min(0.5 / ymin, 1.0))
            else:
                ky = min(ky, max(10.0 / ymax, 1.0))
            ks.append((kx, ky))
        out["k_%d" % j] = np.array(ks, np.float64)
    np.savez_compressed(os.path.join(GOLD, "stretch_params.npz"), **out)
    print("stretch_params: %d rooms written" % len(cors))


def gen_postproc():
    """misc/post_proc.py + inference.py:65-141 on synthetic Manhattan rooms (oracle/synth_rooms.py): per-function
    outputs of the reference, and its whole ``inference()`` driven by a signal-decoding stand-in network."""
    import inference as ref_inf                      # reference inference.py
    from misc import post_proc as ref_pp             # reference misc/post_proc.py
    from oracle import synth_rooms as sr
    import contextlib, io
    rng = np.random.RandomState(5)
    out, meta = {}, []
    H, W = 512, 1024
    j = 0
    for case in range(64):
        n_corners = [4, 4, 6, 6, 8, 8, 10, 4][case % 8]
        noise = [0.0, 0.3, 0.6, 1.2][(case // 8) % 4]
        poly = sr.manhattan_polygon(rng, n_corners)
        rows, cor, cols = sr.render(poly, rng.uniform(1.0, 1.6), rng.uniform(1.2, 1.7), noise, rng)
        kind = "clean"
        if case >= 32 and case % 3 == 0 and n_corners > 4:
            cor = sr.drop_corner(cor, cols, rng.randint(len(cols))); kind = "dropped"
        elif case >= 32 and case % 3 == 1:
            cor = sr.add_corner(cor, rng.uniform(0, W)); kind = "spurious"
        x = sr.encode_image(rows, cor)
        rec = {"case": j, "n_corners": n_corners, "noise": noise, "kind": kind, "runs": []}
        # -- per-function outputs on the signals exactly as inference.py:89-93 prepares them
        y_bon = (sr.SignalNet()(x)[0][0].numpy() / np.pi + 0.5) * H - 0.5
        y_bon[0] = np.clip(y_bon[0], 1, H / 2 - 1)
        y_bon[1] = np.clip(y_bon[1], H / 2 + 1, H - 2)
        y_cor = torch.sigmoid(sr.SignalNet()(x)[1])[0, 0].numpy()
        refined, z1 = ref_pp.np_refine_by_fix_z(y_bon[0], y_bon[1], 50)
        out["x_%d" % j] = x[0, :, 0, :].numpy()
        out["ycor_%d" % j] = y_cor                   # y_bon is exactly-rounded elementwise arithmetic on x: tests recompute it
        out["refined_%d" % j] = refined[::8]
        out["z1_%d" % j] = np.float64(z1)
        out["coory_%d" % j] = ref_pp.infer_coory(y_bon[0], z1 - 50, 50)[::8]
        tol = abs(0.16 * z1 / 1.6)
        for cub in (True, False):
            tag = "cub" if cub else "gen"
            pk = ref_inf.find_N_peaks(y_cor, r=int(round(W * 0.05 / 2)), min_v=0 if cub else 0.05, N=4 if cub else None)[0]
            out["pk_%s_%d" % (tag, j)] = pk
            try:
                c, walls = ref_pp.gen_ww(pk, y_bon[0], 50, tol=tol, force_cuboid=cub)
                out["ww_%s_%d" % (tag, j)] = c
                out["wtype_%s_%d" % (tag, j)] = np.array([w["type"] for w in walls], np.int64)
                out["wval_%s_%d" % (tag, j)] = np.array([w["val"] for w in walls], np.float64)
                out["wscore_%s_%d" % (tag, j)] = np.array([w["score"] for w in walls], np.float64)
                rec["actions_" + tag] = [w.get("action", "") for w in walls]
            except AssertionError:
                rec["actions_" + tag] = "AssertionError"
        # -- the whole reference inference() with the decoding stand-in network
        runs = [dict(force_cuboid=False), dict(force_cuboid=True), dict(force_raw=True),
                dict(force_cuboid=False, flip=True, rotate=[0.25, -0.1]), dict(force_cuboid=False, min_v=0.3, r=0.1)]
        for k, kw in enumerate(runs):
            err = io.StringIO()
            try:
                with contextlib.redirect_stderr(err):
                    cor_id, z0, z1r, _ = ref_inf.inference(sr.SignalNet(), x, "cpu", **kw)
                out["inf%d_%d" % (k, j)] = cor_id
                out["infz1_%d_%d" % (k, j)] = np.float64(z1r)
                rec["runs"].append({"kw": kw, "ok": True, "fallback": "fallback" in err.getvalue()})
            except AssertionError:
                rec["runs"].append({"kw": kw, "ok": False})
        meta.append(rec)
        j += 1
    # -- hard cases: strong noise, missed and spurious corners; general layout only (exercises the cuboid fallback)
    rng = np.random.RandomState(11)
    for case in range(96):
        poly = sr.manhattan_polygon(rng, [6, 8, 10, 12][case % 4])
        rows, cor, cols = sr.render(poly, 1.2, 1.5, rng.choice([0.5, 1.5, 3.0]), rng)
        for _ in range(rng.randint(0, 3)):
            cor = sr.drop_corner(cor, cols, rng.randint(len(cols)))
        for _ in range(rng.randint(0, 3)):
            cor = sr.add_corner(cor, rng.uniform(0, W))
        x = sr.encode_image(rows, cor)
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            cor_id, z0, z1r, _ = ref_inf.inference(sr.SignalNet(), x, "cpu", force_cuboid=False)
        fallback = "fallback" in err.getvalue()
        if not fallback and case % 3:
            continue                                 # keep every fallback and a third of the rest
        out["x_%d" % j] = x[0, :, 0, :].numpy()
        out["ycor_%d" % j] = torch.sigmoid(sr.SignalNet()(x)[1])[0, 0].numpy()
        out["inf0_%d" % j] = cor_id
        out["infz1_0_%d" % j] = np.float64(z1r)
        meta.append({"case": j, "kind": "hard", "runs": [{"kw": {"force_cuboid": False}, "ok": True, "fallback": fallback}]})
        j += 1
    np.savez_compressed(os.path.join(GOLD, "postproc.npz"), **out)
    with open(os.path.join(GOLD, "postproc.json"), "w") as f:
        json.dump(meta, f, indent=0)
    acts = {}
    for m in meta:
        a = m.get("actions_gen", [])
        for t in ([a] if isinstance(a, str) else a):
            acts[t] = acts.get(t, 0) + 1
    print("postproc: %d rooms; general-layout wall actions %s; fallbacks %d; failed runs %d" % (
        len(meta), acts, sum(r.get("fallback", False) for m in meta for r in m["runs"]),
        sum(not r["ok"] for m in meta for r in m["runs"])))


DATASET_CONFIGS = [dict(flip=True, rotate=True, gamma=True, stretch=True), dict(),
                   dict(stretch=True), dict(gamma=True), dict(flip=True, rotate=True)]


def gen_dataset():
    """dataset.py:12-134 (PanoCorBonDataset.__getitem__) on a tiny synthetic dataset in the reference's on-disk format
    (committed under tests/golden/synth_ds), plus eval_general.layout_2_depth / eval_cuboid.eval_3diou on its rooms."""
    import shutil
    import dataset as ref_ds                         # reference dataset.py
    import eval_general as ref_eg
    import eval_cuboid as ref_ec
    from scipy.spatial.distance import cdist as _cdist
    from oracle import synth_rooms as sr, dataset_ref
    from horizonnet_amd import dataset as my_ds      # host-only helpers (label parsing, draw order)
    from PIL import Image
    # dataset.py:109-117 passes p=1 to cdist's default euclidean metric, which SciPy >= 1.9 rejects; on 1-D points the
    # euclidean and the intended Minkowski-1 distance are the same number, so drop the argument (environment shim)
    ref_ds.cdist = lambda a, b, p=None: _cdist(a, b)
    root = os.path.join(GOLD, "synth_ds")
    shutil.rmtree(root, ignore_errors=True)
    sr.write_dataset(root, 4, seed=4, visible_only=False, noise=0.6)
    out, meta = {}, []
    for ci, cfg in enumerate(DATASET_CONFIGS):
        ds = ref_ds.PanoCorBonDataset(root, return_cor=True, **cfg)
        for i in range(len(ds)):
            for rep in range(2 if ci == 0 else 1):
                seed = 1000 * ci + 10 * i + rep
                np.random.seed(seed)
                x, bon, y_cor, cor = ds[i]
                key = "c%d_i%d_r%d" % (ci, i, rep)
                # pin the oracle's image pipeline (the GPU tests' checker) on the same draws
                np.random.seed(seed)
                c0, _ = my_ds.read_label(os.path.join(root, "label_cor", ds.txt_fnames[i]))
                full = dict(flip=False, rotate=False, gamma=False, stretch=False)
                full.update(cfg)
                a = my_ds.draw_augmentation(c0, 1024, full["flip"], full["rotate"], full["gamma"], full["stretch"])
                src = np.array(Image.open(os.path.join(root, "img", ds.img_fnames[i])))
                mine = dataset_ref.augment_image(src, a["kx"], a["ky"], a["flip"], a["roll"] if a["rotate"] else None, a["gamma"])
                assert np.array_equal(mine, x.numpy()), "oracle image pipeline differs from the reference: " + key
                out["x_" + key] = x.numpy()[:, 3::16, 5::16].copy()
                out["xsum_" + key] = np.float64(x.numpy().astype(np.float64).sum())
                out["bon_" + key] = bon.numpy()
                out["ycor_" + key] = y_cor.numpy()
                out["cor_" + key] = np.asarray(cor)
                meta.append({"key": key, "cfg": cfg, "index": i, "seed": seed})
    # metrics: depth maps of the ground-truth layouts and 3D IoU between jittered cuboids (Qhull reference)
    def ref_pp_np_refine(cor):
        return ref_ec.post_proc.np_refine_by_fix_z(cor[0::2, 1].astype(np.float64), cor[1::2, 1].astype(np.float64), 50)

    rng = np.random.RandomState(9)
    for i in range(4):
        cor = np.loadtxt(os.path.join(root, "label_cor", "room_%05d.txt" % i)).astype(np.float32)
        cor = np.roll(cor, -2 * np.argmin(cor[::2, 0]), 0)
        try:
            out["depth_%d" % i] = ref_eg.layout_2_depth(cor, 512, 1024)[::8, ::8].copy()
        except AssertionError:
            pass
    ious = []
    for k in range(24):
        polys = []
        base = sr.manhattan_polygon(rng, 4)
        for _ in range(2):
            q = base * rng.uniform(0.8, 1.25, (1, 2)) + rng.uniform(-0.3, 0.3, (1, 2))
            polys.append(sr.room_corners(q, rng.uniform(1.0, 1.6), 1.6).astype(np.float64))
        a, b = polys
        iou = ref_ec.eval_3diou(a[1::2], a[0::2], b[1::2], b[0::2])
        out["pair_a_%d" % k], out["pair_b_%d" % k] = a, b
        ious.append(iou)
    out["pair_iou3d"] = np.array(ious, np.float64)
    # eval_cuboid.test end to end: a predicted cuboid as inference() would emit it (corners from the jittered box,
    # floor rows re-derived from z1 as inference.py:129 does) against the ground truth
    cub = []
    for k in range(24):
        gt, dt = out["pair_a_%d" % k].astype(np.float32), out["pair_b_%d" % k].astype(np.float32)
        z0 = 50
        _, z1 = ref_pp_np_refine(dt)
        dt[1::2, 1] = ref_ec.post_proc.infer_coory(dt[0::2, 1], z1 - z0, z0)
        losses = {"CE": [], "PE": [], "3DIoU": []}
        ref_ec.test(dt, z0, z1, gt, 1024, 512, losses)
        out["cub_dt_%d" % k], out["cub_z1_%d" % k] = dt, np.float64(z1)
        cub.append([losses["CE"][0], losses["PE"][0], losses["3DIoU"][0]])
    out["cub_metrics"] = np.array(cub, np.float64)
    np.savez_compressed(os.path.join(GOLD, "dataset.npz"), **out)
    with open(os.path.join(GOLD, "dataset.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("dataset: %d samples written; cuboid-pair 3D IoU range %.1f..%.1f %%" % (len(meta), min(ious), max(ious)))


def gen_train():
    """The training step of reference train.py:44-58 (feed_forward: L1 + BCE-with-logits on net(x) in train mode) and
    its backward pass on the unmodified reference module: pins oracle.forward_train (batch-statistics BN, running-stat
    update) and its autograd -- the checker of the engine's training step -- against the reference."""
    import torch.nn.functional as F
    import model as ref_model
    B = 2
    sd = make_state_dict(31, "random")
    g = torch.Generator().manual_seed(32)
    x = torch.rand(B, 3, 512, 1024, generator=g)
    y_bon = (torch.rand(B, 2, 1024, generator=g) - 0.5) * 1.2
    y_cor = (torch.rand(B, 1, 1024, generator=g) < 0.05).float()
    net = ref_model.HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.bi_rnn.dropout = 0.0             # dropout off: deterministic arithmetic (masks are checked statistically on the GPU)
    net.drop_out.p = 0.0
    bon, cor = net(x)
    loss = F.l1_loss(bon, y_bon) + F.binary_cross_entropy_with_logits(cor, y_cor)     # train.py:53-56
    loss.backward()
    ref_g = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    ref_sd_after = {k: v.detach().clone() for k, v in net.state_dict().items()}

    osd = {k: v.clone() for k, v in sd.items()}
    for k, v in osd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    obon, ocor = horizonnet_ref.forward_train(x, osd, 0.1)
    oloss = F.l1_loss(obon, y_bon) + F.binary_cross_entropy_with_logits(ocor, y_cor)
    oloss.backward()
    worst, worst_k = 0.0, ""
    for k, gref in ref_g.items():
        n = float(gref.norm())
        if n < 1e-9 or k.endswith("layers.0.1.bias"):      # conv bias in front of a batch-statistics BN: true gradient 0, noise only
            continue
        e = float((osd[k].grad - gref).norm()) / n
        if e > worst:
            worst, worst_k = e, k
    d_rm = max(float((osd[k] - ref_sd_after[k]).abs().max()) for k in osd if k.endswith("running_mean") or k.endswith("running_var"))
    print("train step oracle-vs-reference: loss %.7f vs %.7f, outputs %.2e / %.2e, worst gradient L2-rel %.2e (%s), running stats %.2e" % (
        float(oloss), float(loss), float((obon - bon).abs().max()), float((ocor - cor).abs().max()), worst, worst_k, d_rm))
    assert abs(float(oloss) - float(loss)) < 1e-5 and worst < 5e-3 and d_rm < 1e-4, "oracle training step deviates from the reference"
    out = {"loss": np.float64(float(loss)), "bon": bon.detach().numpy(), "cor": cor.detach().numpy()}
    names = list(ref_g.keys())
    out["grad_norm"] = np.array([float(ref_g[k].double().norm()) for k in names], np.float64)
    out["grad_sum"] = np.array([float(ref_g[k].double().sum()) for k in names], np.float64)
    for k in ("linear.weight", "bi_rnn.bias_ih_l1", "reduce_height_module.ghc_lst.3.layer.3.layers.0.1.weight",
              "feature_extractor.encoder.layer4.2.bn3.weight", "feature_extractor.encoder.conv1.1.weight"):
        t = ref_g[k].flatten()
        out["grad:" + k] = t[:: max(1, t.numel() // 4096)].numpy().copy()
    out["rm:bn1"] = ref_sd_after["feature_extractor.encoder.bn1.running_mean"].numpy()
    out["rv:bn1"] = ref_sd_after["feature_extractor.encoder.bn1.running_var"].numpy()
    np.savez_compressed(os.path.join(GOLD, "train_step_seed31.npz"), **out)
    with open(os.path.join(GOLD, "train_step_seed31.json"), "w") as f:
        json.dump({"names": names, "B": B, "weights_seed": 31, "data_seed": 32}, f)


def gen_train_big():
    """One training step (train.py:44-58: L1 + BCE-with-logits, train-mode BatchNorm, dropout off) of the UNMODIFIED reference in float32
    at the LARGEST batch this 62 GB build container holds (B = 16; the reference's autograd keeps ~2 GB per panorama, so BASELINE
    configs[2]'s B = 64 does not fit), on the TRAINED config-5 checkpoint and 16 of its synthetic Structured3D-shaped panoramas: loss,
    outputs and the norm of every parameter gradient.  The engine's bf16 training step at the same batch is held against it
    (tests/test_gpu_train.py: test_train_step_bf16_b16_vs_reference) -- a well-conditioned network, unlike the B = 2 seeded-random
    fixtures where fifty batch-statistics BatchNorms over two panoramas amplify any rounding."""
    import torch.nn.functional as F
    import model as ref_model
    from tools import c5_common as c5
    B = int(os.environ.get("HN_GOLDEN_BIG_B", "16"))
    sd = c5.decode_state_dict()
    rooms = list(range(B))
    imgs = np.stack([c5.make_room(c5.room_jobs(1, c5.VAL_SEED0, i)[0])[0] for i in rooms])
    x = torch.FloatTensor(imgs.transpose(0, 3, 1, 2) / 255)
    g = torch.Generator().manual_seed(52)
    y_bon = (torch.rand(B, 2, 1024, generator=g) - 0.5) * 1.2
    y_cor = (torch.rand(B, 1, 1024, generator=g) < 0.05).float()
    net = ref_model.HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    bon, cor = net(x)
    loss = F.l1_loss(bon, y_bon) + F.binary_cross_entropy_with_logits(cor, y_cor)
    loss.backward()
    names = [k for k, _ in net.named_parameters()]
    grads = {k: p.grad.detach() for k, p in net.named_parameters()}
    out = {"loss": np.float64(float(loss)), "bon": bon.detach().numpy(), "cor": cor.detach().numpy(),
           "grad_norm": np.array([float(grads[k].double().norm()) for k in names], np.float64),
           "rooms": np.array(rooms), "crc": np.array([c5.image_crc(im) for im in imgs], np.uint32)}
    for k in ("linear.weight", "feature_extractor.encoder.layer4.2.conv3.weight", "feature_extractor.encoder.layer1.1.conv3.weight",
              "feature_extractor.encoder.layer1.0.downsample.0.weight", "feature_extractor.encoder.conv1.1.weight"):
        t = grads[k].flatten()
        out["grad:" + k] = t[:: max(1, t.numel() // 4096)].numpy().copy()
    print("train step B=%d on the trained checkpoint: loss %.6f, |grad| total %.4e" % (B, float(loss), float(np.sqrt((out["grad_norm"] ** 2).sum()))))
    np.savez_compressed(os.path.join(GOLD, "train_step_trained_b%d.npz" % B), **out)
    with open(os.path.join(GOLD, "train_step_trained_b%d.json" % B), "w") as f:
        json.dump({"names": names, "B": B, "data_seed": 52}, f)


def gen_train_frozen():
    """train.py:200-208,245-256 (--freeze_earlier_blocks 1): the parameters of blocks 0..1 (stem, layer1) have
    requires_grad False AND those modules are put in eval() every epoch, so their BatchNorms normalise with the running
    statistics and leave them untouched, while the rest of the network is in train mode.  One step of the unmodified
    reference in that state: outputs, loss, gradient norms of the live parameters, running statistics after the step."""
    import torch.nn.functional as F
    import model as ref_model
    B = 2
    sd = make_state_dict(33, "random")
    g = torch.Generator().manual_seed(34)
    x = torch.rand(B, 3, 512, 1024, generator=g)
    y_bon = (torch.rand(B, 2, 1024, generator=g) - 0.5) * 1.2
    y_cor = (torch.rand(B, 1, 1024, generator=g) < 0.05).float()
    net = ref_model.HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    blocks = list(net.feature_extractor.list_blocks())
    for i in range(2):
        for m in blocks[i]:
            for p_ in m.parameters():
                p_.requires_grad = False
    net.train()
    for i in range(2):
        for m in blocks[i]:
            m.eval()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    bon, cor = net(x)
    loss = F.l1_loss(bon, y_bon) + F.binary_cross_entropy_with_logits(cor, y_cor)
    loss.backward()
    names = [k for k, p_ in net.named_parameters()]
    live = [k for k, p_ in net.named_parameters() if p_.requires_grad]
    assert all(p_.grad is None for k, p_ in net.named_parameters() if not p_.requires_grad)
    gr = dict((k, p_.grad.detach()) for k, p_ in net.named_parameters() if p_.requires_grad)
    after = net.state_dict()
    frozen_bn = [k for k in after if ("running_" in k or "num_batches" in k) and
                 (k.startswith("feature_extractor.encoder.bn1") or k.startswith("feature_extractor.encoder.layer1"))]
    assert all(torch.equal(after[k], sd[k]) for k in frozen_bn), "frozen BN buffers moved in the reference?"
    out = {"loss": np.float64(float(loss)), "bon": bon.detach().numpy(), "cor": cor.detach().numpy(),
           "grad_norm": np.array([float(gr[k].double().norm()) for k in live], np.float64),
           "grad_sum": np.array([float(gr[k].double().sum()) for k in live], np.float64)}
    for k in ("linear.weight", "feature_extractor.encoder.layer2.0.conv1.weight", "feature_extractor.encoder.layer2.0.bn1.weight",
              "reduce_height_module.ghc_lst.0.layer.0.layers.0.1.weight"):
        t = gr[k].flatten()
        out["grad:" + k] = t[:: max(1, t.numel() // 4096)].numpy().copy()
    out["rm:layer2.0.bn1"] = after["feature_extractor.encoder.layer2.0.bn1.running_mean"].numpy()
    out["rv:layer2.0.bn1"] = after["feature_extractor.encoder.layer2.0.bn1.running_var"].numpy()
    out["nbt:layer2.0.bn1"] = after["feature_extractor.encoder.layer2.0.bn1.num_batches_tracked"].numpy()
    np.savez_compressed(os.path.join(GOLD, "train_step_frozen_seed33.npz"), **out)
    with open(os.path.join(GOLD, "train_step_frozen_seed33.json"), "w") as f:
        json.dump({"names": names, "live": live, "frozen_bn_buffers": frozen_bn, "B": B, "weights_seed": 33, "data_seed": 34,
                   "freeze_earlier_blocks": 1}, f)
    print("train step with frozen blocks 0..1: loss %.7f, %d live / %d parameters" % (float(loss), len(live), len(names)))


def gen_traincurve():
    """BASELINE configs[2] in miniature, from the unmodified reference: train.py's loop (:246-286 -- poly learning rate
    via misc.utils.adjust_learning_rate, feed_forward's L1 + BCE losses, Adam) over batches drawn through the reference's
    own PanoCorBonDataset with every augmentation on (dataset.py:70-105: Pano-Stretch, flip, roll, gamma) from the
    committed synthetic dataset.  Dropout off, float32, CPU.  The engine's bench / tests
    redo the same seeds through DeviceBatcher + the HIP training step and compare the loss curve."""
    import argparse
    import torch.nn.functional as F
    import model as ref_model
    import dataset as ref_ds
    from misc.utils import adjust_learning_rate
    from scipy.spatial.distance import cdist as _cdist
    ref_ds.cdist = lambda a, b, p=None: _cdist(a, b)          # SciPy >= 1.9 shim, see gen_dataset
    root = os.path.join(GOLD, "synth_ds")
    ds = ref_ds.PanoCorBonDataset(root, flip=True, rotate=True, gamma=True, stretch=True)
    B, K = 4, 5
    sd = make_state_dict(41, "random")
    net = ref_model.HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.bi_rnn.dropout = 0.0
    net.drop_out.p = 0.0
    args = argparse.Namespace(lr=1e-4, warmup_lr=1e-6, warmup_iters=0, max_iters=300, lr_pow=0.9, cur_iter=0, running_lr=1e-4)
    opt = torch.optim.Adam(filter(lambda p_: p_.requires_grad, net.parameters()), lr=args.lr, betas=(0.9, 0.999), weight_decay=0)
    curve, xsums = [], []
    for k in range(K):
        adjust_learning_rate(opt, args)
        args.cur_iter += 1
        np.random.seed(7000 + k)
        idx = np.random.randint(len(ds), size=B)
        samples = [ds[int(i)] for i in idx]
        x = torch.stack([s_[0] for s_ in samples])
        y_bon = torch.stack([s_[1] for s_ in samples])
        y_cor = torch.stack([s_[2] for s_ in samples])
        opt.zero_grad()
        bon, cor = net(x)
        l_bon, l_cor = F.l1_loss(bon, y_bon), F.binary_cross_entropy_with_logits(cor, y_cor)
        (l_bon + l_cor).backward()
        opt.step()
        curve.append([float(l_bon), float(l_cor), args.running_lr])
        xsums.append(float(x.double().sum()))
        print("traincurve step %d: bon %.6f cor %.6f lr %.3e" % (k, float(l_bon), float(l_cor), args.running_lr), flush=True)
    np.savez_compressed(os.path.join(GOLD, "traincurve_seed41.npz"), curve=np.array(curve, np.float64),
                        xsum=np.array(xsums, np.float64), B=np.int64(B), weights_seed=np.int64(41), data_seed0=np.int64(7000),
                        max_iters=np.int64(args.max_iters))


def gen_trained():
    """Full taps on TRAINED weights (the forward goldens above use seeded random weights only): the config-5
    checkpoint (tests/golden/config5/ckpt_q.npz, trained on synthetic Structured3D-shaped rooms) through the UNMODIFIED
    reference on two of the config-5 panoramas -- outputs, strided stem / pool / C1..C4, feature and bi-LSTM taps."""
    import model as ref_model
    from tools import c5_common as c5
    net = ref_model.HorizonNet("resnet50", True).eval()
    sd = c5.decode_state_dict()
    net.load_state_dict(sd, strict=True)
    rooms = [0, 7]
    imgs = np.stack([c5.make_room(c5.room_jobs(1, c5.VAL_SEED0, i)[0])[0] for i in rooms])
    x = torch.FloatTensor(imgs.transpose(0, 3, 1, 2) / 255)                      # inference.py:199-200
    with torch.no_grad():
        bon, cor = net(x)
        xn = net._prepare_x(x)
        c = net.feature_extractor(xn)
        feature = net.reduce_height_module(c, 256)
        lstm_out, _ = net.bi_rnn(feature.permute(2, 0, 1))
    taps = {}
    obon, ocor = horizonnet_ref.forward(x, sd, taps)
    d_bon, d_cor = (obon - bon).abs().max().item(), (ocor - cor).abs().max().item()
    print("trained_c5 oracle-vs-reference: bon %.2e cor %.2e |bon|max %.3f |cor|max %.3f" % (d_bon, d_cor, bon.abs().max().item(), cor.abs().max().item()))
    assert d_bon < 2e-5 and d_cor < 2e-5
    out = {"bon": bon.numpy(), "cor": cor.numpy(), "feature": feature.numpy()[:, ::8], "lstm": lstm_out.numpy()[::8],
           "rooms": np.array(rooms), "crc": np.array([c5.image_crc(im) for im in imgs], np.uint32)}
    for k, st in TAP_STEPS.items():
        src = taps[k] if k in ("stem", "pool") else c[int(k[1]) - 1]
        out["tap_" + k] = sample(src, st)
    np.savez_compressed(os.path.join(GOLD, "forward_trained_c5.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    from oracle.hostinfo import usable_cores
    torch.set_num_threads(usable_cores())
    which = sys.argv[1:] or ["model", "panostretch", "peaks", "stretch", "postproc", "dataset", "train"]
    if "trained" in which:
        gen_trained()
    if "train" in which:
        gen_train()
    if "train_frozen" in which:
        gen_train_frozen()
    if "train_big" in which:
        gen_train_big()
    if "traincurve" in which:
        gen_traincurve()
    if "dataset" in which:
        gen_dataset()
    if "postproc" in which:
        gen_postproc()
    if "stretch" in which:
        gen_stretch_params()
    if "panostretch" in which:
        gen_panostretch()
    if "peaks" in which:
        gen_peaks()
    if "model" in which:
        gen_model()
