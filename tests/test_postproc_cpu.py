"""Layout post-processing (horizonnet_amd/postproc.py, inference.py host half) against fixtures generated from the
unmodified reference (oracle/gen_golden.py: gen_postproc).  Host numpy only -- runs without a GPU; the peak finder is
injected (oracle restatement here, the HIP kernel in tests/test_gpu_parity.py)."""
import json
import os

import numpy as np
import pytest
import torch

from horizonnet_amd import postproc as pp
from horizonnet_amd.inference import augment, augment_undo, inference, inference_batch, layout_from_signals
from oracle import peaks_ref, synth_rooms as sr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "postproc.npz")), json.load(open(os.path.join(GOLD, "postproc.json")))


def image_of(g, j):
    return torch.from_numpy(np.broadcast_to(g["x_%d" % j][:, None, :], (3, 512, 1024)).copy())[None]


def rows_of(g, j):
    y = (sr.SignalNet()(image_of(g, j))[0][0].numpy() / np.pi + 0.5) * 512 - 0.5
    y[0] = np.clip(y[0], 1, 255)
    y[1] = np.clip(y[1], 257, 510)
    return y


def test_height_fit_and_floor_rows_bit_exact(gold):
    g, meta = gold
    for m in meta:
        if m["kind"] == "hard":
            continue
        j = m["case"]
        y = rows_of(g, j)
        refined, z1 = pp.refine_by_fix_z(y[0], y[1], 50)
        assert z1 == g["z1_%d" % j]
        assert np.array_equal(refined[::8], g["refined_%d" % j])
        assert np.array_equal(pp.infer_coory(y[0], z1 - 50, 50)[::8], g["coory_%d" % j])


@pytest.mark.parametrize("cuboid", [True, False])
def test_wall_fit_bit_exact(gold, cuboid):
    g, meta = gold
    tag = "cub" if cuboid else "gen"
    seen = set()
    for m in meta:
        if m["kind"] == "hard":
            continue
        j = m["case"]
        y = rows_of(g, j)
        z1 = float(g["z1_%d" % j])
        cor, walls = pp.gen_ww(g["pk_%s_%d" % (tag, j)], y[0], 50, tol=abs(0.16 * z1 / 1.6), force_cuboid=cuboid)
        assert np.array_equal(cor, g["ww_%s_%d" % (tag, j)]), j
        assert [w["type"] for w in walls] == g["wtype_%s_%d" % (tag, j)].tolist()
        assert np.array_equal(np.array([w["val"] for w in walls]), g["wval_%s_%d" % (tag, j)])
        assert np.array_equal(np.array([w["score"] for w in walls], np.float64), g["wscore_%s_%d" % (tag, j)])
        if not cuboid:
            assert [w["action"] for w in walls] == m["actions_gen"]
            seen.update(m["actions_gen"])
    if not cuboid:      # the fixtures reach every branch of the general-layout state machine
        assert seen == {"ori", "forced infer", "forced change"}


def test_vote_matches_dense_formulation():
    """The two-pointer vote against a direct transcription of the N x N search it replaces."""
    rng = np.random.RandomState(0)
    for t in range(300):
        L = rng.randint(1, 40)
        vec = np.round(rng.normal(0, rng.choice([0.5, 3, 20]), L), rng.randint(0, 3))    # rounding makes ties
        tol = float(rng.choice([0.0, 0.5, 3.0, 7.5, 50.0]))
        v = np.sort(vec)
        best = None
        for i in range(L):
            for j in range(L):
                span = j - i + 1.0
                d = 0.0 if i == j else abs(v[i] - v[j]) + 1e-9
                if span < L * 0.4 or d > tol:
                    continue
                if best is None or span > best[0]:
                    best = (span, i, j)
        if best is None or L < tol:
            want = (np.median(v), 0)
        elif best[2] <= best[1]:
            with pytest.raises(AssertionError):
                pp.vote(vec, tol)
            continue
        else:
            want = (v[best[1]:best[2] + 1].mean(), (best[2] - best[1] + 1) / L)
        got = pp.vote(vec, tol)
        assert got[0] == want[0] and got[1] == want[1], (t, L, tol)
        assert got[2] == np.abs(v - want[0]).mean()


def test_inference_end_to_end_bit_exact(gold):
    """inference() with the signal-decoding stand-in network == the reference's inference() on the same input,
    for cuboid / general / raw outputs, flip + rotate test-time augmentation, custom min_v / r, and the cases where
    the general layout self-intersects and the cuboid fallback is taken."""
    g, meta = gold
    fallbacks = 0
    for m in meta:
        j = m["case"]
        x = image_of(g, j)
        for k, run in enumerate(m["runs"]):
            assert run["ok"]
            cor_id, z0, z1, vis = inference(sr.SignalNet(), x, "cpu", peaks_fn=peaks_ref.find_N_peaks, **run["kw"])
            assert z0 == 50 and vis is None
            assert z1 == g["infz1_%d_%d" % (k, j)]
            assert cor_id.dtype == np.float32 and np.array_equal(cor_id, g["inf%d_%d" % (k, j)]), (j, run)
            fallbacks += bool(run.get("fallback"))
    assert fallbacks >= 5


def test_inference_batch_equals_single(gold):
    g, meta = gold
    js = [m["case"] for m in meta[:6]]
    xb = torch.cat([image_of(g, j) for j in js], 0)
    res = inference_batch(sr.SignalNet(), xb, "cpu", flip=True, rotate=[0.125], peaks_fn=peaks_ref.find_N_peaks)
    for j, (cor_id, z0, z1) in zip(js, res):
        one = inference(sr.SignalNet(), image_of(g, j), "cpu", flip=True, rotate=[0.125], peaks_fn=peaks_ref.find_N_peaks)
        assert np.array_equal(cor_id, one[0]) and z1 == one[2]


def test_device_side_augmentation_mean_is_numpys_mean():
    """inference_batch / inference_stream undo the test-time augmentation and average the views on the device
    (_undo_mean_device) instead of numpy (augment_undo(...).mean(0), inference.py:46-62,84-85): same bits for 1..5 views."""
    from horizonnet_amd.inference import _undo_mean_device
    g = torch.Generator().manual_seed(5)
    for flip, rotate in ((False, []), (True, []), (True, [0.125]), (False, [0.3, -0.2]), (True, [0.1, 0.2, 0.33])):
        x = torch.rand(3, 3, 4, 1024, generator=g)
        xa, tags = augment(x, flip, rotate)
        y = (torch.rand(xa.shape[0], 2, 1024, generator=g) - 0.5) * 3.0
        want = augment_undo(y, tags).mean(0)
        got = _undo_mean_device(y, tags).numpy()
        assert want.dtype == np.float32 and got.dtype == np.float32 and np.array_equal(want, got), (flip, rotate)


def test_augment_round_trip_and_visualize(gold):
    g, _ = gold
    x = image_of(g, 0)
    xa, tags = augment(x, True, [0.25, -0.1])
    assert tags == ["", "flip", "rotate 256", "rotate -102"] and xa.shape[0] == 4
    assert torch.equal(xa[1], torch.flip(x[0], dims=[-1])) and torch.equal(xa[2], torch.roll(x[0], 256, dims=-1))
    back = augment_undo(xa[:, :, 0, :], tags)
    assert back.shape == (4, 1, 3, 1024) and all(np.array_equal(back[i, 0], x[0, :, 0].numpy()) for i in range(4))
    with pytest.raises(NotImplementedError):
        augment_undo(xa[:, :, 0, :], ["", "shear", "", ""])
    vis = inference(sr.SignalNet(), x, "cpu", visualize=True, force_raw=True)[3]
    assert vis.shape == (30 + 3 + 512, 1024, 3) and vis.dtype == np.uint8


def test_polygon_validity_rule():
    assert pp.polygon_is_simple([(0, 0), (2, 0), (2, 2), (0, 2)])
    assert not pp.polygon_is_simple([(0, 0), (2, 2), (2, 0), (0, 2)])           # bow tie
    assert not pp.polygon_is_simple([(0, 0), (2, 0), (2, 2), (1, 0), (0, 2)])    # vertex touching an edge
    assert not pp.polygon_is_simple([(0, 0), (1, 0), (2, 0)])                    # no area
    # adversarial rings, answered as the OGC / GEOS validity rules shapely documents answer them (the library itself is absent offline):
    assert not pp.polygon_is_simple([(0, 0), (4, 0), (2, 2), (4, 4), (0, 4), (2, 2)])                 # ring touching itself at a vertex ("ring self-intersection")
    assert not pp.polygon_is_simple([(0, 0), (4, 0), (4, 4), (2, 4), (2, 6), (2, 4), (0, 4)])        # zero-area spike: an edge run forth and back
    assert not pp.polygon_is_simple([(0, 0), (4, 0), (4, 4), (2, 4), (2, 0), (0, 4)])                 # edge through a vertex of another edge + crossing
    assert pp.polygon_is_simple([(0, 0), (2, 0), (2, 2), (0, 2), (0, 0)])                            # explicitly closed ring: the repeated point is fine
    assert pp.polygon_is_simple([(0, 0), (2, 0), (2, 0), (2, 2), (0, 2)])                            # a repeated consecutive point is fine
    assert pp.polygon_is_simple([(0, 0), (4, 0), (4, 1), (1, 1), (1, 3), (4, 3), (4, 4), (0, 4)])    # non-convex "C": valid
    assert pp.polygon_is_simple([(0, 0), (0, 2), (2, 2), (2, 0)])                                    # clockwise orientation: valid (orientation is not a validity rule)
    assert not pp.polygon_is_simple([(0, 0), (2, 0), (2, 2), (0, 2), (0, 0), (2, 0), (2, 2), (0, 2)])  # the same ring traversed twice: every edge is shared
    from oracle.standins.shapely.geometry import Polygon
    rng = np.random.RandomState(3)
    for _ in range(500):
        ring = rng.randint(0, 6, (rng.randint(3, 9), 2)).astype(float)
        assert pp.polygon_is_simple(ring) == Polygon(ring).is_valid


def test_product_never_imports_oracle_postproc():
    import importlib
    for name in ("horizonnet_amd.inference", "horizonnet_amd.postproc"):
        src = open(importlib.import_module(name).__file__).read()
        assert "import oracle" not in src and "from oracle" not in src


def test_native_vote_scan_equals_interpreted_loop():
    """hn_vote_scan (host function of the native library) makes exactly the decisions of the interpreted two-pointer loop --
    which test_vote_equals_dense_reference_formulation ties to the reference's N x N span matrix (post_proc.py:75-98) -- on
    clustered / uniform / constant / tiny vectors and zero / negative / huge tolerances."""
    import ctypes
    from horizonnet_amd import _lib
    rng = np.random.RandomState(3)
    L = _lib.load()
    best = (ctypes.c_int32 * 3)()
    n = 0
    for trial in range(400):
        k = int(rng.choice([1, 2, 3, 5, 17, 64, 200]))
        kind = trial % 4
        if kind == 0:
            v = rng.normal(0, 1, k)
        elif kind == 1:
            v = np.concatenate([rng.normal(5, 0.01, k), rng.uniform(-50, 50, max(1, k // 3))])
        elif kind == 2:
            v = np.full(k, 3.25)
        else:
            v = np.round(rng.uniform(0, 4, k), 1)                       # many exact ties
        v = np.sort(v.astype(np.float64))
        for tol in (0.0, -1.0, 1e-9, 0.05, 0.5, 3.0, 1e6, float(len(v)) + 0.5):
            want = pp._vote_scan_py(v, tol)
            _lib.check(L.hn_vote_scan(v.ctypes.data, len(v), float(tol), ctypes.addressof(best)), "hn_vote_scan")
            assert (int(best[0]), int(best[1]), int(best[2])) == want, (v, tol)
            n += 1
    assert n == 3200
    # and the float32 route stays on the interpreted loop (numpy scalar arithmetic in the array's own type)
    v32 = np.sort(rng.normal(0, 1, 50).astype(np.float32))
    assert pp.vote(v32, 0.3)[1] == pp._vote_scan_py(v32, 0.3)[0] / 50 or pp.vote(v32, 0.3)[1] == 0


def test_host_half_pooled_and_shared_memory_equal_inline(gold):
    """_finish_host (the host half of inference_batch / inference_stream) on signals laid out as the device half leaves them:
    inline, pooled with pickled jobs, and pooled with the shared-memory hand-over return the same layouts, bit for bit,
    including ragged batch sizes and a second batch through the same shared buffers."""
    from horizonnet_amd import inference as inf_mod
    import importlib
    inf_mod = importlib.import_module("horizonnet_amd.inference")
    g, meta = gold
    js = [m["case"] for m in meta][:40]
    xs = torch.cat([image_of(g, j) for j in js], 0)
    with torch.no_grad():
        yb, yc = sr.SignalNet()(xs.float())
    yc = torch.sigmoid(yc)[:, 0].contiguous()

    class Done:
        def synchronize(self):
            pass

    def staged(lo, hi):
        keys = [0.05, 0.0]
        mask = torch.zeros((2, hi - lo, 1024), dtype=torch.uint8)
        for i, k in enumerate(keys):
            for b in range(lo, hi):
                mask[i, b - lo, peaks_ref.find_N_peaks(yc[b].numpy(), r=26, min_v=k, N=None)[0]] = 1
        return {"event": Done(), "bon": yb[lo:hi].contiguous(), "cor": yc[lo:hi].contiguous(), "mask": mask, "keys": keys, "B": hi - lo}

    for lo, hi in ((0, 33), (33, 40)):
        st = staged(lo, hi)
        inline = inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 0, native=False)
        pooled = inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 4, native=False)
        shared = inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 4, shared_signals=True, native=False)
        native = inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 3)          # the default: hn_layout_fit_batch
        assert len(inline) == len(pooled) == len(shared) == len(native) == hi - lo
        for a, b, c, d in zip(inline, pooled, shared, native):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0]) and a[1:] == b[1:] == c[1:]
            assert np.array_equal(a[0], d[0]) and d[0].dtype == np.float32 and a[1:] == d[1:]
    # and they are the layouts the single-panorama entry point produces from the same signals
    one = inference(sr.SignalNet(), image_of(g, js[0]), "cpu", peaks_fn=peaks_ref.find_N_peaks)
    st = staged(0, 1)
    assert np.array_equal(inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 0, native=False)[0][0], one[0])
    assert np.array_equal(inf_mod._finish_host(st, 512, 1024, False, False, None, 0.05, 0)[0][0], one[0])


def _signals(g, meta, kinds=None):
    js = [m["case"] for m in meta if kinds is None or m["kind"] in kinds]
    xs = torch.cat([image_of(g, j) for j in js], 0)
    with torch.no_grad():
        yb, yc = sr.SignalNet()(xs.float())
    return js, yb.numpy(), torch.sigmoid(yc)[:, 0].contiguous().numpy()


def _peak_masks(yc, keys, r=26):
    mask = np.zeros((len(keys),) + yc.shape, np.uint8)
    for i, k in enumerate(keys):
        for b in range(yc.shape[0]):
            mask[i, b, peaks_ref.find_N_peaks(yc[b], r=r, min_v=k, N=None)[0]] = 1
    return mask


def _per_panorama(yb, yc, mask_main, mask_zero, cuboid):
    from horizonnet_amd.inference import _MaskPeaks
    out = []
    for b in range(yb.shape[0]):
        keys = {(0.0 if cuboid else 0.05): mask_main[b], 0.0: mask_zero[b] if not cuboid else mask_main[b]}
        try:
            out.append(layout_from_signals(yb[b].copy(), yc[b].copy(), 512, 1024, cuboid, False, None, 0.05, _MaskPeaks(keys)))
        except AssertionError:
            out.append(None)
    return out


@pytest.mark.parametrize("cuboid", [False, True])
def test_native_batched_fit_is_bit_identical_to_the_reference_pinned_path(gold, cuboid):
    """hn_layout_fit_batch + the batched numpy stages (inference.layouts_from_signals_batch) against the per-panorama
    restatement, which test_inference_end_to_end_bit_exact pins to the UNMODIFIED reference: all 102 fixture rooms -- the
    three branches of the general-layout state machine, the self-intersection -> cuboid fallbacks, the 'hard' rooms with
    tied corner probabilities -- as ONE batch, as ragged sub-batches, on 1 and 5 threads: same cor_id bits, same z1."""
    from horizonnet_amd.inference import layouts_from_signals_batch
    g, meta = gold
    js, yb, yc = _signals(g, meta)
    keys = [0.0] if cuboid else [0.05, 0.0]
    masks = _peak_masks(yc, keys)
    want = _per_panorama(yb, yc, masks[0], masks[-1], cuboid)
    assert all(w is not None for w in want)
    for lo, hi, threads in ((0, len(js), 5), (0, 1, 1), (1, 34, 1), (34, len(js), 3)):
        got = layouts_from_signals_batch(yb[lo:hi], yc[lo:hi], masks[0][lo:hi], masks[-1][lo:hi], 512, 1024, cuboid, threads)
        assert len(got) == hi - lo
        for a, b in zip(got, want[lo:hi]):
            assert a[0].dtype == np.float32 and np.array_equal(a[0], b[0]) and a[1] == b[1] == 50
            assert a[2] == b[2] and type(a[2]) is type(b[2])


def test_native_batched_fit_on_perturbed_signals_and_failures(gold):
    """Decisions away from the fixtures: boundary noise, dropped and spurious corner peaks (300 perturbed rooms).  Wherever
    the per-panorama path produces a layout the batched path produces the same bits; where the reference's own assertions
    fire (too few peaks, degenerate votes) the batched path raises too, naming the panorama."""
    from horizonnet_amd.inference import layouts_from_signals_batch
    g, meta = gold
    js, yb, yc = _signals(g, meta)
    rng = np.random.RandomState(7)
    n_ok = n_bad = n_fallback_like = 0
    for rep in range(3):
        yb2 = yb + rng.normal(0, 0.01 * (rep + 1), yb.shape).astype(np.float32)
        masks = _peak_masks(yc, [0.05, 0.0])
        for b in range(len(js)):                                           # drop / add peaks
            pk = np.nonzero(masks[0, b])[0]
            if len(pk) > 4 and rng.rand() < 0.5:
                masks[0, b, rng.choice(pk)] = 0
            if rng.rand() < 0.3:
                masks[0, b, rng.randint(1024)] = 1
        want = _per_panorama(yb2, yc, masks[0], masks[1], False)
        good = [b for b in range(len(js)) if want[b] is not None]
        bad = [b for b in range(len(js)) if want[b] is None]
        got = layouts_from_signals_batch(yb2[good], yc[good], masks[0][good], masks[1][good], 512, 1024, False, 4)
        for a, b in zip(got, [want[b] for b in good]):
            assert np.array_equal(a[0], b[0]) and a[2] == b[2]
        n_ok += len(good)
        for b in bad:
            with pytest.raises(AssertionError):
                layouts_from_signals_batch(yb2[b:b + 1], yc[b:b + 1], masks[0][b:b + 1], masks[1][b:b + 1], 512, 1024, False, 1)
            n_bad += 1
    print("perturbed rooms: %d fitted identically, %d rejected by both paths" % (n_ok, n_bad))
    assert n_ok >= 250


def test_native_batched_fit_more_walls_than_one_call_holds(gold):
    """a saturated corner signal (plateau columns are all peaks) gives the general layout more walls than the 64 output rows
    of hn_layout_fit_batch.  The reference has no cap: those panoramas must come back as the per-panorama path returns them (a large
    polygon, or the cuboid fallback when that polygon is invalid), inside a batch whose other panoramas keep their native results --
    not as an AssertionError for the whole batch."""
    from horizonnet_amd.inference import layouts_from_signals_batch
    g, meta = gold
    js, yb, yc = _signals(g, meta, kinds=None)
    yb, yc = yb[:6].copy(), yc[:6].copy()
    masks = _peak_masks(yc, [0.05, 0.0])
    rng = np.random.RandomState(3)
    crowded = (1, 4)
    for b, npk in zip(crowded, (90, 140)):
        cols = np.sort(rng.choice(1024, npk, replace=False))
        masks[0, b] = 0
        masks[0, b, cols] = 1
        yc[b, cols] = 1.0                                  # plateau: equal maxima
    want = _per_panorama(yb, yc, masks[0], masks[1], False)
    assert all(w is not None for w in want)
    assert max(len(want[b][0]) for b in crowded) // 2 > 0
    got = layouts_from_signals_batch(yb, yc, masks[0], masks[1], 512, 1024, False, 3)
    assert len(got) == 6
    for a, b in zip(got, want):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
    print("corners of the crowded panoramas:", [len(want[b][0]) // 2 for b in crowded])
