"""Host half of the training-data pipeline (labels.py, dataset.py draw order) and the layout metrics (evaluation.py)
against fixtures from the unmodified reference (oracle/gen_golden.py: gen_dataset).  No GPU."""
import json
import os

import numpy as np
import pytest
import torch

from horizonnet_amd import dataset as ds
from horizonnet_amd import evaluation as ev
from horizonnet_amd import labels

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.join(GOLD, "synth_ds")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "dataset.npz")), json.load(open(os.path.join(GOLD, "dataset.json")))


def test_labels_bit_exact_through_all_augmentations(gold):
    """Seeded like the reference run: same random draws, same bon / y_cor / corner list, bit for bit."""
    g, meta = gold
    names = sorted(os.listdir(os.path.join(ROOT, "label_cor")))
    for m in meta:
        cfg = dict(flip=False, rotate=False, gamma=False, stretch=False)
        cfg.update(m["cfg"])
        cor, occ = ds.read_label(os.path.join(ROOT, "label_cor", names[m["index"]]))
        np.random.seed(m["seed"])
        aug = ds.draw_augmentation(cor, 1024, cfg["flip"], cfg["rotate"], cfg["gamma"], cfg["stretch"])
        bon, y_cor, cor2 = ds.make_labels(cor, occ, aug, 512, 1024)
        k = m["key"]
        assert np.array_equal(torch.FloatTensor(bon.copy()).numpy(), g["bon_" + k]), k
        assert np.array_equal(torch.FloatTensor(y_cor.copy()).numpy(), g["ycor_" + k]), k
        assert cor2.dtype == g["cor_" + k].dtype and np.array_equal(cor2, g["cor_" + k]), k


def test_connect_points_degenerate_and_wrap():
    same = labels.pano_connect_points(np.float32([10, 100]), np.float32([10, 300]))
    assert same.dtype == np.float32 and same.shape == (2, 2)
    pts = labels.pano_connect_points(np.float32([1000.5, 180]), np.float32([20.25, 190]), z=-50)   # crosses the border
    assert pts[0, 0] == 1001 and pts[-1, 0] == 20 and len(pts) == 44
    assert np.all((pts[:, 1] > 150) & (pts[:, 1] < 200))


def test_corner_probability_is_periodic():
    y = labels.corner_probability(np.float32([2.0, 700.5]))[0]
    assert y.shape == (1024,) and y[2] == 1.0
    assert y[1023] == 0.96 ** 3 and y[0] == 0.96 ** 2


def test_layout_depth_matches_reference(gold):
    g, _ = gold
    names = sorted(os.listdir(os.path.join(ROOT, "label_cor")))
    checked = 0
    for i, n in enumerate(names):
        if "depth_%d" % i not in g:
            continue
        cor = np.loadtxt(os.path.join(ROOT, "label_cor", n)).astype(np.float32)
        cor = np.roll(cor, -2 * np.argmin(cor[::2, 0]), 0)
        assert np.array_equal(ev.layout_2_depth(cor, 512, 1024)[::8, ::8], g["depth_%d" % i])
        checked += 1
    assert checked >= 2


def test_iou3d_matches_qhull_reference_on_cuboids(gold):
    """eval_general's polygon-area x height 3D IoU (here without shapely) == eval_cuboid's half-space / convex-hull
    3D IoU on box rooms, where both definitions coincide."""
    g, _ = gold
    for k, want in enumerate(g["pair_iou3d"]):
        m = ev.layout_metrics(g["pair_a_%d" % k], g["pair_b_%d" % k])
        assert abs(100 * m["iou3d"] - want) < 1e-6 * want, (k, 100 * m["iou3d"], want)
        assert 0 < m["iou2d"] <= 1 and m["n_corners"] == 4


def test_polygon_intersection_area():
    sq = [(0, 0), (2, 0), (2, 2), (0, 2)]
    ell = [(0, 0), (4, 0), (4, 1), (1, 1), (1, 4), (0, 4)]
    assert ev.polygon_intersection_area(sq, [(1, 1), (3, 1), (3, 3), (1, 3)]) == 1.0
    assert ev.polygon_intersection_area(sq, sq[::-1]) == 4.0                      # orientation does not matter
    assert ev.polygon_intersection_area(sq, [(5, 5), (6, 5), (6, 6)]) == 0.0
    assert abs(ev.polygon_intersection_area(ell, [(0.5, 0.5), (3, 0.5), (3, 3), (0.5, 3)]) - 2.25) < 1e-12
    r = 2 ** 0.5
    diamond = [(1 + r, 1), (1, 1 + r), (1 - r, 1), (1, 1 - r)]
    assert abs(ev.polygon_intersection_area(sq, diamond) - (4 - 2 * (2 - r) ** 2)) < 1e-12
    rng = np.random.RandomState(0)                                                   # Monte-Carlo cross-check
    a = np.array([(0, 0), (5, 0), (5, 2), (2, 2), (2, 5), (0, 5)], float)
    b = np.array([(1, -1), (6, 1), (3, 6), (1.5, 3)], float)
    pts = rng.uniform(-1, 6, (200000, 2))

    def inside(poly, p):
        x, y = p[:, 0], p[:, 1]
        c = np.zeros(len(p), bool)
        for (x0, y0), (x1, y1) in zip(poly, np.roll(poly, -1, 0)):
            c ^= ((y0 > y) != (y1 > y)) & (x < (x1 - x0) * (y - y0) / (y1 - y0 + 1e-300) + x0)
        return c
    mc = (inside(a, pts) & inside(b, pts)).mean() * 49
    assert abs(ev.polygon_intersection_area(a, b) - mc) < 0.1


def test_test_general_bookkeeping(gold):
    g, _ = gold
    losses = {k: {"2DIoU": [], "3DIoU": [], "rmse": [], "delta_1": []} for k in ("4", "6", "8", "10+", "odd", "overall")}
    a = g["pair_a_0"]
    ev.test_general(a, a, 1024, 512, losses)
    assert abs(losses["4"]["3DIoU"][0] - 1) < 1e-12 and abs(losses["overall"]["2DIoU"][0] - 1) < 1e-12
    assert losses["4"]["rmse"] == [0.0] and losses["4"]["delta_1"] == [1.0]


def test_cuboid_metrics_match_reference(gold):
    """eval_cuboid.test (corner error, pixel surface error, Qhull 3D IoU) on 24 predicted / ground-truth box pairs."""
    g, _ = gold
    for k, (ce, pe, iou) in enumerate(g["cub_metrics"]):
        losses = {"CE": [], "PE": [], "3DIoU": []}
        ev.test_cuboid(g["cub_dt_%d" % k], 50, float(g["cub_z1_%d" % k]), g["pair_a_%d" % k].astype(np.float32), 1024, 512, losses)
        assert losses["CE"][0] == ce and losses["PE"][0] == pe, k
        assert abs(losses["3DIoU"][0] - iou) < 1e-9 * iou, k
