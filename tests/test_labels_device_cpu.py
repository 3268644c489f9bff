"""The device label path's HOST half (labels.device_label_record) and the kernel's selection rule (csrc/labels.hip), checked on the CPU:
a numpy restatement of the kernel -- per column the minimum of the reference's sort key over the edges that cover it -- applied to the
records must reproduce labels.make_labels (= reference dataset.py:84-120, pinned by tests/golden/dataset.npz) BIT FOR BIT: same libm,
same operation order, so any difference would be a difference of algorithm, not of rounding."""
import numpy as np
import pytest

from horizonnet_amd import labels as lb
from horizonnet_amd.dataset import draw_augmentation, make_labels
from horizonnet_amd.panostretch import _stretch_corners
from tools import synth_rooms as sr

H, W = 512, 1024


def kernel_in_numpy(rec, p_base=0.96):
    """csrc/labels.hip, statement by statement, vectorised over the columns."""
    ne, nc, flip, roll = (int(rec[0]), int(rec[1])), int(rec[2]), int(rec[3]), int(rec[4])
    cols = np.arange(W)
    bon = np.zeros((2, W), np.float32)
    missing = False
    for bd in range(2):
        cand = []                                             # (mask over columns, rows)
        extra = []
        for k in range(ne[bd]):
            e = rec[8 + (bd * lb.LABEL_MAX_EDGES + k) * 8:][:8]
            if e[0] != 0:
                xe = np.float64(e[1])             # a vertical edge: an integral x, modulo W (csrc/labels.hip rows_at); a fractional x touches no column
                m = (cols.astype(np.float64) == np.mod(xe, W)) if xe == np.floor(xe) else np.zeros(len(cols), bool)
                cand += [(m, np.full(W, np.float64(e[2]))), (m, np.full(W, np.float64(e[3])))]
                extra += [np.float64(e[2]), np.float64(e[3])]
                continue
            first, cnt = int(e[5]), int(e[6])
            m = ((cols - first) % W) < cnt
            x1, y1, dx, dy, z = (np.float64(v) for v in (e[1], e[2], e[3], e[4], e[7]))
            t = np.tan(((cols + 0.5) / W - 0.5) * 2 * np.pi)
            with np.errstate(all="ignore"):
                s = (t * x1 - y1) / (dy - t * dx)
                rng = np.sqrt((x1 + s * dx) ** 2 + (y1 + s * dy) ** 2)
                cand.append((m, (np.arctan2(z, rng) / np.pi + 0.5) * H - 0.5))
        ymax = max([float(r[m].max()) for m, r in cand if m.any()] + [float(v) for v in extra])
        sgn = 1.0 if bd == 0 else -1.0
        best = np.zeros(W)
        bkey = np.full(W, np.inf)
        found = np.zeros(W, bool)
        for m, r in cand:
            with np.errstate(all="ignore"):
                key = cols + r / ymax * sgn
            take = m & (~found | (key < bkey))
            best[take], bkey[take], found[take] = r[take], key[take], True
        missing |= not found.all()
        lat = ((best + 0.5) / H - 0.5) * np.pi
        j = (W - 1 - cols) if flip else cols
        j = (j + roll) % W
        bon[bd, j] = lat.astype(np.float32)
    cx = rec[8 + 2 * lb.LABEL_MAX_EDGES * 8:][:nc].astype(np.float64).reshape(-1, 1)
    c = cols.reshape(1, -1)
    d = np.minimum(np.minimum(np.abs(cx - c), np.abs(cx - (c + W))), np.abs(cx - (c - W))).min(0)
    return bon, (p_base ** d).astype(np.float32).reshape(1, -1), missing


def rooms(n):
    """Star-shaped Manhattan rooms (tools/synth_rooms.py) and, every third one, an L-shaped room seen from inside one arm: the far
    arm's corners are hidden, its wall traces overlap the near walls' -- the case the one-row-per-column rule exists for."""
    out = []
    for i in range(n):
        r = np.random.RandomState(100 + i)
        if i % 3 == 2:
            a, b, c, d = r.uniform(3.5, 5.0), r.uniform(0.8, 1.3), r.uniform(1.5, 2.5), r.uniform(3.0, 5.0)
            poly = np.array([(-a, -b), (1.0, -b), (1.0, b), (-c, b), (-c, d), (-a, d)]) + r.uniform(-0.2, 0.2, 2)
            if r.randint(2):
                poly = poly[::-1] * np.array([1.0, -1.0])     # the mirrored room, same orientation
        else:
            poly = sr.manhattan_polygon(r, [4, 6, 8, 10, 12][i % 5])
        cor = np.asarray(sr.room_corners(poly), np.float32)
        out.append((cor, lb.find_occlusion(cor[::2].copy()).repeat(2)))
    return out


def test_records_reproduce_the_host_labels_bit_for_bit():
    rng = np.random.RandomState(7)
    n_occluded = 0
    for cor, occ in rooms(120):
        a = draw_augmentation(cor, W, True, True, True, True, 2.0, rng)
        want_bon, want_cor, _ = make_labels(cor, occ, a, H, W)
        c = cor if (a["kx"] == 1.0 and a["ky"] == 1.0) else _stretch_corners(cor, a["kx"], a["ky"], W, H)
        rec, closed = lb.device_label_record(c, occ, a["flip"], a["roll"] if a["rotate"] else None, H, W)
        assert closed
        bon, y_cor, missing = kernel_in_numpy(rec)
        assert not missing
        n_occluded += int(occ.any())
        assert np.array_equal(bon, want_bon.astype(np.float32))
        assert np.array_equal(y_cor, want_cor.astype(np.float32))
    assert n_occluded > 10                                    # rooms with hidden corners (overlapping traces) were among them


def test_records_reproduce_the_reference_fixtures(golden_dir):
    """tests/golden/dataset.npz: bon / y_cor written by the UNMODIFIED reference dataset.__getitem__ (oracle/gen_golden.py) on the synthetic
    dataset under every augmentation switch; the record + the kernel's rule reproduce them bit for bit."""
    import json
    import os
    from horizonnet_amd import dataset as ds
    g, meta = np.load(os.path.join(golden_dir, "dataset.npz")), json.load(open(os.path.join(golden_dir, "dataset.json")))
    root = os.path.join(golden_dir, "synth_ds")
    names = sorted(os.listdir(os.path.join(root, "label_cor")))
    assert len(meta) >= 8
    for m in meta:
        cfg = dict(flip=False, rotate=False, gamma=False, stretch=False)
        cfg.update(m["cfg"])
        cor, occ = ds.read_label(os.path.join(root, "label_cor", names[m["index"]]))
        np.random.seed(m["seed"])
        a = ds.draw_augmentation(cor, W, cfg["flip"], cfg["rotate"], cfg["gamma"], cfg["stretch"])
        c = cor if (a["kx"] == 1.0 and a["ky"] == 1.0) else _stretch_corners(cor, a["kx"], a["ky"], W, H)
        rec, closed = lb.device_label_record(c, occ, a["flip"], a["roll"] if a["rotate"] else None, H, W)
        bon, y_cor, missing = kernel_in_numpy(rec)
        assert closed and not missing
        assert np.array_equal(bon, g["bon_" + m["key"]]), m["key"]
        assert np.array_equal(y_cor, g["ycor_" + m["key"]]), m["key"]


def test_open_outline_is_sent_to_the_host():
    """Two corners: both 'walls' take the short way round over the same columns, the rest of the panorama has no trace point --
    np.interp would interpolate there; the record says so and DeviceBatcher rasterises that panorama with labels.cor_2_1d."""
    cor = np.array([[100, 200], [100, 300], [400, 210], [400, 310]], np.float32)
    rec, closed = lb.device_label_record(cor, np.zeros(4, bool), 0, None, H, W)
    assert not closed
    _, _, missing = kernel_in_numpy(rec)
    assert missing


def test_degenerate_edge_on_an_integer_column_counts_as_covered():
    cor = np.array([[10, 200], [10, 300], [10, 190], [10, 310], [400, 205], [400, 305], [800, 195], [800, 315]], np.float32)
    rec, closed = lb.device_label_record(cor, np.zeros(8, bool), 0, None, H, W)
    assert rec[8] == 1.0 and closed
    bon, _, missing = kernel_in_numpy(rec)
    assert not missing and np.array_equal(bon, lb.cor_2_1d(cor, H, W).astype(np.float32))
