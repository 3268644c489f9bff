"""Training-label rasterisation -- host restatement of reference ``dataset.py:84,108-120,137-169``
(``cor_2_1d``, ``sort_xy_filter_unique``, the 1-D wall-wall probability) and ``misc/panostretch.py:51-78``
(``pano_connect_points``).

Per-image scalar work on <= ~30 corners and 1024 columns; host numpy as in the reference, same arithmetic and the same
numpy dtype flow (float32 corner scalars, float64 column vectors), pinned by ``tests/golden/dataset.npz`` (written by the reference's own ``__getitem__``)."""
import numpy as np


def _lon(x, w):
    return ((x + 0.5) / w - 0.5) * 2 * np.pi


def _lat(y, h):
    return ((y + 0.5) / h - 0.5) * np.pi


def edge_scalars(p1, p2, z=-50, w=1024, h=512):
    """The per-corner half of ``pano_connect_points`` (panostretch.py:51-70), in the reference's float32 scalar flow: the edge's first
    end point (x1, y1) and direction (dx, dy) on the plane z, and the integer columns it covers (first, last; the short way round)."""
    u1, v1, u2, v2 = _lon(p1[0], w), _lat(p1[1], h), _lon(p2[0], w), _lat(p2[1], h)
    r1, r2 = z / np.tan(v1), z / np.tan(v2)
    x1, y1 = r1 * np.cos(u1), r1 * np.sin(u1)
    x2, y2 = r2 * np.cos(u2), r2 * np.sin(u2)
    lo, hi = min(p1[0], p2[0]), max(p1[0], p2[0])
    if abs(p1[0] - p2[0]) < w / 2:
        first, last = np.ceil(lo), np.floor(hi)
    else:                                   # the short way round crosses the image border
        first, last = np.ceil(hi), np.floor(lo + w)
    return x1, y1, x2 - x1, y2 - y1, first, last


def pano_connect_points(p1, p2, z=-50, w=1024, h=512):
    """Image-space trace, one point per integer column, of the straight wall edge joining the corners p1 and p2 on
    the horizontal plane z (ceiling z<0, floor z>0) -- the great-arc between them."""
    if p1[0] == p2[0]:
        return np.array([p1, p2], np.float32)
    x1, y1, dx, dy, first, last = edge_scalars(p1, p2, z, w, h)
    cols = (np.arange(first, last + 1) % w).astype(np.float64)
    t = np.tan(_lon(cols, w))
    s = (t * x1 - y1) / (dy - t * dx)       # where the column's ray meets the edge
    rng = np.sqrt((x1 + s * dx) ** 2 + (y1 + s * dy) ** 2)
    rows = (np.arctan2(z, rng) / np.pi + 0.5) * h - 0.5
    return np.stack([cols, rows], axis=-1)


def _one_row_per_column(xs, ys, upper_first):
    xs, ys = np.array(xs), np.array(ys)
    order = np.argsort(xs + ys / ys.max() * (int(upper_first) * 2 - 1))
    xs, ys = xs[order], ys[order]
    _, first = np.unique(xs, return_index=True)
    xs, ys = xs[first], ys[first]
    assert np.all(np.diff(xs) > 0)
    return xs, ys


def cor_2_1d(cor, H, W):
    """Corner list (ceiling / floor rows alternating) -> bon [2, W] latitude (radians) of the ceiling-wall and
    floor-wall boundary at every column."""
    n = len(cor)
    traces = []
    for first, z, upper_first in ((0, -50, True), (1, 50, False)):
        xs, ys = [], []
        for i in range(n // 2):
            pts = pano_connect_points(cor[i * 2 + first], cor[(i * 2 + 2 + first) % n], z=z, w=W, h=H)
            xs.extend(pts[:, 0])
            ys.extend(pts[:, 1])
        traces.append(_one_row_per_column(xs, ys, upper_first))
    bon = np.zeros((2, W))
    bon[0] = np.interp(np.arange(W), traces[0][0], traces[0][1], period=W)
    bon[1] = np.interp(np.arange(W), traces[1][0], traces[1][1], period=W)
    return ((bon + 0.5) / H - 0.5) * np.pi


def corner_probability(corx, W=1024, p_base=0.96):
    """y_cor [1, W]: p_base ** (circular column distance to the nearest visible wall-wall corner)."""
    corx = np.asarray(corx, np.float64).reshape(-1, 1)
    cols = np.arange(W).reshape(1, -1)
    d = np.minimum(np.minimum(np.abs(corx - cols), np.abs(corx - (cols + W))), np.abs(corx - (cols - W)))
    return (p_base ** d.min(0)).reshape(1, -1)


LABEL_MAX_EDGES = 64       # wall edges per boundary in a device record (general layouts have <= ~16)
LABEL_MAX_CORNERS = 128    # visible wall-wall corners
LABEL_REC_FLOATS = 8 + 2 * LABEL_MAX_EDGES * 8 + LABEL_MAX_CORNERS


def device_label_record(cor, occlusion, flip, roll, H, W, out=None):
    """One panorama's record for ``hn_labels_rasterise`` (csrc/labels.hip): the per-corner scalars of every wall edge (``edge_scalars``,
    the reference's float32 flow) + the visible corner columns after flip / roll (dataset.py:88-97,113-118).  `cor`: the corners AFTER
    the stretch, BEFORE flip / roll (cor_2_1d runs there, dataset.py:84).  Returns (record float32 [LABEL_REC_FLOATS], closed): closed
    False = some column is covered by no edge of a boundary -> rasterise this panorama with ``cor_2_1d`` instead."""
    rec = np.zeros(LABEL_REC_FLOATS, np.float32) if out is None else out
    rec[:] = 0
    n = len(cor)
    closed = n // 2 <= LABEL_MAX_EDGES
    for bd, z in ((0, -50), (1, 50)):
        covered = np.zeros(W, bool)
        k = 0
        for i in range(n // 2):
            if k >= LABEL_MAX_EDGES:
                break
            p1, p2 = cor[i * 2 + bd], cor[(i * 2 + 2 + bd) % n]
            e = rec[8 + (bd * LABEL_MAX_EDGES + k) * 8:8 + (bd * LABEL_MAX_EDGES + k + 1) * 8]
            if p1[0] == p2[0]:
                e[0], e[1], e[2], e[3] = 1, p1[0], p1[1], p2[1]
                if float(p1[0]) == int(p1[0]):
                    covered[int(p1[0]) % W] = True
            else:
                x1, y1, dx, dy, first, last = edge_scalars(p1, p2, z, W, H)
                cnt = max(0, int(last) - int(first) + 1)
                e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7] = 0, x1, y1, dx, dy, first, cnt, z
                covered[(int(first) + np.arange(cnt)) % W] = True
            k += 1
        rec[bd] = k
        closed = closed and bool(covered.all())
    cx = np.asarray(cor[:, 0], np.float32)
    if flip:
        cx = W - 1 - cx                                   # dataset.py:89 (float32 array arithmetic, as flip_labels)
    if roll is not None:
        cx = (cx + roll) % W                              # dataset.py:96
    cx = cx[~np.asarray(occlusion, bool)]
    nc = min(len(cx), LABEL_MAX_CORNERS)
    closed = closed and len(cx) <= LABEL_MAX_CORNERS
    rec[2], rec[3], rec[4] = nc, 1 if flip else 0, 0 if roll is None else int(roll) % W
    rec[8 + 2 * LABEL_MAX_EDGES * 8:8 + 2 * LABEL_MAX_EDGES * 8 + nc] = cx[:nc]
    return rec, closed


def flip_labels(bon, cor, W):
    bon = np.flip(bon, axis=1)
    cor = cor.copy()
    cor[:, 0] = W - 1 - cor[:, 0]
    return bon, cor


def roll_labels(bon, cor, dx, W):
    bon = np.roll(bon, dx, axis=1)
    cor = cor.copy()
    cor[:, 0] = (cor[:, 0] + dx) % W
    return bon, cor


def find_occlusion(coor):
    """For each ceiling corner [N,2]: is the straight line from the camera to it crossed (or touched) by the layout
    outline through the other corners?  (dataset.py:172-186, there with shapely's ``LineString.intersects``.)"""
    coor = np.asarray(coor)
    u, v = _lon(coor[:, 0], 1024), _lat(coor[:, 1], 512)
    r = -50 / np.tan(v)
    px = [float(t) for t in r * np.cos(u)]
    py = [float(t) for t in r * np.sin(u)]
    n = len(px)

    def side(ax, ay, bx, by, cx, cy):
        s = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
        return (s > 0) - (s < 0)

    def in_box(ax, ay, bx, by, cx, cy):
        return min(ax, bx) <= cx <= max(ax, bx) and min(ay, by) <= cy <= max(ay, by)

    def meet(ax, ay, bx, by, cx, cy, dx, dy):
        d1, d2 = side(ax, ay, bx, by, cx, cy), side(ax, ay, bx, by, dx, dy)
        d3, d4 = side(cx, cy, dx, dy, ax, ay), side(cx, cy, dx, dy, bx, by)
        if d1 != d2 and d3 != d4:
            return True
        return ((d1 == 0 and in_box(ax, ay, bx, by, cx, cy)) or (d2 == 0 and in_box(ax, ay, bx, by, dx, dy)) or
                (d3 == 0 and in_box(cx, cy, dx, dy, ax, ay)) or (d4 == 0 and in_box(cx, cy, dx, dy, bx, by)))

    out = []
    for i in range(n):
        chain = list(range(i + 1, n)) + list(range(0, i))
        out.append(any(meet(0.0, 0.0, px[i], py[i], px[a], py[a], px[b], py[b]) for a, b in zip(chain[:-1], chain[1:])))
    return np.array(out)
