"""Manhattan layout fitting from the 1-D signals -- host-side restatement of reference ``misc/post_proc.py``
(the functions ``inference.py:96-129`` calls: ``np_refine_by_fix_z`` ``:109-123``, ``infer_coory`` ``:126-131``,
``gen_ww`` ``:337-359`` with ``gen_ww_cuboid`` ``:205-240`` / ``gen_ww_general`` ``:243-334``, ``vote`` ``:75-98``).

Scalar, data-dependent geometry on 1024-long float64 vectors (tens of microseconds to a millisecond per image):
it stays on the CPU, as in the reference: numpy, plus the decision loop of ``vote`` in the native library
(``hn_vote_scan``, a host function).  Numerics follow the reference operation by operation so that the
outputs are identical (pinned by ``tests/golden/postproc.npz``, generated from the unmodified reference); the wall
voting is a linear two-pointer scan instead of the reference's dense N x N distance matrix.

Conventions: ``u`` longitude of an image column, ``v`` latitude of an image row (positive up), floor-plan coordinates
in a 1024 x 512 "ceiling view" image centred on the camera, ceiling plane ``z`` units above the camera.
"""
import ctypes

import numpy as np

from . import _lib

PI = float(np.pi)
_BEST3 = ctypes.c_int32 * 3


# ---- coordinate transforms (post_proc.py:22-66) ---------------------------------------------------------------------
def col_to_u(coorx, coorW=1024):
    return ((coorx + 0.5) / coorW - 0.5) * 2 * PI


def row_to_v(coory, coorH=512):
    return -((coory + 0.5) / coorH - 0.5) * PI


def pano_to_plan(coor, z=50, coorW=1024, coorH=512, floorW=1024, floorH=512):
    """[N,2] (col,row) of points on the plane z above the camera -> [N,2] floor-plan (x, y)."""
    coor = np.array(coor)
    u = col_to_u(coor[:, 0], coorW)
    v = row_to_v(coor[:, 1], coorH)
    c = z / np.tan(v)
    x = c * np.sin(u) + floorW / 2 - 0.5
    y = -c * np.cos(u) + floorH / 2 - 0.5
    return np.hstack([x[:, None], y[:, None]])


def plan_to_pano(xy, z=50, coorW=1024, coorH=512, floorW=1024, floorH=512):
    x = xy[:, 0] - floorW / 2 + 0.5
    y = xy[:, 1] - floorH / 2 + 0.5
    u = np.arctan2(x, -y)
    v = np.arctan(z / np.sqrt(x ** 2 + y ** 2))
    coorx = (u / (2 * PI) + 0.5) * coorW - 0.5
    coory = (-v / PI + 0.5) * coorH - 0.5
    return np.hstack([coorx[:, None], coory[:, None]])


def y_where_ray_meets_x(x, u, floorW=1024, floorH=512):
    """On the floor-plan ray of longitude u: the y at which it reaches the vertical line x."""
    c = (x - floorW / 2 + 0.5) / np.sin(u)
    return -c * np.cos(u) + floorH / 2 - 0.5


def x_where_ray_meets_y(y, u, floorW=1024, floorH=512):
    c = -(y - floorH / 2 + 0.5) / np.cos(u)
    return c * np.sin(u) + floorW / 2 - 0.5


# ---- robust statistics (post_proc.py:69-98) -------------------------------------------------------------------------
def interquartile_mean(vec, p1=25, p2=75):
    lo = np.percentile(vec, p1)
    hi = np.percentile(vec, p2)
    return vec[(lo <= vec) & (vec <= hi)].mean()


def _vote_scan_py(vec, tol):
    """The decision loop of `vote` for inputs that are not float64 (numpy scalar arithmetic in the array's own type)."""
    L = len(vec)
    best_span, best_i, best_j = -1, -1, -1
    j = 0
    for i in range(L):
        if j < i:
            j = i
        # largest j with (v_j - v_i) + 1e-9 <= tol   (j == i has distance exactly 0)
        while j + 1 < L and not ((vec[j + 1] - vec[i]) + 1e-9 > tol):
            j += 1
        jj = j
        if jj == i and 0.0 > tol:
            continue
        span = jj - i + 1
        if not (span < L * 0.4) and span > best_span:
            best_span, best_i, best_j = span, i, jj
    return best_span, best_i, best_j


def vote(vec, tol):
    """Longest run of the sorted samples whose spread stays within tol (and covers >= 40 % of them): its mean, the
    fraction of samples it covers, and the mean absolute deviation of all samples from it.

    Same decision as the reference's N x N formulation (span matrix n[i][j] = j - i + 1, distance |v_i - v_j| + 1e-9,
    first maximum in row-major order), found with one pointer per end because the samples are sorted."""
    vec = np.sort(vec)
    L = len(vec)
    if vec.dtype == np.float64 and vec.flags.c_contiguous:
        # the decision loop in the native library (hn_vote_scan: same comparisons on the same doubles); it was 2/3 of the host
        # time of the Manhattan fit as an interpreted loop
        best = _BEST3()
        _lib.check(_lib.load().hn_vote_scan(vec.ctypes.data, L, float(tol), ctypes.addressof(best)), "hn_vote_scan")
        best_span, best_i, best_j = int(best[0]), int(best[1]), int(best[2])
    else:
        best_span, best_i, best_j = _vote_scan_py(vec, tol)
    if best_span < 0 or L < tol:
        best_fit = np.median(vec)
        p_score = 0
    else:
        assert best_j > best_i
        best_fit = vec[best_i:best_j + 1].mean()
        p_score = (best_j - best_i + 1) / L
    l1_score = np.abs(vec - best_fit).mean()
    return best_fit, p_score, l1_score


# ---- ceiling / floor consistency (post_proc.py:101-131) -------------------------------------------------------------
def refine_by_fix_z(coory0, coory1, z0=50, coorH=512):
    """Rows coory0 lie on the plane z0; find the single height z1 the rows coory1 agree on and re-project them."""
    v0 = row_to_v(coory0, coorH)
    v1 = row_to_v(coory1, coorH)
    c0 = z0 / np.tan(v0)
    z1 = c0 * np.tan(v1)
    z1_mean = interquartile_mean(z1)
    v1_refine = np.arctan2(z1_mean, c0)
    return (-v1_refine / PI + 0.5) * coorH - 0.5, z1_mean


def infer_coory(coory0, h, z0=50, coorH=512):
    v0 = row_to_v(coory0, coorH)
    c0 = z0 / np.tan(v0)
    v1 = np.arctan2(z0 + h, c0)
    return (-v1 / PI + 0.5) * coorH - 0.5


# ---- wall fitting (post_proc.py:134-359) ----------------------------------------------------------------------------
def wall_groups(peak_cols, coorW):
    """Column -> index of the wall segment it belongs to; the segment that wraps around the image border is 0."""
    g = np.zeros(coorW)
    g[np.round(peak_cols).astype(int)] = 1
    g = np.cumsum(g).astype(int)
    g[g == g[-1]] = 0
    return g


def _wall(axis, val, score, action="ori", gpid=-1, u0=-1, u1=-1, tbd=False):
    # dict keys are the reference's (inference.py:116-119 reads 'type' and 'val')
    return {"type": axis, "val": val, "score": score, "action": action, "gpid": gpid, "u0": u0, "u1": u1, "tbd": tbd}


def _vote_axis(xy, sel, tol):
    """Is the segment a constant-x (0) or constant-y (1) wall?  -> (axis, value, score)"""
    vx, sx, lx = vote(xy[sel, 0], tol)
    vy, sy, ly = vote(xy[sel, 1], tol)
    if (sx, -lx) > (sy, -ly):
        return 0, vx, sx
    return 1, vy, sy


def walls_cuboid(xy, gpid, tol):
    assert len(np.unique(gpid)) == 4
    walls = [_wall(*_vote_axis(xy, gpid == j, tol)) for j in range(4)]
    # a cuboid alternates x / y walls: keep the alternation the votes support most
    balance = [0, 0]
    for j, w in enumerate(walls):
        balance[j % 2] += w["score"] if w["type"] == 0 else -w["score"]
    first = 0 if balance[0] > balance[1] else 1
    for j, w in enumerate(walls):
        w["type"] = (first + j) % 2
    for w in walls:     # the reference's cuboid records carry exactly these keys
        for k in ("action", "gpid", "u0", "u1", "tbd"):
            del w[k]
    return walls


def walls_general(peak_cols, xy, gpid, tol):
    n = len(peak_cols)
    assert n == len(np.unique(gpid))
    walls = []
    for j in range(n):
        axis, val, score = _vote_axis(xy, gpid == j, tol)
        walls.append(_wall(axis, val, score, "ori", j, col_to_u(peak_cols[(j - 1 + n) % n]), col_to_u(peak_cols[j]), True))

    def corner_wall(src, u_key):
        """Wall perpendicular to `src`, through the point where src meets the corner ray u_key."""
        if src["type"] == 0:
            return 1, y_where_ray_meets_x(src["val"], src[u_key])
        return 0, x_where_ray_meets_y(src["val"], src[u_key])

    while True:
        # settle the undetermined wall with the best score
        cur = -1
        for i, w in enumerate(walls):
            if w["tbd"] and (cur == -1 or w["score"] > walls[cur]["score"]):
                cur = i
        if cur == -1:
            break
        w = walls[cur]
        w["tbd"] = False
        pi, ni = (cur - 1 + len(walls)) % len(walls), (cur + 1) % len(walls)
        prv, nxt = walls[pi], walls[ni]
        open_neighbours = prv["tbd"] + nxt["tbd"]
        if open_neighbours == 2:
            continue
        if open_neighbours == 1:
            clash = (not prv["tbd"] and prv["type"] == w["type"]) or (not nxt["tbd"] and nxt["type"] == w["type"])
            if clash:
                if w["score"] >= -1:
                    w["tbd"] = True              # decide it later, after its other neighbour
                    w["score"] -= 100
                else:                            # second visit: force a perpendicular wall next to the settled neighbour
                    if not prv["tbd"]:
                        at = cur
                        axis, val = corner_wall(prv, "u1")
                    else:
                        at = ni
                        axis, val = corner_wall(nxt, "u0")
                    walls.insert(at, _wall(axis, val, 0, "forced infer"))
            continue
        # both neighbours settled
        if prv["type"] == nxt["type"]:
            if w["type"] == prv["type"]:         # three parallel walls in a row: turn the middle one
                w["type"] = (w["type"] + 1) % 2
                w["action"] = "forced change"
                w["val"] = xy[gpid == w["gpid"], w["type"]].mean()
        else:                                    # neighbours perpendicular to each other: replace by two walls meeting at a corner
            a0, v0 = nxt["type"], corner_wall(prv, "u1")[1]
            a1, v1 = prv["type"], corner_wall(nxt, "u0")[1]
            walls = walls[:cur] + [_wall(a0, v0, 0, "forced infer"), _wall(a1, v1, 0, "forced infer")] + walls[cur + 1:]
    return walls


def gen_ww(init_coorx, coory, z=50, coorW=1024, coorH=512, floorW=1024, floorH=512, tol=3, force_cuboid=True):
    """Wall-wall corners of the layout: (cor [N,2] pano (col,row) of the ceiling corners, walls list)."""
    gpid = wall_groups(init_coorx, coorW)
    coor = np.hstack([np.arange(coorW)[:, None], coory[:, None]])
    xy = pano_to_plan(coor, z, coorW, coorH, floorW, floorH)
    walls = walls_cuboid(xy, gpid, tol) if force_cuboid else walls_general(init_coorx, xy, gpid, tol)
    pts = []
    for j, w in enumerate(walls):
        nx = walls[(j + 1) % len(walls)]
        pts.append((nx["val"], w["val"]) if w["type"] == 1 else (w["val"], nx["val"]))
    cor = plan_to_pano(np.array(pts), z, coorW, coorH, floorW, floorH)
    cor = np.roll(cor, -2 * cor[::2, 0].argmin(), axis=0)
    return cor, walls


def polygon_is_simple(xy):
    """What ``shapely.geometry.Polygon(xy).is_valid`` answers at inference.py:120 for a single ring: non-zero area and
    no point shared by two non-adjacent edges (crossing or touching).  All edge pairs at once, O(n^2) on n <= ~20."""
    p = np.asarray(xy, np.float64).reshape(-1, 2)
    # repeated consecutive points (an explicitly closed ring repeats its first point) do not make a ring invalid (OGC SFS / the GEOS
    # validity rules shapely documents): drop them before looking at edge pairs, a zero-length edge would separate true neighbours
    if len(p) > 1:
        p = p[(p != np.roll(p, 1, axis=0)).any(axis=1)]
    n = len(p)
    if n < 3:
        return False
    q = np.roll(p, -1, axis=0)
    twice_area = np.dot(p[:, 0], q[:, 1]) - np.dot(p[:, 1], q[:, 0])
    if twice_area == 0:
        return False
    e = q - p

    def turn(c):          # [i, j] = sign of cross(edge_i, c_j - p_i)
        w = c[None, :, :] - p[:, None, :]
        return np.sign(e[:, None, 0] * w[..., 1] - e[:, None, 1] * w[..., 0])

    def boxed(c):         # [i, j] = c_j inside the bounding box of edge i
        lo, hi = np.minimum(p, q)[:, None, :], np.maximum(p, q)[:, None, :]
        return ((lo <= c[None]) & (c[None] <= hi)).all(-1)

    sp, sq = turn(p), turn(q)                    # edge i versus start / end of edge j
    cross = (sp != sq) & (sp.T != sq.T)
    touch = ((sp == 0) & boxed(p)) | ((sq == 0) & boxed(q))
    meet = cross | touch | touch.T
    idx = np.arange(n)
    gap = (idx[None, :] - idx[:, None]) % n
    far = (gap >= 2) & (gap <= n - 2)
    return not bool((meet & far).any())


# ---- the reference's names (misc/post_proc.py), so `from horizonnet_amd import postproc as post_proc` is a drop-in ----
np_coorx2u, np_coory2v = col_to_u, row_to_v
np_coor2xy, np_xy2coor = pano_to_plan, plan_to_pano
np_x_u_solve_y, np_y_u_solve_x = y_where_ray_meets_x, x_where_ray_meets_y
mean_percentile = interquartile_mean
np_refine_by_fix_z = refine_by_fix_z
get_gpid = wall_groups
