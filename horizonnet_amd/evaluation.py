"""Layout metrics -- restatement of reference ``eval_general.py:14-122`` (``layout_2_depth``, ``test_general``:
2D IoU, 3D IoU, depth RMSE, delta_1) without shapely.

The reference intersects the two floor-plan polygons with GEOS (shapely, absent offline).  Here the intersection area
of two simple polygons is computed exactly by vertical-slab decomposition: between consecutive "event" abscissae
(vertices of either polygon and crossings between their edges) every edge is a straight, non-crossing segment, so the
length of the common cross-section is linear in x and its integral over the slab is width x value at the midpoint.
Pinned against the Qhull half-space 3D IoU of ``eval_cuboid.py:49-92`` on convex rooms (tests/golden/evalgen.npz)."""
import numpy as np

from . import postproc
from .labels import cor_2_1d


def polygon_area(p):
    p = np.asarray(p, np.float64)
    return 0.5 * abs(float(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1))))


def _edge_crossings_x(a, b):
    """x of every proper crossing between an edge of ring a and an edge of ring b."""
    a0, a1 = a, np.roll(a, -1, axis=0)
    b0, b1 = b, np.roll(b, -1, axis=0)
    r = (a1 - a0)[:, None, :]
    s = (b1 - b0)[None, :, :]
    qp = b0[None, :, :] - a0[:, None, :]
    den = r[..., 0] * s[..., 1] - r[..., 1] * s[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (qp[..., 0] * s[..., 1] - qp[..., 1] * s[..., 0]) / den
        u = (qp[..., 0] * r[..., 1] - qp[..., 1] * r[..., 0]) / den
    hit = (den != 0) & (t > 0) & (t < 1) & (u > 0) & (u < 1)
    return (a0[:, None, 0] + np.where(hit, t, 0.0) * r[..., 0])[hit]


def _cross_section(ring, x):
    """Sorted y where the vertical line at x (not through a vertex) crosses the ring."""
    p0, p1 = ring, np.roll(ring, -1, axis=0)
    lo, hi = np.minimum(p0[:, 0], p1[:, 0]), np.maximum(p0[:, 0], p1[:, 0])
    sel = (lo < x) & (x < hi)
    t = (x - p0[sel, 0]) / (p1[sel, 0] - p0[sel, 0])
    return np.sort(p0[sel, 1] + t * (p1[sel, 1] - p0[sel, 1]))


def polygon_intersection_area(a, b):
    """Area common to two simple polygons (even-odd interior), vertices [N,2] in order (either orientation)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    xs = np.unique(np.concatenate([a[:, 0], b[:, 0], _edge_crossings_x(a, b)]))
    total = 0.0
    for x0, x1 in zip(xs[:-1], xs[1:]):
        xm = 0.5 * (x0 + x1)
        ya, yb = _cross_section(a, xm), _cross_section(b, xm)
        if len(ya) < 2 or len(yb) < 2:
            continue
        overlap = 0.0
        for i in range(0, len(ya) - 1, 2):
            for j in range(0, len(yb) - 1, 2):
                overlap += max(0.0, min(ya[i + 1], yb[j + 1]) - max(ya[i], yb[j]))
        total += overlap * (x1 - x0)
    return total


def layout_2_depth(cor_id, h, w, return_mask=False):
    """Per-pixel depth of the layout described by its corners, camera 1.6 above the floor (eval_general.py:14-53)."""
    vc, vf = cor_2_1d(cor_id, h, w)
    vc, vf = vc[None, :], vf[None, :]
    assert (vc > 0).sum() == 0
    assert (vf < 0).sum() == 0
    vs = np.repeat((((np.arange(h) + 0.5) / h - 0.5) * np.pi)[:, None], w, axis=1)
    cam_h = 1.6
    wall_range = cam_h / np.tan(vf)                          # horizontal distance to the wall in each column
    ceil_h = np.abs(wall_range * np.tan(vc))
    floor_mask, ceil_mask = vs > vf, vs < vc
    wall_mask = (~floor_mask) & (~ceil_mask)
    depth = np.zeros([h, w], np.float32)
    depth[floor_mask] = np.abs(cam_h / np.sin(vs))[floor_mask]
    depth[ceil_mask] = np.abs(ceil_h / np.sin(vs))[ceil_mask]
    depth[wall_mask] = np.abs(wall_range / np.cos(vs))[wall_mask]
    assert (depth == 0).sum() == 0
    if return_mask:
        return depth, floor_mask, ceil_mask, wall_mask
    return depth


def layout_metrics(dt_cor_id, gt_cor_id, w=1024, h=512):
    """-> dict(iou2d, iou3d, rmse, delta_1, n_corners) for one detection / ground-truth pair of corner lists in pixel
    coordinates (ceiling / floor rows alternating), or None when the ground truth polygon is invalid
    (eval_general.py:56-103)."""
    dt_cor_id, gt_cor_id = np.asarray(dt_cor_id), np.asarray(gt_cor_id)
    dt_floor, dt_ceil = dt_cor_id[1::2], dt_cor_id[0::2]
    gt_floor, gt_ceil = gt_cor_id[1::2], gt_cor_id[0::2]
    assert (dt_floor[:, 0] != dt_ceil[:, 0]).sum() == 0
    assert (gt_floor[:, 0] != gt_ceil[:, 0]).sum() == 0
    cam = -1.6
    dt_xy = postproc.pano_to_plan(dt_floor, cam, 1024, 512, floorW=1, floorH=1)
    gt_xy = postproc.pano_to_plan(gt_floor, cam, 1024, 512, floorW=1, floorH=1)
    if not postproc.polygon_is_simple(gt_xy):
        return None
    try:
        area_dt, area_gt = polygon_area(dt_xy), polygon_area(gt_xy)
        area_inter = polygon_intersection_area(dt_xy, gt_xy)
        iou2d = area_inter / (area_gt + area_dt - area_inter)
    except Exception:
        iou2d = 0
    try:
        h_dt = abs(get_z1(dt_floor[:, 1], dt_ceil[:, 1], cam, 512).mean() - cam)
        h_gt = abs(get_z1(gt_floor[:, 1], gt_ceil[:, 1], cam, 512).mean() - cam)
        inter3d = area_inter * min(h_dt, h_gt)
        iou3d = inter3d / (area_dt * h_dt + area_gt * h_gt - inter3d)
    except Exception:
        iou3d = 0
    gt_depth = layout_2_depth(gt_cor_id, h, w)
    try:
        dt_depth = layout_2_depth(dt_cor_id, h, w)
    except Exception:
        dt_depth = np.zeros_like(gt_depth)
    rmse = ((gt_depth - dt_depth) ** 2).mean() ** 0.5
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.maximum(gt_depth / dt_depth, dt_depth / gt_depth)
    return {"iou2d": iou2d, "iou3d": iou3d, "rmse": rmse, "delta_1": (ratio < 1.25).mean(), "n_corners": len(gt_floor)}


def get_z1(coory0, coory1, z0=50, coorH=512):
    """Height of the plane through rows coory1 given rows coory0 lie on plane z0 (post_proc.py:101-106)."""
    c0 = z0 / np.tan(postproc.row_to_v(coory0, coorH))
    return c0 * np.tan(postproc.row_to_v(coory1, coorH))


def test_general(dt_cor_id, gt_cor_id, w, h, losses):
    """Reference signature: appends the four metrics to losses[<corner-count bucket>] and losses['overall']."""
    m = layout_metrics(dt_cor_id, gt_cor_id, w, h)
    if m is None:
        print("Skip ground truth invalid")
        return
    n = m["n_corners"]
    bucket = "odd" if n % 2 == 1 else (str(n) if n < 10 else "10+")
    for key in (bucket, "overall"):
        losses[key]["2DIoU"].append(m["iou2d"])
        losses[key]["3DIoU"].append(m["iou3d"])
        losses[key]["rmse"].append(m["rmse"])
        losses[key]["delta_1"].append(m["delta_1"])


test_general.__test__ = False      # not a pytest test despite the reference's name


# ---- cuboid metrics (eval_cuboid.py:13-145) -------------------------------------------------------------------------
def _inward_halfspace(pa, pb, p):
    """Plane through p, pa, pb as [a, b, c, d] with a*x + b*y + c*z + d <= 0 on the origin's side."""
    n = np.cross(pa - p, pb - p)
    if -n @ p > 0:
        n = -n
    return [n[0], n[1], n[2], -n @ p]


def _room_halfspaces(floor_xyz, ceil_xyz):
    """Half-spaces (3 at every floor corner, 3 at every ceiling corner) whose intersection is the room seen from the
    camera at the origin (eval_cuboid.py:24-46)."""
    N = len(floor_xyz)
    hs = []
    for i in range(N):
        a, b = (i - 1) % N, (i + 1) % N
        f, c = floor_xyz[i], ceil_xyz[i]
        hs += [_inward_halfspace(floor_xyz[a], floor_xyz[b], f), _inward_halfspace(floor_xyz[a], c, f),
               _inward_halfspace(c, floor_xyz[b], f), _inward_halfspace(ceil_xyz[a], ceil_xyz[b], c),
               _inward_halfspace(ceil_xyz[a], f, c), _inward_halfspace(f, ceil_xyz[b], c)]
    return np.array(hs)


def eval_3diou(dt_floor_coor, dt_ceil_coor, gt_floor_coor, gt_ceil_coor, ch=-1.6, coorW=1024, coorH=512):
    """3D IoU in percent of two rooms given by floor / ceiling corner pixels, camera 1.6 above the floor: volumes by
    half-space intersection + convex hull (Qhull through SciPy, as the reference: eval_cuboid.py:49-92)."""
    from scipy.spatial import ConvexHull, HalfspaceIntersection

    def room(floor, ceil):
        floor, ceil = np.array(floor), np.array(ceil)
        assert (floor[:, 0] != ceil[:, 0]).sum() == 0
        xy = postproc.pano_to_plan(floor, ch, coorW, coorH, floorW=1, floorH=1)
        fxyz = np.hstack([xy, np.zeros((len(xy), 1)) + ch])
        cxyz = fxyz.copy()
        cxyz[:, 2] = np.sqrt((fxyz[:, :2] ** 2).sum(1)) * np.tan(postproc.row_to_v(ceil[:, 1], coorH))
        return _room_halfspaces(fxyz, cxyz)

    def volume(halfspaces):
        return ConvexHull(HalfspaceIntersection(halfspaces, np.zeros(3)).intersections).volume

    dt_h, gt_h = room(dt_floor_coor, dt_ceil_coor), room(gt_floor_coor, gt_ceil_coor)
    v_in, v_dt, v_gt = volume(np.concatenate([dt_h, gt_h])), volume(dt_h), volume(gt_h)
    return 100 * v_in / (v_dt + v_gt - v_in)


def _boundary_rows(points, w):
    """Row of a boundary at every column from scattered (col, row) samples (eval_cuboid.py:95-97)."""
    points = points[np.argsort(points[:, 0])]
    return np.interp(np.arange(w), points[:, 0], points[:, 1], period=w)


def cuboid_metrics(dt_cor_id, z0, z1, gt_cor_id, w=1024, h=512):
    """Corner error (% of the image diagonal), pixel surface error (% of pixels whose ceiling / wall / floor label
    differs) and 3D IoU (%) of a predicted against a ground-truth cuboid (eval_cuboid.py:100-145)."""
    from .labels import pano_connect_points
    dt_cor_id, gt_cor_id = np.asarray(dt_cor_id), np.asarray(gt_cor_id)
    ce = 100 * np.sqrt(((gt_cor_id - dt_cor_id) ** 2).sum(1)).mean() / np.sqrt(w ** 2 + h ** 2)

    def trace(cor, first, z):
        return _boundary_rows(np.concatenate([pano_connect_points(cor[j * 2 + first], cor[(j * 2 + 2 + first) % 8], z)
                                              for j in range(4)], 0), w)

    ceil_dt = trace(dt_cor_id, 0, -z0)
    floor_dt = postproc.infer_coory(ceil_dt, z1 - z0, z0)
    ceil_gt, floor_gt = trace(gt_cor_id, 0, -z0), trace(gt_cor_id, 1, z0)

    def surface(ceil_rows, floor_rows):
        s = np.zeros((h, w), dtype=np.int32)
        s[np.round(ceil_rows).astype(int), np.arange(w)] = 1
        s[np.round(floor_rows).astype(int), np.arange(w)] = 1
        return np.cumsum(s, axis=0)

    pe = 100 * (surface(ceil_dt, floor_dt) != surface(ceil_gt, floor_gt)).sum() / (h * w)
    return {"CE": ce, "PE": pe, "3DIoU": eval_3diou(dt_cor_id[1::2], dt_cor_id[0::2], gt_cor_id[1::2], gt_cor_id[0::2])}


def test_cuboid(dt_cor_id, z0, z1, gt_cor_id, w, h, losses):
    """Reference signature of ``eval_cuboid.test``: appends CE / PE / 3DIoU to `losses`."""
    m = cuboid_metrics(dt_cor_id, z0, z1, gt_cor_id, w, h)
    for k in ("CE", "PE", "3DIoU"):
        losses[k].append(m[k])


test_cuboid.__test__ = False
