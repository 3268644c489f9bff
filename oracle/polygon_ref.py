"""TEST INFRASTRUCTURE (oracle/): exact area of the intersection of two simple polygons, in rational arithmetic.

The reference intersects the predicted and the ground-truth floor plans with GEOS through shapely
(``eval_general.py:69-79``: ``Polygon(dt).intersection(Polygon(gt)).area``); shapely / GEOS are absent offline, so the
product (``horizonnet_amd/evaluation.py::polygon_intersection_area``) uses its own vertical-slab decomposition in float64.
This file is an INDEPENDENT restatement of what GEOS' overlay computes for two valid simple polygons -- the area of the
point-set intersection -- by a different algorithm and without rounding:

  * every coordinate is converted to a ``fractions.Fraction`` (floats are dyadic rationals: exact);
  * both rings are oriented counter-clockwise; the boundary of A n B consists of the pieces of dA strictly inside B, the
    pieces of dB strictly inside A, and the pieces where dA and dB coincide WITH THE SAME DIRECTION (interiors on the same
    side; counted once) -- coinciding pieces of opposite direction bound a contact of measure zero;
  * every edge is split at every point where it meets the other ring (proper crossings, touches, ends of collinear
    overlaps), each piece is classified by its midpoint, and Green's theorem sums 1/2 (x0 y1 - x1 y0) over the kept pieces.

Parity status: NOT pinned against GEOS itself (no shapely in the build container or on the GPU box).  It is pinned against
closed forms (rectangles, translated copies, nested / disjoint / edge-sharing rooms) in tests/test_round2_cpu.py, and the
product is then checked against it on random non-convex and rectilinear rooms, including the degenerate contacts Manhattan
layouts produce all the time (shared walls, identical polygons)."""
from fractions import Fraction as Fr


def _ring(p):
    pts = [(Fr(float(x)), Fr(float(y))) for x, y in p]
    if len(pts) >= 2 and pts[0] == pts[-1]:
        pts = pts[:-1]
    if _area2(pts) < 0:
        pts.reverse()
    return pts


def _area2(pts):
    s = Fr(0)
    for (x0, y0), (x1, y1) in zip(pts, pts[1:] + pts[:1]):
        s += x0 * y1 - x1 * y0
    return s


def _cross(ax, ay, bx, by):
    return ax * by - ay * bx


def _meet_params(p0, p1, q0, q1):
    """Parameters t in [0,1] along p0->p1 where the segment meets segment q0->q1 (one value for a crossing or touch, the two
    ends of the common part for a collinear overlap)."""
    rx, ry = p1[0] - p0[0], p1[1] - p0[1]
    sx, sy = q1[0] - q0[0], q1[1] - q0[1]
    den = _cross(rx, ry, sx, sy)
    qpx, qpy = q0[0] - p0[0], q0[1] - p0[1]
    if den != 0:
        t = _cross(qpx, qpy, sx, sy) / den
        u = _cross(qpx, qpy, rx, ry) / den
        return [t] if 0 <= t <= 1 and 0 <= u <= 1 else []
    if _cross(qpx, qpy, rx, ry) != 0:
        return []                                        # parallel, not collinear
    rr = rx * rx + ry * ry
    t0 = (qpx * rx + qpy * ry) / rr
    t1 = ((q1[0] - p0[0]) * rx + (q1[1] - p0[1]) * ry) / rr
    lo, hi = max(min(t0, t1), Fr(0)), min(max(t0, t1), Fr(1))
    return [lo, hi] if lo <= hi else []


def _locate(pt, ring):
    """'in' / 'out', or ('on', direction of the ring edge through pt) for a point of the plane."""
    x, y = pt
    inside = False
    for (x0, y0), (x1, y1) in zip(ring, ring[1:] + ring[:1]):
        if _cross(x1 - x0, y1 - y0, x - x0, y - y0) == 0 and min(x0, x1) <= x <= max(x0, x1) and min(y0, y1) <= y <= max(y0, y1):
            return ("on", (x1 - x0, y1 - y0))
        if (y0 > y) != (y1 > y):                          # half-open rule: an edge counts when it spans y
            xc = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
            if xc > x:
                inside = not inside
    return "in" if inside else "out"


def _kept_area2(a, b, keep_shared):
    """Sum of (x0 y1 - x1 y0) over the pieces of ring a that bound a n b."""
    total = Fr(0)
    for p0, p1 in zip(a, a[1:] + a[:1]):
        ts = {Fr(0), Fr(1)}
        for q0, q1 in zip(b, b[1:] + b[:1]):
            ts.update(_meet_params(p0, p1, q0, q1))
        ts = sorted(ts)
        dx, dy = p1[0] - p0[0], p1[1] - p0[1]
        for t0, t1 in zip(ts[:-1], ts[1:]):
            tm = (t0 + t1) / 2
            where = _locate((p0[0] + tm * dx, p0[1] + tm * dy), b)
            if where == "out":
                continue
            if where != "in":                            # on b's boundary: a collinear shared piece
                ex, ey = where[1]
                if not (keep_shared and dx * ex + dy * ey > 0):
                    continue
            xa, ya = p0[0] + t0 * dx, p0[1] + t0 * dy
            xb, yb = p0[0] + t1 * dx, p0[1] + t1 * dy
            total += xa * yb - xb * ya
    return total


def intersection_area_exact(a, b):
    """Area of (polygon a) n (polygon b) as a Fraction; a, b: [N,2] vertex lists of simple polygons, either orientation."""
    ra, rb = _ring(a), _ring(b)
    return (_kept_area2(ra, rb, True) + _kept_area2(rb, ra, False)) / 2


def area_exact(a):
    return abs(_area2(_ring(a))) / 2
