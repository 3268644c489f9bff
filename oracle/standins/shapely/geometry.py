"""Stand-in for the two shapely names the reference imports (inference.py:10, dataset.py:9, eval_general.py).

TEST INFRASTRUCTURE ONLY.  shapely is not installed offline.  Two calls are restated (parity unpinned against shapely
itself): ``Polygon(...).is_valid`` (inference.py:120, "for fear self-intersection") for a single closed ring -- valid
iff the ring has a non-zero area and no two non-adjacent edges share a point; and ``LineString.intersects``
(dataset.py:172-186, corner occlusion) -- two polylines share at least one point.  Everything else raises.
"""
import numpy as np


def _segments_meet(p, q, r, s):
    """Closed segments pq and rs share at least one point."""
    def side(a, b, c):
        v = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
        return int(v > 0) - int(v < 0)

    def within(a, b, c):     # c collinear with ab: inside its bounding box?
        return min(a[0], b[0]) <= c[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= c[1] <= max(a[1], b[1])

    d1, d2, d3, d4 = side(p, q, r), side(p, q, s), side(r, s, p), side(r, s, q)
    if d1 != d2 and d3 != d4:
        return True
    return ((d1 == 0 and within(p, q, r)) or (d2 == 0 and within(p, q, s)) or
            (d3 == 0 and within(r, s, p)) or (d4 == 0 and within(r, s, q)))


class Polygon:
    def __init__(self, shell=None, holes=None):
        if holes:
            raise NotImplementedError("shapely stand-in: holes are out of scope")
        ring = np.asarray(shell, np.float64).reshape(-1, 2)
        # repeated consecutive points (an explicitly closed shell repeats its first point) are legal in OGC SFS / GEOS rings and
        # change neither area nor validity: drop them, a zero-length edge would otherwise hide two true neighbours from each other
        if len(ring) > 1:
            ring = ring[(ring != np.roll(ring, 1, axis=0)).any(axis=1)]
        self._ring = ring

    @property
    def area(self):
        x, y = self._ring[:, 0], self._ring[:, 1]
        return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))

    @property
    def is_valid(self):
        ring = self._ring
        n = len(ring)
        if n < 3 or self.area == 0.0:
            return False
        for i in range(n):
            for j in range(i + 2, n):
                if i == 0 and j == n - 1:
                    continue                                   # adjacent through the closing edge
                if _segments_meet(ring[i], ring[(i + 1) % n], ring[j], ring[(j + 1) % n]):
                    return False
        return True

    def __getattr__(self, name):
        raise NotImplementedError("shapely stand-in: Polygon.%s is out of scope" % name)


class LineString:
    """Open polyline; only ``intersects`` (dataset.py:172-186, occlusion test) is provided."""

    def __init__(self, coords):
        self._pts = np.asarray(coords, np.float64).reshape(-1, 2)

    def intersects(self, other):
        a, b = self._pts, other._pts
        for i in range(len(a) - 1):
            for j in range(len(b) - 1):
                if _segments_meet(a[i], a[i + 1], b[j], b[j + 1]):
                    return True
        return False

    def __getattr__(self, name):
        raise NotImplementedError("shapely stand-in: LineString.%s is out of scope" % name)
