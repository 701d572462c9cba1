"""Coordinate holders with the handful of methods the reference's module-level code and
constructors call. Geometry predicates that GEOS would answer are NOT provided."""
import math
import numpy as np


class BaseGeom:
    pass


class Point(BaseGeom):
    def __init__(self, *args):
        if len(args) == 1:
            a = args[0]
            if isinstance(a, Point):
                self._x, self._y = a.x, a.y
            else:
                self._x, self._y = a[0], a[1]
        else:
            self._x, self._y = args[0], args[1]

    @property
    def x(self):
        return self._x

    @property
    def y(self):
        return self._y

    @property
    def coords(self):
        return [(self._x, self._y)]

    def distance(self, other):
        dx = self._x - other.x
        dy = self._y - other.y
        return math.sqrt(dx * dx + dy * dy)


class LineString(BaseGeom):
    def __init__(self, coords):
        self._coords = [tuple(map(float, c)) for c in coords]

    @property
    def coords(self):
        return self._coords

    def intersection(self, ring):
        """ray(0,0)->end against a closed ring: returns the hit point(s) holder.
        Stub-computed (parametric segment/segment); only used by constructors."""
        (x0, y0), (x1, y1) = self._coords[0], self._coords[-1]
        pts = []
        rc = ring.coords
        for (ax, ay), (bx, by) in zip(rc[:-1], rc[1:]):
            rx, ry = x1 - x0, y1 - y0
            sx, sy = bx - ax, by - ay
            den = rx * sy - ry * sx
            if den == 0:
                continue
            t = ((ax - x0) * sy - (ay - y0) * sx) / den
            u = ((ax - x0) * ry - (ay - y0) * rx) / den
            if 0 <= t <= 1 and 0 <= u <= 1:
                pts.append((x0 + t * rx, y0 + t * ry))
        return MultiPoint(pts)


class MultiPoint(BaseGeom):
    def __init__(self, pts=()):
        self._pts = list(pts)

    def distance(self, p):
        return min(math.sqrt((x - p.x) ** 2 + (y - p.y) ** 2) for x, y in self._pts)


class LinearRing(LineString):
    def __init__(self, coords=()):
        cs = [tuple(map(float, c)) for c in coords]
        if cs and cs[0] != cs[-1]:
            cs.append(cs[0])
        self._coords = cs

    # shapely-1.x pickles a LinearRing as WKB bytes of a LineString
    def __setstate__(self, state):
        import struct
        b = state
        assert b[0] == 1
        gtype, n = struct.unpack_from('<II', b, 1)
        arr = np.frombuffer(b, '<f8', 2 * n, offset=9).reshape(n, 2)
        self._coords = [tuple(map(float, c)) for c in arr]

    def __reduce__(self):
        raise NotImplementedError


class Polygon(BaseGeom):
    def __init__(self, shell=None):
        self.exterior = shell
