"""shapely-1.8 `affine_transform` arithmetic for the 6-parameter 2-D matrix
[a, b, d, e, xoff, yoff]:  x' = a*x + b*y + xoff ; y' = d*x + e*y + yoff
(Python-float evaluation order of shapely 1.8's `affine_pts`)."""
from .geometry import LinearRing, LineString, Point


def affine_transform(geom, matrix):
    a, b, d, e, xoff, yoff = matrix
    out = []
    for x, y in geom.coords:
        out.append((a * x + b * y + xoff, d * x + e * y + yoff))
    if isinstance(geom, LinearRing):
        r = LinearRing.__new__(LinearRing)
        r._coords = out
        return r
    if isinstance(geom, Point):
        return Point(out[0])
    return LineString(out)
