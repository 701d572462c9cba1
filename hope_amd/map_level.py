"""Difficulty label of a parking map: `get_map_level` (src/env/map_level.py:27-112), which `ParkingMapDLP.reset` stores as
`map.map_level` (src/env/parking_map_dlp.py:84) and the evaluation buckets its statistics by (src/evaluation/eval_utils.py:
69-72,99).  Host-side, reset-time only.

The reference asks shapely for five things; they are restated here as small numpy routines (GEOS is not available in this
image, so -- like the rest of the GEOS slice -- the geometry is parity-unpinned; the CONTROL FLOW is pinned to the
reference's own source by tests/test_map_level.py):
  Point.distance(LinearRing)        -> point_ring_distance     (distance to the boundary curve)
  LinearRing.distance(LinearRing)   -> ring_ring_distance      (0 when the boundaries cross)
  LinearRing.equals(LinearRing)     -> identity of the obstacle (the reference compares members of one list)
  MultiPoint.minimum_rotated_rectangle -> min_area_rectangle   (shapely 1.8: smallest-AREA rectangle over the hull edges)
  Polygon.intersects(LinearRing)    -> polygon_meets_ring      (boundary crossing, or the ring inside the polygon)
"""
import math

import numpy as np

from . import tables as T

LEVEL_NORMAL, LEVEL_COMPLEX, LEVEL_EXTREM = 'Normal', 'Complex', 'Extrem'
LENGTH = T.WHEEL_BASE + T.FRONT_HANG + T.REAR_HANG
WIDTH = T.WIDTH
MAX_DRIVE_DISTANCE = 15.0                                              # configs.py:74
MIN_LOT_LEN_NORMAL, MIN_LOT_WID_NORMAL = LENGTH * 1.25, WIDTH + 0.85      # configs.py:43-52
BAY_WALL_NORMAL, PARA_WALL_NORMAL = 7.0, 4.5                            # configs.py:58-65
EXTREM_PARK_LOT_LENGTH = min(LENGTH * 1.2, LENGTH + 0.9)                # map_level.py:11


# ---- geometry ----------------------------------------------------------------------------------------------------
def _pt_seg(p, a, b):
    ax, ay, bx, by = a[0], a[1], b[0], b[1]
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    if l2 == 0.0:
        return math.hypot(p[0] - ax, p[1] - ay)
    r = ((p[0] - ax) * dx + (p[1] - ay) * dy) / l2
    if r <= 0.0:
        return math.hypot(p[0] - ax, p[1] - ay)
    if r >= 1.0:
        return math.hypot(p[0] - bx, p[1] - by)
    return abs((ay - p[1]) * dx - (ax - p[0]) * dy) / math.sqrt(l2)


def _edges(ring):
    n = len(ring)
    return [(ring[i], ring[(i + 1) % n]) for i in range(n)]


def point_ring_distance(p, ring):
    return min(_pt_seg(p, a, b) for a, b in _edges(ring))


def _orient(a, b, c):
    d = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    return int(d > 0) - int(d < 0)


def _segs_meet(p1, p2, q1, q2):
    if max(p1[0], p2[0]) < min(q1[0], q2[0]) or max(q1[0], q2[0]) < min(p1[0], p2[0]) or \
       max(p1[1], p2[1]) < min(q1[1], q2[1]) or max(q1[1], q2[1]) < min(p1[1], p2[1]):
        return False
    o1, o2 = _orient(p1, p2, q1), _orient(p1, p2, q2)
    if o1 * o2 > 0:
        return False
    o3, o4 = _orient(q1, q2, p1), _orient(q1, q2, p2)
    return o3 * o4 <= 0


def rings_cross(a, b):
    return any(_segs_meet(p1, p2, q1, q2) for p1, p2 in _edges(a) for q1, q2 in _edges(b))


def ring_ring_distance(a, b):
    if rings_cross(a, b):
        return 0.0
    return min(min(point_ring_distance(p, b) for p in a), min(point_ring_distance(p, a) for p in b))


def _convex_hull(pts):
    pts = sorted(set((float(x), float(y)) for x, y in pts))
    if len(pts) <= 2:
        return pts

    def half(seq):
        h = []
        for p in seq:
            while len(h) >= 2 and ((h[-1][0] - h[-2][0]) * (p[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (p[0] - h[-2][0])) <= 0:
                h.pop()
            h.append(p)
        return h
    lo, up = half(pts), half(reversed(pts))
    return lo[:-1] + up[:-1]


def min_area_rectangle(pts):
    """smallest-area enclosing rectangle with one side on a convex-hull edge -> 4 corners (CCW)."""
    hull = _convex_hull(pts)
    best, best_area = None, math.inf
    for (ax, ay), (bx, by) in _edges(hull):
        ex, ey = bx - ax, by - ay
        n = math.hypot(ex, ey)
        if n == 0:
            continue
        ux, uy = ex / n, ey / n
        s = [(x - ax) * ux + (y - ay) * uy for x, y in hull]
        t = [-(x - ax) * uy + (y - ay) * ux for x, y in hull]
        s0, s1, t0, t1 = min(s), max(s), min(t), max(t)
        area = (s1 - s0) * (t1 - t0)
        if area < best_area:
            best_area = area
            best = [(ax + ux * a - uy * b, ay + uy * a + ux * b) for a, b in ((s0, t0), (s1, t0), (s1, t1), (s0, t1))]
    return best


def _inside_convex(p, poly):
    sgn = [_orient(a, b, p) for a, b in _edges(poly)]
    return all(v >= 0 for v in sgn) or all(v <= 0 for v in sgn)


def polygon_meets_ring(poly, ring):
    """filled convex polygon vs a closed curve"""
    return rings_cross(poly, ring) or any(_inside_convex(p, poly) for p in ring)


# ---- map_level.py restated ---------------------------------------------------------------------------------------------
def _create_box(pose):
    px, py, c, s = float(pose[0]), float(pose[1]), math.cos(pose[2]), math.sin(pose[2])
    return [(c * float(x) - s * float(y) + px, s * float(x) + c * float(y) + py) for x, y in T.VEHICLE_BOX]   # rb, rf, lf, lb


def _mid(a, b):
    return ((a[0] + b[0]) / 2, (a[1] + b[1]) / 2)


def _translate(pt, heading, dist):
    return (pt[0] + math.cos(heading) * dist, pt[1] + math.sin(heading) * dist)


def _nearest(pt, rings, max_min_dist, skip):
    """_get_nearest_obstacle (:139-154): index of the nearest ring closer than max_min_dist, not in `skip`; or None"""
    best, best_d = None, max_min_dist
    for i, r in enumerate(rings):
        if i in skip:
            continue
        d = point_ring_distance(pt, r)
        if d < best_d:
            best_d, best = d, i
    return best


def _surrounding(dest, rings):
    """_get_surrounding_obstacle (:13-25): nearest obstacle within LENGTH / 2 of the dest box's left / right / front / back
    edge mid-points, each obstacle used at most once -> indices (left, right, front, back) or None"""
    rb, rf, lf, lb = _create_box(dest)
    found = []
    for pt in (_mid(lf, lb), _mid(rf, rb), _mid(lf, rf), _mid(lb, rb)):
        found.append(_nearest(pt, rings, LENGTH / 2, {f for f in found if f is not None}))
    return found


def _has_enough_space(pos, rings, width=None, length=None):
    box = _create_box(pos)
    ok_w = ok_l = True
    if width is not None:
        left, right, _, _ = _surrounding(pos, rings)
        if left is not None and right is not None:
            ok_w = not (ring_ring_distance(rings[left], box) + ring_ring_distance(rings[right], box) + WIDTH < width)
    if length is not None:
        _, _, front, back = _surrounding(pos, rings)
        if front is not None and back is not None:
            ok_l = not (ring_ring_distance(rings[front], box) + ring_ring_distance(rings[back], box) + LENGTH < length)
    return ok_w and ok_l


def _check_extrem(start, dest, rings):
    left, right, front, back = _surrounding(dest, rings)
    if math.hypot(start[0] - dest[0], start[1] - dest[1]) > 30.0:
        if front is not None and back is not None and not _has_enough_space(dest, rings, length=MIN_LOT_LEN_NORMAL):
            return True
        if left is not None and right is not None and not _has_enough_space(dest, rings, width=MIN_LOT_WID_NORMAL):
            return True
    return front is not None and back is not None and not _has_enough_space(dest, rings, length=EXTREM_PARK_LOT_LENGTH)


def get_map_level(start, dest, obstacles):
    """start, dest: (x, y, heading); obstacles: list of rings, each an (n, 2) array of its open vertex list."""
    rings = [[(float(x), float(y)) for x, y in np.asarray(r, dtype=np.float64)] for r in obstacles]
    if len(rings) <= 1:
        return LEVEL_NORMAL
    if _check_extrem(start, dest, rings):
        return LEVEL_EXTREM
    far = math.hypot(start[0] - dest[0], start[1] - dest[1]) > MAX_DRIVE_DISTANCE
    left, right, front, back = _surrounding(dest, rings)
    rb, rf, lf, lb = _create_box(dest)
    if left is not None and right is not None and front is None:                     # bay parking
        if far or not _has_enough_space(dest, rings, width=MIN_LOT_WID_NORMAL):
            return LEVEL_COMPLEX
        h = dest[2]
        pts = [_translate(lf, h, 0.2), _translate(rf, h, 0.2), _translate(lf, h, BAY_WALL_NORMAL - 0.5),
               _translate(rf, h, BAY_WALL_NORMAL - 0.5), (start[0], start[1])]
        free = min_area_rectangle(pts)
        ok = not any(polygon_meets_ring(free, r) for i, r in enumerate(rings) if i not in (left, right))
        return LEVEL_NORMAL if ok else LEVEL_COMPLEX
    if front is not None and back is not None:                                        # parallel parking
        if far or not _has_enough_space(dest, rings, length=MIN_LOT_LEN_NORMAL):
            return LEVEL_COMPLEX
        out = dest[2] + math.pi / 2
        if math.cos(out) * (start[0] - dest[0]) + math.sin(out) * (start[1] - dest[1]) < 0:
            out += math.pi
            kf, kb = rf, rb
        else:
            kf, kb = lf, lb
        pts = [_translate(kf, out, 0.2), _translate(kb, out, 0.2), _translate(kf, out, PARA_WALL_NORMAL - 0.5),
               _translate(kb, out, PARA_WALL_NORMAL - 0.5)] + _create_box(start) + [(start[0], start[1])]
        free = min_area_rectangle(pts)
        ok = not any(polygon_meets_ring(free, r) for i, r in enumerate(rings) if i not in (back, front))
        return LEVEL_NORMAL if ok else LEVEL_COMPLEX
    if (left is None or right is None) and (front is None or back is None):
        return LEVEL_NORMAL
    return LEVEL_COMPLEX
