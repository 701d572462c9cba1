"""Analytic known-answer tests for the GEOS-delegated predicates (shapely/GEOS
is absent, so these are PARITY-UNPINNED against the reference and pinned only here):
LinearRing.intersects (boundary-only, touch counts, containment does not), convex quad
overlap area, point-ring distance cull, exact orientation fallback."""
import math

import numpy as np

from oracle import oracle as O


def rect(cx, cy, w, h, th=0.0):
    c, s = math.cos(th), math.sin(th)
    pts = [(-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2)]
    return np.array([(cx + c * x - s * y, cy + s * x + c * y) for x, y in pts])


def test_orientation_exact_fallback():
    assert O.orient((0, 0), (1, 0), (0.5, 1)) == 1
    assert O.orient((0, 0), (1, 0), (0.5, -1)) == -1
    assert O.orient((0, 0), (1, 1), (2, 2)) == 0
    # near-collinear: filter fails, exact expansion decides.  (12,12),(24,24) with 0.5+k*ulp offsets
    u = 2.0 ** -53
    for k in (-3, -1, 0, 1, 3):
        got = O.orient((0.5, 0.5), (12.0, 12.0), (24.0, 24.0 + k * 64 * u))
        assert got == (k > 0) - (k < 0), k
    # Kettner et al. style grid: sign must match exact rational arithmetic
    from fractions import Fraction as F
    rng = np.random.default_rng(1)
    for _ in range(2000):
        a = (0.5 + rng.integers(0, 64) * u, 0.5 + rng.integers(0, 64) * u)
        b, c = (12.0, 12.0), (24.0, 24.0)
        ex = (F(a[0]) - F(c[0])) * (F(b[1]) - F(c[1])) - (F(a[1]) - F(c[1])) * (F(b[0]) - F(c[0]))
        assert O.orient(a, b, c) == (ex > 0) - (ex < 0)


def test_segment_intersection_cases():
    si = O.segments_intersect
    assert si((0, 0), (2, 2), (0, 2), (2, 0))            # proper crossing
    assert si((0, 0), (1, 1), (1, 1), (2, 0))            # touching end points
    assert si((0, 0), (2, 0), (1, 0), (1, 5))            # T touch
    assert si((0, 0), (2, 0), (1, 0), (3, 0))            # collinear overlap
    assert not si((0, 0), (1, 0), (2, 0), (3, 0))        # collinear disjoint
    assert not si((0, 0), (1, 0), (0, 1), (1, 1))        # parallel
    assert not si((0, 0), (1, 1), (2, 0), (3, -1))       # disjoint envelopes
    assert not si((0, 0), (4, 4), (3, 0), (4, 1))        # envelopes overlap, no hit
    assert si((0, 0), (0, 0), (-1, 0), (1, 0))           # degenerate point on a segment
    assert not si((0, 1), (0, 1), (-1, 0), (1, 0))       # degenerate point off it


def test_ring_intersects_semantics():
    car = rect(0, 0, 4.69, 1.94)
    big = rect(0, 0, 40, 40)
    assert not O.ring_intersects(car, big)                # wholly inside a big obstacle: NOT a hit
    assert not O.ring_intersects(car, rect(10, 0, 2, 2))
    assert O.ring_intersects(car, rect(2.345 + 1, 0, 2, 2))       # touching side (x = 2.345)
    assert O.ring_intersects(car, rect(2.0, 0.5, 2, 2, 0.3))
    tri = np.array([(2.0, -3.0), (2.6, 3.0), (6.0, 0.0)])
    assert O.ring_intersects(car, tri)
    assert not O.ring_intersects(car, tri + [3.0, 0])
    # corner-to-corner touch
    assert O.ring_intersects(rect(0, 0, 2, 2), rect(2, 2, 2, 2))
    # detect_collision over a padded (n,4,2) tile incl. a triangle stored with a repeated vertex
    verts = np.stack([rect(10, 0, 2, 2), np.vstack([tri, tri[2:3]])])
    assert O.detect_collision(car, verts, [4, 3])
    assert not O.detect_collision(car, verts[:1], [4])


def test_quad_overlap_area():
    a = rect(0, 0, 4.69, 1.94)
    assert abs(O.quad_area(a) - 4.69 * 1.94) < 1e-14
    assert abs(O.quad_intersection_area(a, a) - 4.69 * 1.94) < 1e-13
    # axis-aligned shift: overlap = (w-dx)*(h-dy)
    for dx, dy in [(0.1, 0.0), (0.5, 0.3), (4.0, 1.0), (4.69, 0), (5.0, 0), (0, 2.0)]:
        want = max(4.69 - dx, 0) * max(1.94 - dy, 0)
        got = O.quad_intersection_area(a, rect(dx, dy, 4.69, 1.94))
        assert abs(got - want) < 1e-12, (dx, dy)
    # 90-degree rotated copy about the centre: overlap is the 1.94 x 1.94 square
    assert abs(O.quad_intersection_area(a, rect(0, 0, 4.69, 1.94, math.pi / 2)) - 1.94 ** 2) < 1e-12
    # unit square vs 45-degree unit square: regular octagon, area 2(sqrt2 - 1)
    assert abs(O.quad_intersection_area(rect(0, 0, 1, 1), rect(0, 0, 1, 1, math.pi / 4)) -
               2 * (math.sqrt(2) - 1)) < 1e-13
    # symmetry and rigid-motion invariance at DLP-scale coordinates
    rng = np.random.default_rng(3)
    for _ in range(200):
        p = rect(rng.uniform(-1, 1), rng.uniform(-1, 1), 4.69, 1.94, rng.uniform(-3, 3))
        q = rect(rng.uniform(-1, 1), rng.uniform(-1, 1), 4.69, 1.94, rng.uniform(-3, 3))
        ab, ba = O.quad_intersection_area(p, q), O.quad_intersection_area(q, p)
        assert abs(ab - ba) < 1e-12
        sh = np.array([120.0, -30.0])
        assert abs(O.quad_intersection_area(p + sh, q + sh) - ab) < 1e-10
        # Monte-Carlo-free check: inclusion-exclusion bound
        assert -1e-12 <= ab <= 4.69 * 1.94 + 1e-12


def raycast(rings_ego, rng=10.0):
    """independent parametric ray/segment caster (test-side), ego frame."""
    out = np.full(120, rng)
    for i in range(120):
        th = 2 * math.pi * i / 120
        dx, dy = math.cos(th), math.sin(th)
        for r in rings_ego:
            n = len(r)
            for k in range(n):
                (ax, ay), (bx, by) = r[k], r[(k + 1) % n]
                sx, sy = bx - ax, by - ay
                den = dx * sy - dy * sx
                if abs(den) < 1e-14:
                    continue
                t = (ax * sy - ay * sx) / den
                u = (ax * dy - ay * dx) / den
                if t >= 0 and 0 <= u <= 1:
                    out[i] = min(out[i], t)
    return out


def test_lidar_ring_cull_and_transform():
    """ring kept iff its boundary comes within 10 m of the ego origin (lidar_simulator.py:69).
    Rings are tilted in the ego frame: an edge that is (nearly) axis-aligned THERE is missed by the
    reference's tolerance-free bbox test -- a faithful quirk covered by the golden lidar fixture."""
    pose = np.array([3.0, -2.0, 0.7])
    c, s = math.cos(0.7), math.sin(0.7)

    def world(ego_rect):
        return np.array([(3.0 + c * x - s * y, -2.0 + s * x + c * y) for x, y in ego_rect])
    ego = [rect(9.0, 0, 2, 2, 0.2), rect(12.0, 1.0, 2, 2, 0.4), rect(7.6, 7.6, 2, 2, 0.3), rect(-4, 3, 3, 1, 1.1)]
    dmin = [min(math.hypot(*p) for p in r) for r in ego]
    assert dmin[1] > 10.2 and dmin[2] < 9.9        # ring 1 is out of range, ring 2 only by its corner
    hb = O.tables()['hull_base']
    out = O.lidar_observation(pose, np.stack([world(r) for r in ego]), [4, 4, 4, 4])
    want = raycast([ego[0], ego[2], ego[3]])
    assert np.abs(out + hb - want).max() < 1e-9
    assert (want < 10).sum() >= 15
    out2 = O.lidar_observation(pose, np.stack([world(ego[1])]), [4])
    assert np.allclose(out2 + hb, 10.0)
    # a ring wholly containing the vehicle is still seen from inside
    big = rect(0, 0, 12, 12, 0.2)
    out3 = O.lidar_observation(pose, np.stack([world(big)]), [4])
    assert np.abs(out3 + hb - raycast([big])).max() < 1e-9
    assert ((out3 + hb) < 10.0).all()
    # triangle stored as (v0,v1,v2) with nvert 3
    tri = np.array([(5.0, -1.0), (6.0, 2.5), (8.0, 0.3)])
    w = np.vstack([world(tri), world(tri)[2:3]])
    out4 = O.lidar_observation(pose, w[None], [3])
    assert np.abs(out4 + hb - raycast([tri])).max() < 1e-9
