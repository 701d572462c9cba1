"""Exact-arithmetic specification check of the oracle's GEOS slice (the ring cull of a-8).

shapely / GEOS cannot be obtained in this image, so these functions have no reference-generated vector ("parity
unpinned").  What CAN be checked is GEOS's *contract*: `LinearRing.intersects` is decided by
RobustLineIntersector, whose only inexact ingredient -- `Orientation::index` -- is defined to return the EXACT sign of
the orientation determinant (CGAlgorithmsDD falls back to extended precision).  So "do two closed segments with these
double coordinates share a point" has one right answer, computable with `fractions.Fraction` on the exact binary
values of the inputs.  This file compares the oracle with that answer on > 10^5 adversarial cases (collinear overlaps,
touching endpoints, T-junctions, 1-ulp perturbations of all of them, DLP-sized coordinate offsets), and bounds the
continuous functions (convex-quad overlap area, point-segment distance) against exact rational evaluation.
"""
import math
from fractions import Fraction as Fr

import numpy as np
import pytest

from oracle import oracle as O

OFFSETS = [(0.0, 0.0), (147.25, -36.5), (-83.0, 61.125), (1e4, 1e4)]


# ---- exact predicates on the binary values of the inputs -------------------------------------------------------------
def F(p):
    return (Fr(float(p[0])), Fr(float(p[1])))


def orient_exact(a, b, c):
    d = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    return (d > 0) - (d < 0)


def on_seg(a, b, c):
    """c collinear with ab: inside the closed segment?"""
    return min(a[0], b[0]) <= c[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= c[1] <= max(a[1], b[1])


def seg_intersect_exact(p1, p2, q1, q2):
    p1, p2, q1, q2 = F(p1), F(p2), F(q1), F(q2)
    o1, o2 = orient_exact(p1, p2, q1), orient_exact(p1, p2, q2)
    o3, o4 = orient_exact(q1, q2, p1), orient_exact(q1, q2, p2)
    if o1 * o2 < 0 and o3 * o4 < 0:
        return True
    return ((o1 == 0 and on_seg(p1, p2, q1)) or (o2 == 0 and on_seg(p1, p2, q2)) or
            (o3 == 0 and on_seg(q1, q2, p1)) or (o4 == 0 and on_seg(q1, q2, p2)))


def ulp_nudge(rng, pts):
    """move one random coordinate of one random point by +-1..2 ulp"""
    pts = [list(p) for p in pts]
    i, j = int(rng.integers(len(pts))), int(rng.integers(2))
    for _ in range(int(rng.integers(1, 3))):
        pts[i][j] = float(np.nextafter(pts[i][j], math.inf if rng.random() < 0.5 else -math.inf))
    return [tuple(p) for p in pts]


def dyadic(rng, lo, hi, bits=10):
    """a double that is an exact multiple of 2^-bits (sums / differences of a few of them are exact)"""
    return float(rng.integers(int(lo * 2 ** bits), int(hi * 2 ** bits))) / 2 ** bits


def adversarial_segments(rng, n):
    """yields (p1, p2, q1, q2) tuples"""
    for k in range(n):
        ox, oy = OFFSETS[k % len(OFFSETS)]
        kind = k % 8
        if kind == 0:                                       # generic random pair in a 6 m window
            pts = [(ox + rng.uniform(-3, 3), oy + rng.uniform(-3, 3)) for _ in range(4)]
        elif kind in (1, 2):                                # exactly collinear (dyadic direction x small integers)
            bx, by, dx, dy = dyadic(rng, -3, 3), dyadic(rng, -3, 3), dyadic(rng, -1, 1), dyadic(rng, -1, 1)
            ts = rng.integers(-6, 7, 4)
            pts = [(ox + bx + int(t) * dx, oy + by + int(t) * dy) for t in ts]
            if ox == 1e4:
                pts = [(bx + int(t) * dx, by + int(t) * dy) for t in ts]      # keep the sums exact
        elif kind == 3:                                     # shared endpoint
            pts = [(ox + rng.uniform(-3, 3), oy + rng.uniform(-3, 3)) for _ in range(3)]
            pts = [pts[0], pts[1], pts[1], pts[2]] if rng.random() < 0.5 else [pts[0], pts[1], pts[2], pts[0]]
        elif kind == 4:                                     # T-junction: q1 exactly the midpoint of a dyadic p
            a = (dyadic(rng, -3, 3, 8), dyadic(rng, -3, 3, 8))
            d = (dyadic(rng, -2, 2, 8), dyadic(rng, -2, 2, 8))
            p1, p2 = a, (a[0] + 2 * d[0], a[1] + 2 * d[1])
            mid = (a[0] + d[0], a[1] + d[1])
            pts = [p1, p2, mid, (mid[0] + rng.uniform(-2, 2), mid[1] + rng.uniform(-2, 2))]
        elif kind == 5:                                     # axis-aligned walls (the DLP lots are full of them)
            x = dyadic(rng, -3, 3)
            pts = [(ox + x, oy + rng.uniform(-3, 3)), (ox + x, oy + rng.uniform(-3, 3)),
                   (ox + rng.uniform(-3, 3), oy + dyadic(rng, -3, 3)), (ox + rng.uniform(-3, 3), oy + dyadic(rng, -3, 3))]
            if rng.random() < 0.5:
                pts[2] = (ox + x, pts[2][1])                # endpoint exactly on the wall's line
        elif kind == 6:                                     # nearly parallel, nearly touching
            a = np.array([ox + rng.uniform(-3, 3), oy + rng.uniform(-3, 3)])
            d = rng.normal(size=2)
            e = d * (1 + 1e-13 * rng.normal()) + 1e-13 * rng.normal(size=2)
            pts = [tuple(a), tuple(a + d), tuple(a + 0.5 * d + 1e-14 * rng.normal(size=2)), tuple(a + 0.5 * d + e)]
        else:                                               # a vertex of one segment 1e-16-close to the other's line
            a = np.array([ox + rng.uniform(-3, 3), oy + rng.uniform(-3, 3)])
            d = rng.normal(size=2)
            t = rng.uniform(-0.2, 1.2)
            pts = [tuple(a), tuple(a + d), tuple(a + t * d), tuple(a + t * d + rng.normal(size=2))]
        yield tuple(pts)
        if kind in (1, 2, 3, 4, 5, 7):
            yield tuple(ulp_nudge(rng, pts))


@pytest.mark.parametrize('seed', [0, 1])
def test_segments_intersect_is_the_exact_predicate(seed):
    """a-3: 2 x ~87k adversarial pairs, oracle == exact rational answer on every one."""
    rng = np.random.default_rng(seed)
    n = hits = 0
    kinds = {True: 0, False: 0}
    for p1, p2, q1, q2 in adversarial_segments(rng, 50000):
        want = seg_intersect_exact(p1, p2, q1, q2)
        got = O.segments_intersect(p1, p2, q1, q2)
        assert got == want, (p1, p2, q1, q2, got, want)
        kinds[want] += 1
        n += 1
    assert n > 80000 and min(kinds.values()) > 15000           # both answers well represented


def test_orientation_is_the_exact_sign():
    rng = np.random.default_rng(5)
    zero = 0
    for k in range(40000):
        ox, oy = OFFSETS[k % len(OFFSETS)]
        a = np.array([ox + rng.uniform(-3, 3), oy + rng.uniform(-3, 3)])
        d = rng.normal(size=2)
        t = rng.uniform(-1, 2)
        c = a + t * d                                           # on the line up to rounding: the filter must give way
        if k % 3 == 0:                                          # exactly collinear dyadic triple
            a = np.array([dyadic(rng, -3, 3), dyadic(rng, -3, 3)])
            d = np.array([dyadic(rng, -1, 1), dyadic(rng, -1, 1)])
            c = a + 3 * d
        b = a + d
        want = orient_exact(F(a), F(b), F(c))
        assert O.orient(a, b, c) == want
        zero += want == 0
    assert zero > 10000


def test_ring_intersects_boundary_semantics_exact():
    """LinearRing.intersects(LinearRing): boundaries only -- any edge pair shares a point; containment is False."""
    rng = np.random.default_rng(2)
    from hope_amd.scenes import create_box
    n_true = n_false = 0
    for k in range(6000):
        ox, oy = OFFSETS[k % 3]
        hull = create_box((ox + rng.uniform(-2, 2), oy + rng.uniform(-2, 2), rng.uniform(-4, 4)))
        nv = 3 if k % 5 == 0 else 4
        if k % 4 == 0:                                          # big ring around the hull (containment) or far away
            c = np.array([ox, oy]) + (0 if k % 8 == 0 else 40)
            ring = c + np.array([[-12, -12], [12, -12], [12, 12], [-12, 12]], float)[:nv]
        elif k % 4 == 1:                                        # ring sharing exactly one hull corner
            ring = hull[int(rng.integers(4))] + np.vstack([[0, 0], rng.uniform(-3, 3, (nv - 1, 2))])
        else:
            ang = rng.uniform(0, 2 * np.pi)
            r = rng.uniform(0.3, 2.5)
            c = np.array([ox, oy]) + rng.uniform(-5, 5, 2)
            ring = c + r * np.column_stack([np.cos(ang + np.arange(nv) * 2 * np.pi / nv), np.sin(ang + np.arange(nv) * 2 * np.pi / nv)])
        want = any(seg_intersect_exact(hull[i], hull[(i + 1) % 4], ring[j], ring[(j + 1) % nv])
                   for i in range(4) for j in range(nv))
        assert O.ring_intersects(hull, ring) == want, k
        n_true += want
        n_false += not want
    assert n_true > 1500 and n_false > 1500


# ---- continuous functions: bounded against exact rational evaluation ----------------------------------------------------
def clip_area_exact(A, B):
    """Sutherland-Hodgman of convex CCW quad A by the half-planes of convex CCW quad B, shoelace, all in Fraction."""
    poly = [F(p) for p in A]
    Bq = [F(p) for p in B]
    for e in range(4):
        c1, c2 = Bq[e], Bq[(e + 1) % 4]
        ex, ey = c2[0] - c1[0], c2[1] - c1[1]
        out = []
        for i in range(len(poly)):
            s, t = poly[i], poly[(i + 1) % len(poly)]
            ds = ex * (s[1] - c1[1]) - ey * (s[0] - c1[0])
            dt = ex * (t[1] - c1[1]) - ey * (t[0] - c1[0])
            if ds >= 0:
                out.append(s)
            if (ds >= 0) != (dt >= 0):
                r = ds / (ds - dt)
                out.append((s[0] + r * (t[0] - s[0]), s[1] + r * (t[1] - s[1])))
        poly = out
        if not poly:
            return Fr(0)
    a = sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly)))
    return abs(a) / 2


def test_quad_overlap_area_against_exact_rational():
    """a-4 / box-union term of a-12: |hull ∩ dest| for two car-sized rectangles.  The oracle's float64 clip must agree
    with the exact area to 1e-12 m^2 (area ~ 9.1 m^2), and the `> 0.95` arrival decision may only differ from the exact
    one inside that band."""
    from hope_amd.scenes import create_box
    rng = np.random.default_rng(3)
    worst, close = 0.0, 0
    for k in range(3000):
        ox, oy = OFFSETS[k % 3]
        dest = (ox + rng.uniform(-2, 2), oy + rng.uniform(-2, 2), rng.uniform(-4, 4))
        if k % 3 == 0:                                           # near-arrival poses: ratios around 0.95
            ego = (dest[0] + rng.normal() * 0.06, dest[1] + rng.normal() * 0.06, dest[2] + rng.normal() * 0.01)
        elif k % 3 == 1:
            ego = (dest[0] + rng.uniform(-5, 5), dest[1] + rng.uniform(-3, 3), dest[2] + rng.uniform(-1, 1))
        else:                                                    # same heading, pure translation (parallel edges)
            ego = (dest[0] + rng.uniform(-5, 5), dest[1] + rng.uniform(-2, 2), dest[2])
        A, B = create_box(ego), create_box(dest)
        exact = clip_area_exact(A, B)
        got = O.quad_intersection_area(A, B)
        err = abs(got - float(exact))
        worst = max(worst, err)
        area_b = O.quad_area(B)
        ratio_exact = float(exact) / area_b
        if abs(ratio_exact - 0.95) > 1e-12:
            assert (got / area_b > 0.95) == (ratio_exact > 0.95), k
        close += abs(ratio_exact - 0.95) < 0.02
    assert worst < 1e-12, worst
    assert close > 100
    # closed forms: identical boxes -> the box area; disjoint -> 0; half overlap along the axis
    B = create_box((3.0, -2.0, 0.0))
    assert abs(O.quad_intersection_area(B, B) - 4.69 * 1.94) < 1e-13 and abs(O.quad_area(B) - 4.69 * 1.94) < 1e-13
    assert O.quad_intersection_area(create_box((40.0, 0.0, 0.3)), B) == 0.0
    assert abs(O.quad_intersection_area(create_box((3.0 + 4.69 / 2, -2.0, 0.0)), B) - 4.69 * 1.94 / 2) < 1e-13


def test_point_segment_distance_against_exact_rational():
    """ring cull of a-8 (`LinearRing.distance(Point) < 10`): Distance::pointToSegment to a few ulp of the exact value,
    so the keep/drop decision can only differ from the exact one within ~1e-14 m of the 10 m boundary."""
    rng = np.random.default_rng(4)
    worst = 0.0
    for k in range(20000):
        ox, oy = OFFSETS[k % 3]
        p = (ox + rng.uniform(-1, 1), oy + rng.uniform(-1, 1))
        a = (p[0] + rng.uniform(-14, 14), p[1] + rng.uniform(-14, 14))
        b = (a[0] + rng.uniform(-6, 6), a[1] + rng.uniform(-6, 6)) if k % 10 else a          # degenerate segment too
        P, A, B = F(p), F(a), F(b)
        dx, dy = B[0] - A[0], B[1] - A[1]
        l2 = dx * dx + dy * dy
        if l2 == 0:
            d2 = (P[0] - A[0]) ** 2 + (P[1] - A[1]) ** 2
        else:
            r = ((P[0] - A[0]) * dx + (P[1] - A[1]) * dy) / l2
            r = min(max(r, Fr(0)), Fr(1))
            cx, cy = A[0] + r * dx, A[1] + r * dy
            d2 = (P[0] - cx) ** 2 + (P[1] - cy) ** 2
        want = math.sqrt(float(d2))
        got = O.pt_seg_dist(p, a, b)
        rel = abs(got - want) / max(want, 1e-300)
        worst = max(worst, rel)
        if abs(want - 10.0) > 1e-13:
            assert (got < 10.0) == (d2 < 100), k
    assert worst < 2e-14, worst          # perpendicular foot near an endpoint of a ~20 m lever arm: a handful of ulps
