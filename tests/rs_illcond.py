"""Classifier for Reeds-Shepp search disagreements between two float64 implementations with different libm's.

The reference's `find_rs_path` is ill-conditioned in two structural ways (not bugs of either implementation):

 (T) equal-length twins: e.g. the CCC words LRL [t,u,v] and RLR from the backwards family describe mirror/time
     reversed manoeuvres of mathematically EQUAL length; their float lengths differ in the last 1-2 ulps, so
     which one heapdict pops first is rounding noise (`path.L` from atan2/asin/hypot of the libm in use);
 (A) axis-aligned luck: `is_traj_valid` accepts a hit only if the line-line intersection point lies inside BOTH
     edge boxes with NO tolerance (car_parking_base.py:518-526); for an exactly axis-aligned obstacle edge
     (Normal/Complex/Extrem walls) that needs raw_y == y_wall bit-for-bit, a coin flip decided by the last bit
     of cos/sin.  A path that crosses only such edges is "valid" or "invalid" by luck.

`allowed_results` enumerates every outcome reachable under arbitrary luck/tie-breaking; a disagreement is
ill-conditioned (excused) iff BOTH results are in that set.
"""
import numpy as np

from hope_amd import tables as T
from hope_amd.rs_path import MAXC
from oracle import oracle as O

_CAR = np.vstack([T.VEHICLE_BOX, T.VEHICLE_BOX[:1]])


def _edges(verts, nvert):
    for v, n in zip(verts, nvert):
        for j in range(int(n)):
            yield v[j], v[(j + 1) % int(n)]


def path_class(traj, verts, nvert, bbox):
    """'invalid' (hit on a non-axis-aligned edge or out of the map box: every implementation agrees),
    'fragile' (geometrically crosses exactly axis-aligned edges only) or 'valid'."""
    if (traj[:, 0] < bbox[0]).any() or (traj[:, 0] > bbox[1]).any() or (traj[:, 1] < bbox[2]).any() or (traj[:, 1] > bbox[3]).any():
        return 'invalid'
    fragile = False
    edges = list(_edges(verts, nvert))
    for x, y, th in traj:
        c, s = np.cos(th), np.sin(th)
        hx = c * _CAR[:, 0] - s * _CAR[:, 1] + x
        hy = s * _CAR[:, 0] + c * _CAR[:, 1] + y
        hmin, hmax = (hx.min(), hy.min()), (hx.max(), hy.max())
        for p, q in edges:
            if min(p[0], q[0]) > hmax[0] or max(p[0], q[0]) < hmin[0] or min(p[1], q[1]) > hmax[1] or max(p[1], q[1]) < hmin[1]:
                continue
            axis = p[0] == q[0] or p[1] == q[1]
            for k in range(4):
                if O.segments_intersect((hx[k], hy[k]), (hx[k + 1], hy[k + 1]), p, q):
                    if not axis:
                        return 'invalid'
                    fragile = True
    return 'fragile' if fragile else 'valid'


def allowed_results(pose, dest, verts, nvert, bbox):
    """set of words (tuples of type codes, () = no path) reachable under arbitrary tie order / axis-aligned luck."""
    r = O.rs_all_paths(pose, dest, MAXC)
    n = r['n']
    if n == 0:
        return {()}
    order = sorted(range(n), key=lambda i: r['L'][i])
    lmin = r['L'][order[0]]
    words = {i: tuple(int(t) for t in r['ctypes'][i][:r['nseg'][i]]) for i in range(n)}
    cls = {}
    allowed = set()

    def walk(seq):
        """one admissible pop order: returns the outcomes reachable by luck along it"""
        out = set()
        for idx, i in enumerate(seq, 1):
            if r['L'][i] > 1.6 * lmin * (1 + 1e-12) and idx > 2:
                break
            if i not in cls:
                cls[i] = path_class(O.rs_path_samples(pose, dest, MAXC, int(i)), verts, nvert, bbox)
            if cls[i] == 'valid':
                out.add(words[i])
                return out
            if cls[i] == 'fragile':
                out.add(words[i])          # may be accepted by luck; otherwise the walk continues
        out.add(())
        return out
    # tie groups: paths whose lengths agree to 1e-9 relative may pop in either order
    seqs = [order]
    for a in range(len(order) - 1):
        i, j = order[a], order[a + 1]
        if abs(r['L'][i] - r['L'][j]) <= 1e-9 * max(1.0, r['L'][i]):
            sw = list(order)
            sw[a], sw[a + 1] = sw[a + 1], sw[a]
            seqs.append(sw)
    for sq in seqs:
        allowed |= walk(sq)
    return allowed
