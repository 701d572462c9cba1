"""Host-side scene sources (reset-time, not on the per-step hot path).

* `DlpScenePool`  -- the 248 Dragon-Lake-Parking cases of the reference's data/dlp.data, decoded once
  to the shapely-free `data/dlp_scenes.npz` (tests/golden/make_golden.py); `sample()` follows
  `ParkingMapDLP.reset` (src/env/parking_map_dlp.py:38-86): random start candidate with jitter,
  map bbox = floor/ceil(min/max(start,dest) -/+ 20), obstacle cull by that bbox, 50 % flips.
* `Scene` / `pack_scenes` -- the flat arrays `hope_env_set_scenes` takes.

Every obstacle is a 3- or 4-vertex ring held in a 4-vertex slot (triangles repeat their last vertex).
"""
import math
import os
from dataclasses import dataclass

import numpy as np

from . import tables as T

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_DLP = os.path.join(_ROOT, 'data', 'dlp_scenes.npz')


@dataclass
class Scene:
    start: np.ndarray      # (3,) x, y, heading
    dest: np.ndarray       # (3,)
    bbox: np.ndarray       # (4,) xmin, xmax, ymin, ymax
    verts: np.ndarray      # (n, 4, 2)
    nvert: np.ndarray      # (n,) 3 or 4
    level: str = 'dlp'
    case_id: int = -1

    @property
    def n_obst(self):
        return len(self.verts)

    @property
    def map_level(self):
        """`map.map_level`: the generator's level for Normal / Complex / Extrem scenes; for DLP scenes the label
        ParkingMapDLP.reset computes with get_map_level (parking_map_dlp.py:84)"""
        if self.level != 'dlp':
            return self.level
        from .map_level import get_map_level
        return get_map_level(self.start, self.dest, [v[:int(n)] for v, n in zip(self.verts, self.nvert)])


def create_box(pose):
    """State.create_box (vehicle.py:32-36): 4 hull corners of a pose."""
    c, s = math.cos(pose[2]), math.sin(pose[2])
    return np.array([(c * x + (-s) * y + pose[0], s * x + c * y + pose[1]) for x, y in T.VEHICLE_BOX])


def flip_orientation(pose):
    """_flip_box_orientation (parking_map_dlp.py:117-123): same box, heading + pi."""
    cen = np.mean(create_box(pose), axis=0)
    return np.array([2 * cen[0] - pose[0], 2 * cen[1] - pose[1], pose[2] + np.pi])


def pad_ring(coords):
    """(3|4, 2) ring -> one 4-vertex slot."""
    c = np.asarray(coords, dtype=np.float64)
    if c.shape == (4, 2):
        return c, 4
    if c.shape == (3, 2):
        return np.vstack([c, c[2:3]]), 3
    raise ValueError('obstacle rings must have 3 or 4 vertices (reference scenes only contain those)')


def cull_obstacles(verts, nvert, bbox):
    """ParkingMapDLP.filter_obstacles (parking_map_dlp.py:88-101)."""
    xmin, xmax, ymin, ymax = bbox
    x, y = verts[:, :, 0], verts[:, :, 1]
    out = (x.max(1) <= xmin) | (x.min(1) >= xmax) | (y.max(1) <= ymin) | (y.min(1) >= ymax)
    return np.nonzero(~out)[0]


class DlpScenePool:
    def __init__(self, path=DEFAULT_DLP):
        d = np.load(path)
        self.set_verts, self.set_nvert, self.set_off = d['set_verts'], d['set_nvert'].astype(np.int32), d['set_off']
        self.case_set, self.dest, self.starts, self.start_off = d['case_set'], d['dest'], d['starts'], d['start_off']
        self.multi_start = True                     # data/dlp.data holds a list of start candidates per case

    def __len__(self):
        return len(self.case_set)

    def obstacles(self, case):
        s = int(self.case_set[case % len(self)])
        a, b = int(self.set_off[s]), int(self.set_off[s + 1])
        return self.set_verts[a:b], self.set_nvert[a:b]

    def candidates(self, case):
        a, b = self.start_off[case], self.start_off[case + 1]
        return self.starts[a:b]

    def sample(self, case=None, rng=None, jitter=True, flips=True):
        rng = np.random.default_rng() if rng is None else rng
        case = int(rng.integers(len(self))) if case is None else int(case) % len(self)
        cand = self.candidates(case)
        start = cand[int(rng.integers(len(cand)))].copy()
        if jitter and getattr(self, 'multi_start', True):          # parking_map_dlp.py:60-63: only multi-start data is jittered
            start = start + rng.standard_normal(3) * np.array([0.05, 0.05, 0.02])
        dest = self.dest[case].copy()
        bbox = np.array([np.floor(min(start[0], dest[0]) - 20), np.ceil(max(start[0], dest[0]) + 20),
                         np.floor(min(start[1], dest[1]) - 20), np.ceil(max(start[1], dest[1]) + 20)])
        v, nv = self.obstacles(case)
        keep = cull_obstacles(v, nv, bbox)
        if flips and rng.random() > 0.5:
            dest = flip_orientation(dest)
        if flips and rng.random() > 0.5:
            start = flip_orientation(start)
        return Scene(start=start, dest=dest, bbox=bbox, verts=v[keep].copy(), nvert=nv[keep].copy(), level='dlp',
                     case_id=case)


def pack_scenes(scenes, max_obst):
    """list[Scene] -> (start[n,3], dest[n,3], bbox[n,4], verts[n,max_obst,4,2], n_obst[n], nvert[n,max_obst])."""
    n = len(scenes)
    start, dest, bbox = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 4))
    verts = np.zeros((n, max_obst, 4, 2))
    nvert = np.full((n, max_obst), 4, np.int32)
    nob = np.zeros(n, np.int32)
    for k, s in enumerate(scenes):
        if s.n_obst > max_obst:
            raise ValueError(f'scene has {s.n_obst} obstacles > max_obstacles={max_obst}')
        start[k], dest[k], bbox[k] = s.start, s.dest, s.bbox
        verts[k, :s.n_obst] = s.verts
        nvert[k, :s.n_obst] = s.nvert
        nob[k] = s.n_obst
    return start, dest, bbox, verts, nob, nvert


# =================================================================================================
# Normal / Complex / Extrem scene generator (host side)
#
# Produces scenes from the same distribution as the reference's rejection samplers
# `generate_bay_parking_case` / `generate_parallel_parking_case` (src/env/parking_map_normal.py:40-457)
# and `ParkingMapNormal.reset` (:474-494).  The reference delegates `distance` / `intersects` to
# shapely; here they are small numpy routines.  A numpy Generator replaces the global RNG.
# =================================================================================================
LENGTH = T.WHEEL_BASE + T.FRONT_HANG + T.REAR_HANG
# configs.py:43-70
_MIN_LOT_LEN = {'Extrem': LENGTH + 0.6, 'Complex': LENGTH + 0.9, 'Normal': LENGTH * 1.25}
_MAX_LOT_LEN = {'Extrem': LENGTH + 0.9, 'Complex': LENGTH * 1.25, 'Normal': LENGTH * 1.25 + 0.5}
_MIN_LOT_WID = {'Complex': T.WIDTH + 0.4, 'Normal': T.WIDTH + 0.85}
_MAX_LOT_WID = {'Complex': T.WIDTH + 0.85, 'Normal': T.WIDTH + 1.2}
_PARA_WALL = {'Extrem': 3.5, 'Complex': 4.0, 'Normal': 4.5}
_BAY_WALL = {'Complex': 6.0, 'Normal': 7.0}
_N_OBST = {'Extrem': 8, 'Complex': 5, 'Normal': 3}
_GAP = 0.1                     # MIN_DIST_TO_OBST
_P_WALL, _N_EXTRA, _P_EXTRA = 0.5, 3, 0.7     # parking_map_normal.py:20-22


def _segs(ring):
    r = np.asarray(ring, dtype=np.float64)
    return r, np.roll(r, -1, axis=0)


def _cross(ax, ay, bx, by):
    return ax * by - ay * bx


def rings_intersect(a, b):
    """LinearRing.intersects(LinearRing): any boundary segment pair shares a point."""
    p1, p2 = _segs(a)
    q1, q2 = _segs(b)
    P1, P2 = p1[:, None, :], p2[:, None, :]
    Q1, Q2 = q1[None, :, :], q2[None, :, :]
    d1 = _cross(P2[..., 0] - P1[..., 0], P2[..., 1] - P1[..., 1], Q1[..., 0] - P1[..., 0], Q1[..., 1] - P1[..., 1])
    d2 = _cross(P2[..., 0] - P1[..., 0], P2[..., 1] - P1[..., 1], Q2[..., 0] - P1[..., 0], Q2[..., 1] - P1[..., 1])
    d3 = _cross(Q2[..., 0] - Q1[..., 0], Q2[..., 1] - Q1[..., 1], P1[..., 0] - Q1[..., 0], P1[..., 1] - Q1[..., 1])
    d4 = _cross(Q2[..., 0] - Q1[..., 0], Q2[..., 1] - Q1[..., 1], P2[..., 0] - Q1[..., 0], P2[..., 1] - Q1[..., 1])
    box = (np.minimum(P1, P2) <= np.maximum(Q1, Q2)).all(-1) & (np.minimum(Q1, Q2) <= np.maximum(P1, P2)).all(-1)
    opp = (np.sign(d1) * np.sign(d2) <= 0) & (np.sign(d3) * np.sign(d4) <= 0)
    return bool((box & opp).any())


def _pt_seg(p, a, b):
    ab = b - a
    den = (ab * ab).sum(-1)
    t = np.where(den > 0, ((p - a) * ab).sum(-1) / np.where(den > 0, den, 1), 0.0)
    t = np.clip(t, 0, 1)
    c = a + t[..., None] * ab
    return np.sqrt(((p - c) ** 2).sum(-1))


def rings_distance(a, b):
    """LinearRing.distance(LinearRing): 0 when they meet, else the closest vertex-to-edge gap."""
    if rings_intersect(a, b):
        return 0.0
    p1, p2 = _segs(a)
    q1, q2 = _segs(b)
    d_ab = _pt_seg(p1[:, None, :], q1[None], q2[None]).min()
    d_ba = _pt_seg(q1[:, None, :], p1[None], p2[None]).min()
    return float(min(d_ab, d_ba))


def _clipn(rng, mean, std, lo, hi):
    return float(np.clip(rng.standard_normal() * std + mean, lo, hi))


def _uni(rng, lo, hi):
    return float(rng.random() * (hi - lo) + lo)


def _polar(rng, origin, a0, a1, r0, r1):
    ang = _clipn(rng, (a0 + a1) / 2, (a1 - a0) / 4, a0, a1)
    rad = _clipn(rng, (r0 + r1) / 2, (r1 - r0) / 4, r0, r1)
    return (origin[0] + math.cos(ang) * rad, origin[1] + math.sin(ang) * rad)


def _case(level, bay, rng):
    """one rejection-sampling attempt; returns (start, dest, rings) or None."""
    half = 15.0 if bay else 18.0
    if bay:
        space_hi, space_lo = _MAX_LOT_WID[level] - T.WIDTH, _MIN_LOT_WID[level] - T.WIDTH
        wall, yaw0, pitch = _BAY_WALL[level], math.pi / 2, T.WIDTH
        yaw_lo, yaw_hi = math.pi * 5 / 12, math.pi * 7 / 12
        low_pair = (0, 3)                    # rear-right, rear-left corners touch the back wall
        n_extra = _N_EXTRA
    else:
        space_hi, space_lo = _MAX_LOT_LEN[level] - LENGTH, _MIN_LOT_LEN[level] - LENGTH
        wall, yaw0, pitch = _PARA_WALL[level], 0.0, LENGTH
        yaw_lo, yaw_hi = -math.pi / 12, math.pi / 12
        low_pair = (0, 1)                    # rear-right, front-right
        n_extra = _N_EXTRA - 1
    back = np.array([(half, 0.0), (half, -1.0), (-half, -1.0), (-half, 0.0)])

    def slot_pose(x):
        yaw = _clipn(rng, yaw0, math.pi / 36, yaw_lo, yaw_hi)
        b = create_box((x, 0.0, yaw))
        y_min = -min(b[low_pair[0], 1], b[low_pair[1], 1]) + _GAP
        y = _clipn(rng, y_min + 0.4, 0.2, y_min, y_min + 0.8)
        return np.array([x, y, yaw])

    dest = slot_pose(0.0)
    rb, rf, lf, lb = create_box(dest)
    dest_ring = np.array([rb, rf, lf, lb])
    ok = True
    extras = []

    def side(sign, near_a, near_b, d_lo, d_hi):
        """obstacle next to the slot on side `sign` (-1 left, +1 right): a wall-like quad or a parked car
        followed by further parked cars (kept with probability .7)."""
        if rng.random() < _P_WALL:
            a0, a1 = (math.pi * 11 / 12, math.pi * 13 / 12) if sign < 0 else (-math.pi / 12, math.pi / 12)
            pa = _polar(rng, near_a, a0, a1, d_lo, d_hi)
            pb = _polar(rng, near_b, a0, a1, d_lo, d_hi)
            if sign < 0:
                return np.array([pa, pb, (-half, 0.0), (-half, pa[1])])
            return np.array([(half, pa[1]), (half, 0.0), pb, pa])
        x = sign * (pitch + _uni(rng, d_lo, d_hi))
        pose = slot_pose(x)
        first = create_box(pose)
        for _ in range(n_extra):
            x += sign * (pitch + _GAP + _uni(rng, d_lo, d_hi))
            y = pose[1] + _clipn(rng, 0, 0.05, -0.1, 0.1)
            pose = np.array([x, y, _clipn(rng, yaw0, math.pi / 36, yaw_lo, yaw_hi)])
            ring = create_box(pose)
            if rng.random() < _P_EXTRA:
                extras.append(ring)
        return first

    if bay:
        left = side(-1, lf, lb, space_hi / 5 * 1, space_hi / 5 * 4)
    else:
        left = side(-1, lb, rb, space_lo / 5 * 1, space_hi / 5 * 4)
    gap_l = rings_distance(dest_ring, left)
    d_lo = max(space_lo - gap_l, 0) + _GAP
    d_hi = max(space_hi - gap_l, 0) + _GAP
    right = side(+1, rf, rb, d_lo, d_hi) if bay else side(+1, lf, rf, d_lo, d_hi)
    gap_r = rings_distance(dest_ring, right)
    if gap_r + gap_l < space_lo or gap_r + gap_l > space_hi or gap_l < _GAP or gap_r < _GAP:
        ok = False
    rings = [back, left, right] + extras
    if any(rings_intersect(r, dest_ring) for r in rings):
        ok = False

    top = max(float(np.max(r[:, 1])) for r in rings) + _GAP
    far = []
    if rng.random() < 0.2:                                   # only a thin wall across the aisle
        y0 = wall + top + _GAP
        far = [np.array([(-half, y0), (half, y0), (half, y0 + 0.1), (-half, y0 + 0.1)])]
    else:
        zone = np.array([(-half, wall + top), (half, wall + top), (half, wall + top + 8), (-half, wall + top + 8)])
        for _ in range(_N_OBST[level]):
            pose = (_uni(rng, -half + 2, half - 2), _uni(rng, wall + top + 2, wall + top + 6), rng.random() * math.pi * 2)
            ring = create_box(pose) + 0.5 * rng.random((4, 2))
            if rings_intersect(ring, zone) or any(rings_intersect(ring, o) for o in far):
                continue
            far.append(ring)
    rings = rings + far

    while True:                                              # start pose in the aisle, clear of everything
        sx = _uni(rng, -half / 2, half / 2)
        sy = _uni(rng, top + 1, wall + top - 1)
        syaw = _clipn(rng, 0, math.pi / 6, -math.pi / 2, math.pi / 2)
        if rng.random() < 0.5:
            syaw += math.pi
        sbox = create_box((sx, sy, syaw))
        if not any(rings_intersect(r, sbox) for r in rings) and not rings_intersect(dest_ring, sbox):
            break
    start = np.array([sx, sy, syaw])
    if not bay and math.cos(syaw) < 0:                       # parallel: face the slot the way the car arrives
        dest = flip_orientation(dest)
    if not ok:
        return None
    return start, dest, rings


def generate_scene(level, rng=None, case_id=None):
    """ParkingMapNormal.reset (parking_map_normal.py:474-494): bay or parallel case, bbox = +-10 m."""
    rng = np.random.default_rng() if rng is None else rng
    if level not in ('Normal', 'Complex', 'Extrem'):
        raise ValueError(level)
    bay = (case_id == 0 or (rng.random() > 0.5 and case_id != 1)) and level in ('Normal', 'Complex')
    while True:
        got = _case(level, bay, rng)
        if got is not None:
            break
    start, dest, rings = got
    bbox = np.array([np.floor(min(start[0], dest[0]) - 10), np.ceil(max(start[0], dest[0]) + 10),
                     np.floor(min(start[1], dest[1]) - 10), np.ceil(max(start[1], dest[1]) + 10)])
    verts = np.stack(rings)
    return Scene(start=start, dest=dest, bbox=bbox, verts=verts, nvert=np.full(len(rings), 4, np.int32), level=level,
                 case_id=0 if bay else 1)


class SceneSource:
    """uniform mix over difficulty levels, as the reference's SceneChoose does before its curriculum kicks
    in (src/train/train_HOPE_sac.py:23-29)."""

    def __init__(self, levels=('Normal', 'Complex', 'Extrem', 'dlp'), seed=42, dlp_path=DEFAULT_DLP):
        self.levels = tuple(levels)
        self.rng = np.random.default_rng(seed)
        self.pool = DlpScenePool(dlp_path) if 'dlp' in self.levels else None

    def draw(self, level=None):
        level = self.levels[int(self.rng.integers(len(self.levels)))] if level is None else level
        if level == 'dlp':
            return self.pool.sample(rng=self.rng)
        return generate_scene(level, self.rng)
