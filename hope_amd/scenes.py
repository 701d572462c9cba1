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
        if jitter:
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
