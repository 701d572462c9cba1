"""Reads a reference-format DLP pickle (data/dlp.data: list of (start candidates, dest, [LinearRing]))
without shapely: shapely-1.x pickles a LinearRing as the WKB bytes of a LineString."""
import pickle
import struct

import numpy as np

from .scenes import DlpScenePool


class _Ring:
    def __setstate__(self, state):
        b = state
        if b[0] != 1:
            raise ValueError('big-endian WKB is not supported')
        _gtype, n = struct.unpack_from('<II', b, 1)
        self.coords = np.frombuffer(b, '<f8', 2 * n, offset=9).reshape(n, 2).copy()


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('shapely') and name in ('LinearRing', 'LineString'):
            return _Ring
        return super().find_class(module, name)


def load_cases(path):
    with open(path, 'rb') as f:
        return _Unpickler(f).load()


def pool_from_pickle(path):
    """build an in-memory DlpScenePool from a dlp.data-style pickle."""
    cases = load_cases(path)
    # ParkingMapDLP.reset (:43-46): start candidates given as a LIST per case = multi-start data (one is drawn per episode
    # and jittered, :60-63); a single start pose per case is used as it is
    multi_start = isinstance(cases[0][0], list)
    verts, nvert, set_off, case_set, dest, starts, start_off = [], [], [0], [], [], [], [0]
    for ci, case in enumerate(cases):
        cand, dst, rings = case[:3]
        if isinstance(cand, tuple):
            cand = [cand]
        for r in rings:
            c = np.asarray(r.coords)[:-1]
            v = np.zeros((4, 2))
            v[:len(c)] = c
            if len(c) == 3:
                v[3] = c[2]
            elif len(c) != 4:
                raise ValueError('obstacle rings must have 3 or 4 vertices')
            verts.append(v)
            nvert.append(len(c))
        set_off.append(len(verts))
        case_set.append(ci)
        dest.append([float(v) for v in dst])
        starts.extend([[float(v) for v in s] for s in cand])
        start_off.append(len(starts))
    pool = DlpScenePool.__new__(DlpScenePool)
    pool.set_verts, pool.set_nvert = np.array(verts), np.array(nvert, np.int32)
    pool.set_off, pool.case_set = np.array(set_off), np.array(case_set)
    pool.dest, pool.starts, pool.start_off = np.array(dest), np.array(starts), np.array(start_off)
    pool.multi_start = bool(multi_start)
    return pool
