"""f-2: hope_amd.map_level.get_map_level against labels produced by the reference's own get_map_level source
(src/env/map_level.py:27-112, run unmodified by tests/golden/make_golden_r2.py `maplevel`)."""
import numpy as np

from hope_amd import map_level as M
from hope_amd.scenes import DlpScenePool, SceneSource

NAMES = ['Normal', 'Complex', 'Extrem']


def test_map_level_matches_reference_control_flow(gold):
    g = gold('map_level.npz')
    pool = DlpScenePool()
    rng = np.random.default_rng(17)
    k = 0
    for case in range(len(pool)):
        for _ in range(2):
            sc = pool.sample(case=case, rng=rng)
            assert np.array_equal(sc.start, g['dlp_start'][k]) and np.array_equal(sc.dest, g['dlp_dest'][k])
            assert sc.map_level == NAMES[int(g['dlp_label'][k])], (case, k)
            k += 1
    src = SceneSource(levels=('Normal', 'Complex', 'Extrem'), seed=123)
    conf = np.zeros((3, 3), int)
    for j in range(len(g['gen_label'])):
        sc = src.draw()
        lab = M.get_map_level(sc.start, sc.dest, [v[:int(n)] for v, n in zip(sc.verts, sc.nvert)])
        assert lab == NAMES[int(g['gen_label'][j])], j
        conf[NAMES.index(sc.level), NAMES.index(lab)] += 1
    # the classifier recovers the generator's level for most generated scenes (Extrem = narrow parallel slot)
    assert conf[2, 2] > 0.9 * conf[2].sum() and conf[0, 0] > 0.8 * conf[0].sum()


def test_geometry_helpers():
    sq = [(0, 0), (2, 0), (2, 2), (0, 2)]
    assert M.point_ring_distance((1, 1), sq) == 1.0 and M.point_ring_distance((3, 1), sq) == 1.0      # boundary, not area
    assert M.ring_ring_distance(sq, [(3, 0), (4, 0), (4, 1), (3, 1)]) == 1.0
    assert M.ring_ring_distance(sq, [(1, 1), (3, 1), (3, 3), (1, 3)]) == 0.0
    r = M.min_area_rectangle([(0, 0), (1, 1), (2, 0), (1, -1), (1, 0)])                                  # a diamond: side sqrt(2)
    a = np.array(r)
    assert abs(np.linalg.norm(a[1] - a[0]) * np.linalg.norm(a[2] - a[1]) - 2.0) < 1e-12
    assert M.polygon_meets_ring(sq, [(0.5, 0.5), (1.5, 0.5), (1.5, 1.5)])                               # ring inside the polygon
    assert not M.polygon_meets_ring([(0.5, 0.5), (1.5, 0.5), (1.5, 1.5), (0.5, 1.5)], [(-5, -5), (5, -5), (5, 5), (-5, 5)])
    assert M.get_map_level((0, 10, 0), (0, 0, 0), []) == 'Normal'
