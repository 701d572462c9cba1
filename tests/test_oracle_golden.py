"""The CPU oracle (oracle/hope_oracle.c) against vectors produced by the REFERENCE itself
(tests/golden/make_golden.py imported /root/reference in the build container).

Tolerances: the reference mixes Python `math` (libm) with numpy's SIMD sin/cos/tan; the oracle
is libm throughout, so continuous values may differ in the last ulps.  Discrete outputs
(path words, point counts, mask steps, validity flags) must be identical.
"""
import math

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(autouse=True, params=['hope_math', 'libm'])
def math_flavour(request):
    """every test runs on both oracle builds: `libm` (glibc sin/cos/atan2 -- what Python's math gave the reference
    run) and `hope_math` (the deterministic functions shared with the HIP kernels)."""
    O.use_libm(request.param == 'libm')
    yield request.param
    O.use_libm(False)


def case_obstacles(dlp, case, keep=None):
    s = int(dlp['case_set'][case])
    a, b = int(dlp['set_off'][s]), int(dlp['set_off'][s + 1])
    v = dlp['set_verts'][a:b]
    nv = dlp['set_nvert'][a:b].astype(np.int32)
    if keep is not None:
        v, nv = v[keep], nv[keep]
    return v, nv


def test_constants_and_tables(gold):
    g = gold('constants.npz')
    t = O.tables()
    assert np.array_equal(t['actions'], g['discrete_actions'])          # incl. np.arange drift
    assert t['actions'][10, 0] == 4.440892098500626e-16
    assert np.abs(t['boxes'] - g['vehicle_boxes']).max() < 1e-12
    assert np.abs(t['dist_star'][::7] - g['dist_star_every7']).max() < 1e-12
    assert np.abs(t['dist_star'][0] - g['dist_star_beam0']).max() < 1e-12
    assert np.abs(t['dist_star'][600] - g['dist_star_beam600']).max() < 1e-12
    assert abs(t['dist_star'].sum() - float(g['dist_star_sum'])) < 1e-6
    assert abs(t['dist_star'].max() - float(g['dist_star_max'])) < 1e-12
    assert (t['dist_star'] > 0).all()
    assert np.abs(t['beam_a'] - g['beam_a']).max() < 1e-15
    assert np.abs(t['beam_b'] - g['beam_b']).max() < 1e-15


def test_hull_base_closed_form():
    """ray/rectangle ranges: beam 0 -> 3.76, beam 30 -> 0.97, beam 60 -> 0.93 (SURVEY §8c KATs)."""
    hb = O.tables()['hull_base']
    assert hb[0] == 3.76
    assert abs(hb[30] - 0.97) < 1e-15
    assert abs(hb[60] - 0.93) < 1e-15
    assert abs(hb[90] - 0.97) < 1e-15
    th = np.arange(120) * np.pi / 120 * 2
    c, s = np.cos(th), np.sin(th)
    with np.errstate(divide='ignore'):
        tx = np.where(c > 0, 3.76 / c, np.where(c < 0, -0.93 / c, np.inf))
        ty = 0.97 / np.abs(s)
    assert np.abs(hb - np.minimum(tx, ty)).max() < 1e-14


def test_ksmodel(gold):
    g = gold('ksmodel.npz')
    for i in range(len(g['pose'])):
        p, ss = O.ks_step(g['pose'][i], g['action'][i])
        assert np.abs(p - g['out1'][i, :3]).max() < 1e-12
        assert ss[0] == g['out1'][i, 3] and ss[1] == g['out1'][i, 4]
        q = g['pose'][i].copy()
        for _ in range(10):
            q, _ss = O.ks_step(q, g['action'][i])
        assert np.abs(q - g['out10'][i]).max() < 1e-11
    # KATs quoted in SURVEY.md §8a-1
    q = np.array([1, 2, 0.3])
    for _ in range(10):
        q, _ = O.ks_step(q, [0.5, 2.0])
    assert np.abs(q - [1.92073748943096, 2.3861014311805238, 0.4951080320870657]).max() < 1e-13


def test_lidar_fast(gold):
    g, c = gold('lidar.npz'), gold('constants.npz')
    off = g['ring_off']
    own = O.tables()
    try:
        for table in ('reference', 'libm'):
            if table == 'reference':      # the beam sin/cos table captured from the reference run
                O.set_tables(beam_a=c['beam_a'], beam_b=c['beam_b'])
            else:
                O.set_tables(beam_a=own['beam_a'], beam_b=own['beam_b'])
            for k in range(len(off) - 1):
                v = g['ring_verts'][off[k]:off[k + 1]]
                # triangles were stored with the last vertex repeated
                nv = np.where((v[:, 3] == v[:, 2]).all(axis=1), 3, 4).astype(np.int32)
                out = O.lidar_fast(v, nv)
                if table == 'reference':
                    assert np.array_equal(out, g['lidar_raw'][k]), k      # bit-exact
                elif int(g['case'][k]) >= 0:
                    # DLP rings (no exactly axis-aligned edges): libm table agrees to rounding
                    assert np.abs(out - g['lidar_raw'][k]).max() < 1e-9, k
    finally:
        O.set_tables(beam_a=own['beam_a'], beam_b=own['beam_b'])
    assert (g['lidar_raw'][-1] == 10.0).all()


def test_action_mask(gold):
    """get_steps/post_process.  The straight arcs (actions 10, 31) sweep a box whose side coincides
    with the hull side, so with a touching obstacle (lidar <= 0) `dist_star <= d` is an exact TIE
    decided by the table's last ulp.  The oracle is therefore first pinned to the reference's own
    coarse table (its upsampler must reproduce the reference table bit-for-bit: sha256), then
    every mask must match exactly."""
    import hashlib
    g, c = gold('action_mask.npz'), gold('constants.npz')
    own = O.tables()
    try:
        O.set_dist_star_coarse(c['dist_star_coarse'])
        t = O.tables()
        assert hashlib.sha256(t['dist_star'].tobytes()).hexdigest() == str(c['dist_star_sha256'])
        # the hull base the reference run added was stub-computed; feed the same array back in
        for k in range(len(g['scan'])):
            m = O.get_steps(g['scan'][k], hull_base=g['hull_base_used'])
            assert np.array_equal(m, g['mask'][k]), k
    finally:
        O.set_tables(dist_star=own['dist_star'])
    assert np.abs(own['hull_base'] - g['hull_base_used']).max() < 1e-14
    # with the oracle's OWN tables only tie cases (lidar <= 0 on a hull-side beam) may differ
    for k in range(len(g['scan'])):
        if (g['scan'][k] > 1e-9).all():
            assert np.array_equal(O.get_steps(g['scan'][k]), g['mask'][k]), k
    assert (g['mask'][-1] >= 0).all() and set(np.unique(g['mask'])) <= {0, .01, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1.}


def test_reeds_shepp_all_paths(gold, math_flavour):
    g = gold('reeds_shepp.npz')
    maxc = float(g['maxc'])
    assert maxc == 0.3327130214085973 == math.tan(0.75) / 2.8
    off = g['path_off']
    for i in range(len(g['q0'])):
        r = O.rs_all_paths(g['q0'][i], g['q1'][i], maxc)
        a, b = off[i], off[i + 1]
        assert r['n'] == b - a, i
        assert np.array_equal(r['ctypes'], g['ctypes'][a:b]), i
        assert np.abs(r['lengths'] - g['lengths'][a:b]).max() < 1e-9, i
        assert np.abs(r['L'] - g['L'][a:b]).max() < 1e-9, i
        # a word with a segment of ~zero length (start exactly on the slot axis) has a direction/point count that is
        # the SIGN of libm noise; only the glibc build (= Python's math) is required to reproduce those
        degen = (np.abs(g['lengths'][a:b]) < 1e-9) & (g['ctypes'][a:b] >= 0)
        strict = ~degen.any(axis=1) if math_flavour == 'hope_math' else np.ones(b - a, bool)
        assert np.array_equal(r['npts'][strict], g['npts'][a:b][strict]), i
        assert np.abs(r['first3'][strict] - g['first3'][a:b][strict]).max(initial=0) < 1e-9, i
        assert np.abs(r['last3'][strict] - g['last3'][a:b][strict]).max(initial=0) < 1e-9, i
        assert np.abs(r['sums'][strict] - g['sums'][a:b][strict]).max(initial=0) < 1e-6, i
    # SURVEY §8a-14 KAT: (0,0,0)->(5,3,1.0): 6 paths, first SLS 6.3605.../65 pts
    r = O.rs_all_paths([0, 0, 0], [5, 3, 1.0], maxc)
    assert r['n'] == 6 and list(r['npts']) == [65, 61, 190, 163, 115, 118]
    assert abs(r['L'][0] - 6.360574570820278) < 1e-12


def test_is_traj_valid(gold, dlp):
    g, s = gold('traj_valid.npz'), gold('rs_search.npz')
    off = g['traj_off']
    for k in range(len(g['rec'])):
        ri = int(g['rec'][k])
        keep = s['keep'][s['keep_off'][ri]:s['keep_off'][ri + 1]]
        v, nv = case_obstacles(dlp, int(s['case'][ri]), keep)
        ok = O.is_traj_valid(g['traj'][off[k]:off[k + 1]], v, nv, s['bbox'][ri])
        assert ok == bool(g['valid'][k]), k


def test_find_rs_path(gold, dlp):
    s = gold('rs_search.npz')
    for ri in range(len(s['case'])):
        keep = s['keep'][s['keep_off'][ri]:s['keep_off'][ri + 1]]
        v, nv = case_obstacles(dlp, int(s['case'][ri]), keep)
        r = O.find_rs_path(s['pose'][ri], s['dest'][ri], v, nv, s['bbox'][ri])
        assert r['found'] == bool(s['found'][ri]), ri
        assert r['n_tested'] == int(s['n_tested'][ri]), ri
        if r['found']:
            assert np.array_equal(r['ctypes'], s['ctypes'][ri]), ri
            assert np.abs(r['lengths'] - s['lengths'][ri]).max() < 1e-9
            assert abs(r['L'] - s['L'][ri]) < 1e-9


def test_target_reward_wrapper(gold):
    g = gold('wrapper_reward.npz')
    for i in range(len(g['act_in'])):
        # reference evaluates action_rescale in float32 when handed a float32 Box (documented deviation)
        assert np.abs(O.action_rescale(g['act_in'][i]) - g['act_rescaled'][i]).max() < 2e-7
    for i in range(len(g['reward_info'])):
        for j, st in enumerate([1, 2, 3, 4, 5]):
            assert abs(O.reward_shaping(g['reward_info'][i], st) - g['shaped'][i, j]) < 1e-15
    for i in range(len(g['ego'])):
        assert np.abs(O.target_repr(g['ego'][i], g['dest'][i]) - g['target'][i]).max() < 1e-13
        out, acc = O.reward_terms(g['prev'][i], g['ego'][i], g['dest'][i], g['start'][i], g['t'][i],
                                  g['union_area'][i], float(g['dest_area']), g['accum_in'][i])
        assert np.abs(out - g['reward_list'][i]).max() < 1e-13
        assert abs(acc - g['accum_out'][i]) < 1e-15


def test_find_rs_path_genuine_heapdict(gold, dlp):
    """the same 160 searches, reference run with the GENUINE heapdict 1.0.1 (pure python, found under /opt/conda in the
    build container) instead of tests/golden/refstubs/heapdict.py: pop order of equal-length words is the real one."""
    s, h = gold('rs_search.npz'), gold('rs_search_heapdict.npz')
    assert bool(h['identical_to_stub_run'])
    for ri in range(len(s['case'])):
        keep = s['keep'][s['keep_off'][ri]:s['keep_off'][ri + 1]]
        v, nv = case_obstacles(dlp, int(s['case'][ri]), keep)
        r = O.find_rs_path(s['pose'][ri], s['dest'][ri], v, nv, s['bbox'][ri])
        assert r['found'] == bool(h['found'][ri]) and r['n_tested'] == int(h['n_tested'][ri]), ri
        if r['found']:
            assert np.array_equal(r['ctypes'], h['ctypes'][ri]), ri
            assert np.abs(r['lengths'] - h['lengths'][ri]).max() < 1e-9


def test_step_driver_control_flow(gold):
    """CarParking.step / _check_status / get_reward / Vehicle.step / retreat executed from the reference's own source
    (tests/golden/make_golden_r2.py gold_step_driver; only the GEOS predicates were answered by this oracle's
    restatement): sub-step order, arrive-before-collide, retreat, status priority, t, reward gating, accumulator and
    trajectory bookkeeping of the oracle's batch step must equal the reference's control flow on every step."""
    g = gold('step_driver.npz')
    n, mo = g['start'].shape[0], g['verts'].shape[1]
    orc = O.BatchOracle(n, mo)
    orc.set_scenes(np.arange(n), g['start'], g['dest'], g['bbox'], g['verts'], g['nvert'], g['n_obst'])
    orc.t[:] = g['t0']
    seen = set()
    for k in range(g['actions'].shape[0] + 1):
        o = orc.reset_obs(with_rs=False) if k == 0 else orc.step(g['actions'][k - 1], with_rs=False)
        assert np.array_equal(o['status'], g['status'][k]), (k, np.nonzero(o['status'] != g['status'][k])[0])
        assert np.array_equal(orc.t, g['t'][k])
        assert np.abs(orc.pose - g['pose'][k]).max() < 1e-11, k
        assert np.abs(o['reward_info'] - g['reward_info'][k]).max() < 1e-11, k
        assert np.abs(orc.accum - g['accum'][k]).max() < 1e-12, k
        tl = np.array([len(t) for t in orc.traj])
        assert np.array_equal(np.minimum(tl, 21), np.minimum(g['traj_len'][k], 21)), k
        if k > 0:                                   # a step that kept no sub-step leaves the trajectory alone
            assert np.array_equal(o['substeps'] > 0, g['traj_len'][k] > g['traj_len'][k - 1]), k
        seen |= set(o['status'].tolist())
    assert seen == {1, 2, 3, 4, 5}
    # the RS gate fired exactly where the reference's did (t > 1, CONTINUE, within RS_MAX_DIST)
    d = np.hypot(g['pose'][..., 0] - g['dest'][None, :, 0], g['pose'][..., 1] - g['dest'][None, :, 1])
    assert np.array_equal(g['rs_called'] > 0, (g['t'] > 1) & (g['status'] == 1) & (d < 10.0))
