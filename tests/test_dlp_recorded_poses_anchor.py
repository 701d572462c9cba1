"""A reference-HELD anchor for row a-3 (`CarParking._detect_collision`, car_parking_base.py:153-158), whose arithmetic the reference
delegates to GEOS (`LinearRing.intersects`: absent from this image, so the predicate itself is spec-checked, not fixture-pinned).

`data/dlp.data` (decoded bit for bit into data/dlp_scenes.npz by tests/golden/make_golden.py) is the reference's own data: 248
Dragon-Lake cases with 7 .. 3 183 RECORDED drive poses each (the start candidates of ParkingMapDLP.reset, parking_map_dlp.py:43-63)
and one dest.  A recorded pose whose hull ring intersected an obstacle ring would end every episode started from it at t = 1 with
COLLIDED (`reset` runs an action-less `step`, car_parking_base.py:138,175-177): the reference's authors trained on these starts,
so every (hull of an un-jittered candidate or dest, obstacle set of its case) pair is a reference-held NEGATIVE of the predicate:
36 845 poses x 162 .. 312 rings.  The anchor is not vacuous: hundreds of the poses stand within 0.3 m of a neighbouring car.

CPU half: `orc_detect_collision` (oracle/hope_oracle.c) on every pair.  GPU half (-m gpu): the HIP motion launch's status check on
the same poses through the C ABI (reset observation status != COLLIDED), culled exactly as ParkingMapDLP.reset culls.
Exceptions, if any, are reported by (case, candidate) with the distance between the rings."""
import numpy as np
import pytest

from hope_amd.scenes import DlpScenePool, Scene, cull_obstacles, create_box


def _ring_clearance(box, verts, nvert):
    """min distance between the hull ring and the obstacle rings' vertices / edges (both directions), numpy"""
    def seg_pt(a, b, p):                       # distance of points p [m, 2] to segments a -> b [k, 2]
        ab = b - a
        l2 = (ab ** 2).sum(1)
        t = np.clip(((p[:, None, :] - a[None]) * ab[None]).sum(2) / np.where(l2 > 0, l2, 1.0)[None], 0.0, 1.0)
        q = a[None] + t[..., None] * ab[None]
        return np.sqrt(((p[:, None, :] - q) ** 2).sum(2)).min()
    ha, hb = box, np.roll(box, -1, axis=0)
    best = np.inf
    cen = box.mean(0)
    near = np.nonzero(np.hypot(*(verts[:, 0] - cen).T) < 12.0)[0]
    for o in near:
        r = verts[o, :int(nvert[o])]
        best = min(best, seg_pt(ha, hb, r), seg_pt(r, np.roll(r, -1, axis=0), box))
    return best


def _poses(pool):
    for c in range(len(pool)):
        cand = pool.candidates(c)
        for k in range(len(cand)):
            yield c, k, cand[k]
        yield c, -1, pool.dest[c]


def test_recorded_dlp_poses_do_not_collide_under_the_oracle():
    from oracle import oracle as O
    pool = DlpScenePool()
    n = 0
    exceptions = []
    close = 0
    rng = np.random.default_rng(0)
    for c, k, pose in _poses(pool):
        v, nv = pool.obstacles(c)
        box = O.create_box(pose)
        n += 1
        if O.detect_collision(box, v, nv):
            exceptions.append((c, k, _ring_clearance(box, v, nv)))
        elif rng.random() < 0.02:              # how near the negatives are (a sample: the numpy distance is slow)
            close += _ring_clearance(box, v, nv) < 0.3
    print(f'recorded poses checked: {n}; colliding under orc_detect_collision: {exceptions}; of a 2 % sample, {close} within 0.3 m of an obstacle')
    assert n == 36597 + 248
    assert not exceptions, exceptions
    assert close >= 5                          # the negatives are tight ones


@pytest.mark.gpu
def test_recorded_dlp_poses_do_not_collide_under_the_hip_motion_launch():
    import torch
    from hope_amd import ParkingBatch
    pool = DlpScenePool()
    items = list(_poses(pool))
    scenes = []
    for c, k, pose in items:
        start, dest = np.array(pose, dtype=np.float64), pool.dest[c].copy()
        bbox = np.array([np.floor(min(start[0], dest[0]) - 20), np.ceil(max(start[0], dest[0]) + 20),
                         np.floor(min(start[1], dest[1]) - 20), np.ceil(max(start[1], dest[1]) + 20)])
        v, nv = pool.obstacles(c)
        keep = cull_obstacles(v, nv, bbox)     # ParkingMapDLP.filter_obstacles (:88-101)
        scenes.append(Scene(start=start, dest=dest, bbox=bbox, verts=v[keep], nvert=nv[keep], level='dlp', case_id=c))
    mo = max(s.n_obst for s in scenes)
    assert mo <= 128
    n = len(scenes)
    env = ParkingBatch(n, 128, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    env.reset_obs()                            # CarParking.reset's action-less step: status of the start pose
    torch.cuda.synchronize()
    status = env.status.cpu().numpy()
    bad = np.nonzero(status == 3)[0]
    exc = [(items[i][0], items[i][1], _ring_clearance(create_box(scenes[i].start), scenes[i].verts, scenes[i].nvert)) for i in bad]
    print(f'recorded poses through the HIP motion launch: {n}; COLLIDED: {exc}; status histogram: {np.bincount(status, minlength=6).tolist()}')
    assert not exc, exc
    # (the dest pose of a case is ARRIVED by construction; recorded starts are CONTINUE, ARRIVED or -- a start outside the +-20 m box
    # cannot happen -- never OUTBOUND)
    assert int((status == 4).sum()) == 0
    env.close()
