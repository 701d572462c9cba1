#!/usr/bin/env python3
"""Generate the committed golden fixtures by IMPORTING the reference (build container only).

Run from the repo root:   python tests/golden/make_golden.py

What it does:
  * puts `tests/golden/refstubs` (import-satisfying stand-ins for shapely/gym/pygame/
    heapdict/cv2 -- my code, not reference code) and `/root/reference/src` on sys.path,
  * imports the reference's pure numpy/math functions UNMODIFIED and records
    (input, output) pairs as small .npz files next to this script,
  * decodes `/root/reference/data/dlp.data` (a pickle of shapely-1.x rings) into the
    shapely-free `data/dlp_scenes.npz`.

Nothing here travels to the GPU box except the produced .npz files (data, not source).
Geometry that only a stub computes (hull ranges via the stub's ray/ring intersection) is
never recorded as reference output.
"""
import hashlib
import math
import os
import pickle
import sys
import textwrap
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(HERE, 'refstubs'))
sys.path.insert(0, os.path.join(REF, 'src'))
warnings.filterwarnings('ignore')

import configs as C                                    # noqa: E402
from env.vehicle import KSModel, State, Status         # noqa: E402
from env.lidar_simulator import LidarSimlator          # noqa: E402
from model.action_mask import ActionMask               # noqa: E402
import env.reeds_shepp as rs                           # noqa: E402
import env.car_parking_base as cpb                     # noqa: E402
import env.env_wrapper as ew                           # noqa: E402
from gym.spaces import Box                             # noqa: E402

CT = {'S': 0, 'L': 1, 'R': 2}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f'  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB')


# --------------------------------------------------------------------------- DLP decode
def decode_dlp():
    d = pickle.load(open(os.path.join(REF, 'data', 'dlp.data'), 'rb'))
    sets, set_key = [], {}
    case_set = np.zeros(len(d), np.int32)
    dest = np.zeros((len(d), 3))
    starts, start_off = [], [0]
    for ci, case in enumerate(d):
        cand, dst, obs = case[:3]
        rings = []
        for o in obs:
            c = np.array(o.coords)
            assert np.array_equal(c[0], c[-1])
            rings.append(c[:-1])
        key = hashlib.md5(b''.join(r.tobytes() for r in rings)).hexdigest()
        if key not in set_key:
            set_key[key] = len(sets)
            sets.append(rings)
        case_set[ci] = set_key[key]
        dest[ci] = [float(v) for v in dst]
        for s in cand:
            starts.append([float(v) for v in s])
        start_off.append(len(starts))
    verts, nvert, set_off = [], [], [0]
    for rings in sets:
        for r in rings:
            assert r.shape[0] in (3, 4)
            v = np.zeros((4, 2))
            v[:r.shape[0]] = r
            if r.shape[0] == 3:
                v[3] = r[2]          # degenerate 4th slot (repeat last vertex)
            verts.append(v)
            nvert.append(r.shape[0])
        set_off.append(len(verts))
    out = os.path.join(REPO, 'data', 'dlp_scenes.npz')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, set_verts=np.array(verts), set_nvert=np.array(nvert, np.int8),
                        set_off=np.array(set_off, np.int32), case_set=case_set, dest=dest,
                        starts=np.array(starts), start_off=np.array(start_off, np.int32))
    print(f'  wrote data/dlp_scenes.npz: {os.path.getsize(out)/1024:.1f} KiB '
          f'({len(d)} cases, {len(sets)} obstacle sets, {len(starts)} starts)')
    return d


# --------------------------------------------------------------------------- helpers
def pt_seg_dist(px, py, ax, ay, bx, by):
    """point-to-segment distance (used only to choose which rings go INTO a lidar fixture)."""
    if ax == bx and ay == by:
        return math.hypot(px - ax, py - ay)
    len2 = (bx - ax) ** 2 + (by - ay) ** 2
    r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2
    if r <= 0:
        return math.hypot(px - ax, py - ay)
    if r >= 1:
        return math.hypot(px - bx, py - by)
    s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2
    return abs(s) * math.sqrt(len2)


class Ring:
    def __init__(self, coords):
        self.coords = [tuple(c) for c in coords]


def ego_rings(case, pose, rng_keep=10.0):
    """world rings -> ego frame with the reference's affine matrix (lidar_simulator.py:55-64),
    keep rings whose boundary comes within rng_keep of the origin."""
    x, y, th = pose
    a, b = math.cos(th), math.sin(th)
    xo, yo = -x * a - y * b, x * b - y * a
    out = []
    for o in case[2]:
        cs = [(a * px + b * py + xo, -b * px + a * py + yo) for px, py in o.coords]
        dmin = min(pt_seg_dist(0, 0, *p, *q) for p, q in zip(cs[:-1], cs[1:]))
        if dmin < rng_keep:
            out.append(Ring(cs))
    return out


def flat_rings(rings):
    """ragged rings -> (verts[n,4,2] with triangles padded by repeating the last vertex)."""
    v = np.zeros((len(rings), 4, 2))
    for i, r in enumerate(rings):
        c = np.array(r.coords[:-1])
        v[i, :len(c)] = c
        if len(c) == 3:
            v[i, 3] = c[2]
    return v


# --------------------------------------------------------------------------- fixtures
def gold_constants(am, lidar):
    # capture the table handed to _linear_interpolate inside precompute() (action_mask.py:142)
    captured = []
    orig = am._linear_interpolate
    am._linear_interpolate = lambda x, r=None: (captured.append(np.array(x)), orig(x, r))[1]
    again = am.precompute()
    am._linear_interpolate = orig
    assert np.array_equal(again, am.dist_star)
    coarse = captured[0]
    theta = np.array([a * math.pi / 120 * 2 for a in range(120)])
    sub = am.dist_star[::7]
    h = hashlib.sha256(np.ascontiguousarray(am.dist_star).tobytes()).hexdigest()
    save('constants.npz',
         discrete_actions=np.array(C.discrete_actions, dtype=np.float64),
         vehicle_box=np.array(C.VehicleBox.coords),
         vehicle_boxes=am.vehicle_boxes,
         dist_star_coarse=coarse, dist_star_every7=sub, dist_star_sum=np.array(am.dist_star.sum()),
         dist_star_max=np.array(am.dist_star.max()), dist_star_min=np.array(am.dist_star.min()),
         dist_star_sha256=np.array(h),
         dist_star_beam0=am.dist_star[0], dist_star_beam600=am.dist_star[600],
         beam_a=np.sin(theta), beam_b=-np.cos(theta),
         scalars=np.array([C.WHEEL_BASE, C.FRONT_HANG, C.REAR_HANG, C.LENGTH, C.WIDTH,
                           C.VALID_SPEED[0], C.VALID_SPEED[1], C.VALID_STEER[0], C.VALID_STEER[1],
                           C.NUM_STEP, C.STEP_LENGTH, C.LIDAR_RANGE, C.LIDAR_NUM, C.TOLERANT_TIME,
                           C.RS_MAX_DIST, C.PRECISION, C.REWARD_RATIO, float(C.ENV_COLLIDE)]),
         reward_weight=np.array(list(C.REWARD_WEIGHT.values()), dtype=np.float64))


def gold_ksmodel(rng):
    ks = KSModel(C.WHEEL_BASE, C.STEP_LENGTH, C.NUM_STEP, C.VALID_SPEED, C.VALID_STEER)
    n = 600
    pose = np.column_stack([rng.uniform(-20, 150, n), rng.uniform(-40, 80, n), rng.uniform(-7, 7, n)])
    act = np.column_stack([rng.uniform(-1.0, 1.0, n), rng.uniform(-3.2, 3.2, n)])
    act[:20] = [[0.5, 2.0]] * 10 + [[3.0, 9.0]] * 5 + [[-3.0, -9.0]] * 5
    pose[0] = [1, 2, 0.3]
    pose[10] = [1, 2, 0.3]
    act[20] = [0.0, 0.0]
    act[21] = [0.75, 2.5]
    act[22] = [-0.75, -2.5]
    out1 = np.zeros((n, 5))
    out10 = np.zeros((n, 3))
    for i in range(n):
        st = State([float(pose[i, 0]), float(pose[i, 1]), float(pose[i, 2]), 0, 0])
        a = [float(act[i, 0]), float(act[i, 1])]
        s1 = ks.step(st, a, step_time=1)
        out1[i] = [s1.loc.x, s1.loc.y, s1.heading, s1.speed, s1.steering]
        s = st
        for _ in range(10):
            s = ks.step(s, a, step_time=1)
        out10[i] = [s.loc.x, s.loc.y, s.heading]
    save('ksmodel.npz', pose=pose, action=act, out1=out1, out10=out10)


def gold_lidar_mask(d, am, lidar, rng):
    cases, offs, verts, outs, poses = [], [0], [], [], []
    for k in range(70):
        ci = int(rng.integers(len(d)))
        cand = d[ci][0]
        s = cand[int(rng.integers(len(cand)))]
        pose = (float(s[0]) + rng.normal() * 0.3, float(s[1]) + rng.normal() * 0.3,
                float(s[2]) + rng.normal() * 0.3)
        if k % 5 == 0:   # start near the destination slot (tight surroundings)
            dst = d[ci][1]
            pose = (float(dst[0]) + rng.normal() * 0.4, float(dst[1]) + rng.normal() * 0.4,
                    float(dst[2]) + rng.normal() * 0.1)
        rings = ego_rings(d[ci], pose)
        out = lidar._fast_calc_lidar_obs(rings)
        v = flat_rings(rings)
        verts.append(v)
        offs.append(offs[-1] + len(v))
        outs.append(out)
        poses.append(pose)
        cases.append(ci)
    # hand-made: axis-aligned box around the origin, single far edge, empty
    specials = [
        [Ring([(2, -1), (4, -1), (4, 1), (2, 1), (2, -1)])],
        [Ring([(-3, -3), (3, -3), (3, 3), (-3, 3), (-3, -3)])],
        [Ring([(5.5, 0.25), (7.5, 1.0), (6.75, 3.0), (4.75, 2.25), (5.5, 0.25)]),
         Ring([(-1.5, 2.0), (1.0, 2.5), (0.5, 4.0), (-1.5, 2.0)])],
        [],
    ]
    for rings in specials:
        out = lidar._fast_calc_lidar_obs(rings)
        v = flat_rings(rings) if rings else np.zeros((0, 4, 2))
        verts.append(v)
        offs.append(offs[-1] + len(v))
        outs.append(out)
        poses.append((0, 0, 0))
        cases.append(-1)
    outs = np.array(outs)
    save('lidar.npz', ring_verts=np.concatenate(verts), ring_off=np.array(offs, np.int32),
         lidar_raw=outs, pose=np.array(poses), case=np.array(cases, np.int32))

    # action mask: scans = (lidar - stub hull base is NOT reference) -> feed raw-minus-base
    # using closed-form base computed by the oracle side; here we feed scans directly.
    base = am.vehicle_lidar_base      # stub-computed; recorded only as the INPUT that was added
    scans = [o - base for o in outs]
    scans += [np.full(120, 10.0) - base, np.zeros(120), np.full(120, -0.05), np.full(120, 0.3),
              np.full(120, 1.0)]
    for q in range(4):
        s = np.full(120, 10.0) - base
        s[q * 30 + 7] = 0.05
        scans.append(s)
    for _ in range(40):
        s = rng.uniform(0, 3.0, 120)
        scans.append(s)
    for _ in range(40):
        s = np.full(120, 10.0) - base
        i0 = int(rng.integers(120))
        w = int(rng.integers(1, 40))
        idx = (i0 + np.arange(w)) % 120
        s[idx] = rng.uniform(0.0, 1.5, w)
        scans.append(s)
    scans = np.array(scans)
    masks = np.array([am.get_steps(s.copy()) for s in scans])
    save('action_mask.npz', scan=scans, hull_base_used=base, mask=masks)


def enc_path(p, npmax=5):
    ct = np.full(npmax, -1, np.int8)
    ln = np.zeros(npmax)
    for i, (c, l) in enumerate(zip(p.ctypes, p.lengths)):
        ct[i] = CT[c]
        ln[i] = l
    return ct, ln


def gold_rs(rng):
    maxc = math.tan(C.VALID_STEER[-1]) / C.WHEEL_BASE
    n = 1500
    q0 = np.column_stack([rng.uniform(-20, 150, n), rng.uniform(-40, 80, n), rng.uniform(-4, 4, n)])
    r = rng.uniform(0.3, 10.0, n)
    ang = rng.uniform(0, 2 * math.pi, n)
    q1 = np.column_stack([q0[:, 0] + r * np.cos(ang), q0[:, 1] + r * np.sin(ang), rng.uniform(-4, 4, n)])
    # near-degenerate heading differences and aligned poses
    q1[:60, 2] = q0[:60, 2] + rng.choice([0.0, math.pi, -math.pi, 1e-9, math.pi / 2], 60)
    q0[60:80] = [0, 0, 0]
    q1[60:80, 0] = rng.uniform(-8, 8, 20)
    q1[60:80, 1] = 0.0
    q1[60:80, 2] = 0.0
    q0[80] = [0, 0, 0]
    q1[80] = [5, 3, 1.0]
    path_off, ct, ln, L, npts, first3, last3, sums, opt = [0], [], [], [], [], [], [], [], []
    for i in range(n):
        paths = rs.calc_all_paths(*[float(v) for v in q0[i]], *[float(v) for v in q1[i]], maxc, 0.1)
        for p in paths:
            c, l = enc_path(p)
            ct.append(c)
            ln.append(l)
            L.append(p.L)
            npts.append(len(p.x))
            xyz = np.column_stack([p.x, p.y, p.yaw])
            f = np.zeros((3, 3))
            la = np.zeros((3, 3))
            m = min(3, len(xyz))
            f[:m] = xyz[:m]
            la[3 - m:] = xyz[-m:]
            first3.append(f)
            last3.append(la)
            sums.append([math.fsum(p.x), math.fsum(p.y), math.fsum(p.yaw), float(sum(p.directions))])
        path_off.append(len(L))
        if paths:
            minL, mini = paths[0].L, 0
            for j in range(len(paths)):
                if paths[j].L <= minL:
                    minL, mini = paths[j].L, j
            opt.append(mini)
        else:
            opt.append(-1)
    save('reeds_shepp.npz', q0=q0, q1=q1, maxc=np.array(maxc), path_off=np.array(path_off, np.int32),
         ctypes=np.array(ct), lengths=np.array(ln), L=np.array(L), npts=np.array(npts, np.int32),
         first3=np.array(first3), last3=np.array(last3), sums=np.array(sums), optimal=np.array(opt, np.int32))


def method_source(cls, name, nxt):
    src = open(os.path.join(REF, 'src', 'env', 'car_parking_base.py')).read()
    a = src.index(f'    def {name}')
    b = src.index(f'    def {nxt}')
    return textwrap.dedent(src[a:b])


def gold_traj_and_rs_search(d, rng):
    """is_traj_valid (car_parking_base.py:452) and find_rs_path (:413) run on DLP scenes.
    Obstacles are identified by (case id, kept obstacle indices) into data/dlp_scenes.npz."""
    ns = {'np': np, 'VehicleBox': C.VehicleBox, 'Area': cpb.Area, 'math': math, 'rsCurve': rs,
          'heapdict': cpb.heapdict, 'VALID_STEER': C.VALID_STEER, 'WHEEL_BASE': C.WHEEL_BASE}
    exec(method_source(cpb.CarParking, 'is_traj_valid', 'close'), ns)
    exec(method_source(cpb.CarParking, 'find_rs_path', 'is_traj_valid'), ns)
    maxc = math.tan(C.VALID_STEER[-1]) / C.WHEEL_BASE

    recs = dict(case=[], pose=[], dest=[], bbox=[], keep_off=[0], keep=[], n_paths=[],
                found=[], ct=[], ln=[], L=[], n_tested=[])
    tv = dict(rec=[], path_idx=[], valid=[], traj_off=[0], traj=[])
    n_target = 160
    tries = 0
    while len(recs['case']) < n_target:
        tries += 1
        ci = int(rng.integers(len(d)))
        cand, dst, obs = d[ci][:3]
        dst = [float(v) for v in dst]
        if rng.random() < 0.5:                # DLP reset flips dest with p=.5 (parking_map_dlp.py:80)
            bx = State(dst + [0, 0]).create_box()
            cen = np.mean(np.array(bx.coords[:-1]), axis=0)
            dst = [2 * cen[0] - dst[0], 2 * cen[1] - dst[1], dst[2] + np.pi]
        rr = rng.uniform(0.5, 9.5)
        aa = rng.uniform(0, 2 * math.pi)
        pose = [dst[0] + rr * math.cos(aa), dst[1] + rr * math.sin(aa), dst[2] + rng.normal() * 0.8]
        if len(recs['case']) % 4 == 0:        # approach roughly along the slot axis: more feasible paths
            fwd = rng.uniform(3.0, 9.0) * rng.choice([-1, 1])
            pose = [dst[0] + fwd * math.cos(dst[2]) + rng.normal() * 0.2,
                    dst[1] + fwd * math.sin(dst[2]) + rng.normal() * 0.2, dst[2] + rng.normal() * 0.1]
        s0 = cand[int(rng.integers(len(cand)))]
        xmin = np.floor(min(float(s0[0]), dst[0]) - 20)
        xmax = np.ceil(max(float(s0[0]), dst[0]) + 20)
        ymin = np.floor(min(float(s0[1]), dst[1]) - 20)
        ymax = np.ceil(max(float(s0[1]), dst[1]) + 20)
        keep = []
        for oi, o in enumerate(obs):           # parking_map_dlp.py:88-101
            c = np.array(o.coords)
            if not (c[:, 0].max() <= xmin or c[:, 0].min() >= xmax or
                    c[:, 1].max() <= ymin or c[:, 1].min() >= ymax):
                keep.append(oi)
        kept = [obs[i] for i in keep]
        fake = types.SimpleNamespace()
        fake.map = types.SimpleNamespace(xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, obstacles=kept,
                                         dest=State(dst + [0, 0]))
        fake.vehicle = types.SimpleNamespace(state=State(pose + [0, 0]))
        tested = []

        def itv(traj, _f=fake, _t=tested):
            r = ns['is_traj_valid'](_f, traj)
            _t.append((traj, r))
            return r
        fake.is_traj_valid = itv
        path = ns['find_rs_path'](fake, Status.CONTINUE)
        all_paths = rs.calc_all_paths(*pose, *dst, maxc, 0.1)
        ri = len(recs['case'])
        recs['case'].append(ci)
        recs['pose'].append(pose)
        recs['dest'].append(dst)
        recs['bbox'].append([xmin, xmax, ymin, ymax])
        recs['keep'].extend(keep)
        recs['keep_off'].append(len(recs['keep']))
        recs['n_paths'].append(len(all_paths))
        recs['n_tested'].append(len(tested))
        recs['found'].append(path is not None)
        if path is not None:
            c, l = enc_path(path)
            recs['ct'].append(c)
            recs['ln'].append(l)
            recs['L'].append(path.L)
        else:
            recs['ct'].append(np.full(5, -1, np.int8))
            recs['ln'].append(np.zeros(5))
            recs['L'].append(0.0)
        if ri < 60:                            # keep explicit trajectories for a subset (size)
            for k, (traj, r) in enumerate(tested):
                t = np.array(traj)
                if len(t) > 1500:
                    continue
                tv['rec'].append(ri)
                tv['path_idx'].append(k)
                tv['valid'].append(r)
                tv['traj'].append(t)
                tv['traj_off'].append(tv['traj_off'][-1] + len(t))
    print(f'  rs search: {sum(recs["found"])}/{n_target} feasible, {tries} tries')
    save('rs_search.npz', case=np.array(recs['case'], np.int32), pose=np.array(recs['pose']),
         dest=np.array(recs['dest']), bbox=np.array(recs['bbox']),
         keep=np.array(recs['keep'], np.int16), keep_off=np.array(recs['keep_off'], np.int32),
         n_paths=np.array(recs['n_paths'], np.int32), n_tested=np.array(recs['n_tested'], np.int32),
         found=np.array(recs['found']), ctypes=np.array(recs['ct']), lengths=np.array(recs['ln']),
         L=np.array(recs['L']))
    save('traj_valid.npz', rec=np.array(tv['rec'], np.int32), path_idx=np.array(tv['path_idx'], np.int32),
         valid=np.array(tv['valid']), traj=np.concatenate(tv['traj']),
         traj_off=np.array(tv['traj_off'], np.int32))


def gold_wrapper_reward(rng):
    space = Box(np.array([C.VALID_STEER[0], C.VALID_SPEED[0]]).astype(np.float32),
                np.array([C.VALID_STEER[1], C.VALID_SPEED[1]]).astype(np.float32))
    acts = np.vstack([rng.uniform(-1.5, 1.5, (200, 2)), [[1, 1], [-1, -1], [0, 0], [0.3, -0.7]]])
    resc = np.array([ew.action_rescale(a.copy(), space) for a in acts], dtype=np.float64)
    # reward_shaping for every status with random reward_info
    infos = rng.uniform(-1, 1, (50, 5))
    keys = list(C.REWARD_WEIGHT.keys())
    shaped = np.zeros((50, 5))
    for i in range(50):
        ri = dict(zip(keys, [float(v) for v in infos[i]]))
        for j, st in enumerate([Status.CONTINUE, Status.ARRIVED, Status.COLLIDED, Status.OUTBOUND,
                                Status.OUTTIME]):
            shaped[i, j] = ew.reward_shaping(None, ri, st, {})[1]

    # _get_targt_repr (car_parking_base.py:372) and the non-GEOS arithmetic of _get_reward (:186)
    ns = {'np': np, 'math': math, 'rsCurve': rs, 'REWARD_WEIGHT': C.REWARD_WEIGHT,
          'TOLERANT_TIME': C.TOLERANT_TIME, 'VALID_STEER': C.VALID_STEER, 'WHEEL_BASE': C.WHEEL_BASE,
          'State': State}

    class FakePoly:                     # area supplied as an INPUT (GEOS overlay is not available)
        def __init__(self, tag):
            self.tag = tag

        def intersection(self, other):
            return types.SimpleNamespace(area=FakePoly.union_area)

        @property
        def area(self):
            return FakePoly.dest_area
    ns['Polygon'] = FakePoly
    exec(method_source(cpb.CarParking, '_get_targt_repr', 'render'), ns)
    exec(method_source(cpb.CarParking, '_get_reward', 'get_reward'), ns)
    n = 300
    ego = np.column_stack([rng.uniform(-20, 150, n), rng.uniform(-40, 80, n), rng.uniform(-7, 7, n)])
    prv = ego + rng.normal(0, 0.5, (n, 3))
    dst = ego + np.column_stack([rng.uniform(-15, 15, n), rng.uniform(-15, 15, n), rng.uniform(-4, 4, n)])
    sta = dst + np.column_stack([rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.uniform(-4, 4, n)])
    tt = rng.integers(1, 201, n).astype(np.float64)
    ua = rng.uniform(0, 9.0986, n)
    ua[::3] = 0.0
    acc_in = rng.uniform(0, 0.6, n)
    acc_in[::2] = 0.0
    tgt = np.zeros((n, 5))
    rew = np.zeros((n, 5))
    acc_out = np.zeros(n)
    for i in range(n):
        fake = types.SimpleNamespace()
        fake.map = types.SimpleNamespace(dest=State(list(map(float, dst[i])) + [0, 0]),
                                         start=State(list(map(float, sta[i])) + [0, 0]), dest_box='d')
        fake.vehicle = types.SimpleNamespace(state=State(list(map(float, ego[i])) + [0, 0]), box='v')
        fake.t = float(tt[i])
        fake.accum_arrive_reward = float(acc_in[i])
        FakePoly.union_area = float(ua[i])
        FakePoly.dest_area = 9.0986
        tgt[i] = ns['_get_targt_repr'](fake)
        rew[i] = ns['_get_reward'](fake, State(list(map(float, prv[i])) + [0, 0]), fake.vehicle.state)
        acc_out[i] = fake.accum_arrive_reward
    save('wrapper_reward.npz', act_in=acts, act_rescaled=resc, reward_info=infos, shaped=shaped,
         ego=ego, prev=prv, dest=dst, start=sta, t=tt, union_area=ua, dest_area=np.array(9.0986),
         accum_in=acc_in, accum_out=acc_out, target=tgt, reward_list=rew)


def gold_agent_glue(rng):
    """agent-side glue right after the env step: RsPlanner expansion, ActionMask.choose_action
    probabilities, StateNorm recurrence."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('parking_agent', os.path.join(REF, 'src', 'model', 'agent', 'parking_agent.py'))
    pa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pa)
    from model.state_norm import StateNorm
    g = np.load(os.path.join(HERE, 'reeds_shepp.npz'))
    names = {0: 'S', 1: 'L', 2: 'R'}
    words, lens, acts, off = [], [], [], [0]
    special = [([1, 0, 2], [1.25, 2.5, -1.25]), ([0], [0.001]), ([1, 2], [3.75, -0.0005]), ([2, 0, 1, 2, 1], [-5.1, 0.3, 2.5000001, -1.2499999, 9.99])]
    cases = [(list(g['ctypes'][k][g['ctypes'][k] >= 0]), list(g['lengths'][k][g['ctypes'][k] >= 0])) for k in range(0, 1200, 3)] + special
    for ct, ln in cases:
        path = types.SimpleNamespace(ctypes=[names[int(c)] for c in ct], lengths=[float(v) for v in ln])
        pl = pa.RsPlanner(0.05 * 10 * 2.5)
        pl.set_rs_path(path)
        w = np.full(5, -1, np.int8); w[:len(ct)] = ct
        l = np.zeros(5); l[:len(ln)] = ln
        words.append(w); lens.append(l)
        acts.extend([[float(a[0]), float(a[1])] for a in pl.actions])
        off.append(len(acts))
    # choose_action probabilities (np.random.choice intercepted)
    am = ActionMask()
    n = 200
    mean = rng.uniform(-1.2, 1.2, (n, 2)); std = rng.uniform(0.05, 1.5, (n, 2))
    mask = rng.integers(0, 11, (n, 42)) / 10.0
    mask[::7] = 0.01
    probs = np.zeros((n, 42))
    orig = np.random.choice
    for i in range(n):
        cap = {}
        np.random.choice = lambda a, p=None, _c=cap: (_c.__setitem__('p', np.array(p)), 0)[1]
        am.choose_action(mean[i].copy(), std[i].copy(), mask[i].copy())
        probs[i] = cap['p']
    np.random.choice = orig
    # StateNorm recurrence
    sn = StateNorm({'lidar': (120,), 'target': (5,), 'action_mask': (42,)},
                   {'img': False, 'lidar': True, 'target': True, 'action_mask': False})
    seq_l = rng.uniform(0, 10, (300, 120)); seq_t = rng.normal(0, 3, (300, 5))
    outs = []
    for i in range(300):
        o = sn.state_norm({'lidar': seq_l[i].copy(), 'target': seq_t[i].copy(), 'action_mask': np.zeros(42)}, update=True)
        outs.append(o['target'].copy())
    save('agent_glue.npz', words=np.array(words), lengths=np.array(lens), actions=np.array(acts), action_off=np.array(off, np.int32),
         pa_mean=mean, pa_std=std, pa_mask=mask, pa_probs=probs, sn_lidar=seq_l, sn_target=seq_t,
         sn_mean_lidar=np.asarray(sn.state_mean['lidar'], np.float64), sn_std_lidar=np.asarray(sn.state_std['lidar'], np.float64),
         sn_mean_target=np.asarray(sn.state_mean['target'], np.float64), sn_std_target=np.asarray(sn.state_std['target'], np.float64),
         sn_norm_target=np.array(outs), sn_n=np.array(sn.n_state))


def main():
    rng = np.random.default_rng(20240807)
    print('decoding dlp.data')
    d = decode_dlp()
    print('building reference ActionMask / LidarSimlator')
    am = ActionMask()
    lidar = LidarSimlator(C.LIDAR_RANGE, C.LIDAR_NUM)
    gold_constants(am, lidar)
    gold_ksmodel(rng)
    gold_lidar_mask(d, am, lidar, rng)
    gold_rs(rng)
    gold_traj_and_rs_search(d, rng)
    gold_wrapper_reward(rng)
    gold_agent_glue(np.random.default_rng(7))


if __name__ == '__main__':
    main()
