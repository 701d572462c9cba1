"""ctypes front-end of oracle/hope_oracle.c (CPU oracle -- test infrastructure only).

Every function below is a thin numpy wrapper; the arithmetic lives in the C file, whose
functions cite the reference file:line they restate.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NBEAM, NACT, NITER, NL = 120, 42, 10, 1200
CT_NAMES = {0: 'S', 1: 'L', 2: 'R'}


def build(force=False):
    so = os.path.join(_HERE, '_build', 'libhope_oracle_libm.so')
    srcs = [os.path.join(_HERE, 'hope_oracle.c'), os.path.join(_HERE, 'hope_oracle_img.c'),
            os.path.join(os.path.dirname(_HERE), 'hope_amd', 'csrc', 'hope_math.h')]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


_libs = {}
_flavour = {'libm': False}


def use_libm(flag):
    """select the glibc-math build (strict pinning against the reference vectors) for the single-thread library;
    default is the hope_math.h build that is bit-compatible with the HIP kernels."""
    _flavour['libm'] = bool(flag)


def lib(omp=False):
    key = ('libm_omp' if _flavour['libm'] else 'omp') if omp else ('libm' if _flavour['libm'] else 'st')
    if key not in _libs:
        build()
        name = {'omp': 'libhope_oracle_omp.so', 'st': 'libhope_oracle.so', 'libm': 'libhope_oracle_libm.so', 'libm_omp': 'libhope_oracle_libm_omp.so'}[key]
        L = C.CDLL(os.path.join(_HERE, '_build', name))
        L.orc_init()
        L.orc_quad_intersection_area.restype = C.c_double
        L.orc_quad_area.restype = C.c_double
        L.orc_reward_shaping.restype = C.c_double
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def tables():
    L = lib()
    t = dict(actions=np.zeros((NACT, 2)), boxes=np.zeros((NACT, NITER, 4, 2)), hull_base=np.zeros(NBEAM),
             beam_a=np.zeros(NBEAM), beam_b=np.zeros(NBEAM), dist_star=np.zeros((NL, NACT, NITER)))
    L.orc_get_tables(_p(t['actions']), _p(t['boxes']), _p(t['hull_base']), _p(t['beam_a']), _p(t['beam_b']),
                     _p(t['dist_star']))
    return t


def set_tables(hull_base=None, beam_a=None, beam_b=None, dist_star=None, omp=None):
    """inject tables into the oracle (both library flavours unless omp is given)."""
    args = [None if a is None else _f64(a) for a in (hull_base, beam_a, beam_b, dist_star)]
    for flavour in ((False, True) if omp is None else (omp,)):
        lib(flavour).orc_set_tables(*[_p(a) for a in args])


def set_dist_star_coarse(coarse):
    c = _f64(coarse)
    assert c.shape == (NBEAM, NACT, NITER)
    for flavour in (False, True):
        lib(flavour).orc_set_dist_star_coarse(_p(c))


def math_fn(fn, a, b=None):
    """evaluate one of hope_math.h's functions on the host (fn: see orc_math in hope_oracle.c)."""
    a = _f64(a)
    bb = _f64(b) if b is not None else None
    out = np.zeros_like(a)
    lib().orc_math(C.c_int(fn), C.c_int(a.size), _p(a), _p(bb), _p(out))
    return out


def ks_step(pose, action):
    p = _f64(pose).copy()
    ss = np.zeros(2)
    lib().orc_ks_step(_p(p), _p(_f64(action)), _p(ss))
    return p, ss


def create_box(pose):
    b = np.zeros((4, 2))
    lib().orc_create_box(_p(_f64(pose)), _p(b))
    return b


def orient(a, b, c):
    L = lib()
    L.orc_orient.argtypes = [C.c_double] * 6
    return L.orc_orient(*map(float, (*a, *b, *c)))


def segments_intersect(p1, p2, q1, q2):
    L = lib()
    L.orc_segments_intersect.argtypes = [C.c_double] * 8
    return bool(L.orc_segments_intersect(*map(float, (*p1, *p2, *q1, *q2))))


def pt_seg_dist(p, a, b):
    L = lib()
    L.orc_pt_seg_dist.argtypes = [C.c_double] * 6
    L.orc_pt_seg_dist.restype = C.c_double
    return L.orc_pt_seg_dist(*map(float, (*p, *a, *b)))


def ring_intersects(box, ring):
    ring = _f64(ring)
    nv = len(ring)
    r = np.zeros((4, 2))
    r[:nv] = ring
    return bool(lib().orc_ring_intersects(_p(_f64(box)), _p(r), C.c_int(nv)))


def detect_collision(box, verts, nvert):
    verts, nvert = _f64(verts), _i32(nvert)
    return bool(lib().orc_detect_collision(_p(_f64(box)), _p(verts), _p(nvert), C.c_int(len(nvert))))


def quad_intersection_area(a, b):
    return lib().orc_quad_intersection_area(_p(_f64(a)), _p(_f64(b)))


def quad_area(a):
    return lib().orc_quad_area(_p(_f64(a)))


def lidar_fast(ring_verts, nvert):
    rv, nv = _f64(ring_verts), _i32(nvert)
    out = np.zeros(NBEAM)
    lib().orc_lidar_fast(_p(rv), _p(nv), C.c_int(len(nv)), _p(out))
    return out


def lidar_observation(pose, verts, nvert):
    v, nv = _f64(verts), _i32(nvert)
    out = np.zeros(NBEAM)
    lib().orc_lidar_observation(_p(_f64(pose)), _p(v), _p(nv), C.c_int(len(nv)), _p(out))
    return out


def get_steps(scan, hull_base=None, dist_star=None):
    out = np.zeros(NACT)
    hb = _f64(hull_base) if hull_base is not None else None
    ds = _f64(dist_star) if dist_star is not None else None
    lib().orc_get_steps(_p(_f64(scan)), _p(hb), _p(ds), _p(out))
    return out


def rs_all_paths(q0, q1, maxc, step=0.1, max_paths=64):
    nseg = np.zeros(max_paths, np.int32)
    ct = np.zeros((max_paths, 5), np.int32)
    ln = np.zeros((max_paths, 5))
    Ls = np.zeros(max_paths)
    npts = np.zeros(max_paths, np.int32)
    f3 = np.zeros((max_paths, 3, 3))
    l3 = np.zeros((max_paths, 3, 3))
    sums = np.zeros((max_paths, 4))
    n = lib().orc_rs_all_paths(_p(_f64(q0)), _p(_f64(q1)), C.c_double(maxc), C.c_double(step), C.c_int(max_paths),
                               _p(nseg), _p(ct), _p(ln), _p(Ls), _p(npts), _p(f3), _p(l3), _p(sums))
    return dict(n=n, nseg=nseg[:n], ctypes=ct[:n], lengths=ln[:n], L=Ls[:n], npts=npts[:n], first3=f3[:n],
                last3=l3[:n], sums=sums[:n])


def rs_path_samples(q0, q1, maxc, idx, step=0.1, cap=40000):
    xyz = np.zeros((cap, 3))
    n = lib().orc_rs_path_samples(_p(_f64(q0)), _p(_f64(q1)), C.c_double(maxc), C.c_double(step), C.c_int(idx),
                                  C.c_int(cap), _p(xyz))
    return xyz[:max(n, 0)]


def is_traj_valid(traj, verts, nvert, bbox):
    traj, v, nv = _f64(traj), _f64(verts), _i32(nvert)
    return bool(lib().orc_is_traj_valid(_p(traj), C.c_int(len(traj)), _p(v), _p(nv), C.c_int(len(nv)),
                                        _p(_f64(bbox))))


def find_rs_path(pose, dest, verts, nvert, bbox):
    v, nv = _f64(verts), _i32(nvert)
    nseg, ntest = C.c_int32(0), C.c_int32(0)
    ct = np.zeros(5, np.int32)
    ln = np.zeros(5)
    Lr = C.c_double(0)
    found = lib().orc_find_rs_path(_p(_f64(pose)), _p(_f64(dest)), _p(v), _p(nv), C.c_int(len(nv)), _p(_f64(bbox)),
                                   C.byref(nseg), _p(ct), _p(ln), C.byref(Lr), C.byref(ntest))
    return dict(found=bool(found), nseg=nseg.value, ctypes=ct, lengths=ln, L=Lr.value, n_tested=ntest.value)


def target_repr(ego, dest):
    out = np.zeros(5)
    lib().orc_target_repr(_p(_f64(ego)), _p(_f64(dest)), _p(out))
    return out


def reward_terms(prev, cur, dest, start, t, union_area, dest_area, accum):
    out = np.zeros(5)
    acc = C.c_double(accum)
    L = lib()
    L.orc_reward_terms(_p(_f64(prev)), _p(_f64(cur)), _p(_f64(dest)), _p(_f64(start)), C.c_double(t),
                       C.c_double(union_area), C.c_double(dest_area), C.byref(acc), _p(out))
    return out, acc.value


def reward_shaping(info, status):
    return lib().orc_reward_shaping(_p(_f64(info)), C.c_int(int(status)))


def action_rescale(act):
    out = np.zeros(2)
    lib().orc_action_rescale(_p(_f64(act)), _p(out))
    return out


# ---- bird's-eye image observation (hope_oracle_img.c; PARITY UNPINNED, see its header) -------------------------
def bev_world(verts, nvert, start, dest, bbox, pose, traj):
    """the 500x500 world raster of CarParking._render as palette ids; traj = vehicle.trajectory (poses, oldest first)"""
    verts, nvert = _f64(verts), _i32(nvert)
    traj = _f64(traj).reshape(-1, 3)
    last = np.ascontiguousarray(traj[-20:])
    world = np.zeros((500, 500), np.uint8)
    lib().orc_bev_world(_p(verts), _p(nvert), C.c_int(len(nvert)), _p(_f64(start)), _p(_f64(dest)), _p(_f64(bbox)),
                        _p(_f64(pose)), _p(last), C.c_int(len(last)), C.c_int(len(traj)), _p(world))
    return world


def bev_raw(world, bbox, pose):
    """_get_img_observation: [256,256,3] uint8"""
    raw = np.zeros((256, 256, 3), np.uint8)
    lib().orc_bev_raw(_p(np.ascontiguousarray(world, np.uint8)), _p(_f64(bbox)), _p(_f64(pose)), _p(raw))
    return raw


def bev_process(raw):
    """Obs_Processor.process_img * 255: [64,64,3] uint8"""
    out = np.zeros((64, 64, 3), np.uint8)
    lib().orc_bev_process(_p(np.ascontiguousarray(raw, np.uint8)), _p(out))
    return out


def bev_image(verts, nvert, start, dest, bbox, pose, traj, omp=False):
    """obs['img'] * 255 after the wrapper's transpose: [3,64,64] uint8"""
    verts, nvert = _f64(verts), _i32(nvert)
    traj = _f64(traj).reshape(-1, 3)
    last = np.ascontiguousarray(traj[-20:])
    out = np.zeros((3, 64, 64), np.uint8)
    lib(omp).orc_bev_image(_p(verts), _p(nvert), C.c_int(len(nvert)), _p(_f64(start)), _p(_f64(dest)), _p(_f64(bbox)),
                           _p(_f64(pose)), _p(last), C.c_int(len(last)), C.c_int(len(traj)), _p(out))
    return out


class BatchOracle:
    """N independent scenes stepped by the C oracle (fixed stride of max_obst obstacles/scene).
    Mirrors the product's batch interface so parity tests can run both on identical inputs."""

    def __init__(self, n, max_obst, omp=False, track_traj=True):
        self.n, self.max_obst = n, max_obst
        self.track_traj = track_traj                  # False: no vehicle.trajectory bookkeeping (a Python loop over the scenes; bench.py's timed baseline)
        self.L = lib(omp)
        self.n_obst = np.zeros(n, np.int32)
        self.verts = np.zeros((n, max_obst, 4, 2))
        self.nvert = np.full((n, max_obst), 4, np.int32)
        self.start = np.zeros((n, 3))
        self.dest = np.zeros((n, 3))
        self.bbox = np.zeros((n, 4))
        self.pose = np.zeros((n, 3))
        self.t = np.zeros(n)
        self.accum = np.zeros(n)
        self.out = dict(lidar=np.zeros((n, NBEAM)), mask=np.zeros((n, NACT)), target=np.zeros((n, 5)),
                        reward_info=np.zeros((n, 5)), reward=np.zeros(n), status=np.zeros(n, np.int32),
                        rs_found=np.zeros(n, np.int32), rs_ctypes=np.zeros((n, 5), np.int32),
                        rs_lengths=np.zeros((n, 5)), substeps=np.zeros(n, np.int32))
        self.traj = [[] for _ in range(n)] if track_traj else None        # vehicle.trajectory (vehicle.py:114,132-133,144,158), last 20 kept

    def set_scenes(self, ids, start, dest, bbox, verts, nvert, n_obst):
        ids = np.asarray(ids)
        self.start[ids] = start
        self.dest[ids] = dest
        self.bbox[ids] = bbox
        self.n_obst[ids] = n_obst
        for k, i in enumerate(ids):
            m = int(n_obst[k])
            self.verts[i, :m] = verts[k][:m]
            self.nvert[i, :m] = nvert[k][:m]
        self.restart(ids)

    def restart(self, ids):
        """CarParking.reset's state part (:128-136): accumulators zeroed, vehicle.reset(start), trajectory = [start]"""
        ids = np.asarray(ids)
        self.pose[ids] = self.start[ids]
        self.t[ids] = 0.0
        self.accum[ids] = 0.0
        if self.track_traj:
            for i in ids:
                self.traj[int(i)] = [self.start[int(i)].copy()]

    def _call(self, actions, with_rs):
        o = self.out
        a = _f64(actions) if actions is not None else None
        self.L.orc_batch_step(C.c_int(self.n), C.c_int(self.max_obst), _p(self.n_obst), _p(self.verts), _p(self.nvert),
                              _p(self.start), _p(self.dest), _p(self.bbox), _p(self.pose), _p(self.t), _p(self.accum),
                              _p(a), C.c_int(int(with_rs)), _p(o['lidar']), _p(o['mask']), _p(o['target']),
                              _p(o['reward_info']), _p(o['reward']), _p(o['status']), _p(o['rs_found']),
                              _p(o['rs_ctypes']), _p(o['rs_lengths']), _p(o['substeps']))
        if actions is not None and self.track_traj:
            # car_parking_base.py:259-276: of the sub-step states only the last kept one stays in vehicle.trajectory
            for i in np.nonzero(o['substeps'] > 0)[0]:
                tr = self.traj[int(i)]
                tr.append(self.pose[i].copy())
                if len(tr) > 21:                   # keep len > 20 distinguishable from len == 20; only 20 are drawn
                    del tr[0]
        return o

    def image(self, ids=None):
        """obs['img'] * 255 for the given scenes: [len(ids),3,64,64] uint8"""
        ids = np.arange(self.n) if ids is None else np.asarray(ids)
        out = np.zeros((len(ids), 3, 64, 64), np.uint8)
        for k, i in enumerate(ids):
            m = int(self.n_obst[i])
            out[k] = bev_image(self.verts[i, :m], self.nvert[i, :m], self.start[i], self.dest[i], self.bbox[i],
                               self.pose[i], np.array(self.traj[int(i)]))
        return out

    def reset_obs(self, with_rs=True):
        """the action-less step every scene performs at reset (t: 0 -> 1)."""
        return self._call(None, with_rs)

    def step(self, actions, with_rs=True):
        return self._call(actions, with_rs)
