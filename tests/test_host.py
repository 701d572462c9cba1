"""CPU-side tests (-m "not gpu"): host tables, scene sources, the C-ABI library's exported symbols,
wrapper arithmetic, path sampler.  No kernel is launched here."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_tables_bit_identical_to_reference(gold):
    """hope_amd.tables restates ActionMask.__init__ with numpy: on the numpy build that produced the
    fixtures the tables carry the reference's own rounding (sha256 of the 4 MB dist_star)."""
    from hope_amd import tables as T
    g = gold('constants.npz')
    t = T.all_tables()
    assert np.array_equal(t['actions'], g['discrete_actions'])
    assert np.abs(T.swept_boxes() - g['vehicle_boxes']).max() < 1e-12
    assert np.abs(T.dist_star_coarse() - g['dist_star_coarse']).max() < 1e-12
    assert np.abs(t['dist_star'][::7] - g['dist_star_every7']).max() < 1e-12
    assert abs(t['dist_star'].sum() - float(g['dist_star_sum'])) < 1e-6
    assert np.abs(t['beam_ab'][:, 0] - g['beam_a']).max() < 1e-15 and np.abs(t['beam_ab'][:, 1] - g['beam_b']).max() < 1e-15
    if np.array_equal(T.dist_star_coarse(), g['dist_star_coarse']):        # same numpy SIMD path as the fixture
        assert hashlib.sha256(t['dist_star'].tobytes()).hexdigest() == str(g['dist_star_sha256'])
    assert t['hull_base'][0] == 3.76 and abs(t['hull_base'][30] - 0.97) < 1e-15 and abs(t['hull_base'][60] - 0.93) < 1e-15


def test_tables_agree_with_oracle():
    from hope_amd import tables as T
    from oracle import oracle as O
    t, o = T.all_tables(), O.tables()
    assert np.array_equal(t['actions'], o['actions'])
    assert np.abs(t['dist_star'] - o['dist_star']).max() < 1e-12
    assert np.abs(t['hull_base'] - o['hull_base']).max() < 1e-14
    assert np.abs(t['beam_ab'][:, 0] - o['beam_a']).max() < 1e-15


def test_library_exports_every_declared_symbol():
    """the C-ABI library loads (no GPU needed for dlopen) and exports exactly what include/hope_env.h declares."""
    from hope_amd import build_extension, lib_path
    from hope_amd import _lib
    build_extension()
    hdr = open(os.path.join(ROOT, 'include', 'hope_env.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(hope_[a-z_0-9]+)\s*\(', hdr))
    declared = {d for d in declared if not d.endswith('_t')}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    L = _lib.load_library()
    assert L.hope_abi_version() == _lib.ABI_VERSION == 8


def test_no_cpu_fallback():
    """without a HIP device the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from hope_amd import _lib
    L = _lib.load_library()
    h = ctypes.c_void_p()
    rc = L.hope_env_create(ctypes.byref(h), 4, 16, 0, 0)
    assert rc == -2 and b'no CPU fallback' in L.hope_last_error()
    from hope_amd import ParkingBatch
    with pytest.raises(_lib.HopeError):
        ParkingBatch(4, 16)
    src = ''.join(open(os.path.join(ROOT, 'hope_amd', f)).read() for f in os.listdir(os.path.join(ROOT, 'hope_amd'))
                  if f.endswith('.py'))
    assert 'oracle' not in src.replace('never route through the oracle', '')


def test_dlp_pool_cull_rule_matches_reference_fixture(gold, dlp):
    """filter_obstacles (parking_map_dlp.py:88-101): the kept-obstacle lists recorded from the reference run."""
    from hope_amd.scenes import DlpScenePool, cull_obstacles
    pool = DlpScenePool()
    s = gold('rs_search.npz')
    for ri in range(len(s['case'])):
        v, nv = pool.obstacles(int(s['case'][ri]))
        keep = cull_obstacles(v, nv, s['bbox'][ri])
        want = s['keep'][s['keep_off'][ri]:s['keep_off'][ri + 1]]
        assert np.array_equal(keep, want), ri


def test_dlp_sample_semantics():
    from hope_amd.scenes import DlpScenePool, create_box, flip_orientation
    pool = DlpScenePool()
    assert len(pool) == 248
    rng = np.random.default_rng(1)
    for _ in range(50):
        s = pool.sample(rng=rng)
        assert 3 <= s.nvert.min() and s.nvert.max() <= 4 and 30 <= s.n_obst <= 128
        assert s.bbox[0] == np.floor(s.bbox[0]) and s.bbox[1] - s.bbox[0] >= 40
    p = np.array([3.0, 4.0, 0.7])
    q = flip_orientation(p)
    assert np.allclose(np.sort(create_box(p), axis=0), np.sort(create_box(q), axis=0))
    assert abs(q[2] - p[2] - np.pi) < 1e-15


def test_normal_generator_scenes_are_valid():
    from hope_amd.scenes import generate_scene, rings_distance, rings_intersect, create_box
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    kinds = set()
    for lv in ('Normal', 'Complex', 'Extrem'):
        for _ in range(25):
            s = generate_scene(lv, rng)
            kinds.add((lv, s.case_id))
            assert not O.detect_collision(create_box(s.start), s.verts, s.nvert)
            assert not O.detect_collision(create_box(s.dest), s.verts, s.nvert)
            assert 3 <= s.n_obst <= 17 and (s.nvert == 4).all()
            d = np.hypot(*(s.start[:2] - s.dest[:2]))
            assert d < 25 and s.bbox[0] <= min(s.start[0], s.dest[0]) - 10
            # slot clearance rule of the rejection sampler
            gl, gr = rings_distance(create_box(s.dest), s.verts[1]), rings_distance(create_box(s.dest), s.verts[2])
            assert gl >= 0.1 - 1e-12 and gr >= 0.1 - 1e-12
    assert ('Extrem', 0) not in kinds and ('Normal', 0) in kinds and ('Normal', 1) in kinds
    # host geometry agrees with the oracle's GEOS-semantics predicate
    for _ in range(200):
        a = create_box((rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-3, 3)))
        b = create_box((rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-3, 3)))
        assert rings_intersect(a, b) == O.ring_intersects(a, b)


def test_wrapper_surface_is_built_on_kernel_outputs():
    """The wrapper's reward_shaping / action_rescale are not re-derived on the host: the arithmetic lives in
    the kernels (pinned to the reference's vectors through the oracle, tests/test_oracle_golden.py, and to the oracle
    on the GPU, tests/test_gpu_dropin.py); the module keeps the reference's class surface and the Status values."""
    import inspect
    from hope_amd import env as E
    assert [s.value for s in E.Status] == [1, 2, 3, 4, 5]
    assert not hasattr(E, 'reward_shaping') and not hasattr(E, 'action_rescale')
    sig = inspect.signature(E.CarParkingWrapper.__init__)
    assert list(sig.parameters)[1:] == ['env', 'action_func', 'reward_func', 'observation_func']     # env_wrapper.py:59-60
    assert list(E.REWARD_WEIGHT.keys()) == ['time_cost', 'rs_dist_reward', 'dist_reward', 'angle_reward', 'box_union_reward']


def test_rs_path_sampler_matches_reference_vectors(gold):
    """hope_amd.rs_path.PATH regenerates x/y/yaw from (ctypes, lengths): compare with calc_all_paths fixtures."""
    from hope_amd.rs_path import PATH, MAXC
    g = gold('reeds_shepp.npz')
    assert MAXC == float(g['maxc'])
    off = g['path_off']
    for i in range(0, 300):
        for k in range(off[i], off[i + 1]):
            n = int((g['ctypes'][k] >= 0).sum())
            p = PATH(g['lengths'][k][:n], ['SLR'[c] for c in g['ctypes'][k][:n]], g['q0'][i])
            assert len(p.x) == int(g['npts'][k]), (i, k)
            m = min(3, len(p.x))
            got = np.column_stack([p.x[-m:], p.y[-m:], p.yaw[-m:]])
            assert np.abs(got - g['last3'][k][3 - m:]).max() < 1e-9
            assert abs(p.L - g['L'][k]) < 1e-9


def test_dlp_pickle_reader_roundtrip(tmp_path, dlp):
    """a reference-format pickle (shapely-1.x LinearRing = WKB LineString bytes) is readable without shapely."""
    import pickle
    import struct
    import sys
    import types
    mod = types.ModuleType('shapely.geometry.polygon')

    class LinearRing:
        def __init__(self, c):
            self.c = np.asarray(c, dtype='<f8')

        def __getstate__(self):          # shapely 1.x: the pickled state is the WKB of a LineString
            c = np.vstack([self.c, self.c[:1]])
            return b'\x01' + struct.pack('<II', 2, len(c)) + c.tobytes()
    LinearRing.__module__ = 'shapely.geometry.polygon'
    LinearRing.__qualname__ = 'LinearRing'
    mod.LinearRing = LinearRing
    pkgs = {n: types.ModuleType(n) for n in ('shapely', 'shapely.geometry')}
    sys.modules.update(pkgs)
    sys.modules['shapely.geometry.polygon'] = mod
    try:
        rings = [LinearRing([(0, 0), (2, 0), (2, 1), (0, 1)]), LinearRing([(5, 5), (6, 5), (5.5, 7)])]
        case = ([(1.0, 2.0, 0.1), (1.5, 2.0, 0.2)], (9.0, 9.0, 1.57), rings)
        f = tmp_path / 'mini.data'
        f.write_bytes(pickle.dumps([case]))
    finally:
        for n in ('shapely', 'shapely.geometry', 'shapely.geometry.polygon'):
            sys.modules.pop(n, None)
    from hope_amd.dlp_io import pool_from_pickle
    pool = pool_from_pickle(str(f))
    v, nv = pool.obstacles(0)
    assert list(nv) == [4, 3] and np.array_equal(v[1, 3], v[1, 2]) and np.array_equal(v[0, 2], [2, 1])
    assert np.allclose(pool.dest[0], [9, 9, 1.57]) and len(pool.candidates(0)) == 2


def _dist_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from hope_amd import dist as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env('gloo')
    lo, hi = D.shard_range(11, r, w)
    torch.manual_seed(0)
    net = torch.nn.Linear(7, 3)
    x = torch.arange(11 * 7, dtype=torch.float32).reshape(11, 7)[lo:hi] / 50
    loss = net(x).pow(2).sum() / 11           # per-rank share of a global-batch loss
    loss.backward()
    nbytes = D.allreduce_gradients(list(net.parameters()), average=False)
    n_local = hi - lo
    rec = D.gather_eval_stats(torch.full((n_local,), 2 if r == 0 else 3, dtype=torch.int32),
                              torch.arange(lo, hi, dtype=torch.int32), torch.ones(n_local), torch.zeros(n_local))
    q.put((r, lo, hi, net.weight.grad.clone().numpy(), nbytes, rec.numpy(), D.success_rate(rec)))
    dist.destroy_process_group()


def test_distributed_helpers_gloo_world2():
    """N > 1 path on CPU: scene sharding, one fused gradient bucket, eval-stat gather (gloo, world_size 2)."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, lo0, hi0, g0, nb0, rec0, sr0), (r1, lo1, hi1, g1, nb1, rec1, sr1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)
    torch.manual_seed(0)
    net = torch.nn.Linear(7, 3)
    x = torch.arange(11 * 7, dtype=torch.float32).reshape(11, 7) / 50
    (net(x).pow(2).sum() / 11).backward()
    assert np.allclose(g0, net.weight.grad.numpy(), rtol=1e-5) and np.allclose(g0, g1)
    assert nb0 == (7 * 3 + 3) * 4
    assert rec0.shape == (11, 4) and np.array_equal(rec0, rec1) and np.array_equal(rec0[:, 1], np.arange(11))
    assert abs(sr0 - 6 / 11) < 1e-6
