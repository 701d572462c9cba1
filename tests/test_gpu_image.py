"""-m gpu: the bird's-eye image observation kernel (HOPE_STAGE_IMG, k_bev_image) against oracle/hope_oracle_img.c.
Integer pixels: the comparison is exact (uint8 equality on every one of the 3 x 64 x 64 values)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def make_img_pair(scenes, max_obst=128):
    from hope_amd import ParkingBatch
    from hope_amd.scenes import pack_scenes
    from oracle import oracle as O
    n = len(scenes)
    env = ParkingBatch(n, max_obst, obs_dtype=torch.float64, action_dtype=torch.float64, image=True)
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, max_obst)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, max_obst)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    return env, orc


def assert_images_equal(env, orc, what, ids=None):
    torch.cuda.synchronize()
    got = env.img.cpu().numpy()
    want = orc.image(ids)
    if ids is not None:
        got = got[ids]
    bad = np.nonzero((got != want).reshape(len(got), -1).any(1))[0]
    assert len(bad) == 0, f'{what}: {len(bad)} of {len(got)} images differ, first scene {bad[0]}, ' \
                          f'{int((got[bad[0]] != want[bad[0]]).sum())} values'
    return got


def restart_done(env, orc):
    """finished episodes start over on the same map, in both implementations"""
    from oracle import oracle as O
    done = env.done.cpu().numpy().astype(bool)
    if not done.any():
        return 0
    ids = np.nonzero(done)[0]
    env.restart(env.done)
    env.reset_obs(active=env.done.clone())
    orc.restart(ids)
    sub = O.BatchOracle(len(ids), orc.max_obst)
    sub.set_scenes(np.arange(len(ids)), orc.start[ids], orc.dest[ids], orc.bbox[ids], orc.verts[ids], orc.nvert[ids], orc.n_obst[ids])
    sub.reset_obs()
    orc.pose[ids] = sub.pose; orc.t[ids] = sub.t; orc.accum[ids] = sub.accum
    return len(ids)


@pytest.mark.parametrize('level', ['mixed', 'dlp'])
def test_image_rollout_matches_oracle(level):
    from hope_amd.scenes import DlpScenePool, SceneSource
    rng = np.random.default_rng(7)
    n = 160
    if level == 'dlp':
        pool = DlpScenePool()
        scenes = [pool.sample(rng=rng) for _ in range(n)]
    else:
        src = SceneSource(seed=17)
        scenes = [src.draw() for _ in range(n)]
    for k in range(0, n, 3):                      # a third start close to the destination
        s = scenes[k]
        r, a = rng.uniform(0.0, 6.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    env, orc = make_img_pair(scenes)
    env.reset_obs(); orc.reset_obs()
    img = assert_images_equal(env, orc, 'reset')
    assert (img[:, :, 32, 32] == np.array([30, 144, 255])).all()          # len(trajectory) == 1: vehicle colour
    restarts = 0
    for it in range(28):                           # > 20 steps: the trajectory ring wraps
        act = rng.uniform(-1.1, 1.1, (n, 2))
        if it < 6:
            act[:, 1] = np.abs(act[:, 1]) * (1 if it % 2 == 0 else -1)
        env.step(torch.from_numpy(act).to(env.device))
        o = orc.step(act)
        torch.cuda.synchronize()
        assert np.array_equal(env.status.cpu().numpy(), o['status'])
        assert np.array_equal(env.pose.cpu().numpy(), orc.pose)
        assert_images_equal(env, orc, f'step {it}')
        restarts += restart_done(env, orc)
        assert_images_equal(env, orc, f'step {it} after restarts')
    lens = np.array([len(t) for t in orc.traj])
    print(f'{level}: restarts {restarts}, trajectory lengths max {lens.max()}')
    assert lens.max() >= 21 and restarts > 0
    env.close()


def test_image_special_headings_and_background_pixel():
    """rotate90 path (heading an exact multiple of 90 degrees in float32), the rotate() background taken from the
    surface's top-left pixel, cars far outside the window, one-pixel-high and degenerate polygons"""
    from hope_amd.scenes import Scene
    quarter = [float(np.float64(np.float32(90.0 * k)) * np.pi / 180.0) for k in (0, 1, 2, 3, -1)]
    scenes = []

    def rect(x0, y0, x1, y1):
        return np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], float)
    obst = [rect(3, 2, 7, 4), rect(-6, -5, -4, 3), np.array([[0, 6], [3, 6.5], [1, 8.0]]), rect(-22, -22, -19, -19),
            rect(-1, -9, 8, -8.93), rect(10, 10, 10.05, 10.05)]
    verts = np.zeros((len(obst), 4, 2))
    nvert = np.array([len(o) for o in obst], np.int32)
    for i, o in enumerate(obst):
        verts[i, :len(o)] = o
        verts[i, len(o):] = o[-1]

    def mk(start, half):
        return Scene(start=np.array(start, float), dest=np.array([6.0, -6.0, 0.4]), bbox=np.array([-half, half, -half, half], float),
                     verts=verts.copy(), nvert=nvert.copy(), level='Normal')
    for h in quarter + [0.3, -2.0, 1e-9, np.pi]:
        scenes.append(mk([0.5, -1.0, h], 25.0))
    # near the surface corner: samples fall outside the source; pixel (0,0) is covered by the 4th obstacle
    for h in (0.7, 2.5, -1.1):
        scenes.append(mk([-15.0, -15.0, h], 25.0))
    # far outside the 500 x 500 surface (map larger than the window): mostly fill colour
    scenes.append(mk([-40.0, 30.0, 0.9], 50.0))
    env, orc = make_img_pair(scenes, max_obst=16)
    env.reset_obs(); orc.reset_obs()
    img = assert_images_equal(env, orc, 'special reset')
    rng = np.random.default_rng(3)
    for it in range(6):
        act = rng.uniform(-1, 1, (len(scenes), 2))
        act[: len(quarter), 0] = 0.0                # straight: the heading stays an exact multiple of 90 degrees
        env.step(torch.from_numpy(act).to(env.device)); orc.step(act)
        assert_images_equal(env, orc, f'special step {it}')
    grey = (img[-2] == 150).all(0).mean()
    assert grey > 0.02, 'the corner scene should show the top-left pixel colour outside the source'
    env.close()


def test_image_with_auto_reset_and_active_mask():
    """HOPE_AUTO_RESET: the image is the NEW episode's first observation, like lidar / mask / target"""
    from hope_amd.scenes import SceneSource
    src = SceneSource(levels=('Normal', 'Complex'), seed=5)
    scenes = [src.draw() for _ in range(96)]
    envA, orc = make_img_pair(scenes, max_obst=32)
    envB, _ = make_img_pair(scenes, max_obst=32)
    envA.reset_obs(); envB.reset_obs(); orc.reset_obs()
    rng = np.random.default_rng(1)
    n_done = 0
    for it in range(45):
        act = rng.uniform(-1.2, 1.2, (96, 2))
        act[:, 1] = np.sign(act[:, 1] + 1e-9)
        a = torch.from_numpy(act).to(envA.device)
        envA.step(a, auto_reset=True)
        envB.step(a)
        orc.step(act)
        torch.cuda.synchronize()
        n_done += int(envB.done.sum())
        restart_done(envB, orc)
        torch.cuda.synchronize()
        assert torch.equal(envA.img, envB.img)
        assert_images_equal(envA, orc, f'auto-reset step {it}')
    assert n_done >= 4
    # active mask: untouched scenes keep their previous image
    before = envA.img.clone()
    mask = torch.zeros(96, dtype=torch.uint8, device=envA.device)
    mask[::2] = 1
    envA.step(torch.from_numpy(rng.uniform(-1, 1, (96, 2))).to(envA.device), active=mask)
    torch.cuda.synchronize()
    assert torch.equal(envA.img[1::2], before[1::2])
    assert not torch.equal(envA.img[::2], before[::2])
    envA.close(); envB.close()


def test_image_stage_needs_the_image_flag():
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import SceneSource
    env = ParkingBatch(4, 32)
    env.set_scenes(np.arange(4), [SceneSource(seed=1).draw() for _ in range(4)])
    with pytest.raises(L.HopeError):
        env.reset_obs(stages=L.STAGE_ALL | L.STAGE_IMG)
    env.close()


def test_dropin_env_returns_the_image_like_the_reference():
    """CarParking(use_img_observation=True): obs['img'] is (64, 64, 3) float64 in [0, 1] from the raw env and (3, 64, 64)
    after CarParkingWrapper (env_wrapper.py:52-55,68-71); values = oracle image / 255."""
    from hope_amd.env import CarParking, CarParkingWrapper
    from hope_amd.scenes import SceneSource, pack_scenes
    from oracle import oracle as O
    scene = SceneSource(levels=('Normal',), seed=23).draw()
    raw = CarParking(render_mode='rgb_array', fps=100, verbose=False, use_img_observation=True, max_obstacles=32)
    env = CarParkingWrapper(raw)
    assert env.observation_shape['img'] == (3, 64, 64) and raw.observation_space['img'].shape == (64, 64, 3)
    orc = O.BatchOracle(1, 32)
    start, dest, bbox, verts, nob, nvert = pack_scenes([scene], 32)
    orc.set_scenes([0], start, dest, bbox, verts, nvert, nob)
    t = raw._batch.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    obs = raw.reset_to_scene(scene)
    orc.reset_obs()
    assert obs['img'].shape == (64, 64, 3) and obs['img'].dtype == np.float64
    assert np.array_equal(obs['img'], orc.image()[0].transpose(1, 2, 0) / 255.0)
    rng = np.random.default_rng(2)
    for it in range(8):
        a = rng.uniform(-1, 1, 2)
        obs, reward, done, info = env.step(a)
        orc.step(a[None])
        assert list(obs.keys()) == ['img', 'lidar', 'target', 'action_mask'] and obs['img'].shape == (3, 64, 64)
        assert np.array_equal(obs['img'], orc.image()[0] / 255.0)
        assert 0.0 <= obs['img'].min() and obs['img'].max() <= 1.0
    env.close()


def test_image_with_more_than_128_obstacles():
    """max_obstacles > 128: the obstacles beyond the two register-resident chunks are converted again per tile"""
    from hope_amd.scenes import Scene
    rng = np.random.default_rng(4)
    n = 150
    verts = np.zeros((n, 4, 2))
    k = 0
    for gx in range(15):
        for gy in range(10):
            cx, cy = -14 + 2.0 * gx, -9 + 2.0 * gy
            if abs(cx) < 4 and abs(cy) < 3:
                cx += 30                                     # keep the start area free
            a = rng.uniform(0, np.pi)
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            verts[k] = np.array([cx, cy]) + (np.array([[-0.5, -0.3], [0.5, -0.3], [0.5, 0.3], [-0.5, 0.3]]) @ R.T)
            k += 1
    scenes = [Scene(start=np.array([0.0, 0.0, h]), dest=np.array([3.0, 6.5, 1.0]), bbox=np.array([-20.0, 20.0, -20.0, 20.0]),
                    verts=verts.copy(), nvert=np.full(n, 4, np.int32), level='Normal') for h in (0.2, 1.7, -2.4)]
    env, orc = make_img_pair(scenes, max_obst=160)
    env.reset_obs(); orc.reset_obs()
    assert_images_equal(env, orc, '150 obstacles reset')
    for it in range(4):
        act = rng.uniform(-1, 1, (3, 2))
        env.step(torch.from_numpy(act).to(env.device)); o = orc.step(act)
        torch.cuda.synchronize()
        assert np.array_equal(env.status.cpu().numpy(), o['status'])
        assert_images_equal(env, orc, f'150 obstacles step {it}')
    img = env.img.cpu().numpy()
    assert (img == 150).any()                               # obstacles are visible
    env.close()


def test_image_rollout_with_the_per_tile_raster_of_the_moving_boxes(monkeypatch):
    """The trajectory layer replaced the per-tile raster of the moving boxes for plain car-shaped boxes; the raster stays as the
    exact path for any other box shape and must keep producing the same pixels (HOPE_BEV_LEGACY forces it)."""
    monkeypatch.setenv('HOPE_BEV_LEGACY', '1')
    test_image_rollout_matches_oracle('mixed')


def test_image_trajectory_layer_over_a_long_episode():
    """More than 192 trajectory entries without a reset: the codes of the trajectory layer wrap and the layer is cleared and
    repainted every 96 entries (hope_bev.hip: DYN_MOD, DYN_REFRESH).  Every uint8 equal to the oracle's."""
    from hope_amd.scenes import SceneSource
    rng = np.random.default_rng(23)
    n = 48
    src = SceneSource(levels=('Normal',), seed=29)
    scenes = [src.draw() for _ in range(n)]
    env, orc = make_img_pair(scenes, max_obst=32)
    env.reset_obs(); orc.reset_obs()
    mask = torch.ones(n, dtype=torch.uint8, device=env.device)
    checked = 0
    pushes = np.ones(n, np.int64)                   # entries of vehicle.trajectory so far (reset leaves [start])
    for it in range(236):
        # slow wiggles: few collisions, the car moves nearly every step; finished scenes are NOT restarted, they keep stepping
        act = np.stack([rng.uniform(-1, 1, n), 0.3 * np.where((it // 7) % 2 == 0, 1.0, -1.0) * np.ones(n)], 1)
        before = orc.pose.copy()
        env.step(torch.from_numpy(act).to(env.device))
        orc.step(act)
        pushes += (orc.pose != before).any(axis=1)  # (a step that kept >= 1 sub-step appends its final state)
        if it % 9 == 0 or 90 <= it <= 100 or it >= 186:
            assert_images_equal(env, orc, f'step {it}')
            checked += 1
    assert pushes.max() > 200 and checked > 60, (pushes.max(), checked)
    env.close()


def test_image_trajectory_torus_falls_back_when_the_drawn_boxes_span_more_than_256_px():
    """The trajectory layer is a 256 x 256 px torus (hope_bev.hip): twenty consecutive boxes of a car that drives straight at full
    speed span 19 x 15 px + a box = up to ~345 px, more than the torus can hold without aliasing -- the episode must then switch to
    the exact per-tile raster.  Empty 80 m lots, cars driving straight (and diagonally, both axes) for 40 steps at +-2.5 m/s from
    poses all over the surface, plus slow ones that stay on the layer: every uint8 equal to the oracle's on every step."""
    from hope_amd.scenes import Scene
    rng = np.random.default_rng(5)
    scenes = []
    heads = [0.0, np.pi / 2, np.pi / 4, -3 * np.pi / 4, np.pi, 0.3, -1.2, 2.5]
    for k in range(32):
        h = heads[k % len(heads)]
        # the map box decides the surface's offset; start in the middle so that 40 steps of 1.25 m stay inside the 41.6 m surface
        c = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5)])
        start = np.array([c[0] - 18 * np.cos(h), c[1] - 18 * np.sin(h), h])
        dest = np.array([c[0] + 19 * np.cos(h + 1.3), c[1] + 19 * np.sin(h + 1.3), h + 0.7])
        far = np.array([[[60.0, 60.0], [61.0, 60.0], [61.0, 61.0], [60.0, 61.0]]])                    # one obstacle out of the way
        bbox = np.array([min(start[0], dest[0]) - 25, max(start[0], dest[0]) + 25, min(start[1], dest[1]) - 25, max(start[1], dest[1]) + 25])
        scenes.append(Scene(start=start, dest=dest, bbox=np.round(bbox), verts=far, nvert=np.array([4], np.int32), level='Normal'))
    env, orc = make_img_pair(scenes, max_obst=32)
    env.reset_obs(); orc.reset_obs()
    n = len(scenes)
    fast = np.arange(n) % 4 != 3                     # three of four cars at full speed, the fourth crawls (stays on the layer)
    span = 0.0
    p0 = orc.pose.copy()
    for it in range(40):
        act = np.stack([np.zeros(n), np.where(fast, 1.0, 0.15) * np.where((np.arange(n) // 8) % 2 == 0, 1.0, -1.0)], 1)
        env.step(torch.from_numpy(act).to(env.device))
        orc.step(act)
        assert_images_equal(env, orc, f'step {it}')
        span = max(span, float(np.abs(orc.pose[:, :2] - p0[:, :2]).max()))
    assert span * 12 > 300, span                      # some car did move more than the torus is wide
    env.close()


def test_gpu_image_obeys_the_analytic_invariants_of_the_reference_geometry():
    """Row f-1: the HIP image itself -- not only its equality with the restated renderer -- is anchored to what the
    reference's code prescribes: over 3 x 256 random scenes the vehicle / dest / obstacle pixels sit within 1 px of
    31.5 + R(heading) K (P - C) / 4, keep their area at 3 px/m, colours are convex combinations of the palette and the
    background is exactly (0, 0, 0); after driving, the centre shows the newest trajectory colour.  tests/image_invariants.py."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import image_invariants as I
    from hope_amd import ParkingBatch
    rng = np.random.default_rng(12)
    n = 256
    seen = {'dest': 0, 'obstacle': 0}
    for kind in ('vehicle', 'dest', 'obstacle'):
        made = [I.make_scene(rng, kind) for _ in range(n)]
        env = ParkingBatch(n, 32, obs_dtype=torch.float64, action_dtype=torch.float64, image=True)
        env.set_scenes(np.arange(n), [m[0] for m in made])
        env.reset_obs()
        torch.cuda.synchronize()
        imgs = env.img.cpu().numpy()
        for k, (sc, info) in enumerate(made):
            what = f'{kind} {k}'
            if kind == 'vehicle':
                I.check_vehicle_only(imgs[k], what)
            elif kind == 'dest':
                seen['dest'] += I.check_dest(imgs[k], sc.start, info, what) is not None
            else:
                seen['obstacle'] += I.check_obstacle(imgs[k], sc.start, info, what) is not None
        if kind == 'vehicle':
            # three steps straight ahead: the vehicle box is the newest trajectory box -> TRAJ_COLORS[-1] at the centre, and the
            # trail lies BEHIND the car (towards - col), nothing ahead of the nose
            for _ in range(3):
                env.step(torch.tensor([[0.0, 0.6]] * n, dtype=torch.float64, device=env.device))
            torch.cuda.synchronize()
            imgs = env.img.cpu().numpy()
            moved = (env.pose.cpu().numpy()[:, :2] != np.array([m[0].start[:2] for m in made])).any(axis=1)
            assert moved.all()
            for k in range(n):
                assert tuple(imgs[k][:, 31, 31]) == I.TRAJ_NEWEST and tuple(imgs[k][:, 32, 32]) == I.TRAJ_NEWEST, k
                assert not imgs[k][:, 22:42, 42:54].any(), k                # right ahead of the nose (7 px + borders): empty
                assert imgs[k][:, 29:35, 14:24].any(), k                    # behind: the older boxes / the start outline
        env.close()
    assert seen['dest'] > 200 and seen['obstacle'] > 200, seen


def test_image_in_pipelined_steps_equals_the_joined_step():
    """Pipelined steps (HOPE_DEFER_RS with the overlap streams: the image is rendered on the caller's stream, and the static-layer
    rebuild of the scenes that got a NEW map -- k_bev_list + k_bev_static -- runs on a side stream next to k_bev_prep, joined in
    front of the image launches) give the same image bytes as the joined step, every step, with episode turnover on new maps from
    the pool / Dragon-Lake cases; and so do all other outputs."""
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    n = 4096
    arrs = mixed_arrays(n, seed=31, max_obst=128)
    parts = [generate_arrays(lv, 128, seed=40 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
    pool = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6))
    envs = [ParkingBatch(n, 128, overlap=True, image=True) for _ in range(2)]
    for e in envs:
        e.set_scene_arrays(np.arange(n), *arrs[:5])
        e.set_draw_class(np.arange(3, n, 4), 1)
        e.set_pool(pool)
        e.set_dlp_cases()
        e.set_redraw_seed(9)
        e.reset_obs()
    torch.cuda.synchronize()
    assert torch.equal(envs[0].img, envs[1].img)
    g = torch.Generator(device='cuda').manual_seed(3)
    turnovers = 0
    for it in range(40):
        a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
        a[:, 1] = torch.sign(a[:, 1] + 1e-9)                 # full speed: episodes end (collisions, out of bounds) within the run
        envs[0].step(a, auto_reset=True, fresh=True)
        envs[1].step(a, auto_reset=True, fresh=True, defer_rs=True)
        torch.cuda.current_stream().synchronize()            # the image is ordered on the caller's stream
        assert torch.equal(envs[1].img, envs[0].img), it
        for k in ('lidar', 'action_mask', 'target', 'status', 'pose'):
            assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), (it, k)
        turnovers += int(envs[0].done.sum())
    envs[1].wait_rs()
    torch.cuda.synchronize()
    assert torch.equal(envs[1].rs_word, envs[0].rs_word)
    assert turnovers > 100, turnovers                       # that many layers were rebuilt on the side stream
    for e in envs:
        e.close()
