"""Known-answer tests for the image-observation oracle (oracle/hope_oracle_img.c).

pygame / OpenCV are not available here and the reference holds no rendered image, so the oracle's parity is
UNPINNED (see its header).  What can be checked: (1) the restated pygame scan-line / Bresenham rules against
rows worked out by hand from draw.c's rule, (2) exactly predictable images for axis-aligned scenes, (3) agreement
with an independent geometric renderer (ideal ego-frame transform + point-in-polygon) for arbitrary headings.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

L = O.lib()
K = 12.0


def fill(points, n=500):
    """draw_fillpoly on an empty surface via orc_bev_world: one 'obstacle' whose pixel coordinates equal `points`
    (bbox chosen so that bx = by = 0 and k = 12 -> world coordinates = pixel / 12 + tiny offset)."""
    pts = np.asarray(points, float)
    verts = np.zeros((1, 4, 2))
    nv = len(pts)
    verts[0, :nv] = (pts + 0.25) / K            # (int)(12 * v + 0) truncates back to the integer pixel
    verts[0, nv:] = verts[0, nv - 1]
    bbox = np.array([0.0, 500 / K, 0.0, 500 / K])          # bx = 0.5 * (500 - 12 * (xmax + xmin)) = 0
    far = np.array([1e4, 1e4, 0.0])                         # start / dest / vehicle far outside the surface
    w = O.bev_world(verts, np.array([nv], np.int32), far, far, bbox, far, [far])
    return w


def rows_of(w, value=1):
    out = {}
    for y in np.nonzero((w == value).any(1))[0]:
        xs = np.nonzero(w[y] == value)[0]
        assert xs[-1] - xs[0] + 1 == len(xs)
        out[int(y)] = (int(xs[0]), int(xs[-1]))
    return out


def test_fill_axis_aligned_rectangle_is_inclusive():
    w = fill([(10, 5), (20, 5), (20, 9), (10, 9)])
    assert rows_of(w) == {y: (10, 20) for y in range(5, 10)}


def test_fill_diamond_rows():
    # rows worked by hand from draw_fillpoly: widths 1,3,...,21,...,3,1
    w = fill([(10, 0), (20, 10), (10, 20), (0, 10)])
    r = rows_of(w)
    assert r == {y: (10 - min(y, 20 - y), 10 + min(y, 20 - y)) for y in range(21)}
    assert (w == 1).sum() == 221


def test_fill_triangle_floor_ceil_rule():
    # p0=(10,0) p1=(13,10) p2=(7,10): first-found intersection floors, second ceils (draw.c), see rows in the comment
    w = fill([(10, 0), (13, 10), (7, 10)])
    expect = {0: (10, 10), 1: (10, 10), 2: (10, 10), 3: (10, 10), 4: (9, 11), 5: (9, 11), 6: (9, 11), 7: (8, 12),
              8: (8, 12), 9: (8, 12), 10: (7, 13)}
    assert rows_of(w) == expect


def test_fill_is_clipped_to_the_surface():
    w = fill([(-30, -20), (40, -20), (40, 15), (-30, 15)])
    assert rows_of(w) == {y: (0, 40) for y in range(0, 16)}
    w = fill([(480, 490), (700, 490), (700, 800), (480, 800)])
    assert rows_of(w) == {y: (480, 499) for y in range(490, 500)}


def test_outline_is_bresenham():
    # the start box is drawn with width=1: put it where its pixel corners are known; heading 0 box corners
    # (-0.93,-0.97) (3.76,-0.97) (3.76,0.97) (-0.93,0.97) about the pose
    bbox = np.array([0.0, 500 / K, 0.0, 500 / K])
    far = np.array([1e4, 1e4, 0.0])
    start = np.array([10.0, 10.0, 0.0])
    w = O.bev_world(np.zeros((0, 4, 2)), np.zeros(0, np.int32), start, far, bbox, far, [far])
    x0, x1 = int(K * (10 - 0.93)), int(K * (10 + 3.76))
    y0, y1 = int(K * (10 - 0.97)), int(K * (10 + 0.97))
    exp = np.zeros_like(w)
    exp[y0, x0:x1 + 1] = 2; exp[y1, x0:x1 + 1] = 2; exp[y0:y1 + 1, x0] = 2; exp[y0:y1 + 1, x1] = 2
    assert (w == exp).all()
    # tilted: compare every edge with a literal Bresenham walk (err = (dx > dy ? dx : -dy) / 2, C division)
    start = np.array([20.0, 20.0, 0.7])
    w = O.bev_world(np.zeros((0, 4, 2)), np.zeros(0, np.int32), start, far, bbox, far, [far])
    box = O.create_box(start)
    pix = [(int(K * x + 0.0 * y + 0.0), int(0.0 * x + K * y + 0.0)) for x, y in box]
    exp = np.zeros_like(w)
    for (xa, ya), (xb, yb) in zip(pix, pix[1:] + pix[:1]):
        dx, dy = abs(xb - xa), abs(yb - ya)
        sx, sy = (1 if xa < xb else -1), (1 if ya < yb else -1)
        err = int((dx if dx > dy else -dy) / 2)
        x, y = xa, ya
        while (x, y) != (xb, yb):
            exp[y, x] = 2
            e2 = err
            if e2 > -dx: err -= dy; x += sx
            if e2 < dy: err += dx; y += sy
        exp[yb, xb] = 2
    assert (w == exp).all()


def test_draw_order_and_trajectory_colours():
    bbox = np.array([0.0, 500 / K, 0.0, 500 / K])
    far = np.array([1e4, 1e4, 0.0])
    pose = np.array([20.0, 20.0, 0.0])
    # len(trajectory) == 1 -> only the vehicle colour
    w = O.bev_world(np.zeros((0, 4, 2)), np.zeros(0, np.int32), far, far, bbox, pose, [pose])
    assert set(np.unique(w)) == {0, 4}
    # 3 entries: colours TRAJ_COLORS[-3:], newest (= current pose) on top of the vehicle
    traj = [pose - [1.0, 0, 0], pose - [0.5, 0, 0], pose]
    w = O.bev_world(np.zeros((0, 4, 2)), np.zeros(0, np.int32), far, far, bbox, pose, traj)
    assert set(np.unique(w)) == {0, 5 + 17, 5 + 18, 5 + 19}
    assert w[int(K * 20), int(K * 20)] == 5 + 19
    # 25 entries: only the last 20 are drawn, oldest of them with TRAJ_COLORS[0]
    traj = [pose + [0.3 * i, 0, 0] for i in range(25)]
    w = O.bev_world(np.zeros((0, 4, 2)), np.zeros(0, np.int32), far, far, bbox, traj[-1], traj)
    assert set(np.unique(w)) == {0} | set(range(5, 25))
    assert w[int(K * 20), int(K * (20 + 0.3 * 5 - 0.9))] == 5      # rear end of the oldest drawn box (entry 5)


def test_axis_aligned_scene_image_is_predictable():
    """heading 0 -> rotate90 path (0 turns); every pixel of the 64x64 image follows from rectangle arithmetic"""
    bbox = np.array([-10.0, 10.0, -10.0, 10.0])         # bx = by = 250
    pose = np.array([0.0, 0.0, 0.0])
    far = np.array([1e4, 1e4, 0.0])
    verts = np.array([[[5.0, 2.0], [7.0, 2.0], [7.0, 6.0], [5.0, 6.0]]])
    world = np.zeros((500, 500), np.uint8)

    def rect(x0, x1, y0, y1, v):
        world[int(K * y0 + 250):int(K * y1 + 250) + 1, int(K * x0 + 250):int(K * x1 + 250) + 1] = v
    rect(5, 7, 2, 6, 1)
    rect(-0.93, 3.76, -0.97, 0.97, 4)
    w = O.bev_world(verts, np.array([4], np.int32), far, far, bbox, pose, [pose])
    assert (w == world).all()
    # centroid (1.415, 0) -> vehicle centre pixel (266.98, 250) -> blit offset (int(-16.98), 0) = (-16, 0)
    rgb = np.zeros((500, 500, 3), np.uint8)             # background white -> black by change_bg_color
    rgb[world == 1] = (150, 150, 150)
    rgb[world == 4] = (30, 144, 255)
    crop = rgb[122:122 + 256, 138:138 + 256].astype(np.int32)
    exp = (crop[1::4, 1::4] + crop[1::4, 2::4] + crop[2::4, 1::4] + crop[2::4, 2::4] + 2) >> 2
    img = O.bev_image(verts, np.array([4], np.int32), far, far, bbox, pose, [pose])
    assert (img.transpose(1, 2, 0) == exp).all()
    assert img[:, 32, 32].tolist() == [30, 144, 255]    # the car sits at the image centre


@pytest.mark.parametrize('turns', [1, 2, 3])
def test_quarter_turn_headings_use_exact_pixel_rotation(turns):
    h = np.float64(np.float32(90.0 * turns)) * np.pi / 180.0
    if np.float32(np.rad2deg(h)) % 90 != 0:
        pytest.skip('heading does not round-trip to an exact multiple of 90 degrees in float32')
    bbox = np.array([-10.0, 10.0, -10.0, 10.0])
    pose = np.array([1.0, -2.0, h])
    far = np.array([1e4, 1e4, 0.0])
    verts = np.array([[[5.0, 2.0], [7.0, 2.5], [6.5, 6.0], [4.0, 5.0]]])
    w = O.bev_world(verts, np.array([4], np.int32), far, far, bbox, pose, [pose])
    raw = O.bev_raw(w, bbox, pose)
    # np.rot90 rotates counter-clockwise in array (row, col) terms = pygame's rotate90 on the (y, x) raster
    rot = np.rot90(w, turns)
    assert rot.shape == (500, 500)
    box = O.create_box(pose)
    cx, cy = box.mean(0)
    vcx, vcy = K * cx + 250, K * cy + 250
    ddx = (vcx - 250) * np.cos(h) + (vcy - 250) * np.sin(h)
    ddy = -(vcx - 250) * np.sin(h) + (vcy - 250) * np.cos(h)
    ox, oy = int(-ddx), int(-ddy)
    ids = np.zeros((256, 256), np.uint8)
    for y in range(256):
        for x in range(0, 256):
            rx, ry = 122 + x - ox, 122 + y - oy
            if 0 <= rx < 500 and 0 <= ry < 500:
                ids[y, x] = rot[ry, rx]
    mask = raw.reshape(256, 256, 3).astype(int).sum(-1) != 765
    assert (mask == (ids != 0)).all()


def ideal_mask(verts, nvert, dest, pose):
    """independent renderer: ideal ego-frame transform + even-odd point-in-polygon at the four sample centres"""
    box = O.create_box(pose)
    cx, cy = box.mean(0)
    polys = [v[:n] for v, n in zip(verts, nvert)] + [O.create_box(dest), box]
    u = np.arange(64)
    samples = np.stack(np.meshgrid(np.r_[4 * u + 1, 4 * u + 2], np.r_[4 * u + 1, 4 * u + 2]), -1).reshape(-1, 2) + 0.5
    ex, ey = (samples[:, 0] - 128) / K, (samples[:, 1] - 128) / K
    c, s = np.cos(pose[2]), np.sin(pose[2])
    wx, wy = cx + ex * c - ey * s, cy + ex * s + ey * c
    inside = np.zeros(len(samples), bool)
    for P in polys:
        acc = np.zeros(len(samples), bool)
        for (x1, y1), (x2, y2) in zip(P, np.roll(P, -1, 0)):
            cond = (y1 > wy) != (y2 > wy)
            xi = x1 + (wy - y1) * (x2 - x1) / np.where(y2 == y1, 1, y2 - y1)
            acc ^= cond & (wx < xi)
        inside |= acc
    m = np.zeros((256, 256), bool)
    m[(samples[:, 1] - 0.5).astype(int), (samples[:, 0] - 0.5).astype(int)] = inside
    return m


def test_general_heading_agrees_with_geometric_renderer():
    rng = np.random.default_rng(5)
    agree = []
    for trial in range(12):
        pose = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-np.pi, np.pi)])
        dest = np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-np.pi, np.pi)])
        n = 6
        verts = np.zeros((n, 4, 2))
        for o in range(n):
            c = rng.uniform(-9, 9, 2)
            a = rng.uniform(0, np.pi)
            hw, hh = rng.uniform(0.8, 2.5), rng.uniform(0.5, 1.2)
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            verts[o] = c + (np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]) @ R.T)
        nvert = np.full(n, 4, np.int32)
        bbox = np.array([-12.0, 12.0, -12.0, 12.0])
        far = np.array([1e4, 1e4, 0.0])
        w = O.bev_world(verts, nvert, far, dest, bbox, pose, [pose])
        raw = O.bev_raw(w, bbox, pose).astype(int)
        got = raw.sum(-1) != 765
        want = ideal_mask(verts, nvert, dest, pose)
        sel = np.zeros((256, 256), bool)
        u = np.arange(64)
        idx = np.r_[4 * u + 1, 4 * u + 2]
        sel[np.ix_(idx, idx)] = True
        agree.append((got[sel] == want[sel]).mean())
        img = O.bev_process(raw.astype(np.uint8))
        assert img[32, 32].tolist() == [30, 144, 255]
    assert min(agree) > 0.985, agree        # differences only along polygon borders (integer vertices, nearest sampling)


def test_batch_oracle_tracks_the_trajectory_like_vehicle_py():
    from hope_amd.scenes import SceneSource, pack_scenes
    src = SceneSource(levels=('Normal',), seed=11)
    scenes = [src.draw() for _ in range(4)]
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, 32)
    orc = O.BatchOracle(4, 32)
    orc.set_scenes(np.arange(4), start, dest, bbox, verts, nvert, nob)
    orc.reset_obs()
    assert all(len(t) == 1 for t in orc.traj)
    rng = np.random.default_rng(0)
    for it in range(5):
        o = orc.step(rng.uniform(-1, 1, (4, 2)))
        for i in range(4):
            assert (orc.traj[i][-1] == orc.pose[i]).all()
    moved = [len(t) for t in orc.traj]
    assert max(moved) == 6
    img = orc.image()
    assert img.shape == (4, 3, 64, 64) and img.dtype == np.uint8
    assert (img[:, :, 32, 32] == np.array([10, 10, 200])).all()      # newest trajectory box covers the centre


def test_image_oracle_regression_hashes():
    """guards the oracle itself against accidental edits (it is the checker of tests/test_gpu_image.py): images of 6
    seeded mixed scenes after 1, 6 and 24 random steps (hope_math build; the same arithmetic as the HIP path)"""
    import hashlib
    from hope_amd.scenes import SceneSource, pack_scenes
    src = SceneSource(seed=123)
    scenes = [src.draw() for _ in range(6)]
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, 128)
    orc = O.BatchOracle(6, 128)
    orc.set_scenes(np.arange(6), start, dest, bbox, verts, nvert, nob)
    orc.reset_obs(with_rs=False)
    rng = np.random.default_rng(9)
    got = []
    for it in range(24):
        orc.step(rng.uniform(-1, 1, (6, 2)), with_rs=False)
        if it in (0, 5, 23):
            got.append(hashlib.sha256(orc.image().tobytes()).hexdigest())
    assert got == ['878a38e87d132f13373469c5b16a352e264e6d4eb888a5210b0cfa508aa9a728',
                   '65a611dc552bdb5983c4a2be71000d95f24acb3c5848894a31d4b891a1cc04e1',
                   'e55e0297f0dcf6c578c36af1335e12f65f406d1e09bcebafb7e065ca6ea58a7c']


def test_oracle_image_obeys_the_analytic_invariants_of_the_reference_geometry():
    """Row f-1's statistical anchor: 3 x 200 random scenes (vehicle alone / dest box in view / one obstacle in view); the rendered
    shapes sit where car_parking_base.py:139-147, :322-346 put them (centroid within 1 px of 31.5 + R(heading) K (P - C) / 4), keep
    their area at 3 px/m (- 5 % .. + 16 %: filled polygons include their border pixels), the vehicle is centred with its nose to
    + col, pixels are convex combinations of the palette, the background is exactly (0, 0, 0).  tests/image_invariants.py."""
    import image_invariants as I
    rng = np.random.default_rng(11)
    seen = {'dest': 0, 'obstacle': 0}
    for kind in ('vehicle', 'dest', 'obstacle'):
        for k in range(200):
            sc, info = I.make_scene(rng, kind)
            img = O.bev_image(sc.verts, sc.nvert, sc.start, sc.dest, sc.bbox, sc.start, sc.start[None])
            what = f'{kind} {k}'
            if kind == 'vehicle':
                I.check_vehicle_only(img, what)
            elif kind == 'dest':
                seen['dest'] += I.check_dest(img, sc.start, info, what) is not None
            else:
                seen['obstacle'] += I.check_obstacle(img, sc.start, info, what) is not None
    assert seen['dest'] > 150 and seen['obstacle'] > 150, seen
