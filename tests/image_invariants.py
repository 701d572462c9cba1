"""Analytic invariants of obs['img'] ("parity only statistical").  pygame / OpenCV are absent from the image,
so no reference-rendered pixel exists; what CAN be stated without them is the geometry the reference's code prescribes
(car_parking_base.py:139-147 coord_transform_matrix, :301-350 _render / _get_img_observation, observation_processor.py:11-23):
  * a world point P lands at output pixel (col, row) = 31.5 + R(heading) K (P - C) / 4, with C the vehicle box's centroid,
    K = 12 px/m, R(t)(u, v) = (u cos t + v sin t, -u sin t + v cos t) -- pygame's counter-clockwise rotation on a y-down surface,
    the very expression the reference uses for its own blit offset (:331-334) -- so the vehicle sits at the centre, nose to +col;
  * shapes keep their area: 3 px/m after the 4x INTER_LINEAR reduction, i.e. a car box (4.69 m x 1.94 m) covers 81.9 px^2;
  * the white background becomes exactly (0, 0, 0) (change_bg_color) and cv2.resize only averages: an output pixel is a convex
    combination of the palette colours of its source pixels.
These checks run on the CPU oracle's image (tests/test_oracle_image.py) and on the HIP kernel's (tests/test_gpu_image.py): both
are anchored to the same independent statement instead of only to each other."""
import numpy as np

K = 12.0
VEH = np.array([30, 144, 255], float)       # COLOR_POOL[0]  (vehicle.py:117)
DEST = np.array([69, 139, 0], float)        # DEST_COLOR     (configs.py:83)
OBST = np.array([150, 150, 150], float)     # OBSTACLE_COLOR
TRAJ_NEWEST = (10, 10, 200)                 # TRAJ_COLORS[-1]
CAR_L, CAR_W = 0.96 + 2.8 + 0.93, 1.94      # VehicleBox (configs.py:13-24)
CAR_MID = 0.5 * (3.76 - 0.93)               # box centre ahead of the rear axle
CAR_AREA_PX = CAR_L * CAR_W * (K / 4) ** 2  # 81.9


def box_centre(pose):
    return np.array([pose[0] + CAR_MID * np.cos(pose[2]), pose[1] + CAR_MID * np.sin(pose[2])])


def to_image(P, pose):
    """(col, row) of world point P in the 64 x 64 observation of a vehicle at `pose`"""
    s = K * (np.asarray(P, float) - box_centre(pose))
    c, sn = np.cos(pose[2]), np.sin(pose[2])
    return np.array([31.5 + (s[0] * c + s[1] * sn) / 4, 31.5 + (-s[0] * sn + s[1] * c) / 4])


def make_scene(rng, kind):
    """kind: 'vehicle' (nothing else inside the crop), 'dest' (the dest box 6.5 .. 8 m away), 'obstacle' (one 2.4 m square 5.5 .. 8 m away)"""
    from hope_amd.scenes import Scene
    h = rng.uniform(-np.pi, np.pi)
    start = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), h])
    bearing = rng.uniform(-np.pi, np.pi)
    d_dest = rng.uniform(6.5, 8.0) if kind == 'dest' else 18.5
    c0 = box_centre(start)
    dest_h = rng.uniform(-np.pi, np.pi)
    dc = c0 + d_dest * np.array([np.cos(bearing), np.sin(bearing)])           # centre of the dest box
    dest = np.array([dc[0] - CAR_MID * np.cos(dest_h), dc[1] - CAR_MID * np.sin(dest_h), dest_h])
    if kind == 'obstacle':
        ob = bearing + np.pi + rng.uniform(-1.0, 1.0)
        oc = c0 + rng.uniform(5.5, 8.0) * np.array([np.cos(ob), np.sin(ob)])
    else:
        # far outside the 41.7 m surface (pygame clips it away).  NOT merely outside the crop: an obstacle over the surface's top-left
        # pixel would turn rotate()'s background -- the colour of that pixel -- grey (reproduced, tests/test_gpu_image.py)
        oc = c0 - 45.0 * np.array([np.cos(bearing), np.sin(bearing)])
    a = rng.uniform(0, np.pi / 2)
    r = 1.2 * np.sqrt(2)
    sq = np.array([[oc[0] + r * np.cos(a + np.pi / 4 + k * np.pi / 2), oc[1] + r * np.sin(a + np.pi / 4 + k * np.pi / 2)] for k in range(4)])
    bbox = np.array([np.floor(min(start[0], dest[0]) - 10), np.ceil(max(start[0], dest[0]) + 10),
                     np.floor(min(start[1], dest[1]) - 10), np.ceil(max(start[1], dest[1]) + 10)])
    sc = Scene(start=start, dest=dest, bbox=bbox, verts=sq[None], nvert=np.array([4], np.int32), level='Normal')
    return sc, {'dest_centre': dc, 'obst_centre': oc, 'obst_area_px': 2.4 * 2.4 * (K / 4) ** 2}


def mass(img, colour, channel):
    """per-pixel coverage of a shape of `colour`, from one channel the other shapes of the scene leave at 0"""
    return img[channel].astype(float) / colour[channel]


def centroid(w):
    rows, cols = np.mgrid[0:64, 0:64]
    return np.array([(w * cols).sum(), (w * rows).sum()]) / w.sum()


def check_vehicle_only(img, what=''):
    """reset observation, nothing but the vehicle inside the crop"""
    cov = mass(img, VEH, 2)
    # exactly black outside the box's neighbourhood: half extents 7.0 x 2.9 px + 2 px for the border pixels and the integer vertex / blit truncations
    rows, cols = np.mgrid[0:64, 0:64]
    outside = (np.abs(cols - 31.5) > 9.1) | (np.abs(rows - 31.5) > 5.1)
    assert not img[:, outside].any(), f'{what}: background not exactly (0, 0, 0)'
    # cv2.resize only averages: every pixel is a convex combination a VEH + b START (the start outline, drawn under the vehicle at
    # the same pose, can peek out along the box's border) + (1 - a - b) black; solve (a, b) from R and B, G must agree
    START = np.array([100, 149, 237], float)
    M = np.array([[VEH[0], START[0]], [VEH[2], START[2]]])
    ab = np.linalg.solve(M, np.stack([img[0].ravel().astype(float), img[2].ravel().astype(float)]))
    assert ab.min() >= -0.02 and ab.sum(0).max() <= 1.02, (what, ab.min(), ab.sum(0).max())
    assert np.abs(VEH[1] * ab[0] + START[1] * ab[1] - img[1].ravel()).max() <= 1.5, what
    assert ab[1].sum() <= 0.12 * CAR_AREA_PX, (what, ab[1].sum())                          # the outline is a border effect
    area = cov.sum()
    # filled polygons include their border pixels (about half a source pixel all round: + 6 % for a car box, more when rotated)
    assert 0.95 * CAR_AREA_PX <= area <= 1.16 * CAR_AREA_PX, (what, area, CAR_AREA_PX)
    c = centroid(cov)
    assert np.abs(c - 31.5).max() <= 0.75, (what, c)
    assert tuple(img[:, 31, 31]) == (30, 144, 255) and tuple(img[:, 32, 32]) == (30, 144, 255), what      # the centre is pure vehicle colour
    # nose towards +col: the box is long along the columns
    ext_c = np.ptp(np.nonzero(cov.sum(0) > 0.5)[0]) + 1
    ext_r = np.ptp(np.nonzero(cov.sum(1) > 0.5)[0]) + 1
    assert 13 <= ext_c <= 16 and 5 <= ext_r <= 8, (what, ext_c, ext_r)
    return area, c


def check_dest(img, pose, info, what=''):
    """dest box inside the crop, away from the vehicle: its green mass sits where the geometry says, with the box's area"""
    pure = img[2] == 0                                              # vehicle and start outline have blue; the dest colour has none
    cov = np.where(pure, img[1].astype(float) / DEST[1], 0.0)
    want = to_image(info['dest_centre'], pose)
    if not (4 <= want[0] <= 59 and 4 <= want[1] <= 59):            # (partly outside the crop: not a centroid test)
        return None
    area = cov.sum()
    assert 0.93 * CAR_AREA_PX <= area <= 1.16 * CAR_AREA_PX, (what, area)
    c = centroid(cov)
    assert np.abs(c - want).max() <= 1.0, (what, c, want)
    assert np.abs(np.where(pure, img[0] - cov * DEST[0], 0.0)).max() <= 1.0, what
    return area, c, want


def check_obstacle(img, pose, info, what=''):
    grey = (img[0] == img[1]) & (img[1] == img[2])
    cov = np.where(grey, img[0].astype(float) / OBST[0], 0.0)
    want = to_image(info['obst_centre'], pose)
    if not (4 <= want[0] <= 59 and 4 <= want[1] <= 59):
        return None
    area = cov.sum()
    assert 0.95 * info['obst_area_px'] <= area <= 1.25 * info['obst_area_px'], (what, area, info['obst_area_px'])
    c = centroid(cov)
    assert np.abs(c - want).max() <= 1.0, (what, c, want)
    return area, c, want
