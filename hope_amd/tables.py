"""Host-side tables of the hot path, numpy float64.

These are the products of the reference constructors `ActionMask.__init__`
(src/model/action_mask.py:9-163) and `LidarSimlator.__init__` (src/env/lidar_simulator.py:14-53).
The elementwise arithmetic (operand order, ufuncs applied to whole arrays) follows the reference
so that on the same numpy build the tables carry the reference's own rounding; they are uploaded
to the GPU once with `hope_env_upload_tables` and shared by every scene.
"""
import math

import numpy as np

WHEEL_BASE, FRONT_HANG, REAR_HANG, WIDTH = 2.8, 0.96, 0.93, 1.94          # configs.py:13-17
VALID_STEER, VALID_SPEED = (-0.75, 0.75), (-2.5, 2.5)                     # configs.py:32-33
LIDAR_RANGE, LIDAR_NUM, PRECISION, N_ITER, UPSAMPLE = 10.0, 120, 10, 10, 10
VEHICLE_BOX = np.array([(-REAR_HANG, -WIDTH / 2), (FRONT_HANG + WHEEL_BASE, -WIDTH / 2),
                        (FRONT_HANG + WHEEL_BASE, WIDTH / 2), (-REAR_HANG, WIDTH / 2)])   # configs.py:20-24


def discrete_actions():
    """configs.py:108-115: 21 steers (np.arange drift kept) x {+1, -1} speed -> (42, 2)."""
    hi = VALID_STEER[-1]
    steer = np.arange(hi, -(hi + hi / PRECISION), -hi / PRECISION)
    fwd = np.column_stack([steer, np.full_like(steer, 1.0)])
    bwd = np.column_stack([steer, np.full_like(steer, -1.0)])
    return np.vstack([fwd, bwd])


def swept_boxes(actions=None):
    """action_mask.py:84-112: hull corners after k = 1..10 arc increments of 0.05 m -> (42, 10, 4, 2)."""
    actions = discrete_actions() if actions is None else actions
    radius = 1 / (np.tan(actions[:, 0]) / WHEEL_BASE)
    cx, cy = VEHICLE_BOX[:, 0].reshape(1, -1), VEHICLE_BOX[:, 1].reshape(1, -1)
    ox = 0 - radius * np.sin(0)
    oy = 0 + radius * np.cos(0)
    dphi = 0.5 * actions[:, 1] / 10 / radius
    phi = 0
    out = np.zeros((N_ITER, len(actions), 4, 2))
    for k in range(N_ITER):
        phi = phi + dphi
        px = ox + radius * np.sin(phi)
        py = oy - radius * np.cos(phi)
        c, s = np.cos(phi).reshape(-1, 1), np.sin(phi).reshape(-1, 1)
        out[k, :, :, 0] = c * cx - s * cy + px.reshape(-1, 1)
        out[k, :, :, 1] = s * cx + c * cy + py.reshape(-1, 1)
    return out.transpose(1, 0, 2, 3)


def _edge_hits(p, q, tol=1e-8):
    """action_mask.py:31-82 for m edges p (m,2,2) against n edges q (n,2,2): norm of the hit or inf."""
    x1, x2 = p[:, 0, 0].reshape(-1, 1), p[:, 1, 0].reshape(-1, 1)
    y1, y2 = p[:, 0, 1].reshape(-1, 1), p[:, 1, 1].reshape(-1, 1)
    a, b, c = y2 - y1, x1 - x2, y1 * x2 - x1 * y2
    u1, u2 = q[:, 0, 0].reshape(1, -1), q[:, 1, 0].reshape(1, -1)
    v1, v2 = q[:, 0, 1].reshape(1, -1), q[:, 1, 1].reshape(1, -1)
    d, e, f = v2 - v1, u1 - u2, v1 * u2 - u1 * v2
    det = a * e - b * d
    par = det == 0
    det[par] = 1
    rx = (b * f - c * e) / det
    ry = (c * d - a * f) / det
    bad_x = (rx > np.maximum(x1, x2) + tol) | (rx < np.minimum(x1, x2) - tol) | \
            (rx > np.maximum(u1, u2) + tol) | (rx < np.minimum(u1, u2) - tol) | par
    bad_y = (ry > np.maximum(y1, y2) + tol) | (ry < np.minimum(y1, y2) - tol) | \
            (ry > np.maximum(v1, v2) + tol) | (ry < np.minimum(v1, v2) - tol)
    n = np.sqrt(rx * rx + ry * ry)
    n[bad_x | bad_y] = np.inf
    return n


def circular_upsample(x, rate=UPSAMPLE):
    """action_mask.py:145-163: y[j] = x[j//r]*(1-(j%r)/r) + x[j//r+1]*((j%r)/r), circular on axis 0."""
    x = np.concatenate([x, x[0:1]], axis=0)
    j = np.arange((x.shape[0] - 1) * rate)
    shp = (len(j),) + (1,) * (x.ndim - 1)
    return x[j // rate] * (1 - (j % rate) / rate).reshape(shp) + x[j // rate + 1] * ((j % rate) / rate).reshape(shp)


def dist_star_coarse(boxes=None):
    """action_mask.py:114-141: per beam (length 100), action, increment: farthest hit on the swept box."""
    boxes = swept_boxes() if boxes is None else boxes
    idx = np.arange(LIDAR_NUM)
    far = LIDAR_RANGE * 10
    ends = np.stack([np.cos(idx / LIDAR_NUM * 2 * np.pi) * far, np.sin(idx / LIDAR_NUM * 2 * np.pi) * far], axis=1)
    beams = np.stack([np.zeros_like(ends), ends], axis=1)                     # (120, 2, 2)
    nxt = np.roll(boxes, -1, axis=2)                                          # vertex v+1 first, then v (:131-134)
    edges = np.stack([nxt, boxes], axis=3).reshape(-1, 2, 2)
    n = _edge_hits(beams, edges).reshape(LIDAR_NUM, boxes.shape[0], N_ITER, 4)
    n[n == np.inf] = 0
    return n.max(axis=-1)


def dist_star(coarse=None):
    """(1200, 42, 10) float64, reference layout."""
    return circular_upsample(dist_star_coarse() if coarse is None else coarse)


def hull_base():
    """Range rear-axle -> hull along each beam (lidar_simulator.py:48-53 / action_mask.py:21-29).
    The reference asks GEOS for LineString∩LinearRing; here: parametric ray/rectangle hit, then the
    norm of the hit point."""
    out = np.zeros(LIDAR_NUM)
    for i in range(LIDAR_NUM):
        th = i * math.pi / LIDAR_NUM * 2
        ex, ey = math.cos(th) * LIDAR_RANGE, math.sin(th) * LIDAR_RANGE
        best = math.inf
        for k in range(4):
            ax, ay = VEHICLE_BOX[k]
            bx, by = VEHICLE_BOX[(k + 1) % 4]
            sx, sy = bx - ax, by - ay
            den = ex * sy - ey * sx
            if den == 0:
                continue
            t = (ax * sy - ay * sx) / den
            u = (ax * ey - ay * ex) / den
            if 0 <= t <= 1 and 0 <= u <= 1:
                hx, hy = t * ex, t * ey
                best = min(best, math.sqrt(hx * hx + hy * hy))
        out[i] = best
    return out


def beam_ab():
    """lidar_simulator.py:86-88: (sin theta_i, -cos theta_i), theta_i = i*pi/120*2 -> (120, 2)."""
    theta = np.array([a * math.pi / LIDAR_NUM * 2 for a in range(LIDAR_NUM)])
    return np.ascontiguousarray(np.stack([np.sin(theta), -np.cos(theta)], axis=1))


_cache = {}


def all_tables():
    if not _cache:
        _cache.update(actions=discrete_actions(), dist_star=np.ascontiguousarray(dist_star()),
                      hull_base=hull_base(), beam_ab=beam_ab())
    return _cache
