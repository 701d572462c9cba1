"""The lidar's beam-span filter takes edge directions from `span_angle` (hope_amd/csrc/hope_step_kernel.h), a polynomial
atan2 -- not a reference quantity, but the filter's margins (MARGIN = 2e-3 rad for the spans, 2e-4 + 1e-5 rad for the
back-face cull's "vertex near a beam" test) assume it is within 4e-6 rad of the true direction.  This test restates the device
function in numpy float32 (coefficients parsed from the header, the hardware reciprocal modelled as exact +- 1 ulp) and checks
that bound over the plane, the axes and the quadrant seams."""
import re
from pathlib import Path

import numpy as np

HDR = Path(__file__).resolve().parents[1] / 'hope_amd' / 'csrc' / 'hope_step_kernel.h'


def _coefficients():
    m = re.search(r'SPAN_ANGLE_C\[6\]\s*=\s*\{([^}]*)\}', HDR.read_text())
    assert m, 'SPAN_ANGLE_C not found'
    c = [np.float32(float(t.strip().rstrip('f'))) for t in m.group(1).split(',')]
    assert len(c) == 6
    return c


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def span_angle(y, x, rcp_ulps=0):
    c = _coefficients()
    f = np.float32
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    with np.errstate(divide='ignore', invalid='ignore'):
        rcp = (f(1) / mx).astype(f)
        rcp = (rcp * f(1 + rcp_ulps * 2.0 ** -23)).astype(f)
        t = (mn * rcp).astype(f)
    s = (t * t).astype(f)
    q = _fma(s, np.full_like(s, c[5]), c[4])
    for k in (3, 2, 1, 0):
        q = _fma(q, s, c[k])
    r = (q * t).astype(f)
    r = np.where(ay > ax, (f(1.57079633) - r).astype(f), r)
    r = np.where(x < 0, (f(3.14159265) - r).astype(f), r)
    return np.copysign(r, y).astype(f)


def _err(y, x, **kw):
    got = span_angle(y, x, **kw).astype(np.float64)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    d = np.abs(got - want)
    return np.minimum(d, 2 * np.pi - d)          # -pi and +pi are the same direction


def test_span_angle_is_within_4e_6_rad_of_atan2():
    rng = np.random.default_rng(5)
    r = np.exp(rng.uniform(np.log(0.1), np.log(60.0), 2_000_000))      # vertices nearer than 0.1 m take the all-beams path
    th = rng.uniform(-np.pi, np.pi, r.size)
    x, y = (r * np.cos(th)).astype(np.float32), (r * np.sin(th)).astype(np.float32)
    for ulps in (-1, 0, 1):
        assert _err(y, x, rcp_ulps=ulps).max() < 4e-6


def test_span_angle_on_the_axes_diagonals_and_seams():
    v = np.array([0.1, 0.5, 1.0, 3.25, 10.0, 47.0], np.float32)
    eps = np.float32(1e-6)
    xs, ys = [], []
    for a in v:
        for sx, sy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)):
            for dx in (-eps, 0, eps):
                for dy in (-eps, 0, eps):
                    xs.append(sx * a + dx * a)
                    ys.append(sy * a + dy * a)
    x, y = np.array(xs, np.float32), np.array(ys, np.float32)
    assert _err(y, x).max() < 4e-6
    # the direction of (0, 0) is undefined: NaN, which the caller turns into "all beams"
    assert np.isnan(span_angle(np.zeros(1, np.float32), np.zeros(1, np.float32)))[0]
    # signed zeros follow atan2
    z, one = np.float32(0.0), np.float32(1.0)
    assert span_angle(np.array([-z]), np.array([-one]))[0] == np.float32(-3.14159265)
    assert span_angle(np.array([z]), np.array([-one]))[0] == np.float32(3.14159265)
    assert span_angle(np.array([one]), np.array([-z]))[0] == np.float32(1.57079633)
