"""hope_amd/csrc/hope_math.h: accuracy against libm on the host, and (-m gpu) bit-equality device vs host."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as O

FN = dict(sin=0, cos=1, tan=2, atan2=3, asin=4, acos=5, hypot=6, fmod=7, tanh=8, exp=9, sqrt=10, div=11)


def ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want) + 1e-300)


def inputs(seed=0):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(-100, 100, 100000), rng.uniform(-7, 7, 100000), rng.uniform(-1e4, 1e4, 20000),
                        [0.0, -0.0, 1e-300, math.pi / 2, math.pi, -math.pi, 1e-9, 0.75, -0.75]])
    y, xx = rng.normal(0, 5, len(x)), rng.normal(0, 5, len(x))
    y[:8] = [0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 2.0, 0.0]
    xx[:8] = [1.0, -1.0, 0.0, 0.0, 0.0, -1.0, -2.0, -0.0]
    z = np.concatenate([rng.uniform(-1, 1, len(x) - 4), [1.0, -1.0, 0.0, 0.25]])
    w = np.concatenate([rng.uniform(-0.15, 0.15, len(x) // 2), rng.uniform(-6, 6, len(x) - len(x) // 2)])
    return x, y, xx, z, w


def test_accuracy_against_libm():
    x, y, xx, z, w = inputs()
    assert ulps(O.math_fn(FN['sin'], x), np.sin(x)).max() <= 2
    assert ulps(O.math_fn(FN['cos'], x), np.cos(x)).max() <= 2
    assert ulps(O.math_fn(FN['tan'], x), np.tan(x)).max() <= 4
    assert ulps(O.math_fn(FN['atan2'], y, xx), np.arctan2(y, xx)).max() <= 2
    assert ulps(O.math_fn(FN['asin'], z), np.arcsin(z)).max() <= 3
    assert ulps(O.math_fn(FN['acos'], z), np.arccos(z)).max() <= 3
    assert ulps(O.math_fn(FN['hypot'], y, xx), np.hypot(y, xx)).max() <= 1
    assert np.array_equal(O.math_fn(FN['fmod'], x, np.full_like(x, 2 * math.pi)), np.fmod(x, 2 * math.pi))
    assert ulps(O.math_fn(FN['tanh'], w), np.tanh(w)).max() <= 3
    assert ulps(O.math_fn(FN['exp'], w * 4), np.exp(w * 4)).max() <= 2
    # exact symmetries the kernels rely on (reeds_shepp.py:522-523 uses cos(-x), sin(-x))
    assert np.array_equal(O.math_fn(FN['sin'], -x), -O.math_fn(FN['sin'], x))
    assert np.array_equal(O.math_fn(FN['cos'], -x), O.math_fn(FN['cos'], x))
    # zeros: C semantics
    for a, b in [(0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 0.0)]:
        assert O.math_fn(FN['atan2'], [a], [b])[0] == math.atan2(a, b)


@pytest.mark.gpu
def test_device_equals_host_bit_for_bit():
    """same source, IEEE-exact operations only, contraction off -> identical bits on gfx950 and x86-64
    (also pins the device's float64 sqrt and division as correctly rounded)."""
    torch = pytest.importorskip('torch')
    from hope_amd import _lib as L
    lib = L.load_library()
    x, y, xx, z, w = inputs(1)
    cases = [('sin', x, None), ('cos', x, None), ('tan', x, None), ('atan2', y, xx), ('asin', z, None), ('acos', z, None),
             ('hypot', y, xx), ('fmod', x, np.full_like(x, 2 * math.pi)), ('tanh', w, None), ('exp', w * 4, None),
             ('sqrt', np.abs(x), None), ('div', y, xx)]
    # arguments beyond the 32-bit range of the quadrant conversion (the host's conversion was undefined behaviour there and
    # differed from the GPU's saturating one): every finite and non-finite input must give the same bits on both sides
    big = np.array([3.3e9, 3.4e9, -3.4e9, 1e10, -7.5e12, 2.0 ** 40, 1e19, 1e300, -1e300, np.inf, -np.inf, np.nan, 2147483647.0 * 1.5707963267948966])
    cases += [('sin', big, None), ('cos', big, None), ('tan', big, None)]
    for name, a, b in cases:
        ta = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        tb = torch.from_numpy(np.ascontiguousarray(b)).cuda() if b is not None else None
        out = torch.empty_like(ta)
        L.check(lib.hope_debug_math(FN[name], ta.numel(), C.c_void_p(ta.data_ptr()),
                                    C.c_void_p(tb.data_ptr()) if tb is not None else None,
                                    C.c_void_p(out.data_ptr()), None), 'hope_debug_math')
        torch.cuda.synchronize()
        host = O.math_fn(FN[name], a, b)
        dev = out.cpu().numpy()
        same = (dev.view(np.int64) == host.view(np.int64)) | (np.isnan(dev) & np.isnan(host))
        assert same.all(), (name, int((~same).sum()), a[~same][:3], dev[~same][:3], host[~same][:3])


@pytest.mark.gpu
def test_kinematics_division_by_20_is_the_ieee_quotient():
    """k_kinematics divides its 200 displacement terms per step by MINI_ITER = 20 with q0 = x R, r = fma(-q0, 20, x), q = fma(r, R, q0)
    (R = RN(1/20); proof of correct rounding in hope_step_kernel.h).  Here: the device sequence against numpy's division on 2e6
    values -- random mantissas over 40 binades and both signs, integers near multiples of 5, near-midpoint quotients, the step's
    own magnitudes."""
    torch = pytest.importorskip('torch')
    from hope_amd import _lib as L
    lib = L.load_library()
    rng = np.random.default_rng(5)
    mant = rng.integers(1 << 52, 1 << 53, 1_000_000).astype(np.float64)
    x = np.concatenate([
        np.ldexp(mant, rng.integers(-80, -40, len(mant))) * rng.choice([-1.0, 1.0], len(mant)),
        (np.float64(1 << 52) + np.arange(200_000)), (np.float64(1 << 53) - 1 - np.arange(200_000)),
        20.0 * (np.float64(1 << 52) + rng.integers(0, 1 << 40, 200_000)) + rng.integers(-12, 13, 200_000),     # quotients next to integers
        rng.uniform(-2.5, 2.5, 400_000) * rng.uniform(-1, 1, 400_000) * 5e-2, [0.0, -0.0, 20.0, 1.0, 5e-2, 1e-200, 1e300]])
    ta = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = torch.empty_like(ta)
    L.check(lib.hope_debug_math(12, ta.numel(), C.c_void_p(ta.data_ptr()), None, C.c_void_p(out.data_ptr()), None), 'hope_debug_math')
    torch.cuda.synchronize()
    dev, host = out.cpu().numpy(), x / 20.0
    same = dev.view(np.int64) == host.view(np.int64)
    assert same.all(), (int((~same).sum()), x[~same][:3], dev[~same][:3], host[~same][:3])


@pytest.mark.gpu
def test_mask_step_fraction_is_the_ieee_quotient():
    """the action mask's k / n_iter (action_mask.py:196) is formed without a division or a table: q0 = k R, r = fma(-q0, 10, k),
    q = fma(r, R, q0) with R = RN(0.1) (hope_step_kernel.h mask_fraction): all eleven values against numpy's quotients"""
    torch = pytest.importorskip('torch')
    from hope_amd import _lib as L
    lib = L.load_library()
    x = np.arange(0, 11, dtype=np.float64)
    ta = torch.from_numpy(x).cuda()
    out = torch.empty_like(ta)
    L.check(lib.hope_debug_math(13, ta.numel(), C.c_void_p(ta.data_ptr()), None, C.c_void_p(out.data_ptr()), None), 'hope_debug_math')
    torch.cuda.synchronize()
    dev, host = out.cpu().numpy(), x / 10.0
    assert (dev.view(np.int64) == host.view(np.int64)).all(), (dev, host)


@pytest.mark.gpu
def test_device_math_sweep_against_glibc_ten_million_arguments():
    """hope_math.h sits on both sides of every tolerance-0.0 comparison (kernels and default oracle), so its
    accuracy is checked on its own against glibc (numpy's float64 functions = what Python's `math` gave the reference) over 1e7
    arguments per function evaluated ON THE DEVICE: the ranges the path reaches -- headings of the unwrapped Euler chain
    (vehicle.py:92-93: |h| up to ~1e2 rad and beyond), the Reeds-Shepp solvers' angles, the reward's acos(cos(.)) fold
    (car_parking_base.py:203-206), quotients next to +-1 for asin / acos, atan2 over all quadrants incl. tiny and axis-aligned
    arguments, fmod by 2 pi (reeds_shepp.py:561-568 pi_2_pi), square roots over 600 binades -- asserting <= 3 ulp (sqrt and fmod:
    bit-equal).  The observed maxima are printed for profiles/."""
    torch = pytest.importorskip('torch')
    from hope_amd import _lib as L
    lib = L.load_library()
    rng = np.random.default_rng(77)
    N = 10_000_000
    parts = [rng.uniform(-130, 130, N // 2), rng.uniform(-2 * math.pi, 2 * math.pi, N // 4), rng.normal(0, 1e3, N // 8),
             np.ldexp(rng.uniform(-1, 1, N // 16), rng.integers(-40, 1, N // 16))]
    h = np.concatenate(parts + [np.arange(N - sum(len(p_) for p_ in parts)) * (math.pi / 1024)])      # multiples of pi / 1024: the quadrant boundaries
    unit = np.concatenate([rng.uniform(-1, 1, N // 2), 1 - np.ldexp(rng.uniform(0, 1, N // 4), rng.integers(-50, 0, N // 4)),
                           -1 + np.ldexp(rng.uniform(0, 1, N // 4), rng.integers(-50, 0, N // 4))])
    ya = np.concatenate([rng.normal(0, 5, N // 2), np.ldexp(rng.uniform(-1, 1, N // 4), rng.integers(-60, 10, N // 4)), np.zeros(N // 8), rng.normal(0, 5, N // 8)])
    xa = np.concatenate([rng.normal(0, 5, N // 2), rng.normal(0, 5, N // 4), rng.normal(0, 5, N // 8), np.zeros(N // 8)])
    pos = np.ldexp(rng.uniform(0.5, 1, N), rng.integers(-300, 300, N))
    report = {}

    def dev(fn, a, b=None):
        ta = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        tb = torch.from_numpy(np.ascontiguousarray(b)).cuda() if b is not None else None
        out = torch.empty_like(ta)
        L.check(lib.hope_debug_math(FN[fn], ta.numel(), C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()) if tb is not None else None,
                                    C.c_void_p(out.data_ptr()), None), 'hope_debug_math')
        torch.cuda.synchronize()
        return out.cpu().numpy()
    with np.errstate(all='ignore'):
        for name, got, want, bound in (
                ('sin', dev('sin', h), np.sin(h), 3), ('cos', dev('cos', h), np.cos(h), 3), ('tan(steer)', dev('tan', h[:N // 8] / 130 * 0.8), np.tan(h[:N // 8] / 130 * 0.8), 3),
                ('asin', dev('asin', unit), np.arcsin(unit), 3), ('acos', dev('acos', unit), np.arccos(unit), 3),
                ('atan2', dev('atan2', ya, xa), np.arctan2(ya, xa), 3), ('hypot', dev('hypot', ya, xa), np.hypot(ya, xa), 1),
                ('tanh(t / 2000)', dev('tanh', np.abs(h[:N // 8]) / 1300), np.tanh(np.abs(h[:N // 8]) / 1300), 3)):
            u = ulps(got, want)
            u = u[np.isfinite(u)]
            report[name] = float(u.max())
            assert u.max() <= bound, (name, float(u.max()))
        f = dev('fmod', h, np.full_like(h, 2 * math.pi))
        assert np.array_equal(f, np.fmod(h, 2 * math.pi))
        sq = dev('sqrt', pos)
        assert np.array_equal(sq, np.sqrt(pos))
        # the reward's fold acos(cos(d)) over heading differences of the unwrapped chain (composition error)
        d = h[:N // 4]
        fold = dev('acos', dev('cos', d))
        u = np.abs(fold - np.arccos(np.cos(d)))
        report['acos(cos(d)) abs'] = float(u.max())
        assert u.max() < 1e-10          # (acos is ill-conditioned at +-1; the host build of the same source gives 1.8e-12)
    print('device hope_math.h vs glibc, max ulp over 1e7 arguments (fmod, sqrt bit-equal):', report)
