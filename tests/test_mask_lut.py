"""CPU tests of the action mask's count-interval table (round 6; hope_env_upload_tables -> k_obs_pair's mask stage): the table the
LIBRARY builds (hope_debug_mask_lut, pure host code) brackets the reference's first-exceed counts (src/model/action_mask.py:166-177)
for every scan value that can fall into a bin, and the kernel's two-visit evaluation -- mirrored here in numpy, same float64
expressions -- returns the exact coarse-beam minimum or raises the tie flag whenever an entry lies within 1e-9 of the scan."""
import ctypes as C

import numpy as np
import pytest

from hope_amd import load_library
from hope_amd import tables as T

NB, ROW, NBEAM, NACT, NITER, UPS = 128, 32, 120, 42, 10, 10


@pytest.fixture(scope='module')
def lut():
    t = T.all_tables()
    ds = np.ascontiguousarray(t['dist_star'], np.float64)           # [1200][42][10]
    hb = np.ascontiguousarray(t['hull_base'], np.float64)
    out = np.zeros((NBEAM * NB + 1, ROW), np.uint16)
    sc = np.zeros(NBEAM)
    lib = load_library()
    assert lib.hope_debug_mask_lut(ds.ctypes.data, hb.ctypes.data, out.ctypes.data, sc.ctypes.data) == 0
    tab = np.maximum.accumulate(ds[::UPS], axis=2)                  # [120][42][10] prefix-maxed coarse rows
    return dict(lut=out, scale=sc, tab=tab, hb=hb, pmax=tab.max(axis=(1, 2)))


def unpack(v):
    """-> cnt_lo[.., 42], cnt_hi[.., 42] from rows of 32 uint16 (word hl: forward action hl | backward action 21 + hl)"""
    w = v[..., :NACT // 2].astype(np.int64)
    lo = np.concatenate([w & 15, (w >> 8) & 15], axis=-1)
    hi = np.concatenate([(w >> 4) & 15, w >> 12], axis=-1)
    return lo, hi


def test_padding_and_neutral_row(lut):
    assert (lut['lut'][:, NACT // 2:] == 0xAAAA).all() and (lut['lut'][-1] == 0xAAAA).all()
    lo, hi = unpack(lut['lut'][:-1])
    assert (lo <= hi).all() and hi.max() <= NITER
    # a useful table: most entries are decided by the bin alone
    assert (lo == hi).mean() > 0.93


def bins_of(L, i, x):
    q = (x - (L['hb'][i] - 1e-6)) * L['scale'][i]                   # the kernel's expression, float64
    return q


def test_brackets_the_float64_counts(lut):
    """every scan value, also those on bin edges, table entries and their 1e-9 neighbourhoods"""
    L = lut
    rng = np.random.default_rng(0)
    for i in range(NBEAM):
        lo_i = L['hb'][i] - 1e-6
        xs = [rng.uniform(L['hb'][i], L['pmax'][i] + 1e-3, 4000)]
        edges = lo_i + np.arange(NB + 1) / L['scale'][i]
        for d in (0.0, 1e-12, -1e-12, 1e-9, -1e-9, 3e-9, -3e-9):
            xs.append(edges + d)
            xs.append(L['tab'][i].ravel() + d)
        x = np.concatenate(xs)
        x = x[x >= L['hb'][i]]
        q = bins_of(L, i, x)
        act = q < NB
        # beyond the last bin every entry is <= x - 1e-9 (the beam cannot lower any count)
        assert (L['tab'][i].max() <= x[~act] - 1e-9).all()
        x, q = x[act], q[act]
        b = q.astype(np.int64)
        lo, hi = unpack(L['lut'][i * NB + b])                       # [n, 42]
        cnt = (L['tab'][i][None] <= x[:, None, None]).sum(-1)
        cnt_m = (L['tab'][i][None] <= (x - 1e-9)[:, None, None]).sum(-1)
        assert (lo <= cnt_m).all() and (cnt <= hi).all()


def kernel_mask_counts(L, x):
    """numpy mirror of the mask stage of k_obs_pair (HOPE_MASK_LUT): -> (ms[42], tie)"""
    q = (x - (L['hb'] - 1e-6)) * L['scale']
    act = np.nonzero(q < NB)[0]
    if len(act) == 0:
        return np.full(NACT, NITER), False
    rows = act * NB + q[act].astype(np.int64)
    lo, hi = unpack(L['lut'][rows])                                 # [n_act, 42]
    mhi = hi.min(0).copy()
    tie = False
    if (lo.min(0) < mhi).any():
        for j, i in enumerate(act):
            for a in range(NACT):
                if lo[j, a] < hi[j, a] and lo[j, a] < mhi[a]:
                    c, lim = lo[j, a], min(hi[j, a], mhi[a])
                    while c < lim:
                        t = L['tab'][i, a, c]
                        if t > x[i]:
                            break
                        if t > x[i] - 1e-9:
                            tie = True
                        c += 1
                    mhi[a] = min(mhi[a], c)
    return mhi, tie


def exact_counts(L, x, shift=0.0):
    return (L['tab'] <= (x - shift)[:, None, None]).sum(-1).min(0)


def test_two_visit_evaluation_is_exact_or_ties(lut):
    L = lut
    rng = np.random.default_rng(1)
    n_tie = 0
    for trial in range(3000):
        kind = trial % 4
        x = L['hb'] + rng.uniform(0, 10, NBEAM)                     # nothing near
        k = rng.integers(1, 40)
        near = rng.choice(NBEAM, k, replace=False)
        x[near] = L['hb'][near] + rng.uniform(0, 1.0, k) * (L['pmax'][near] - L['hb'][near]) * 1.2
        if kind == 1:                                               # scan values ON table entries (the structural tie) and just beside them
            for i in near[:4]:
                x[i] = max(L['hb'][i], L['tab'][i].ravel()[rng.integers(NACT * NITER)] + rng.choice([0.0, 5e-10, -5e-10, 2e-9, -2e-9]))
        if kind == 2:                                               # touching obstacle: the scan is the hull range itself
            x[near[:3]] = L['hb'][near[:3]]
        ms, tie = kernel_mask_counts(L, x)
        m, mlow = exact_counts(L, x), exact_counts(L, x, 1e-9)
        if (m != mlow).any():
            assert tie, 'an entry within 1e-9 of the scan must send the scene to the exact evaluation'
        if not tie:
            assert (ms == m).all()
        n_tie += tie
    assert 0 < n_tie < 3000
