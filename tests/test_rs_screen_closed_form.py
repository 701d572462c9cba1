"""The closed-form sample positions of k_rs_validate_f's SCREEN pass (hope_amd/csrc/hope_rs.hip) against the sequential
`pd += d` walk of generate_local_course (src/env/reeds_shepp.py:452-507) they stand for: every sample the screen uses must be
a sample of the walk (same segment, pd equal to ~1e-12), whichever way ties at segment ends fall.  Words come from the oracle's
calc_all_paths on random queries plus hand-made tie cases (segment lengths that are exact multiples of the step).
Pure host arithmetic: a Python mirror of the device loop, kept next to it line by line."""
import math

import numpy as np

from oracle import oracle as O

MAXC = 0.3327130214085973
STEP = 0.1 * MAXC
EPS = 1e-7
SPAN = 128


def walk(lengths):
    """generate_local_course's samples as (segment, pd), the start pose and the final end point left out (hope_rs.hip's generator)."""
    out = []
    ll, lprev, pd = 0.0, 0.0, 0.0
    for i, l in enumerate(lengths):
        d = STEP if l > 0.0 else -STEP
        pd = (-d - ll) if (i >= 1 and lprev * l > 0) else (d - ll)
        while abs(pd) <= abs(l):
            out.append((i, pd))
            pd += d
        ll = l - pd - d
        lprev = l
    return out


def screen_samples(lengths, lanes):
    """the device loop: lane k of a word -> (segment, pd) or None"""
    stride = SPAN // lanes
    inv = 1.0 / STEP
    res = []
    for k in range(lanes):
        a = float((k + 1) * stride) * STEP
        c, u0, r, lprev, si, pdv = 0.0, STEP, 0.0, 0.0, -1, 0.0
        for i, l in enumerate(lengths):
            al = abs(l)
            if i >= 1:
                u0 = r if lprev * l > 0 else -r
            if si < 0 and a < c + al:
                j = max(math.ceil((a - c - u0) * inv), 0.0)
                u = u0 + j * STEP
                si = i if (abs(u0) <= al and EPS < u < al - EPS) else 5
                pdv = u if l > 0.0 else -u
            n = 0.0 if abs(u0) > al else math.floor((al - u0) * inv) + 1.0
            r = u0 + n * STEP - al
            c += al
            lprev = l
        res.append((si, pdv) if 0 <= si < 5 else None)
    return res


def check_word(lengths):
    ref = walk(lengths)
    by_seg = {}
    for i, pd in ref:
        by_seg.setdefault(i, []).append(pd)
    used = 0
    for lanes in (64, 32, 21, 16):
        for smp in screen_samples(lengths, lanes):
            if smp is None:
                continue
            i, pd = smp
            cands = np.array(by_seg.get(i, [np.inf]))
            assert np.abs(cands - pd).min() < 1e-11, (lengths, lanes, i, pd)
            used += 1
    return used


def test_screen_samples_are_samples_of_the_walk_random_words():
    rng = np.random.default_rng(5)
    used = words = 0
    for _ in range(400):
        q0 = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-np.pi, np.pi)])
        q1 = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-np.pi, np.pi)])
        paths = O.rs_all_paths(q0, q1, MAXC)
        for w in range(paths['n']):
            ln = [float(x) for x in paths['lengths'][w][:paths['nseg'][w]]]
            used += check_word(ln)
            words += 1
    assert words > 2000 and used > 50 * words


def test_screen_samples_survive_ties_at_segment_ends():
    """segment lengths that are exact multiples of the step (and one ulp either side), both directions, cusps"""
    used = 0
    for n1 in (1, 3, 17, 40):
        for n2 in (2, 9, 33):
            for s1 in (1, -1):
                for s2 in (1, -1):
                    for bump in (0.0, 1.0, -1.0):
                        l1 = s1 * n1 * STEP
                        l1 = np.nextafter(l1, l1 + bump * 10) if bump else l1
                        for acc in (False, True):
                            if acc:                      # the walk's own n-fold sum of the step instead of the product
                                t = 0.0
                                for _ in range(n1):
                                    t += STEP
                                l1 = s1 * t
                            ln = [float(l1), s2 * n2 * STEP, -s2 * 1.37, s1 * 0.05 * STEP, 2.2]
                            used += check_word(ln)
    assert used > 10000


def test_screen_skips_a_short_segment_behind_a_cusp_with_a_large_remainder():
    """|l_i| < |u0_i| = the remainder the cusp carries in: generate_local_course takes NO lattice sample on that segment
    (reeds_shepp.py:488), so neither may the screen (it selected pd = 0.15 step on segment 2 of the first word)"""
    used = 0
    used += check_word([10.9 * STEP, -5.05 * STEP, 0.3 * STEP])
    for big in (10.9, 3.95, 7.5, 20.05):
        for mid in (-5.05, -0.4, -2.95, 4.05):
            for short in (0.3, -0.3, 0.6, -0.85, 0.05):
                for tail in ((), (6.3,), (-4.2, 9.1)):
                    used += check_word([big * STEP, mid * STEP, short * STEP] + [t * STEP for t in tail])
                    used += check_word([-big * STEP, -mid * STEP, -short * STEP] + [-t * STEP for t in tail])
    assert used > 1000
