#!/usr/bin/env python3
"""CPU study (oracle only, no GPU): how many of the words find_rs_path tests would a coarse, strided look at their first
samples condemn?  Rolls the bench's scene mix to its stationary population with the CPU oracle, then for every scene that
passes the Reeds-Shepp gate samples each tested word (orc_rs_path_samples) and evaluates is_traj_valid per SAMPLE.
    python tools/rs_screen_study.py [--scenes 1024] [--steps 60]
Output: distribution of the first colliding sample, share of invalid words caught by a screen of the first W samples at
stride s, searches whose words are all condemned by the screen.  Design input for k_rs_validate_f's screen pass."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--seed', type=int, default=42)
    args = ap.parse_args()
    from hope_amd import tables as T
    from hope_amd.scene_gen import mixed_arrays
    from oracle import oracle as O
    n, mo = args.scenes, 128
    start, dest, bbox, verts, nob, nvert = mixed_arrays(n, seed=args.seed, max_obst=mo)
    t = T.all_tables()
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'], omp=True)
    orc = O.BatchOracle(n, mo, omp=True)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    rng = np.random.default_rng(args.seed)
    orc.reset_obs(with_rs=False)
    orc.t[:] = rng.integers(1, 201, n)
    for _ in range(args.steps):
        o = orc.step(rng.uniform(-1, 1, (n, 2)), with_rs=False)
        done = np.nonzero(o['status'] != 1)[0]
        if len(done):
            orc.restart(done)
            orc.t[done] = 0
    # one more step: the poses the searches start from
    o = orc.step(rng.uniform(-1, 1, (n, 2)), with_rs=True)
    maxc = 0.3327130214085973
    first_hit, caught = [], {}
    combos = [(128, 32), (128, 16), (96, 24), (64, 16), (160, 32), (64, 1), (64, 4), (96, 3), (128, 4), (128, 6), (128, 8), (192, 9), (192, 12), (256, 8), (256, 12), (256, 16), (160, 10), (160, 5)]
    for c in combos:
        caught[c] = 0
    n_search = n_words = n_invalid = all_dead = {c: 0 for c in combos}
    n_search = n_words = n_invalid = 0
    all_dead = {c: 0 for c in combos}
    seg0_hit = 0
    for i in range(n):
        if not (orc.t[i] > 1 and o['status'][i] == 1 and np.hypot(*(orc.pose[i, :2] - orc.dest[i, :2])) < 10.0):
            continue
        m = int(orc.n_obst[i])
        paths = O.rs_all_paths(orc.pose[i], orc.dest[i], maxc)
        if paths['n'] == 0:
            continue
        order = np.argsort(paths['L'], kind='stable')
        lmin = paths['L'][order[0]]
        tested = [pi for k, pi in enumerate(order) if not (paths['L'][pi] > 1.6 * lmin and k + 1 > 2)]
        # (pop order approximated by a stable sort; the stop rule ends at the FIRST word beyond 1.6 Lmin)
        cut = len(tested)
        for k, pi in enumerate(order):
            if paths['L'][pi] > 1.6 * lmin and k + 1 > 2:
                cut = k
                break
        tested = list(order[:cut])
        n_search += 1
        dead = {c: True for c in combos}
        found = False
        for pi in tested:
            xyz = O.rs_path_samples(orc.pose[i], orc.dest[i], maxc, int(pi))
            n_words += 1
            bad = np.array([not O.is_traj_valid(xyz[k:k + 1], orc.verts[i, :m], orc.nvert[i, :m], orc.bbox[i]) for k in range(min(len(xyz), 400))])
            if not bad.any():
                found = True
                for c in combos:
                    dead[c] = False
                break
            n_invalid += 1
            fh = int(np.argmax(bad))
            first_hit.append(fh)
            for (w, s) in combos:
                idx = np.arange(s - 1, min(w, len(bad)), s)        # samples s-1, 2s-1, ... of the first w
                if bad[idx].any():
                    caught[(w, s)] += 1
                else:
                    dead[(w, s)] = False
        for c in combos:
            all_dead[c] += dead[c]
    fh = np.array(first_hit)
    print(f'scenes {n}, searches {n_search} ({n_search / n:.2f} of the scenes), words tested {n_words} ({n_words / max(n_search, 1):.2f} per search), invalid {n_invalid}')
    print('first colliding sample of an invalid word: median %d, quartiles %d / %d, <16: %.2f  <32: %.2f  <64: %.2f  <128: %.2f' % (
        np.median(fh), np.percentile(fh, 25), np.percentile(fh, 75), (fh < 16).mean(), (fh < 32).mean(), (fh < 64).mean(), (fh < 128).mean()))
    for c in combos:
        print(f'screen first {c[0]:3d} samples, stride {c[1]}: condemns {caught[c] / max(n_invalid, 1):.3f} of the invalid words; '
              f'searches with every tested word condemned: {all_dead[c] / max(n_search, 1):.3f}')


if __name__ == '__main__':
    main()
