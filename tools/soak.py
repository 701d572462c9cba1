#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: many seeds x (fused / two-launch step kernel) x full step incl. Reeds-Shepp search; every
output must be EXACTLY equal (float64 observation mode).  Longer than the -m gpu tests; run by hand on a GPU box:
    python tools/soak.py --seeds 12 --scenes 4096 --steps 12
Prints one line per run and exits non-zero on the first difference."""
import argparse
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run(seed, n, steps, split, level, defer=False):
    import torch
    from hope_amd import ParkingBatch
    from hope_amd.scenes import DlpScenePool, SceneSource, pack_scenes
    from oracle import oracle as O
    if split:
        os.environ['HOPE_SPLIT_MIN'] = '1'
    else:
        os.environ.pop('HOPE_SPLIT_MIN', None)
    rng = np.random.default_rng(seed)
    if level == 'dlp':
        pool = DlpScenePool()
        uniq = [pool.sample(rng=rng) for _ in range(min(n, 512))]
    else:
        src = SceneSource(seed=seed)
        uniq = [src.draw() for _ in range(min(n, 512))]
    scenes = []
    for k in range(n):
        s = copy.copy(uniq[k % len(uniq)])
        if k % 2 == 0:
            r, a = rng.uniform(0.0, 9.0), rng.uniform(0, 2 * np.pi)
            s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.7])
        scenes.append(s)
    mo = 128
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, mo, omp=True)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, mo)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    env.reset_obs()
    orc.reset_obs()
    bad = 0
    searches = found = 0
    for it in range(steps):
        act = rng.uniform(-1.2, 1.2, (n, 2))
        if it % 3 == 0:
            act[:, 1] = np.sign(act[:, 1])
        if defer:                 # HOPE_DEFER_RS: the pipelined launch structure (two sub-chains for a single-class batch), joined before the compare
            env.step(torch.from_numpy(act).to(env.device), defer_rs=True)
            env.wait_rs()
        else:
            env.step(torch.from_numpy(act).to(env.device))
        o = orc.step(act)
        torch.cuda.synchronize()
        pose, tt, acc = env.download_state()
        w = env.rs_word.cpu().numpy()
        checks = {
            'status': np.array_equal(env.status.cpu().numpy(), o['status']),
            'mask': np.array_equal(env.action_mask.cpu().numpy(), o['mask']),
            'lidar': np.array_equal(env.lidar.cpu().numpy(), o['lidar']),
            'target': np.array_equal(env.target.cpu().numpy(), o['target']),
            'reward': np.array_equal(env.reward.cpu().numpy(), o['reward']),
            'reward_info': np.array_equal(env.reward_info.cpu().numpy(), o['reward_info']),
            'pose': np.array_equal(pose, orc.pose), 't': np.array_equal(tt, orc.t.astype(np.int32)), 'accum': np.array_equal(acc, orc.accum),
            'rs_found': np.array_equal(w[:, 6], o['rs_found']), 'rs_word': np.array_equal(w[:, :5], o['rs_ctypes']),
            'rs_lengths': np.array_equal(env.rs_lengths.cpu().numpy(), o['rs_lengths']),
        }
        searches += int(((o['status'] == 1) & (np.hypot(*(orc.pose[:, :2] - dest[:, :2]).T) < 10)).sum())
        found += int(o['rs_found'].sum())
        for k, ok in checks.items():
            if not ok:
                bad += 1
                print(f'  DIFFERENCE seed {seed} step {it}: {k}')
    env.close()
    print(f'seed {seed:3d} {level:6s} split={int(split)} searches {searches:6d} found {found:5d}  differences {bad}')
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--scenes', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--defer', action='store_true', help='deferred steps (HOPE_DEFER_RS) + wait_rs; with HOPE_AUTO_CHAINS=1:1073741824:1:1073741824 the Dragon-Lake runs go through two sub-chains')
    args = ap.parse_args()
    total = 0
    for seed in range(100, 100 + args.seeds):
        for split in (False, True):
            total += run(seed, args.scenes, args.steps, split, 'mixed' if seed % 2 else 'dlp', args.defer)
    print('total differences:', total)
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
