#!/usr/bin/env python3
"""Where does a background pool commit cost the step loop?  Runs the bench's steady loop with per-step device events and host
timestamps, and prints the steps around every commit: host time spent in poll() (the commit call), device time of the step.
    python tools/commit_cost.py [--steps 1200]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=1200)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--pool', type=int, default=8192)
    a = ap.parse_args()
    import torch
    sys.argv = ['bench']
    import bench
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import PoolRefresher, generate_arrays
    from hope_amd.scenes import pack_scenes
    N = 65536
    rng = np.random.default_rng(42)
    uniq = bench.make_scenes(2048, 'mixed', rng)
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    reps = (N + len(uniq) - 1) // len(uniq)
    tile = lambda x: np.concatenate([x] * reps, axis=0)[:N]  # noqa: E731
    env = ParkingBatch(N, 128, overlap=True)
    for s0 in range(0, N, 8192):
        sl = slice(s0, min(N, s0 + 8192))
        env.set_scene_arrays(np.arange(sl.start, sl.stop), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
    levels = ('Normal', 'Complex', 'Extrem')
    parts = [generate_arrays(lv, a.pool // 3, seed=17 + j, max_obst=128) for j, lv in enumerate(levels)]
    env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
    env.set_dlp_cases()
    env.set_redraw_seed(7)
    ref = PoolRefresher(env, a.pool // 3 * 3, levels=levels, seed=11, relaxed=True, threads=a.threads)
    g = torch.Generator(device='cuda').manual_seed(1)
    acts = [torch.rand((N, 2), device='cuda', generator=g) * 2 - 1 for _ in range(8)]
    env.reset_obs()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host_poll = np.zeros(a.steps)
    host_step = np.zeros(a.steps)
    commits = []
    for i in range(100):
        env.step(acts[i % 8], auto_reset=True, fresh=True, defer_rs=True)
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(a.steps):
        t0 = time.perf_counter()
        if i % 9 == 0:
            if ref.poll():
                commits.append(i)
        t1 = time.perf_counter()
        env.step(acts[i % 8], auto_reset=True, fresh=True, defer_rs=True)
        t2 = time.perf_counter()
        ev[i + 1].record()
        host_poll[i] = (t1 - t0) * 1e3
        host_step[i] = (t2 - t1) * 1e3
    torch.cuda.synchronize()
    dev = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)])
    print(f'pool {a.pool}, HSA_ENABLE_SDMA={os.environ.get("HSA_ENABLE_SDMA")}; steps {a.steps}, commits at {commits}')
    print(f'device ms per step: median {np.median(dev):.4f} mean {dev.mean():.4f}; host ms per step call: median {np.median(host_step):.4f}')
    base = np.median(dev)
    for c in commits:
        w = slice(max(0, c - 2), min(a.steps, c + 12))
        print(f'commit at step {c}: host poll {host_poll[c]:.3f} ms; device ms of steps {w.start}..{w.stop - 1}: ' + ' '.join(f'{x:.3f}' for x in dev[w]) +
              f'  | excess over median in the window: {(dev[w] - base).sum():.3f} ms; host step calls: ' + ' '.join(f'{x:.2f}' for x in host_step[w]))
    ref.close()


if __name__ == '__main__':
    main()
