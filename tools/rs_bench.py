#!/usr/bin/env python3
"""Kernel times of the step with SEQUENTIAL launches (no stream overlap) on the bench workload: a stable number per kernel for
A/B experiments on one kernel (environment variables select variants, e.g. HOPE_RS_EXACT=1, HOPE_RS_OCC=5, HOPE_RS_DEBUG=0x8000).
Usage (GPU box):  python tools/rs_bench.py [--scenes 65536] [--steps 30]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    import torch
    sys.argv = ['bench']
    import bench
    from hope_amd import ParkingBatch
    from hope_amd.scenes import pack_scenes
    rng = np.random.default_rng(42)
    uniq = bench.make_scenes(2048, 'mixed', rng)
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    N = args.scenes
    reps = (N + len(uniq) - 1) // len(uniq)
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:N]  # noqa: E731
    env = ParkingBatch(N, 128, overlap=False, profile=True)
    for a in range(0, N, 8192):
        sl = slice(a, min(N, a + 8192))
        env.set_scene_arrays(np.arange(sl.start, sl.stop), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
    g = torch.Generator(device='cuda').manual_seed(1)
    env.reset_obs()
    for _ in range(12):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    env.kernel_ms()
    for _ in range(args.steps):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    km = env.kernel_ms()
    per_step = {k: round(a / args.steps, 4) for k, (a, b) in km.items() if b}
    print(args.tag, 'ms per step, sequential launches:', per_step, 'sum', round(sum(per_step.values()), 4))


if __name__ == '__main__':
    main()
