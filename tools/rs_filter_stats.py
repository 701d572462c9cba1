#!/usr/bin/env python3
"""How often k_rs_validate's float32 filter decides a pass, and (with --check) its self-check against the float64 arithmetic
on the bench workload.  Usage (GPU box):  python tools/rs_filter_stats.py [--scenes 65536] [--steps 30] [--check]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--check', action='store_true', help='self-check: float64 for every sample (slow), count contradictions')
    args = ap.parse_args()
    os.environ['HOPE_RS_DEBUG'] = '0x6000' if args.check else '0x4000'
    import torch
    sys.argv = ['bench']
    import bench
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import pack_scenes
    rng = np.random.default_rng(42)
    uniq = bench.make_scenes(2048, 'mixed', rng)
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    N = args.scenes
    reps = (N + len(uniq) - 1) // len(uniq)
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:N]  # noqa: E731
    env = ParkingBatch(N, 128)
    for a in range(0, N, 8192):
        sl = slice(a, min(N, a + 8192))
        env.set_scene_arrays(np.arange(sl.start, sl.stop), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
    g = torch.Generator(device='cuda').manual_seed(1)
    env.reset_obs()
    lib = L.load_library()
    out = (C.c_uint64 * 16)()
    for _ in range(args.steps):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    lib.hope_debug_rs_filter_stats(out, 1)
    v = list(out)
    print(f'passes {v[0]}: decided by a certain float32 hit {100 * v[1] / max(v[0], 1):.1f} %, float64 re-evaluation in '
          f'{100 * v[2] / max(v[0], 1):.1f} % ({v[3] / max(v[2], 1):.1f} undecided samples each), all clear '
          f'{100 * (v[0] - v[1] - v[2]) / max(v[0], 1):.1f} %')
    names = ('no certain crossing', 'axis-parallel obstacle edge', 'axis-parallel hull', 'hull corner near the edge line', 'shallow angle')
    print('undecided passes by cause (a pass can have several): ' + ', '.join(f'{nm} {v[8 + i]}' for i, nm in enumerate(names)))
    print(f'screen pass: {v[13]} words condemned, {v[14]} searches with every tested word condemned')
    if args.check:
        print(f'screen self-check: condemned words the full walk found valid {v[7]}  (must be 0)')
        print(f'self-check over {v[6]} samples: float32 hit / float64 clear {v[4]}, float32 clear / float64 hit {v[5]}  (both must be 0)')


if __name__ == '__main__':
    main()
