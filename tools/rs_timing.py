#!/usr/bin/env python3
"""Cycle accounting of k_rs_validate (instrumented build, HOPE_RS_TIMING=1): where a search's cycles go.
Usage (GPU box):  HOPE_RS_TIMING=1 python tools/rs_timing.py [--scenes 65536]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HOPE_RS_TIMING', '1')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
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
    env = ParkingBatch(N, 128, overlap=False, profile=True)
    for a in range(0, N, 8192):
        sl = slice(a, min(N, a + 8192))
        env.set_scene_arrays(np.arange(sl.start, sl.stop), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
    g = torch.Generator(device='cuda').manual_seed(1)
    env.reset_obs()
    for _ in range(12):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    out = (C.c_uint64 * 16)()
    lib = L.load_library()
    lib.hope_debug_rs_prof(out, 1)
    env.kernel_ms()
    for _ in range(args.steps):
        env.step(torch.rand((N, 2), device='cuda', generator=g) * 2 - 1, auto_reset=True)
    lib.hope_debug_rs_prof(out, 1)
    km = env.kernel_ms()
    print({k: round(a / max(b, 1), 4) for k, (a, b) in km.items() if b}, 'ms per launch')
    v = np.array(list(out), dtype=np.float64)
    waves = max(v[8], 1)
    names = ['prologue', 'word setup', 'sample generator', 'interpolate + transform', 'pose_hits: hull / union / cull',
             'pose_hits: candidate loop']
    extra = {14: 'screen: wait for obstacle view + tables', 15: 'screen: samples, poses, cull', 7: 'screen: (sample, obstacle) pairs'}
    tot = v[6]
    print(f'searches (waves with work) per step: {waves / args.steps:.0f};  cycles per search: {tot / waves:.0f}')
    acc = 0
    for i, nm in enumerate(names):
        print(f'  {nm:34s} {v[i] / waves:9.0f} cycles/search  {100 * v[i] / tot:5.1f} %')
        acc += v[i]
    for i, nm in extra.items():
        if v[i]:
            print(f'  {nm:34s} {v[i] / waves:9.0f} cycles/search  {100 * v[i] / tot:5.1f} %')
            acc += v[i]
    print(f'  {"rest (bookkeeping, carry, output)":34s} {(tot - acc) / waves:9.0f} cycles/search  {100 * (tot - acc) / tot:5.1f} %')
    print(f'  words tested / search {v[9] / waves:.2f}; generator rounds {v[10] / waves:.2f}; passes {v[11] / waves:.2f}; '
          f'passes with candidates {v[12] / waves:.2f}; candidate obstacles per such pass {v[13] / max(v[12], 1):.2f}')


if __name__ == '__main__':
    main()
