#!/usr/bin/env python3
"""Tie census: over >= 10^8 scene-steps of the BENCH workload (65 536 mixed scenes, random actions, new-map
turnover, device-drawn Dragon-Lake cases), how close does any decision come to a tie whose outcome a conformant-but-different
GEOS (or libm) could flip?  Counted inside the instrumented step kernel (HOPE_STEP_TIMING build, hope_debug_census):
  * _check_arrived       car_parking_base.py:164-170   min |area(hull ∩ dest) / area(dest) - 0.95|
  * lidar ring cull      lidar_simulator.py:69         min |LinearRing.distance(origin) - 10 m|
  * _detect_collision    car_parking_base.py:153-158   (hull edge, obstacle edge) pairs with an orientation determinant inside its
                                                       float64 error bound -- the pairs the exact expansion had to decide
  * ActionMask.get_steps action_mask.py:173            min |dist_star entry - upsampled scan value| over the coarse compares, and
                                                       the scene-steps with an entry within 1e-9 (which take the exact 1200-beam path)
Usage (GPU box):  python tools/tie_census.py [--scene-steps 1e8] > profiles/r04_tie_census.txt"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['HOPE_STEP_TIMING'] = '1'            # the instrumented instantiation of k_env_step (one-launch form)


def census(lib, reset):
    out = (C.c_uint64 * 16)()
    rc = lib.hope_debug_census(out, int(reset))
    assert rc == 0, rc
    v = list(out)
    f = lambda u: struct.unpack('<d', struct.pack('<Q', u))[0]  # noqa: E731
    return {'arrival_evals': v[0], 'arrival_min_abs_ratio_minus_0.95': f(v[1]), 'ring_evals': v[2], 'ring_min_abs_dist_minus_10': f(v[3]),
            'orientation_pairs_undecided_by_filter': v[4], 'mask_compares': v[6], 'mask_min_abs_entry_minus_scan': f(v[7]),
            'mask_scene_steps_exact_path': v[8], 'scene_steps': v[9], 'mask_compares_bit_equal': v[10]}


def run(n_scenes=65536, scene_steps=1e8, seed=42, quiet=False):
    import torch
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scene_gen import generate_arrays, mixed_arrays
    lib = L.load_library()
    N = n_scenes
    levels = ('Normal', 'Complex', 'Extrem', 'dlp')
    n_uniq = min(N, 2048)
    start, dest, bbox, verts, nob, nvert = mixed_arrays(n_uniq, levels=levels, seed=seed, max_obst=128)
    reps = (N + n_uniq - 1) // n_uniq
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:N]  # noqa: E731
    env = ParkingBatch(N, 128, obs_dtype=torch.float32, action_dtype=torch.float32, profile=False)
    for a in range(0, N, 8192):
        b = min(N, a + 8192)
        env.set_scene_arrays(np.arange(a, b), tile(start)[a:b], tile(dest)[a:b], tile(bbox)[a:b], tile(verts)[a:b], tile(nob)[a:b])
    env.set_draw_class(np.arange(3, N, 4), 1)
    parts = [generate_arrays(lv, 8192 // 3, seed=seed * 7919 + 17 + j, max_obst=128) for j, lv in enumerate(levels[:3])]
    env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
    env.set_dlp_cases()
    env.set_redraw_seed(seed * 7919 + 1)
    g = torch.Generator(device=env.device).manual_seed(seed)
    bank = [torch.rand((N, 2), generator=g, device=env.device, dtype=torch.float32) * 2 - 1 for _ in range(64)]
    env.reset_obs()
    env.upload_state(t=np.random.default_rng(seed + 99).integers(1, 201, N).astype(np.int32))
    census(lib, True)
    steps = int(np.ceil(scene_steps / N))
    t0 = time.perf_counter()
    for i in range(steps):
        if i % 64 == 0 and i:                    # fresh actions now and then (the bank is only 64 deep)
            bank[(i // 64) % 64] = torch.rand((N, 2), generator=g, device=env.device, dtype=torch.float32) * 2 - 1
        env.step(bank[i % 64], auto_reset=True, fresh=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = census(lib, False)
    # reset_obs ran before the reset of the counters; every env.step launches both tile classes
    c['steps'] = steps
    c['seconds'] = dt
    c['relative'] = {'arrival': c['arrival_min_abs_ratio_minus_0.95'] / 0.95, 'ring': c['ring_min_abs_dist_minus_10'] / 10.0,
                     'mask_vs_largest_entry_4.37m': c['mask_min_abs_entry_minus_scan'] / 4.372115310333457}
    env.close()
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--scene-steps', type=float, default=1e8)
    ap.add_argument('--seed', type=int, default=42)
    args = ap.parse_args()
    c = run(args.scenes, args.scene_steps, args.seed)
    print(f"tie census over {c['scene_steps']:.3e} scene-steps of the bench workload ({args.scenes} scenes x {c['steps']} steps, seed {args.seed}, {c['seconds']:.1f} s)")
    print(f"  _check_arrived      : {c['arrival_evals']:.3e} polygon-clip evaluations, min |area ratio - 0.95| = {c['arrival_min_abs_ratio_minus_0.95']:.3e}  (relative {c['relative']['arrival']:.2e})")
    print(f"  lidar ring cull     : {c['ring_evals']:.3e} ring distances,           min |distance - 10 m|   = {c['ring_min_abs_dist_minus_10']:.3e} m (relative {c['relative']['ring']:.2e})")
    print(f"  _detect_collision   : {c['orientation_pairs_undecided_by_filter']} (hull edge, obstacle edge) pairs had an orientation determinant inside its float64 error bound (exact expansion decided)")
    print(f"  ActionMask.get_steps: {c['mask_compares']:.3e} coarse table compares,       min non-zero |entry - scan| = {c['mask_min_abs_entry_minus_scan']:.3e} m; {c['mask_compares_bit_equal']} compares with entry == scan bit for bit; {c['mask_scene_steps_exact_path']} scene-steps had an entry within 1e-9 of the scan (exact 1200-beam evaluation)")
    print(json.dumps(c))


if __name__ == '__main__':
    main()
