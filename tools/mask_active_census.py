#!/usr/bin/env python3
"""How many coarse beams the action-mask stage has to probe, on the bench's stationary scene population: per scene the number of
beams whose scan value lies below the table's maximum at that beam (`x_i - 1e-9 < pmax_i`, hope_step_kernel.h), by tile class,
and for pairs of small-tile scenes (k_obs_pair visits max(n_a, n_b) beams).  float64 observations: the same values the kernel sees.
  python tools/mask_active_census.py [--scenes 65536] [--steps 300]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--bank', type=int, default=0, help='cycle through this many fixed action tensors (bench.py: 16) instead of fresh draws every step')
    args = ap.parse_args()
    import torch
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    N = args.scenes
    arrs = mixed_arrays(N, seed=3, max_obst=128)
    env = ParkingBatch(N, 128, overlap=True, obs_dtype=torch.float64)
    env.set_scene_arrays(np.arange(N), *arrs[:5])
    parts = [generate_arrays(lv, 1024, seed=5 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
    env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
    env.set_dlp_cases()
    env.reset_obs()
    env.upload_state(t=np.random.default_rng(1).integers(1, 200, N))
    g = torch.Generator(device=env.device); g.manual_seed(0)
    bank = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(args.bank)]
    for i in range(args.steps):
        a = bank[i % args.bank] if args.bank else torch.rand((N, 2), generator=g, device=env.device) * 2 - 1
        env.step(a, auto_reset=True, fresh=True)
    print('actions:', f'a bank of {args.bank} tensors, cycled' if args.bank else 'fresh U[-1,1]^2 draws every step', '|', args.steps, 'steps,', N, 'scenes')
    torch.cuda.synchronize()
    lid = env.lidar.cpu().numpy()
    nob = env.n_obst_now()
    t = env.tables
    tab = np.maximum.accumulate(t['dist_star'][::10], axis=2)
    pmax = tab.max(axis=(1, 2))
    x = np.clip(lid, 0, 10) + t['hull_base'][None]
    na = (x - 1e-9 < pmax[None]).sum(1)
    for name, sel in (('small tile (<= 32 obstacles)', nob <= 32), ('large tile', nob > 32)):
        v = na[sel]
        print(f'{name}: {sel.sum()} scenes, active coarse beams mean {v.mean():.2f}  none {np.mean(v == 0):.3f}  1-4 {np.mean((v > 0) & (v <= 4)):.3f}  '
              f'5-8 {np.mean((v > 4) & (v <= 8)):.3f}  9-16 {np.mean((v > 8) & (v <= 16)):.3f}  > 16 {np.mean(v > 16):.3f}  max {v.max()}')
    v = na[nob <= 32]
    m = np.maximum(v[0:len(v) // 2 * 2:2], v[1:len(v) // 2 * 2:2])
    print(f'pairs of small-tile scenes (list neighbours): none {np.mean(m == 0):.3f}  1-4 {np.mean((m > 0) & (m <= 4)):.3f}  5-8 {np.mean((m > 4) & (m <= 8)):.3f}  '
          f'> 8 {np.mean(m > 8):.3f}  probe groups of 4 per wave {np.ceil(m / 4).mean():.2f}')
    print('nearest obstacle beyond the hull, percentiles 10/25/50/75/90 (m):', np.percentile(lid[nob <= 32].min(1), [10, 25, 50, 75, 90]).round(2))
    env.close()


if __name__ == '__main__':
    main()
