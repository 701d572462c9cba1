#!/usr/bin/env python3
"""Per-stage cost of the step kernel: times hope_env_step with different stage masks / scene mixes using the
in-library HIP events.  Usage: python tools/stage_times.py [--scenes 32768] [--preroll 200] [--bank 0]
Round 6: the measured poses are those of the bench's stationary population -- random episode ages, `--preroll` full steps with a fresh
U[-1,1]^2 action per scene-step (`--bank 4 --preroll 12` restores rounds 1-5: twelve steps of four repeated action tensors, which leaves
70 % of the generated lots with active mask beams instead of 34 %)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=32768)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--preroll', type=int, default=200)
    ap.add_argument('--bank', type=int, default=0, help='cycle this many fixed action tensors in the pre-roll (0: a fresh draw per step)')
    args = ap.parse_args()
    import torch
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import SceneSource, pack_scenes
    N = args.scenes
    for mix in (('dlp',), ('Normal', 'Complex', 'Extrem')):
        src = SceneSource(levels=mix, seed=3)
        uniq = [src.draw() for _ in range(512)]
        start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
        reps = N // len(uniq)
        tile = lambda a: np.concatenate([a] * reps, axis=0)  # noqa: E731
        env = ParkingBatch(N, 128, profile=True)
        for a in range(0, N, 8192):
            sl = slice(a, a + 8192)
            env.set_scene_arrays(np.arange(a, a + 8192), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl], tile(nob)[sl])
        g = torch.Generator(device=env.device); g.manual_seed(0)
        acts = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(4)]
        draw = lambda i: acts[i % args.bank] if args.bank else torch.rand((N, 2), generator=g, device=env.device) * 2 - 1  # noqa: E731
        pop = None
        for name, st in (('motion', L.STAGE_MOTION), ('motion+reward', L.STAGE_MOTION | L.STAGE_REWARD),
                         ('obs only(no motion)', L.STAGE_OBS), ('motion+obs+reward', L.STAGE_MOTION | L.STAGE_OBS | L.STAGE_REWARD),
                         ('all', L.STAGE_ALL), ('obs -mask', L.STAGE_OBS | 0x2000), ('obs -beams', L.STAGE_OBS | 0x1000),
                         ('obs -beams -mask', L.STAGE_OBS | 0x3000)):
            def full_step(i):          # keep the episode population realistic: finished scenes start over
                env.step(draw(i), stages=L.STAGE_ALL)
                env.restart(env.done)
                env.reset_obs(active=env.done)
            if pop is None:            # one pre-roll per scene mix: every configuration is measured on the same poses
                env.reset_obs(stages=L.STAGE_ALL)
                if args.bank == 0:
                    env.upload_state(t=np.random.default_rng(1).integers(1, 200, N))
                for i in range(args.preroll):
                    full_step(i)
                torch.cuda.synchronize()
                pop = env.download_state()
            pose, tt, acc = pop
            tot = {k: 0.0 for k in L.KERNELS}
            for i in range(args.steps):
                env.upload_state(pose=pose, t=tt, accum=acc)     # same poses for every measured configuration
                env.kernel_ms(reset=True)
                env.step(acts[i % 4], stages=st)
                torch.cuda.synchronize()
                for k, v in env.kernel_ms(reset=True).items():
                    tot[k] += v[0]
            class _K:                                            # adapter for the print below
                pass
            env_kernel = {k: (v, 1) for k, v in tot.items()}
            km = env_kernel
            per = {k: v[0] / args.steps * 1e3 for k, v in km.items()}
            tot = sum(per.values())
            print(f'{"/".join(mix):24s} {name:22s} ' + ' '.join(f'{k[2:]} {v:7.1f}' for k, v in per.items()) +
                  f' us | total {tot:7.1f} us ({N/tot:.1f} M scene-steps/s)', flush=True)
        env.close()


if __name__ == '__main__':
    main()
