#!/usr/bin/env python3
"""End-to-end batched rollout on one GPU (or one rank per GPU under torch.distributed.run): scenes -> env step
(auto-reset) -> StateNorm -> stand-in policy -> mask-weighted sampling / RS-path replay.
usage: python examples/rollout_demo.py [--scenes 16384] [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hope_amd import ParkingBatch
from hope_amd import dist as D
from hope_amd.rollout import BatchedRollout
from hope_amd.scenes import SceneSource


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=16384)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--unique', type=int, default=1024)
    args = ap.parse_args()
    rank, world, local = D.init_from_env()
    dev = f'cuda:{local}'
    src = SceneSource(seed=42 + rank)
    uniq = [src.draw() for _ in range(args.unique)]
    env = ParkingBatch(args.scenes, 128, device=dev)
    env.set_scenes(np.arange(args.scenes), [uniq[i % len(uniq)] for i in range(args.scenes)])
    ro = BatchedRollout(env, seed=rank)
    for _ in range(10):
        ro.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = ro.stats()
    if rank == 0:
        print(f'{world} rank(s) x {args.scenes} scenes: {args.scenes * world * args.steps / dt / 1e6:.2f} M env+agent steps/s, '
              f'{s["episodes"]} episodes on rank 0, success rate {s["success_rate"]:.3f}, RS replay active {s["executing_rs"]:.3f}')
    # evaluation as eval_utils.py does it (one episode per slot, per-level table), records gathered over the ranks
    from hope_amd import agents as A
    from hope_amd import evaluate as E
    ag = A.BatchedPPO(device=dev, use_img=False)
    env.set_scenes(np.arange(args.scenes), [uniq[i % len(uniq)] for i in range(args.scenes)])
    rec = E.BatchedEvaluator(env, ag, seed=rank).run()
    if rank == 0:
        levels = [uniq[i % len(uniq)].level for i in range(args.scenes)] * world
        for k, v in E.summarize(rec, levels).items():
            print(f'  eval {k:8s}: {v["episodes"]} episodes, success {v["success_rate"]:.3f}, steps {v["step_num_mean"]:.1f} +- {v["step_num_std"]:.1f}, '
                  f'path {v["path_length_mean"]:.2f} m')


if __name__ == '__main__':
    main()
