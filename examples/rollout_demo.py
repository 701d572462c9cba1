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
    rec = D.gather_eval_stats(env.status, torch.zeros_like(env.status), env.reward, torch.zeros_like(env.reward))
    if rank == 0:
        print(f'{world} rank(s) x {args.scenes} scenes: {args.scenes * world * args.steps / dt / 1e6:.2f} M env+agent steps/s, '
              f'{s["episodes"]} episodes on rank 0, success rate {s["success_rate"]:.3f}, RS replay active {s["executing_rs"]:.3f}, '
              f'gathered {len(rec)} eval records')


if __name__ == '__main__':
    main()
