#!/usr/bin/env python3
"""Which part of k_env_step produces the memory-side traffic?  Runs the step kernel with different stage masks on a
fixed scene population and writes the launch sequence (config name per k_env_step launch) to a JSON; run it under
`rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE` and join with `--reduce <counter_collection.csv>`.

  python tools/pmc_stage_probe.py --seq gpurun_out/probe_seq.json            # (under rocprofv3 --pmc ...)
  python tools/pmc_stage_probe.py --seq gpurun_out/probe_seq.json --reduce <csv> WRITE_SIZE
"""
import argparse
import collections
import csv
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reduce(seq_path, csv_path, counter):
    seq = json.load(open(seq_path))
    rows = [r for r in csv.DictReader(open(csv_path)) if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in ('k_env_step', 'k_obs_pair', 'k_motion_pair'))]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    assert len(rows) == len(seq['launches']), (len(rows), len(seq['launches']))
    acc = collections.defaultdict(list)
    for r, (name, scenes) in zip(rows, seq['launches']):
        acc[name].append(float(r['Counter_Value']) * (1 if counter.startswith('SQ_') else 1024) / seq['scenes'])
    for name, v in acc.items():
        unit = 'wave-instructions/scene (per launch)' if counter.startswith('SQ_') else 'B/scene'
        print(f'{counter:12s} {name:28s} {sum(v) / len(v):9.1f} {unit}  ({len(v)} launches)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=32768)
    ap.add_argument('--seq', default='gpurun_out/probe_seq.json')
    ap.add_argument('--reduce', nargs=2)
    ap.add_argument('--mix', default='normal', choices=['normal', 'mixed'], help='mixed: the bench scene mix (all four levels)')
    args = ap.parse_args()
    if args.reduce:
        return reduce(args.seq, *args.reduce)
    import torch
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import SceneSource, pack_scenes
    N = args.scenes
    src = SceneSource(levels=('Normal',), seed=3) if args.mix == 'normal' else SceneSource(seed=3)
    uniq = [src.draw() for _ in range(512)]
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    reps = N // len(uniq)
    tile = lambda a: np.concatenate([a] * reps, axis=0)  # noqa: E731
    env = ParkingBatch(N, 128, profile=True)
    env.set_scene_arrays(np.arange(N), tile(start), tile(dest), tile(bbox), tile(verts), tile(nob))
    g = torch.Generator(device=env.device); g.manual_seed(0)
    acts = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(4)]
    launches = []

    def count(name):
        torch.cuda.synchronize()
        n = env.kernel_ms(reset=True)['k_env_step'][1]
        launches.extend([[name, N]] * n)

    env.kernel_ms(reset=True)
    env.reset_obs(stages=L.STAGE_ALL)
    for i in range(6):
        env.step(acts[i % 4], stages=L.STAGE_ALL)
        env.restart(env.done)
        env.reset_obs(active=env.done)
    count('warmup')
    pose, tt, acc = env.download_state()
    for name, st in (('motion', L.STAGE_MOTION), ('motion+reward', L.STAGE_MOTION | L.STAGE_REWARD),
                     ('obs', L.STAGE_OBS), ('obs -mask', L.STAGE_OBS | 0x2000), ('obs -beams -mask', L.STAGE_OBS | 0x3000),
                     ('motion+obs+reward', L.STAGE_MOTION | L.STAGE_OBS | L.STAGE_REWARD), ('all', L.STAGE_ALL)):
        for i in range(3):
            env.upload_state(pose=pose, t=tt, accum=acc)
            env.step(acts[i % 4], stages=st)
        count(name)
    os.makedirs(os.path.dirname(args.seq) or '.', exist_ok=True)
    json.dump({'scenes': N, 'launches': launches}, open(args.seq, 'w'))
    env.close()


if __name__ == '__main__':
    main()
