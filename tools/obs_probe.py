#!/usr/bin/env python3
"""Vector / scalar instructions of the small-tile observation launch by phase, two-scenes-per-wave kernel (k_obs_pair) against the
one-scene kernel (k_env_step<.., 2>, stage bit 0x8000), on the SAME steady-state scene population: full steps with the profiling
switches 0x1000 (no beam pairs) / 0x2000 (no mask).  Run under rocprofv3 --pmc, then --reduce:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d /tmp/op -- python tools/obs_probe.py --seq gpurun_out/obs_probe_seq.json
  python tools/obs_probe.py --seq gpurun_out/obs_probe_seq.json --reduce /tmp/op"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = (('obs', 0), ('obs -mask', 0x2000), ('obs -beams', 0x1000), ('obs -beams -mask', 0x3000))


def reduce(seq_path, d):
    seq = json.load(open(seq_path))
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_obs_pair' in r['Kernel_Name'] or ('k_env_step' in r['Kernel_Name'] and ', 2, ' in r['Kernel_Name'])]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        by[int(r['Dispatch_Id'])]['_k'] = 'pair' if 'k_obs_pair' in r['Kernel_Name'] else 'single'
        by[int(r['Dispatch_Id'])]['_g'] = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
    ids = sorted(by)
    # launches of the small-tile class only: pair kernel launches, and single-kernel launches whose grid is the small class's
    small = seq['small']
    sel = [i for i in ids if by[i]['_k'] == 'pair' or by[i]['_g'] == small * 64]
    tail = sel[-len(seq['launches']):]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, name in zip(tail, seq['launches']):
        for c, v in by[i].items():
            if not c.startswith('_'):
                acc[name][c].append(v / small)
    for name in seq['order']:
        print(f'{name:28s} ' + '  '.join(f'{c} {np.mean(v):9.1f}' for c, v in sorted(acc[name].items())) + '   per scene')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=65536)
    ap.add_argument('--seq', default='gpurun_out/obs_probe_seq.json')
    ap.add_argument('--reduce')
    args = ap.parse_args()
    if args.reduce:
        return reduce(args.seq, args.reduce)
    import torch
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    os.environ['HOPE_SPLIT_MIN'] = '1'
    N = args.scenes
    arrs = mixed_arrays(N, seed=3, max_obst=128)
    env = ParkingBatch(N, 128, overlap=True)
    env.set_scene_arrays(np.arange(N), *arrs[:5])
    parts = [generate_arrays(lv, 1024, seed=5 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
    env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
    env.set_dlp_cases()
    env.reset_obs()
    env.upload_state(t=np.random.default_rng(1).integers(1, 200, N))
    g = torch.Generator(device=env.device); g.manual_seed(0)
    acts = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(8)]
    for i in range(200):
        env.step(acts[i % 8], auto_reset=True, fresh=True)
    torch.cuda.synchronize()
    pose, tt, acc = env.download_state()
    launches, order = [], []
    for kern, bit in (('pair', 0), ('single', 0x8000)):
        for name, sw in VARIANTS:
            order.append(f'{kern}: {name}')
            for i in range(3):
                env.upload_state(pose=pose, t=tt, accum=acc)
                env.step(acts[i % 8], stages=L.STAGE_ALL | sw | bit)
                torch.cuda.synchronize()
                launches.append(f'{kern}: {name}')
    small = int((arrs[4] <= 32).sum())
    os.makedirs(os.path.dirname(args.seq) or '.', exist_ok=True)
    json.dump({'scenes': N, 'small': small, 'launches': launches, 'order': order}, open(args.seq, 'w'))
    env.close()


if __name__ == '__main__':
    main()
