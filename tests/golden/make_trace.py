#!/usr/bin/env python3
"""Config-1 golden trace (BASELINE.json configs[0]): ONE DLP scene, 2000 wrapper steps with recorded
random actions.  The reference env cannot run end-to-end in the build container (shapely / pygame /
gym absent -- SURVEY.md §8c), so this trace is produced by the CPU ORACLE, whose every component is
pinned to reference vectors by tests/test_oracle_golden.py; the tables are the product's numpy tables
(bit-identical to the reference's on this numpy build).  Episodes that end are restarted on the same
scene.  Run from the repo root:  python tests/golden/make_trace.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from hope_amd import tables as T                    # noqa: E402
from hope_amd.scenes import DlpScenePool            # noqa: E402
from oracle import oracle as O                      # noqa: E402


def main(steps=2000, case=0, seed=1):
    pool = DlpScenePool()
    rng = np.random.default_rng(seed)
    sc = pool.sample(case=case, rng=rng, jitter=False, flips=False)
    # start a few metres in front of the slot so that collisions, the RS search and arrivals all occur
    sc.start = np.array([sc.dest[0] + 7.0 * np.cos(sc.dest[2]) + 0.4, sc.dest[1] + 7.0 * np.sin(sc.dest[2]) - 0.3,
                         sc.dest[2] + 0.25])
    t = T.all_tables()
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    mo = 128
    orc = O.BatchOracle(1, mo)
    v = np.zeros((1, mo, 4, 2)); v[0, :sc.n_obst] = sc.verts
    nv = np.full((1, mo), 4, np.int32); nv[0, :sc.n_obst] = sc.nvert
    orc.set_scenes([0], sc.start[None], sc.dest[None], sc.bbox[None], v, nv, [sc.n_obst])
    rec = dict(action=[], pose=[], status=[], reward=[], mask=[], lidar=[], target=[], rs_found=[], rs_ctypes=[],
               rs_lengths=[], reset=[])
    o = orc.reset_obs()
    first = dict(lidar=o['lidar'][0].copy(), mask=o['mask'][0].copy(), target=o['target'][0].copy(), status=int(o['status'][0]))
    planned = []
    for i in range(steps):
        if planned:
            a = planned.pop(0)                         # replay a found RS path for a while (RsPlanner-style)
        else:
            a = rng.uniform(-1, 1, 2)
            if i % 7 == 0:
                a[1] = np.sign(a[1])
        o = orc.step(a[None])
        rec['action'].append(a)
        rec['pose'].append(orc.pose[0].copy())
        rec['status'].append(int(o['status'][0]))
        rec['reward'].append(float(o['reward'][0]))
        rec['mask'].append(np.round(o['mask'][0] * 100).astype(np.uint8))
        rec['lidar'].append(o['lidar'][0].astype(np.float32))
        rec['target'].append(o['target'][0].copy())
        rec['rs_found'].append(int(o['rs_found'][0]))
        rec['rs_ctypes'].append(o['rs_ctypes'][0].astype(np.int8))
        rec['rs_lengths'].append(o['rs_lengths'][0].copy())
        if o['rs_found'][0] and not planned and rng.random() < 0.04:
            for c, l in zip(o['rs_ctypes'][0], o['rs_lengths'][0]):    # parking_agent.py:12-41 unit actions
                if c < 0:
                    break
                steer = {0: 0.0, 1: 1.0, 2: -1.0}[int(c)]
                n = l / 1.25
                while abs(n) > 1:
                    planned.append(np.array([steer, np.sign(n)]))
                    n -= np.sign(n)
                if abs(n) > 1e-3:
                    planned.append(np.array([steer, n]))
        done = o['status'][0] != 1
        rec['reset'].append(bool(done))
        if done:
            planned = []
            orc.pose[0] = orc.start[0]; orc.t[0] = 0; orc.accum[0] = 0
            orc.reset_obs()
    out = {k: np.array(v) for k, v in rec.items()}
    st = out['status']
    print('status counts', {s: int((st == s).sum()) for s in range(1, 6)}, 'rs_found', int(out['rs_found'].sum()))
    np.savez_compressed(os.path.join(HERE, 'trace_config1.npz'), start=sc.start, dest=sc.dest, bbox=sc.bbox,
                        verts=sc.verts, nvert=sc.nvert, first_lidar=first['lidar'], first_mask=first['mask'],
                        first_target=first['target'], first_status=np.array(first['status']), **out)
    print('wrote trace_config1.npz', os.path.getsize(os.path.join(HERE, 'trace_config1.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
