#!/usr/bin/env python3
"""BASELINE configs[0] as a number: wall time per `CarParkingWrapper.step` of the N = 1 look-alike classes
(hope_amd/env.py: one hope_env_step + one packed device-to-host copy + the obs dict) on the recorded 2000-action trace of
DLP case 0 (tests/golden/trace_config1.npz), with and without the image, next to the CPU oracle on one host core for
the same trace (checker only) and the reference's own structural ceiling of 100 steps/s (car_parking_base.py:409, fps = 100
render clock).
    python tools/config1_latency.py > profiles/rNN_config1_latency.txt      (on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_lookalike(g, scene, use_img):
    from hope_amd.env import CarParking, CarParkingWrapper
    raw = CarParking(render_mode='rgb_array', fps=100, verbose=False, use_img_observation=use_img)
    env = CarParkingWrapper(raw)
    raw.reset_to_scene(scene)
    acts = g['action']
    for i in range(50):                       # warm-up (first launches, pinned buffers)
        _, _, done, _ = env.step(acts[i])
        if done:
            raw.reset_to_scene(scene)
    raw.reset_to_scene(scene)
    lat = np.empty(len(acts))
    resets = 0
    t0 = time.perf_counter()
    for i in range(len(acts)):
        ta = time.perf_counter()
        _, _, done, _ = env.step(acts[i])
        lat[i] = time.perf_counter() - ta
        if done:
            raw.reset_to_scene(scene)
            resets += 1
    wall = time.perf_counter() - t0
    env.close()
    return wall, lat, resets


def run_oracle(g, scene):
    from hope_amd.scenes import pack_scenes
    from hope_amd import tables as T
    from oracle import oracle as O
    mo = 128
    start, dest, bbox, verts, nob, nvert = pack_scenes([scene], mo)
    t = T.all_tables()
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    orc = O.BatchOracle(1, mo)
    orc.set_scenes([0], start, dest, bbox, verts, nvert, nob)
    orc.reset_obs(with_rs=True)
    acts = g['action']
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        o = orc.step(acts[i:i + 1].astype(np.float64), with_rs=True)
        if int(o['status'][0]) != 1:            # Status.CONTINUE = 1 (vehicle.py:13-18)
            orc.set_scenes([0], start, dest, bbox, verts, nvert, nob)
            orc.reset_obs(with_rs=True)
    return (time.perf_counter() - t0) / n


def main():
    import torch
    assert torch.cuda.is_available(), 'needs a HIP device: the product has no CPU path'
    from hope_amd.scenes import Scene
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'trace_config1.npz'))
    scene = Scene(start=g['start'], dest=g['dest'], bbox=g['bbox'], verts=g['verts'], nvert=g['nvert'], level='dlp', case_id=0)
    print('config 1: N = 1, DLP case 0, %d recorded wrapper steps (tests/golden/trace_config1.npz), %s' %
          (len(g['action']), torch.cuda.get_device_name(0)))
    for use_img in (True, False):
        wall, lat, resets = run_lookalike(g, scene, use_img)
        us = np.sort(lat) * 1e6
        print('  look-alike CarParkingWrapper.step, img %-5s: %8.1f steps/s  (%.1f us per step mean incl. %d resets; '
              'median %.1f, p90 %.1f, p99 %.1f us)' % (use_img, len(lat) / wall, wall / len(lat) * 1e6, resets,
                                                        us[len(us) // 2], us[int(0.9 * len(us))], us[int(0.99 * len(us))]))
    try:
        per = run_oracle(g, scene)
        print('  CPU oracle (oracle/hope_oracle.c, 1 thread, no image), same trace: %8.1f steps/s (%.1f us per step)' % (1 / per, per * 1e6))
    except Exception as e:                    # the oracle is the checker, not the product: report and go on
        print('  CPU oracle leg skipped:', repr(e))
    print('  reference (Python, CPU): structural ceiling 100 steps/s (fps = 100 render clock, car_parking_base.py:409); cannot run in this image (no shapely / pygame)')


if __name__ == '__main__':
    main()
