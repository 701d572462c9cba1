#!/usr/bin/env python3
"""Runs the stress configuration and dumps every scene-step where the GPU and the oracle disagree on the RS result."""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root (this dev tool lives under tests/: it uses the oracle)
import torch
from hope_amd import ParkingBatch
from hope_amd.scenes import SceneSource, pack_scenes
from oracle import oracle as O

n, mo = 8192, 128
src = SceneSource(seed=77)
uniq = [src.draw() for _ in range(1024)]
rng = np.random.default_rng(78)
scenes = []
for k in range(n):
    s = copy.copy(uniq[k % len(uniq)])
    if k % 3 == 0:
        r, a = rng.uniform(0.0, 8.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.6])
    scenes.append(s)
env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
env.set_scenes(np.arange(n), scenes)
orc = O.BatchOracle(n, mo, omp=True)
start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, mo)
orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
t = env.tables
O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
env.reset_obs(); orc.reset_obs()
out = []
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(steps):
    act = rng.uniform(-1.2, 1.2, (n, 2))
    if it % 3 == 0:
        act[:, 1] = np.sign(act[:, 1]) * 1.0
    env.step(torch.from_numpy(act).to(env.device))
    o = orc.step(act)
    torch.cuda.synchronize()
    w = env.rs_word.cpu().numpy()
    bad = np.nonzero((w[:, 6] != o['rs_found']) | (w[:, :5] != o['rs_ctypes']).any(axis=1))[0]
    pose_g, _, _ = env.download_state()
    for i in bad:
        out.append(dict(it=it, scene=int(i), pose_gpu=pose_g[i].copy(), pose_orc=orc.pose[i].copy(), dest=dest[i], bbox=bbox[i],
                        verts=verts[i, :nob[i]].copy(), nvert=nvert[i, :nob[i]].copy(), gpu_word=w[i].copy(),
                        orc_found=int(o['rs_found'][i]), orc_ct=o['rs_ctypes'][i].copy(), level=scenes[i].level))
        print('mismatch it', it, 'scene', i, scenes[i].level, 'gpu', w[i], 'orc', o['rs_found'][i], o['rs_ctypes'][i], flush=True)
print('total mismatches', len(out))
os.makedirs('gpurun_out', exist_ok=True)
np.save('gpurun_out/rs_mismatch.npy', np.array(out, dtype=object), allow_pickle=True)
