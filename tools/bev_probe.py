import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from hope_amd import ParkingBatch, _lib as L
from hope_amd.scenes import SceneSource, pack_scenes
N = 32768
for mix in (('dlp',), ('Normal', 'Complex', 'Extrem')):
    src = SceneSource(levels=mix, seed=3)
    uniq = [src.draw() for _ in range(512)]
    start, dest, bbox, verts, nob, nvert = pack_scenes(uniq, 128)
    reps = N // len(uniq)
    tile = lambda a: np.concatenate([a] * reps, axis=0)
    env = ParkingBatch(N, 128, profile=True, image=True)
    env.set_scene_arrays(np.arange(N), tile(start), tile(dest), tile(bbox), tile(verts), tile(nob))
    g = torch.Generator(device=env.device); g.manual_seed(0)
    acts = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(4)]
    env.reset_obs()
    for i in range(25):
        env.step(acts[i % 4], auto_reset=True)
    torch.cuda.synchronize()
    for name, dbg in (('full', 0), ('no raster', 1), ('no obstacles', 2), ('no boxes', 4), ('no gather', 8), ('setup only', 9)):
        env.kernel_ms(reset=True)
        for i in range(3):
            env.reset_obs(stages=L.STAGE_IMG | (dbg << 12))
        torch.cuda.synchronize()
        ms, cnt = env.kernel_ms(reset=True)['k_bev_image']
        print(f'{"/".join(mix):22s} {name:14s} {ms / cnt * 1e3:8.1f} us per {N} scenes', flush=True)
    env.close()
