"""k_bev_image stage probe: time per launch with parts of the kernel switched off (HOPE_BEV_DEBUG bits: 1 no raster, 2 no static ids,
4 no boxes, 8 no gather, 16 no block cache, 64 no image stores); one process per setting (the env is read at every step)."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from hope_amd import ParkingBatch, _lib as L
from hope_amd.scene_gen import mixed_arrays
N = 65536
arr = mixed_arrays(N, seed=5, max_obst=128)
env = ParkingBatch(N, 128, profile=True, image=True)
env.set_scene_arrays(np.arange(N), *arr[:5])
g = torch.Generator(device=env.device); g.manual_seed(0)
acts = [torch.rand((N, 2), generator=g, device=env.device) * 2 - 1 for _ in range(4)]
env.reset_obs()
for i in range(40):
    env.step(acts[i % 4], auto_reset=True)
torch.cuda.synchronize()
for name, dbg in (('full', 0), ('no stores', 64), ('no gather', 8), ('no gather, no cache', 24), ('no cache', 16), ('no boxes', 4), ('no boxes no stores', 68), ('first round trip only', 128), ('launch only', 256)):
    os.environ['HOPE_BEV_DEBUG'] = str(dbg)
    env.kernel_ms(reset=True)
    for i in range(4):
        env.reset_obs(stages=L.STAGE_IMG)
    torch.cuda.synchronize()
    ms, cnt = env.kernel_ms(reset=True)['k_bev_image']
    print(f'{name:22s} {ms / cnt * 1e3:8.1f} us per {N} scenes', flush=True)
for name, dbg in (('prep full', 0), ('prep no paint', 512)):
    os.environ['HOPE_BEV_DEBUG'] = str(dbg)
    for i in range(3):
        env.step(acts[i % 4], auto_reset=True)
    torch.cuda.synchronize()
    env.kernel_ms(reset=True)
    for i in range(8):
        env.step(acts[i % 4], auto_reset=True)
    torch.cuda.synchronize()
    k = env.kernel_ms(reset=True)
    print(name, {n: round(v[0] / v[1] * 1e3, 1) for n, v in k.items() if 'bev' in n}, flush=True)
env.close()
