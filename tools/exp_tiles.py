import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from hope_amd import ParkingBatch, _lib as L
from hope_amd.scene_gen import mixed_arrays
N = 16384
arr = mixed_arrays(N, seed=5, max_obst=128)
env = ParkingBatch(N, 128, image=True)
env.set_scene_arrays(np.arange(N), *arr[:5])
g = torch.Generator(device=env.device); g.manual_seed(0)
env.reset_obs()
for i in range(60):
    env.step(torch.rand((N, 2), generator=g, device=env.device) * 2 - 1, auto_reset=True)
torch.cuda.synchronize()
img = env.img.view(N, 3, 4, 16, 4, 16).permute(0, 2, 4, 1, 3, 5).reshape(N, 16, -1)
empty = (img == 0).all(-1)
print('fraction of all-black 16x16 tiles:', float(empty.float().mean()), 'per tile index:', [round(float(x), 2) for x in empty.float().mean(0)])
lv = [('Normal', 'Complex', 'Extrem', 'dlp')[k % 4] for k in range(N)]
for k, name in enumerate(('Normal', 'Complex', 'Extrem', 'dlp')):
    print(name, float(empty[k::4].float().mean()))
