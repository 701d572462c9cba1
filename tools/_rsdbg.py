import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
from test_gpu_parity import make_pair
env, orc, rng = make_pair(512, seed=21)
env.reset_obs(); orc.reset_obs()
shown = 0
for it in range(6):
    act = rng.uniform(-1.2, 1.2, (env.n, 2))
    env.step(torch.from_numpy(act).to(env.device)); o = orc.step(act)
    torch.cuda.synchronize()
    w = env.rs_word.cpu().numpy(); gl = env.rs_lengths.cpu().numpy()
    bad = np.nonzero((w[:, :5] != o['rs_ctypes']).any(1) | (np.abs(gl - o['rs_lengths']) > 0).any(1))[0]
    print('step', it, 'bad', len(bad), 'found', int(o['rs_found'].sum()))
    for i in bad[:4]:
        print('  scene', i, 'gpu', w[i], gl[i], 'orc', o['rs_ctypes'][i], o['rs_lengths'][i], o['rs_found'][i])
