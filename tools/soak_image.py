"""Image soak: the HIP image against oracle/hope_oracle_img.c over many scenes and steps (every uint8, every step), episodes
restarting as they end -- tests/test_gpu_image.py::test_image_rollout_matches_oracle at a larger scale and over several seeds.
  gpurun -- 'python tools/soak_image.py --seeds 4 --scenes 1024 --steps 40 > gpurun_out/soak_image.txt'"""
import argparse, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--seeds', type=int, default=4)
ap.add_argument('--scenes', type=int, default=1024)
ap.add_argument('--steps', type=int, default=40)
args = ap.parse_args()
from test_gpu_image import make_img_pair, restart_done
from hope_amd.scenes import DlpScenePool, SceneSource

tot_img = tot_bad = tot_restart = 0
t0 = time.time()
for seed in range(args.seeds):
    rng = np.random.default_rng(100 + seed)
    n = args.scenes
    src = SceneSource(seed=200 + seed)
    pool = DlpScenePool()
    scenes = [pool.sample(rng=rng) if k % 4 == 3 else src.draw() for k in range(n)]
    for k in range(0, n, 3):                      # a third start close to the destination
        s = scenes[k]
        r, a = rng.uniform(0.0, 6.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    env, orc = make_img_pair(scenes)
    env.reset_obs(); orc.reset_obs()
    bad = 0
    for it in range(args.steps):
        act = rng.uniform(-1.1, 1.1, (n, 2))
        if it % 7 < 3:
            act[:, 1] = np.sign(act[:, 1] + 1e-9)
        env.step(torch.from_numpy(act).to(env.device))
        o = orc.step(act)
        torch.cuda.synchronize()
        assert np.array_equal(env.status.cpu().numpy(), o['status']) and np.array_equal(env.pose.cpu().numpy(), orc.pose)
        got, want = env.img.cpu().numpy(), orc.image(None)
        bad += int((got != want).reshape(n, -1).any(1).sum())
        tot_img += n
        tot_restart += restart_done(env, orc)
        torch.cuda.synchronize()
        got, want = env.img.cpu().numpy(), orc.image(None)
        bad += int((got != want).reshape(n, -1).any(1).sum())
        tot_img += n
    lens = np.array([len(t) for t in orc.traj])
    print(f'seed {seed}: {n} scenes x {args.steps} steps, images differing {bad}, longest trajectory {lens.max()}', flush=True)
    tot_bad += bad
    env.close()
print(f'image soak: {tot_img} images compared (64 x 64 x 3 uint8 each), {tot_bad} differ, {tot_restart} episode restarts, {time.time() - t0:.0f} s')
sys.exit(1 if tot_bad else 0)
