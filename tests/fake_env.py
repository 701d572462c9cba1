"""A CPU stand-in with ParkingBatch's attribute surface, driven by the C oracle (test infrastructure only): lets the
host-side loops in hope_amd/rollout.py (PPOTrainer / SACTrainer) run in `-m "not gpu"` tests and in 2-rank gloo tests."""
import numpy as np
import torch

from oracle import oracle as O


class OracleEnv:
    def __init__(self, scenes, max_obst=32, with_rs=True):
        from hope_amd import tables as T
        from hope_amd.scenes import pack_scenes
        self.n = len(scenes)
        self.device = torch.device('cpu')
        self.action_dtype = torch.float32
        self.with_rs = with_rs
        self.mo = max_obst
        self.packed = pack_scenes(scenes, max_obst)
        start, dest, bbox, verts, nob, nvert = self.packed
        t = T.all_tables()
        O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
        self.orc = O.BatchOracle(self.n, max_obst)
        self.orc.set_scenes(np.arange(self.n), start, dest, bbox, verts, nvert, nob)
        n = self.n
        self.lidar, self.target, self.action_mask = torch.zeros(n, 120), torch.zeros(n, 5), torch.zeros(n, 42)
        self.reward, self.done = torch.zeros(n), torch.zeros(n, dtype=torch.uint8)
        self.status = torch.zeros(n, dtype=torch.int32)
        self.rs_word = torch.full((n, 8), -1, dtype=torch.int8)
        self.rs_lengths = torch.zeros(n, 5)
        self.img = None

    def _publish(self, o, obs_only=None):
        sel = slice(None) if obs_only is None else obs_only
        self.lidar[sel] = torch.from_numpy(o['lidar'][sel]).float()
        self.target[sel] = torch.from_numpy(o['target'][sel]).float()
        self.action_mask[sel] = torch.from_numpy(o['mask'][sel]).float()

    def reset_obs(self):
        o = self.orc.reset_obs(with_rs=self.with_rs)
        self._publish(o)
        return self

    def step(self, actions, auto_reset=False):
        o = self.orc.step(actions.double().numpy(), with_rs=self.with_rs)
        self._publish(o)
        self.reward.copy_(torch.from_numpy(o['reward']).float())
        self.status.copy_(torch.from_numpy(o['status']))
        self.done.copy_(torch.from_numpy((o['status'] != 1).astype(np.uint8)))
        self.rs_word[:, :5] = torch.from_numpy(o['rs_ctypes'].astype(np.int8))
        self.rs_word[:, 5] = torch.from_numpy((o['rs_ctypes'] >= 0).sum(1).astype(np.int8))
        self.rs_word[:, 6] = torch.from_numpy(o['rs_found'].astype(np.int8))
        self.rs_lengths.copy_(torch.from_numpy(o['rs_lengths']).float())
        ids = np.nonzero(o['status'] != 1)[0]
        if auto_reset and len(ids):                 # HOPE_AUTO_RESET: restart on the same map + the action-less step
            start, dest, bbox, verts, nob, nvert = self.packed
            sub = O.BatchOracle(len(ids), self.mo)
            sub.set_scenes(np.arange(len(ids)), start[ids], dest[ids], bbox[ids], verts[ids], nvert[ids], nob[ids])
            so = sub.reset_obs(with_rs=False)
            self.orc.restart(ids)
            self.orc.t[ids] = sub.t
            self.orc.accum[ids] = sub.accum
            self.lidar[ids] = torch.from_numpy(so['lidar']).float()
            self.target[ids] = torch.from_numpy(so['target']).float()
            self.action_mask[ids] = torch.from_numpy(so['mask']).float()
        return self
