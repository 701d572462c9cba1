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
        self.pose = torch.from_numpy(self.orc.pose.copy())

    def _publish(self, o, obs_only=None):
        sel = slice(None) if obs_only is None else obs_only
        self.lidar[sel] = torch.from_numpy(o['lidar'][sel]).float()
        self.target[sel] = torch.from_numpy(o['target'][sel]).float()
        self.action_mask[sel] = torch.from_numpy(o['mask'][sel]).float()

    def reset_obs(self):
        o = self.orc.reset_obs(with_rs=self.with_rs)
        self._publish(o)
        self.pose.copy_(torch.from_numpy(self.orc.pose))
        return self

    def step(self, actions, auto_reset=False, active=None):
        keep = None
        if active is not None:                      # frozen scenes: state and outputs untouched (hope_env_step's `active`)
            keep = ~active.bool().numpy()
            saved = (self.orc.pose[keep].copy(), self.orc.t[keep].copy(), self.orc.accum[keep].copy())
        o = self.orc.step(actions.double().numpy(), with_rs=self.with_rs)
        if keep is not None:
            self.orc.pose[keep], self.orc.t[keep], self.orc.accum[keep] = saved
        sel = slice(None) if keep is None else np.nonzero(~keep)[0]
        self._publish(o, sel)
        self.pose[sel] = torch.from_numpy(self.orc.pose[sel])
        self.reward[sel] = torch.from_numpy(o['reward'][sel]).float()
        self.status[sel] = torch.from_numpy(o['status'][sel])
        self.done[sel] = torch.from_numpy((o['status'][sel] != 1).astype(np.uint8))
        self.rs_word[sel, :5] = torch.from_numpy(o['rs_ctypes'][sel].astype(np.int8))
        self.rs_word[sel, 5] = torch.from_numpy((o['rs_ctypes'][sel] >= 0).sum(1).astype(np.int8))
        self.rs_word[sel, 6] = torch.from_numpy(o['rs_found'][sel].astype(np.int8))
        self.rs_lengths[sel] = torch.from_numpy(o['rs_lengths'][sel]).float()
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
