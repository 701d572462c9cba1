"""Batched counterparts of the agent-side glue that sits right after the env step.
Plain torch tensor code (any device); no custom kernels -- these are a few elementwise ops per step.

  BatchedRsPlanner     <-> RsPlanner                 src/model/agent/parking_agent.py:2-47
  mask_action_probs /
  choose_action        <-> ActionMask.choose_action  src/model/action_mask.py:199-227
  BatchedStateNorm     <-> StateNorm                 src/model/state_norm.py:7-46
  batched_gae          <-> PPO.update's GAE loop     src/model/agent/ppo_agent.py:258-273
  RolloutStorage       <-> ReplayMemory              src/model/replay_memory.py:6-49
"""
import math

import torch

from . import tables as T

STEP_RATIO = 0.05 * 10 * 2.5            # kinetic_model.step_len * n_step * VALID_SPEED[1] (train_HOPE_sac.py:164)


class BatchedRsPlanner:
    """Open-loop replay of a found Reeds-Shepp path as unit actions [steer in {1,0,-1}, signed fraction of a full
    1.25 m step], one queue per scene.  Expansion rule = RsPlanner.set_rs_path (:12-41), including its quirks: a
    segment of exactly +-1 step or of |step| <= 1e-3 is dropped; a longer one becomes ceil(|x|)-1 unit actions plus
    the remainder."""

    def __init__(self, n, device='cpu', step_ratio=STEP_RATIO, max_actions=96):
        self.n, self.step_ratio, self.tmax = n, step_ratio, max_actions
        self.device = torch.device(device)
        self.actions = torch.zeros((n, max_actions, 2), dtype=torch.float64, device=self.device)
        self.length = torch.zeros(n, dtype=torch.int64, device=self.device)
        self.cursor = torch.zeros(n, dtype=torch.int64, device=self.device)

    @property
    def executing(self):
        """bool [N]: scenes currently replaying a path (ParkingAgent.executing_rs)."""
        return self.cursor < self.length

    def reset(self, mask=None):
        if mask is None:
            self.length.zero_(); self.cursor.zero_()
        else:
            self.length[mask] = 0; self.cursor[mask] = 0

    @staticmethod
    def expand(rs_word, rs_lengths, step_ratio=STEP_RATIO, max_actions=96):
        """rs_word int8 [N,8] (types S=0 L=1 R=2, -1 unused; [5]=n_seg), rs_lengths [N,5] metres ->
        (actions [N,max_actions,2] float64, count [N])."""
        n = rs_word.shape[0]
        dev = rs_word.device
        types = rs_word[:, :5].to(torch.int64)
        used = types >= 0
        steer = torch.where(types == 1, 1.0, torch.where(types == 2, -1.0, 0.0)).to(torch.float64)   # L:1 S:0 R:-1
        # a python-float divisor makes the GPU kernel multiply by the reciprocal (not correctly rounded): divide by a
        # device tensor so that x equals the reference's `length / step_ratio` bit for bit on every device
        x = rs_lengths.to(torch.float64) / torch.full((1,), float(step_ratio), dtype=torch.float64, device=dev)
        ax = x.abs()
        k = torch.where(ax > 1, torch.ceil(ax) - 1, torch.zeros_like(ax))             # unit actions
        rem = torch.where(ax > 1, torch.sign(x) * (ax - k), x)
        keep_rem = (ax != 1) & (rem.abs() > 1e-3) & used
        k = torch.where(used, k, torch.zeros_like(k)).to(torch.int64)
        cnt = k + keep_rem.to(torch.int64)                                            # [N,5]
        off = torch.cumsum(cnt, dim=1) - cnt
        total = cnt.sum(dim=1)
        t = torch.arange(max_actions, device=dev).view(1, -1, 1)                      # [1,T,1]
        rel = t - off.unsqueeze(1)                                                    # [N,T,5]
        inseg = (rel >= 0) & (rel < cnt.unsqueeze(1))
        unit = inseg & (rel < k.unsqueeze(1))
        val = torch.where(unit, torch.sign(x).unsqueeze(1).expand(-1, max_actions, -1), rem.unsqueeze(1).expand(-1, max_actions, -1))
        a_step = (val * inseg).sum(dim=2)
        a_steer = (steer.unsqueeze(1) * inseg).sum(dim=2)
        actions = torch.stack([a_steer, a_step], dim=2)
        return actions, torch.clamp(total, max=max_actions)

    def set_paths(self, rs_word, rs_lengths, forced=False):
        """ParkingAgent.set_planner_path (:65-69): adopt a newly found path unless one is being replayed."""
        found = rs_word[:, 6] > 0
        take = found if forced else (found & ~self.executing)
        if bool(take.any()):
            acts, cnt = self.expand(rs_word[take], rs_lengths[take], self.step_ratio, self.tmax)
            self.actions[take] = acts
            self.length[take] = cnt
            self.cursor[take] = 0
        return take

    def get_actions(self):
        """pop the next planned action of every executing scene -> ([N,2] actions, bool [N] which rows are valid)."""
        ex = self.executing
        idx = torch.clamp(self.cursor, max=self.tmax - 1)
        a = self.actions[torch.arange(self.n, device=self.device), idx]
        self.cursor = torch.where(ex, self.cursor + 1, self.cursor)
        done = ex & (self.cursor >= self.length)
        self.length[done] = 0
        self.cursor[done] = 0
        return a, ex


_ACTIONS = None


def _scaled_actions(device, dtype):
    global _ACTIONS
    if _ACTIONS is None:
        _ACTIONS = torch.from_numpy(T.discrete_actions() / [T.VALID_STEER[1], 1.0])      # action_mask.py:217-221
    return _ACTIONS.to(device=device, dtype=dtype)


def mask_action_probs(action_mean, action_std, action_mask):
    """ActionMask.choose_action (:212-224) for a batch: probability of each of the 42 discrete actions under the
    Gaussian policy head, re-weighted by the action mask.  mean/std [N,2], mask [N,42] -> [N,42]."""
    mean, std = action_mean.to(torch.float64), action_std.to(torch.float64)
    acts = _scaled_actions(mean.device, torch.float64)                                     # [42,2]
    z = (acts.unsqueeze(0) - mean.unsqueeze(1)) / std.unsqueeze(1)
    logp = -0.5 * z ** 2 - torch.log(math.sqrt(2 * math.pi) * std).unsqueeze(1)
    prob = torch.clamp(logp, -10, 10).sum(dim=2)
    e = torch.exp(prob) * action_mask.to(torch.float64)
    return e / e.sum(dim=1, keepdim=True)


def choose_action(action_mean, action_std, action_mask, generator=None):
    """sample one discrete action per scene -> [N,2] in the policy's [-1,1] scaling (action_mask.py:225-227)."""
    p = mask_action_probs(action_mean, action_std, action_mask)
    idx = torch.multinomial(p, 1, generator=generator).squeeze(1)
    return _scaled_actions(p.device, torch.float64)[idx], idx


class BatchedStateNorm:
    """StateNorm (state_norm.py:7-46) for batches: running mean / std of the 'lidar' and 'target' observations
    (DEFAULT_UPDATE_MODAL).  The reference folds observations in one at a time (Welford); `update` folds a whole
    batch with the parallel-merge form of the same recurrence, and keeps the reference's first-sample quirk
    (mean = std = first observation)."""

    def __init__(self, shapes=None, update_modal=('lidar', 'target'), device='cpu'):
        shapes = shapes or {'lidar': 120, 'target': 5, 'action_mask': 42}
        self.modal = tuple(update_modal)
        self.n_state = 0
        self.fixed = False
        dev = torch.device(device)
        self.mean = {k: torch.zeros(shapes[k], dtype=torch.float64, device=dev) for k in self.modal}
        self.S = {k: torch.zeros(shapes[k], dtype=torch.float64, device=dev) for k in self.modal}
        self.std = {k: torch.zeros(shapes[k], dtype=torch.float64, device=dev) for k in self.modal}

    def fix_parameters(self):
        self.fixed = True

    def normalize(self, obs):
        return {k: ((v.to(torch.float64) - self.mean[k]) / (self.std[k] + 1e-8) if k in self.modal else v) for k, v in obs.items()}

    def update(self, obs):
        """fold obs[k] of shape [B, dim] into the running statistics (no-op when fixed)."""
        if self.fixed:
            return
        b = next(iter(obs.values())).shape[0]
        start = 0
        if self.n_state == 0:                                   # first sample: mean = std = observation (:27-31)
            for k in self.modal:
                self.mean[k] = obs[k][0].to(torch.float64).clone()
                self.std[k] = obs[k][0].to(torch.float64).clone()
            self.n_state = 1
            start = 1
        m = b - start
        if m <= 0:
            return
        n0 = self.n_state
        for k in self.modal:
            x = obs[k][start:].to(torch.float64)
            bm = x.mean(dim=0)
            bS = ((x - bm) ** 2).sum(dim=0)
            delta = bm - self.mean[k]
            self.mean[k] = self.mean[k] + delta * (m / (n0 + m))
            self.S[k] = self.S[k] + bS + delta ** 2 * (n0 * m / (n0 + m))
            self.std[k] = torch.sqrt(self.S[k] / (n0 + m))
        self.n_state = n0 + m


# ---- PPO / SAC storage over [N, T] -----------------------------------------------------------
def batched_gae(reward, value, next_value, done, gamma=0.98, lam=0.95, use_gae=True):
    """Advantages of PPO.update (src/model/agent/ppo_agent.py:258-273) for N parallel scenes at once.

    The reference keeps ONE env's transitions in time order and runs
        delta = r + gamma * (1 - done) * V(s') - V(s);  gae = delta + gamma * lambda * gae * (1 - done)   (backwards)
    so with N scenes the recurrence runs along T independently per row.  reward / value / next_value / done: [N, T]
    (done as 0/1).  The accumulation is done in float64 like the reference's Python-float loop and returned in
    value's dtype.  Returns (adv [N, T], delta [N, T])."""
    r, v, nv, d = (x.to(torch.float64) for x in (reward, value, next_value, done))
    delta = (r + gamma * (1 - d) * nv - v).to(value.dtype).to(torch.float64)       # deltas are float32 tensors there
    if not use_gae:
        return delta.to(value.dtype), delta.to(value.dtype)
    adv = torch.zeros_like(delta)
    gae = torch.zeros(delta.shape[0], dtype=torch.float64, device=delta.device)
    for t in range(delta.shape[1] - 1, -1, -1):
        gae = delta[:, t] + gamma * lam * gae * (1.0 - d[:, t])
        adv[:, t] = gae
    return adv.to(value.dtype), delta.to(value.dtype)


class RolloutStorage:
    """Device-resident [N, T] ring of transitions: the batched stand-in for ReplayMemory (src/model/replay_memory.py)
    as both agents use it -- PPO reads everything in time order (get_items(arange), ppo_agent.py:241), SAC samples
    uniformly (sample(batch_size), sac_agent.py) with next_state = None at episode ends / at the newest entry
    (:27-30), which is reported here as `next_valid`."""

    def __init__(self, n, horizon, obs_shapes, device='cpu', action_dim=2, extra=('log_prob',)):
        self.n, self.T, self.device = n, horizon, torch.device(device)
        self.obs = {k: torch.zeros((n, horizon) + tuple(s), dtype=torch.uint8 if k == 'img' else torch.float32, device=self.device)
                    for k, s in obs_shapes.items()}
        self.action = torch.zeros((n, horizon, action_dim), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros((n, horizon), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((n, horizon), dtype=torch.float32, device=self.device)
        self.extra = {k: torch.zeros((n, horizon, action_dim), dtype=torch.float32, device=self.device) for k in extra}
        self.head, self.size = 0, 0

    def push(self, obs, action, reward, done, **extra):
        """one step of all N scenes (ReplayMemory.push :12-15)"""
        t = self.head
        for k in self.obs:
            self.obs[k][:, t] = obs[k].to(self.obs[k].dtype)
        self.action[:, t] = action
        self.reward[:, t] = reward
        self.done[:, t] = done.to(torch.float32)
        for k, v in extra.items():
            self.extra[k][:, t] = v
        self.head = (t + 1) % self.T
        self.size = min(self.size + 1, self.T)

    def _time_index(self):
        """column indices oldest -> newest"""
        return (torch.arange(self.size, device=self.device) + (self.head - self.size)) % self.T

    def ordered(self):
        """everything in time order, [N, size, ...] (PPO); next-observation of column j is column j + 1"""
        idx = self._time_index()
        out = {'obs': {k: v[:, idx] for k, v in self.obs.items()}, 'action': self.action[:, idx],
               'reward': self.reward[:, idx], 'done': self.done[:, idx]}
        out.update({k: v[:, idx] for k, v in self.extra.items()})
        return out

    def sample(self, batch_size, generator=None):
        """uniform transitions over (scene, time) like ReplayMemory.sample (:33-35).  next_valid is False where the
        reference returns next_state None: episode end or newest entry (:27-28)."""
        idx = self._time_index()
        s = torch.randint(self.n, (batch_size,), device=self.device, generator=generator)
        j = torch.randint(self.size, (batch_size,), device=self.device, generator=generator)
        t = idx[j]
        nxt = idx[torch.clamp(j + 1, max=self.size - 1)]
        next_valid = (j < self.size - 1) & (self.done[s, t] == 0)
        return {'obs': {k: v[s, t] for k, v in self.obs.items()}, 'next_obs': {k: v[s, nxt] for k, v in self.obs.items()},
                'next_valid': next_valid, 'action': self.action[s, t], 'reward': self.reward[s, t], 'done': self.done[s, t]}

    def clear(self):
        self.head, self.size = 0, 0
