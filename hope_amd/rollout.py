"""Batched rollout / training loops over N scenes: the N >> 1 counterparts of the episode loops in
src/train/train_HOPE_ppo.py:177-213, src/train/train_HOPE_sac.py:177-221 and src/evaluation/eval_utils.py:16-84, with
the hybrid controller of src/model/agent/parking_agent.py (replay a found Reeds-Shepp path, otherwise ask the policy).

  HopeRollout.collect_step   one env step of every scene: state-norm -> actor -> mask-weighted / Gaussian action (or the
                             planner's) -> ParkingBatch.step(auto_reset) -> storage -> planner bookkeeping
  PPOTrainer                 BASELINE config 5 (train_HOPE_ppo.py): T steps of all scenes, then BatchedPPO.update
  SACTrainer                 BASELINE config 4 / train_HOPE_sac.py: [N, T] ring replay, one BatchedSAC.update every
                             `update_every` steps once the ring holds `warmup` columns
  BatchedRollout             round-1 loop with a stand-in MLP policy (kept for the examples)

`env` is a `hope_amd.ParkingBatch` (HIP path) or anything with the same attributes (tests use a CPU fake).
ParkingBatch's observation tensors are persistent buffers that the next step overwrites in place, so every loop
copies what it keeps BEFORE stepping.
"""
import torch

from . import agent_glue as G


class StandInPolicy(torch.nn.Module):
    def __init__(self, hidden=256):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(120 + 5 + 42, hidden), torch.nn.Tanh(),
                                       torch.nn.Linear(hidden, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, 2), torch.nn.Tanh())
        self.log_std = torch.nn.Parameter(torch.zeros(2))

    def forward(self, lidar, target, mask):
        mean = self.net(torch.cat([lidar, target, mask], dim=1))
        return mean, self.log_std.exp().expand_as(mean)


class BatchedRollout:
    def __init__(self, env, policy=None, seed=0, update_norm=True):
        self.env = env
        dev = env.device
        self.policy = (policy or StandInPolicy()).to(dev)
        self.norm = G.BatchedStateNorm(device=dev)
        self.planner = G.BatchedRsPlanner(env.n, device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.update_norm = update_norm
        self.episodes = torch.zeros(env.n, dtype=torch.int64, device=dev)
        self.successes = torch.zeros(env.n, dtype=torch.int64, device=dev)
        self.steps = 0
        env.reset_obs()

    @torch.no_grad()
    def step(self):
        env = self.env
        obs = {'lidar': env.lidar.to(torch.float32), 'target': env.target.to(torch.float32)}
        if self.update_norm:
            self.norm.update(obs)
        nz = self.norm.normalize(obs)
        mean, std = self.policy(nz['lidar'].float(), nz['target'].float(), env.action_mask.float())
        rl_action, _ = G.choose_action(mean, std, env.action_mask, self.gen)          # mask-weighted discrete sampling
        plan_action, executing = self.planner.get_actions()                          # replay of a found RS path
        action = torch.where(executing.unsqueeze(1), plan_action, rl_action).to(env.action_dtype).contiguous()
        env.step(action, auto_reset=True)
        done = env.done.bool()
        self.episodes += done
        self.successes += (env.status == 2)
        self.planner.reset(done)                                                     # ParkingAgent.reset at episode end
        self.planner.set_paths(env.rs_word, env.rs_lengths)                          # info['path_to_dest'] (:212-213)
        self.steps += 1
        return action

    def stats(self):
        ep = int(self.episodes.sum().item())
        return {'steps': self.steps, 'episodes': ep, 'success_rate': float(self.successes.sum().item()) / max(ep, 1),
                'executing_rs': float(self.planner.executing.float().mean().item())}


class TransitionRing:
    """[N, T] device ring of (normalised obs, action, log_prob, reward, done): the batched ReplayMemory
    (src/model/replay_memory.py:6-49).  A column is written in two halves -- what the agent saw and did before the env
    step, what the env answered after it."""

    def __init__(self, n, horizon, keys, device, img_shape=(3, 64, 64)):
        self.n, self.T, self.device = n, horizon, torch.device(device)
        shp = {'lidar': (120,), 'target': (5,), 'action_mask': (42,), 'img': tuple(img_shape)}
        self.obs = {k: torch.zeros((n, horizon) + shp[k], dtype=torch.uint8 if k == 'img' else torch.float32, device=self.device)
                    for k in keys}
        self.action = torch.zeros((n, horizon, 2), dtype=torch.float32, device=self.device)
        self.log_prob = torch.zeros((n, horizon, 2), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros((n, horizon), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((n, horizon), dtype=torch.float32, device=self.device)
        self.head, self.size = 0, 0

    def write_before(self, nobs, action, log_prob):
        t = self.head
        for k, buf in self.obs.items():
            buf[:, t].copy_(nobs[k])
        self.action[:, t].copy_(action)
        self.log_prob[:, t].copy_(log_prob)

    def write_after(self, reward, done):
        t = self.head
        self.reward[:, t].copy_(reward)
        self.done[:, t].copy_(done)
        self.head = (t + 1) % self.T
        self.size = min(self.size + 1, self.T)

    def columns(self):
        """column indices oldest -> newest"""
        return (torch.arange(self.size, device=self.device) + (self.head - self.size)) % self.T

    def ordered(self):
        idx = self.columns()
        if self.size == self.T and self.head == 0:                      # the PPO case: no gather needed
            return self.obs, self.action, self.reward, self.done, self.log_prob
        return ({k: v[:, idx] for k, v in self.obs.items()}, self.action[:, idx], self.reward[:, idx], self.done[:, idx],
                self.log_prob[:, idx])

    def sample(self, batch_size, last_obs, generator=None):
        """uniform (scene, time) transitions like ReplayMemory.sample (:33-35); the observation after the newest column
        is `last_obs` (the one the agent is about to act on)."""
        idx = self.columns()
        s = torch.randint(self.n, (batch_size,), device=self.device, generator=generator)
        j = torch.randint(self.size, (batch_size,), device=self.device, generator=generator)
        t = idx[j]
        newest = j == self.size - 1
        tn = idx[torch.clamp(j + 1, max=self.size - 1)]
        nxt = {}
        for k, v in self.obs.items():
            a = v[s, tn]
            m = newest.view((-1,) + (1,) * (a.dim() - 1))
            nxt[k] = torch.where(m, last_obs[k][s].to(a.dtype), a)
        return {'obs': {k: v[s, t] for k, v in self.obs.items()}, 'next_obs': nxt, 'action': self.action[s, t],
                'reward': self.reward[s, t], 'done': self.done[s, t]}

    def clear(self):
        self.head, self.size = 0, 0


class HopeRollout:
    def __init__(self, env, agent, horizon, use_mask=True, seed=0, use_planner=True, fresh_scenes=False, pool_refresher=None,
                 defer_rs=True):
        """defer_rs: step the env with two completion points (HOPE_DEFER_RS): the next policy forward is enqueued as soon as the
        observation is written, the planner reads rs_word / rs_lengths (after ParkingBatch.wait_rs) just before its override --
        the same actions as with the joined step, the forward overlaps the Reeds-Shepp kernels.
        fresh_scenes: finished episodes continue on a NEW map drawn from the env's device-resident scene pool
        (ParkingBatch.set_pool / set_dlp_cases), as the reference's loop does with `env.reset(...)`; otherwise on the same map.
        pool_refresher: a `scene_gen.PoolRefresher`; the trainers poll it after every update, so the pool of generated lots is
        replaced by new ones in the background (asynchronous upload, no synchronisation with the step loop)."""
        self.env, self.agent, self.use_mask, self.fresh = env, agent, use_mask, fresh_scenes
        self.refresher = pool_refresher
        self.seed = seed
        self.defer_rs = bool(defer_rs) and hasattr(env, 'wait_rs')
        self._plan_pending = False                    # the planner has not yet seen the last step's done / RS outputs
        self._rs_step = None                          # hope_env_last_step() of the deferred step whose search the planner waits for
        dev = env.device
        self.ring = TransitionRing(env.n, horizon, agent.keys, dev)
        self.planner = G.BatchedRsPlanner(env.n, device=dev) if use_planner else None
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.episodes = torch.zeros((), dtype=torch.int64, device=dev)
        self.successes = torch.zeros((), dtype=torch.int64, device=dev)
        self.reward_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.steps = 0
        if fresh_scenes and hasattr(env, 'set_redraw_seed'):
            env.set_redraw_seed(seed * 1000003 + 17)
        env.reset_obs()
        agent.observe(self._raw_obs())

    def _raw_obs(self):
        e = self.env
        o = {'lidar': e.lidar, 'target': e.target, 'action_mask': e.action_mask}
        if 'img' in self.agent.keys:
            o['img'] = e.img
        return o

    def _plan(self):
        """the planner's part of ParkingAgent.choose_action: bookkeeping of the last step (its done flags and RS paths), then
        the actions of the scenes that are replaying a path"""
        if self.planner is None:
            return None, None
        if self._plan_pending:
            env = self.env
            if self.defer_rs:
                # the search of THE step whose words the planner is about to read (a newer step would have replaced them: HOPE_ESTATE)
                env.wait_rs(step=self._rs_step) if self._rs_step is not None and hasattr(env, 'last_step') else env.wait_rs()
            self.planner.reset(env.done.bool())                     # ParkingAgent.reset at episode end
            self.planner.set_paths(env.rs_word, env.rs_lengths)     # info['path_to_dest'] -> set_planner_path
            self._plan_pending = False
        return self.planner.get_actions()

    @torch.no_grad()
    def collect_step(self, random_action=False):
        env, agent = self.env, self.agent
        executing = None
        if random_action:
            planned, executing = self._plan()
            action, log_prob, nobs = agent.act(self._raw_obs(), self.use_mask, self.gen, planned, executing)
        else:
            action, log_prob, nobs = agent.act(self._raw_obs(), self.use_mask, self.gen, plan_fn=self._plan)
        if random_action:                             # train_HOPE_sac.py:196-198: uniform exploration while the memory fills
            rnd = torch.rand(action.shape, device=action.device, generator=self.gen) * 2 - 1
            action = rnd if executing is None else torch.where(executing.unsqueeze(1), action, rnd)
            mean = agent.policy_mean(nobs)
            from .policy import gaussian_log_prob
            log_prob = gaussian_log_prob(mean, agent.log_std.expand_as(mean), action)
        self.ring.write_before(nobs, action, log_prob)              # copies: env.step overwrites the buffers in place
        kw = {'defer_rs': True} if self.defer_rs else {}
        if self.fresh:                                # new map per episode, drawn inside the step kernel (HOPE_AUTO_REDRAW)
            env.step(action.to(env.action_dtype).contiguous(), auto_reset=True, fresh=True, **kw)
        else:
            env.step(action.to(env.action_dtype).contiguous(), auto_reset=True, **kw)
        self.ring.write_after(env.reward, env.done)
        agent.observe(self._raw_obs())                              # push_memory: state_norm(next_obs, update=True)
        done = env.done.bool()
        self.episodes += done.sum()
        self.successes += (env.status == 2).sum()
        self.reward_sum += env.reward.sum(dtype=torch.float64)
        self._plan_pending = self.planner is not None               # (its bookkeeping runs in _plan, before the next override)
        self._rs_step = env.last_step() if (self.defer_rs and hasattr(env, 'last_step')) else None
        self.steps += 1

    def last_obs(self):
        """normalised observation the agent will act on next (value bootstrap / newest next_obs)"""
        return self.agent._norm_obs(self._raw_obs())

    def stats(self):
        ep = int(self.episodes.item())
        return {'steps': self.steps, 'episodes': ep, 'success_rate': float(self.successes.item()) / max(ep, 1),
                'mean_reward': float(self.reward_sum.item()) / max(self.steps * self.env.n, 1)}


class PPOTrainer(HopeRollout):
    """train_HOPE_ppo.py:177-213: act -> step -> push; when the buffer is full (`horizon` steps of all scenes,
    the batched `len(memory) % batch_size == 0`) run PPO.update and clear."""

    def __init__(self, env, agent, horizon=16, seed=0, use_planner=True, fresh_scenes=False, pool_refresher=None, defer_rs=True):
        super().__init__(env, agent, horizon, use_mask=True, seed=seed, use_planner=use_planner, fresh_scenes=fresh_scenes,
                         pool_refresher=pool_refresher, defer_rs=defer_rs)
        self.updates = 0

    def step(self):
        self.collect_step()
        if self.ring.size == self.ring.T:
            obs, action, reward, done, log_prob = self.ring.ordered()
            losses = self.agent.update(obs, action, reward, done, log_prob, self.last_obs(), generator=self.gen)
            self.ring.clear()
            self.updates += 1
            if self.refresher is not None:
                self.refresher.poll()
            return losses
        return None


class SACTrainer(HopeRollout):
    """train_HOPE_sac.py:177-221: uniform random actions until the memory is full, then the policy (plain Gaussian
    sample, no action mask), one SAC update every `update_every` env steps on a uniform batch from the ring."""

    def __init__(self, env, agent, horizon=8, update_every=10, seed=0, use_planner=True, learn=True, fresh_scenes=False,
                 pool_refresher=None, defer_rs=True):
        super().__init__(env, agent, horizon, use_mask=False, seed=seed, use_planner=use_planner, fresh_scenes=fresh_scenes,
                         pool_refresher=pool_refresher, defer_rs=defer_rs)
        self.update_every, self.learn, self.updates = update_every, learn, 0

    def step(self):
        self.collect_step(random_action=self.learn and self.ring.size < self.ring.T)
        if self.learn and self.ring.size == self.ring.T and self.steps % self.update_every == 0:
            batch = self.ring.sample(self.agent.batch_size, self.last_obs(), self.gen)
            self.updates += 1
            if self.refresher is not None and self.updates % 16 == 0:
                self.refresher.poll()
            return self.agent.update(batch)
        return None
