"""Batched PPO and SAC updates for N parallel scenes (BASELINE configs 4 and 5).

The reference trains ONE env with per-sample python lists; these classes keep the same mathematics on [N, T] device
tensors and put the one data-parallel exchange (a fused gradient all-reduce, `hope_amd.dist.allreduce_gradients`)
between backward and the optimiser step.

  BatchedPPO   <-> PPOAgent   src/model/agent/ppo_agent.py   (_init_network :71-118, _actor_forward :120-145,
                               _post_process_action :147-160, push_memory :205-215, update :236-349)
  BatchedSAC   <-> SACAgent   src/model/agent/sac_agent.py   (_init_network :89-139, _get_action_and_log_prob :250-261,
                               update :263-337)
  hyper-parameters            src/configs.py:117-121 (GAMMA .98, LR 5e-6, TAU .1), PPOConfig :16-43, SACConfig :33-59

Deviations forced by batching (everything else follows the reference line by line and is pinned against it on CPU in
tests/test_agents.py with reference-generated post-update weights):
  * mini-batches are drawn from the N x T transitions of all scenes; `mini_batch` defaults to the reference's 32 but is
    meant to be raised with N (one optimiser step per 32 samples would be 10^4 launches per epoch at N = 16 384);
  * GAE runs along T for every scene independently (the reference's single env is the N = 1 case);
  * advantage normalisation uses the statistics of all ranks (sum / sum-of-squares all-reduce) so that a sharded run
    normalises like a single-process run over the same transitions.
"""
import copy
import math

import torch
import torch.nn.functional as F

from . import agent_glue as G
from . import dist as D
from . import policy as P

GAMMA, LR, TAU = 0.98, 5e-6, 0.1                       # configs.py:117-121


def _obs_keys(use_img):
    return ('lidar', 'target', 'action_mask') + (('img',) if use_img else ())


def _soft_update(target, current, tau):
    """AgentBase._soft_update (agent_base.py:67-69): target <- tau * current + (1 - tau) * target."""
    with torch.no_grad():
        tp, cp = list(target.parameters()), list(current.parameters())
        torch._foreach_mul_(tp, 1.0 - tau)
        torch._foreach_add_(tp, cp, alpha=tau)


def _global_mean_std(x):
    """mean and unbiased std of x over all ranks (torch.Tensor.std() of the concatenated tensor)."""
    s = torch.stack([x.sum(dtype=torch.float64), (x.double() ** 2).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
    if D.dist.is_available() and D.dist.is_initialized() and D.dist.get_world_size() > 1:
        D.dist.all_reduce(s)
    n = s[2]
    mean = s[0] / n
    # a single transition has no spread: std = 0 (torch's unbiased std would be NaN and poison the advantages; the
    # normalisation's own epsilon then makes them 0)
    var = (s[1] - n * mean * mean) / torch.clamp(n - 1, min=1)
    return mean.to(x.dtype), torch.sqrt(torch.clamp(var, min=0)).to(x.dtype)


class _AgentCommon:
    def _norm_obs(self, obs):
        o = {k: obs[k] for k in self.keys}
        if self.state_norm is not None:
            nz = self.state_norm.normalize({k: o[k] for k in ('lidar', 'target')})
            o['lidar'], o['target'] = nz['lidar'].float(), nz['target'].float()
        else:
            o['lidar'], o['target'] = o['lidar'].float(), o['target'].float()
        o['action_mask'] = o['action_mask'].float()
        return o

    def observe(self, next_obs):
        """push_memory's `state_norm(next_obs, update=True)` (ppo_agent.py:213): fold the new observations in."""
        if self.state_norm is not None:
            self.state_norm.update({'lidar': next_obs['lidar'], 'target': next_obs['target']})

    # ---- stock-PyTorch levers for the INFERENCE forward (rollout side; off by default; bench.py --policy-fast) -------------
    def enable_fast_policy(self, graph=True, amp=False):
        """graph: replay the actor's inference forward as one captured device graph (static batch: launch overhead of its ~100
        small kernels disappears; same fp32 arithmetic, same results).  amp: run the token mixer and the embedding MLPs under
        bf16 autocast (the image encoder's convolutions stay fp32: `policy.set_img_amp` is the separate lever) -- changes the
        policy's numerics, never the env's.  Updates keep the plain fp32 path."""
        self._fast = {'graph': bool(graph), 'amp': bool(amp), 'g': None, 'in': None, 'out': None, 'replays': 0, 'captures': 0}

    def disable_fast_policy(self):
        """drop the levers and the captured graph (call it, or `enable_fast_policy` again, after anything that replaces the actor
        wholesale; a load_state_dict / `.to()` that re-materialises parameters is also detected: the graph is keyed on their storage)."""
        self._fast = None

    def _actor_key(self):
        """what a captured graph bakes in besides the input shapes: the actor object and the storage of every parameter / buffer."""
        ts = list(self.actor.parameters()) + list(self.actor.buffers())
        return (id(self.actor), tuple(t.data_ptr() for t in ts))

    def _actor_infer(self, nobs):
        f = getattr(self, '_fast', None)
        if f is not None and f['amp'] and next(self.actor.parameters()).is_cuda:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                return self.actor(nobs).float()
        return self.actor(nobs)

    @torch.no_grad()
    def policy_mean(self, nobs):
        f = getattr(self, '_fast', None)
        if f is None or not f['graph'] or not nobs['lidar'].is_cuda:
            return torch.clamp(self._actor_infer(nobs), -1, 1)                   # ppo_agent.py:137
        # the graph replays against the parameter storage it was captured with: the signature holds the input shapes AND the actor's
        # storage, so re-materialised weights (load_state_dict(assign=True), .to(), a swapped module) recapture instead of replaying
        # against freed memory; in-place updates (optimizer steps, load_state_dict's copy) keep the addresses and the graph
        sig = (tuple((k, tuple(v.shape), v.dtype) for k, v in nobs.items()), self._actor_key())
        if f['g'] is None or f['sig'] != sig:                                    # (re)capture for this batch shape / these weights
            f['in'] = {k: torch.empty_like(v) for k, v in nobs.items()}
            for k, v in nobs.items():
                f['in'][k].copy_(v)
            # inference semantics whatever mode the caller left the actor in (a train-mode dropout / batch-norm must not be baked in)
            was_training = self.actor.training
            self.actor.eval()
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):                                           # warm-up outside the capture (lazy inits)
                        torch.clamp(self._actor_infer(f['in']), -1, 1)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    f['out'] = torch.clamp(self._actor_infer(f['in']), -1, 1)
            finally:
                self.actor.train(was_training)
            f['g'], f['sig'] = g, sig
            f['captures'] += 1
        for k, v in nobs.items():
            f['in'][k].copy_(v)
        f['g'].replay()
        f['replays'] += 1
        return f['out'].clone()

    @torch.no_grad()
    def act(self, obs, use_mask=True, generator=None, planned=None, executing=None, plan_fn=None):
        """choose_action (mask-weighted discrete sampling, ppo_agent.py:163-169 + action_mask.py:199-227) or
        get_action (plain Gaussian sample, :171-185), clamped to [-1, 1]; scenes flagged in `executing` take the
        planner's action instead (ParkingAgent.choose_action, parking_agent.py:80-99).  Returns (action [N,2] float32,
        log_prob [N,2] under the current policy, normalised obs dict that was fed to the actor).
        plan_fn: called AFTER the policy forward and the sampling, returns (planned, executing) -- the forward needs the
        observation only, so it is enqueued before the caller waits for the Reeds-Shepp outputs of the env step."""
        nobs = self._norm_obs(obs)
        mean = self.policy_mean(nobs)
        log_std = self.log_std.expand_as(mean)
        if use_mask:
            a, _ = G.choose_action(mean, log_std.exp(), obs['action_mask'], generator)
            a = a.to(mean.dtype)
        else:
            a = mean + log_std.exp() * torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        a = torch.clamp(a, -1, 1)
        if plan_fn is not None:
            planned, executing = plan_fn()
        if planned is not None:
            a = torch.where(executing.unsqueeze(1), planned.to(a.dtype), a)
        return a, P.gaussian_log_prob(mean, log_std, a), nobs


class BatchedPPO(_AgentCommon):
    def __init__(self, device='cpu', use_img=False, lr=LR, gamma=GAMMA, lam=0.95, clip_epsilon=0.2, mini_epoch=10,
                 mini_batch=32, tau=TAU, adv_norm=True, state_norm=True, adam_epsilon=1e-8, img_use_tanh=True,
                 actor_cfg=None, critic_cfg=None):
        self.device = torch.device(device)
        self.keys = _obs_keys(use_img)
        self.gamma, self.lam, self.clip, self.mini_epoch, self.mini_batch, self.tau, self.adv_norm = \
            gamma, lam, clip_epsilon, mini_epoch, mini_batch, tau, adv_norm
        self.actor = P.HopeNet(actor_cfg or P.actor_configs(use_img=use_img), img_use_tanh).to(self.device)
        self.critic = P.HopeNet(critic_cfg or P.critic_configs(use_img=use_img), img_use_tanh).to(self.device)
        self.critic_target = copy.deepcopy(self.critic)
        self.log_std = torch.zeros(1, 2, device=self.device, requires_grad=True)                  # :77-81
        self.actor_opt = torch.optim.Adam([{'params': self.actor.parameters()}, {'params': [self.log_std]}], lr)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr * 5, eps=adam_epsilon)     # lr_critic = 5 lr
        self.state_norm = G.BatchedStateNorm(device=self.device) if state_norm else None
        self.last_losses = (float('nan'), float('nan'))
        self.allreduce_bytes = 0

    def trainable(self):
        return list(self.actor.parameters()) + [self.log_std] + list(self.critic.parameters())

    @torch.no_grad()
    def advantages(self, obs, reward, done, last_obs, chunk=65536):
        """GAE block of PPO.update (:255-273).  obs: normalised observations {k: [N,T,...]}, last_obs {k: [N,...]}
        (the observation after the newest transition), reward / done [N,T].  Returns (adv [N,T], v_target [N,T])."""
        n, t = reward.shape
        flat = {k: v.reshape((n * t,) + v.shape[2:]) for k, v in obs.items()}

        def run(net, x):
            m = next(iter(x.values())).shape[0]
            return torch.cat([net({k: v[i:i + chunk] for k, v in x.items()}) for i in range(0, m, chunk)]).squeeze(1)
        value = run(self.critic, flat).view(n, t)
        v_last = run(self.critic, last_obs)
        next_value = torch.cat([value[:, 1:], v_last.unsqueeze(1)], dim=1)
        adv, _ = G.batched_gae(reward, value, next_value, done, self.gamma, self.lam)
        v_target = adv + run(self.critic_target, flat).view(n, t)
        if self.adv_norm:
            m, s = _global_mean_std(adv)
            adv = (adv - m) / (s + 1e-5)
        return adv, v_target

    def update(self, obs, action, reward, done, old_log_prob, last_obs, generator=None, perms=None):
        """PPO.update (:236-349) over the [N, T] transitions collected since the last update.  `perms` (optional list of
        index permutations, one per epoch) replaces the internal shuffle (tests)."""
        adv, v_target = self.advantages(obs, reward, done, last_obs)
        n, t = reward.shape
        m = n * t
        flat = {k: v.reshape((m,) + v.shape[2:]) for k, v in obs.items()}
        action, old_lp = action.reshape(m, 2), old_log_prob.reshape(m, 2).sum(1, keepdim=True)
        adv, v_target = adv.reshape(m, 1), v_target.reshape(m, 1)
        params = self.trainable()
        mb = min(self.mini_batch, m)
        for ep in range(self.mini_epoch):
            perm = perms[ep] if perms is not None else torch.randperm(m, device=self.device, generator=generator)
            for i in range(0, m, mb):
                ri = perm[i:i + mb]
                st = {k: v[ri] for k, v in flat.items()}
                mean = torch.clamp(self.actor(st), -1, 1)
                lp = P.gaussian_log_prob(mean, self.log_std.expand_as(mean), action[ri]).sum(1, keepdim=True)
                ratio = (lp - old_lp[ri]).exp()
                a = adv[ri]
                actor_loss = -torch.min(ratio * a, torch.clamp(ratio, 1 - self.clip, 1 + self.clip) * a).mean()
                critic_loss = F.mse_loss(self.critic(st), v_target[ri])
                self.actor_opt.zero_grad(set_to_none=True)
                self.critic_opt.zero_grad(set_to_none=True)
                (actor_loss + critic_loss).backward()             # disjoint parameter sets: same grads as two backward()s
                self.allreduce_bytes += D.allreduce_gradients(params)
                self.actor_opt.step()
                self.critic_opt.step()
            _soft_update(self.critic_target, self.critic, self.tau)                                   # :338
        self.last_losses = (float(actor_loss.detach()), float(critic_loss.detach()))
        return self.last_losses


class BatchedSAC(_AgentCommon):
    def __init__(self, device='cpu', use_img=False, lr=LR, gamma=GAMMA, tau=0.005, batch_size=32,
                 initial_temperature=0.01, learn_temperature=True, state_norm=True, img_use_tanh=True):
        self.device = torch.device(device)
        self.keys = _obs_keys(use_img)
        self.gamma, self.tau, self.batch_size, self.learn_temperature = gamma, tau, batch_size, learn_temperature
        self.target_entropy = -2.0                                                                   # -action_dim (:51)
        self.actor = P.HopeNet(P.actor_configs(use_img=use_img), img_use_tanh).to(self.device)
        self.log_std = torch.zeros(1, 2, device=self.device, requires_grad=True)
        self.actor_opt = torch.optim.Adam([{'params': self.actor.parameters()}, {'params': [self.log_std]}], lr)
        self.critic1 = P.SacCritic(P.critic_configs(use_img=use_img), 2, img_use_tanh).to(self.device)
        self.critic2 = P.SacCritic(P.critic_configs(use_img=use_img), 2, img_use_tanh).to(self.device)
        self.critic_target1, self.critic_target2 = copy.deepcopy(self.critic1), copy.deepcopy(self.critic2)
        self.critic_opt1 = torch.optim.Adam(self.critic1.parameters(), lr)
        self.critic_opt2 = torch.optim.Adam(self.critic2.parameters(), lr)
        self.log_alpha = torch.tensor(math.log(initial_temperature), dtype=torch.float64, device=self.device, requires_grad=True)
        self.alpha_opt = torch.optim.Adam([self.log_alpha], lr)
        self.state_norm = G.BatchedStateNorm(device=self.device) if state_norm else None
        self.allreduce_bytes = 0

    @property
    def alpha(self):
        return self.log_alpha.exp()

    def _sample_action(self, nobs, noise=None):
        """_get_action_and_log_prob (:250-261): reparameterised sample, clamp, per-dimension log-probability."""
        mean = torch.clamp(self.actor(nobs), -1, 1)
        log_std = self.log_std.expand_as(mean)
        eps = torch.randn_like(mean) if noise is None else noise
        a = torch.clamp(mean + log_std.exp() * eps, -1, 1)
        return a, P.gaussian_log_prob(mean, log_std, a)

    def update(self, batch, noise=None):
        """one SAC update (:263-337) on a batch {'obs', 'next_obs' (normalised dicts), 'action', 'reward', 'done'}.
        `noise` = (eps for the next-state action, eps for the policy-loss action), tests only."""
        st, nst = batch['obs'], batch['next_obs']
        action, reward, done = batch['action'], batch['reward'].view(-1, 1), batch['done'].view(-1, 1)
        alpha = self.alpha.detach().to(reward.dtype)
        with torch.no_grad():
            na, nlp = self._sample_action(nst, None if noise is None else noise[0])
            nlp = nlp.sum(-1, keepdim=True)
            q_t = torch.min(self.critic_target1(nst, na), self.critic_target2(nst, na))
            q_target = reward + (1 - done) * self.gamma * (q_t - alpha * nlp)
        q1_loss = F.mse_loss(self.critic1(st, action), q_target)
        q2_loss = F.mse_loss(self.critic2(st, action), q_target)
        self.critic_opt1.zero_grad(set_to_none=True)
        self.critic_opt2.zero_grad(set_to_none=True)
        (q1_loss + q2_loss).backward()
        cparams = list(self.critic1.parameters()) + list(self.critic2.parameters())
        self.allreduce_bytes += D.allreduce_gradients(cparams)
        self.critic_opt1.step()
        self.critic_opt2.step()
        for p in cparams:
            p.requires_grad_(False)
        a_, lp = self._sample_action(st, None if noise is None else noise[1])
        lp = lp.sum(-1, keepdim=True)
        actor_loss = (alpha * lp - torch.min(self.critic1(st, a_), self.critic2(st, a_))).mean()
        self.actor_opt.zero_grad(set_to_none=True)
        actor_loss.backward()
        self.allreduce_bytes += D.allreduce_gradients(list(self.actor.parameters()) + [self.log_std])
        self.actor_opt.step()
        for p in cparams:
            p.requires_grad_(True)
        if self.learn_temperature:
            alpha_loss = (self.alpha * (-lp - self.target_entropy).detach()).mean()
            self.alpha_opt.zero_grad(set_to_none=True)
            alpha_loss.backward()
            self.allreduce_bytes += D.allreduce_gradients([self.log_alpha])
            self.alpha_opt.step()
        _soft_update(self.critic_target1, self.critic1, self.tau)
        _soft_update(self.critic_target2, self.critic2, self.tau)
        return float(actor_loss.detach()), float(q1_loss.detach())
