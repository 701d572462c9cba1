"""Batched rollout loop over N scenes: the N >> 1 counterpart of the episode loop in
src/train/train_HOPE_ppo.py:177-213 / src/evaluation/eval_utils.py:16-84 with the hybrid controller of
src/model/agent/parking_agent.py (replay a found Reeds-Shepp path, otherwise ask the policy).

The policy network itself is out of this library's scope (it runs on stock PyTorch-ROCm); `StandInPolicy` is a
small random-init MLP with the same inputs / outputs (lidar 120 + target 5 + action_mask 42 -> Gaussian over
(steer, speed)) so that the loop can be exercised and timed end to end.
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
