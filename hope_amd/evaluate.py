"""Batched evaluator: `eval()` of the reference (src/evaluation/eval_utils.py:16-84, driven by eval_mix_scene.py:86-115) for N
episodes at once -- one evaluation episode per scene slot, all slots stepped together, finished slots frozen.

Per episode, as the reference records them (:28-84): the final status, the step count, the summed wrapper reward and the path
length (sum of the rear-axle displacement per step, :52-53).  The loop keeps the reference's stuck detector (:46-47: when
obs['target'] did not change since the previous step the action is replaced by `env.action_space.sample()`, a uniform draw from
the PHYSICAL action box that the wrapper then clips to [-1, 1]) and hands found Reeds-Shepp paths to the planner (:55-56).
`summarize` produces the numbers result.txt holds (:131-149): success rate, steps of the successful episodes, and per map level
(`env.map.map_level`, :69) success rate, steps and the path length of the episodes shorter than 200 steps; OUTBOUND episodes
count 200 steps in the per-case step record (:73-76).  `BatchedEvaluator.run` returns one record per episode; with
torch.distributed initialised the records of all ranks are gathered (hope_amd.dist.gather_eval_stats: one all-gather)."""
import numpy as np
import torch

from . import agent_glue as G
from . import dist as D
from . import tables as T

LEVELS = ('Normal', 'Complex', 'Extrem', 'dlp')          # map_level labels; DLP lots get get_map_level's label when asked
TOLERANT_TIME = 200


class BatchedEvaluator:
    def __init__(self, env, agent, post_proc_action=True, use_planner=True, seed=0):
        """agent: a hope_amd.agents agent (`act(obs, use_mask, generator, planned, executing)`); post_proc_action: PPO's
        mask-weighted choose_action (eval_utils.py:42-43) instead of the plain sample (:44-45)."""
        self.env, self.agent, self.use_mask = env, agent, bool(post_proc_action)
        self.planner = G.BatchedRsPlanner(env.n, device=env.device) if use_planner else None
        self.gen = torch.Generator(device=env.device)
        self.gen.manual_seed(seed)
        lo = torch.tensor([T.VALID_STEER[0], T.VALID_SPEED[0]], dtype=torch.float64, device=env.device)
        hi = torch.tensor([T.VALID_STEER[1], T.VALID_SPEED[1]], dtype=torch.float64, device=env.device)
        self._lo, self._span = lo, hi - lo

    def _obs(self):
        e = self.env
        o = {'lidar': e.lidar, 'target': e.target, 'action_mask': e.action_mask}
        if 'img' in self.agent.keys:
            o['img'] = e.img
        return o

    @torch.no_grad()
    def run(self, max_steps=TOLERANT_TIME + 2, gather=True):
        """one episode per scene slot from the slots' current maps (the caller has uploaded / drawn them and not stepped yet).
        -> float32 [n_total, 4]: status, steps, reward, path length."""
        env, agent = self.env, self.agent
        n, dev = env.n, env.device
        env.reset_obs()                                               # env.reset(i + 1) -> first observation (:32)
        if self.planner is not None:
            self.planner.reset()                                      # agent.reset() (:33)
        alive = torch.ones(n, dtype=torch.bool, device=dev)
        steps = torch.zeros(n, dtype=torch.int32, device=dev)
        total = torch.zeros(n, dtype=torch.float64, device=dev)
        path = torch.zeros(n, dtype=torch.float64, device=dev)
        status = torch.ones(n, dtype=torch.int32, device=dev)
        last_xy = env.pose[:, :2].clone()
        last_target = env.target.clone()                              # last_obs = obs['target'] (:39)
        first = True
        for _ in range(max_steps):
            if not bool(alive.any()):
                break
            planned, executing = self.planner.get_actions() if self.planner is not None else (None, None)
            action, _, _ = agent.act(self._obs(), self.use_mask, self.gen, planned, executing)
            # stuck detector (:46-47): the very first comparison is obs['target'] with itself -> always a random first action
            same = torch.ones(n, dtype=torch.bool, device=dev) if first else (env.target == last_target).all(dim=1)
            first = False
            rnd = self._lo + self._span * torch.rand((n, 2), device=dev, dtype=torch.float64, generator=self.gen)
            action = torch.where(same.unsqueeze(1), rnd.clamp(-1, 1).to(action.dtype), action)
            last_target = env.target.clone()
            env.step(action.to(env.action_dtype).contiguous(), active=alive.to(torch.uint8))
            steps += alive.to(torch.int32)
            total += torch.where(alive, env.reward.double(), torch.zeros_like(total))
            xy = env.pose[:, :2]
            path += torch.where(alive, (xy - last_xy).norm(dim=1), torch.zeros_like(path))
            last_xy = xy.clone()
            done = alive & env.done.bool()
            status = torch.where(done, env.status, status)
            if self.planner is not None:
                self.planner.reset(done)
                word = env.rs_word.clone()                         # info['path_to_dest'] -> agent.set_planner_path (:55-56)
                word[~(alive & ~done), 6] = 0                      # (frozen slots keep stale outputs)
                self.planner.set_paths(word, env.rs_lengths)
            alive = alive & ~done
        rec = torch.stack([status.float(), steps.float(), total.float(), path.float()], dim=1)
        if gather:
            return D.gather_eval_stats(status, steps, total.float(), path.float())
        return rec


def summarize(records, levels=None):
    """records [n, 4] (status, steps, reward, path length) [+ levels: sequence of n map-level labels] -> the numbers of
    result.txt (eval_utils.py:131-149)"""
    r = records.detach().cpu().numpy() if torch.is_tensor(records) else np.asarray(records)
    status, steps, reward, plen = r[:, 0].astype(int), r[:, 1], r[:, 2], r[:, 3]
    succ = status == 2

    def block(sel, per_case):
        # per_case: the per-case / all-episodes `step_record`, in which an OUTBOUND episode counts 200 steps (eval_utils.py:73-76);
        # the per-level "step num" of result.txt comes from step_num_level = the raw step_num (:71, :143)
        s = sel & succ
        step_rec = np.where(status[sel] == 4, float(TOLERANT_TIME), steps[sel]) if per_case else steps[sel]
        short = sel & (steps < TOLERANT_TIME)
        return {'episodes': int(sel.sum()), 'success_rate': float(succ[sel].mean()) if sel.any() else 0.0,
                'step_num_mean': float(step_rec.mean()) if sel.any() else 0.0, 'step_num_std': float(step_rec.std()) if sel.any() else 0.0,
                'success_step_mean': float(steps[s].mean()) if s.any() else 0.0,
                'path_length_mean': float(plen[short].mean()) if short.any() else 0.0,
                'path_length_std': float(plen[short].std()) if short.any() else 0.0,
                'reward_mean': float(reward[sel].mean()) if sel.any() else 0.0}
    out = {'all': block(np.ones(len(r), bool), True)}
    if levels is not None:
        lv = np.asarray(levels)
        for k in sorted(set(lv.tolist())):
            out[str(k)] = block(lv == k, False)
    return out
