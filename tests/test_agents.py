"""hope_amd.policy / hope_amd.agents vs vectors produced by the REFERENCE's own networks and update() methods
(tests/golden/make_golden_r2.py ran src/model/network.py, ppo_agent.py:236-349 and sac_agent.py:263-337 unmodified).
SURVEY.md §8f row f-3; BASELINE configs 4 and 5.  The same checks run on the GPU in tests/test_gpu_agents.py."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from det_weights import det_fill, probe  # noqa: E402

from hope_amd import agents as A  # noqa: E402
from hope_amd import policy as P  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _obs(g, prefix, dev, keys=('lidar', 'target', 'action_mask', 'img')):
    return {k: torch.from_numpy(g[prefix + k]).to(dev) for k in keys if prefix + k in g}


def check_policy_forward(dev, tol):
    g = _load('policy_forward.npz')
    x = _obs(g, 'x_', dev)
    act = torch.from_numpy(g['action']).to(dev)
    for img in (True, False):
        tag = 'img' if img else 'noimg'
        a = det_fill(P.HopeNet(P.actor_configs(use_img=img)), 0.1).to(dev)
        c = det_fill(P.HopeNet(P.critic_configs(use_img=img)), 0.2).to(dev)
        q = det_fill(P.SacCritic(P.critic_configs(use_img=img)), 0.3).to(dev)
        assert P.count_parameters(a) == int(g['n_actor_' + tag]) and P.count_parameters(c) == int(g['n_critic_' + tag])
        assert P.count_parameters(q) == int(g['n_q_' + tag])
        with torch.no_grad():
            assert np.abs(a(x).cpu().numpy() - g['actor_' + tag]).max() < tol
            assert np.abs(c(x).cpu().numpy() - g['critic_' + tag]).max() < tol * 4
            assert np.abs(q(x, act).cpu().numpy() - g['q_' + tag]).max() < tol * 4
    return True


def test_param_count_and_state_dict_names():
    """909 778 actor parameters and the reference's state_dict key names / shapes (checkpoint loading)."""
    g = _load('policy_forward.npz')
    a = P.HopeNet(P.actor_configs())
    assert P.count_parameters(a) == 909778 == int(g['n_actor_img'])
    assert list(a.state_dict().keys()) == list(g['keys_actor'])
    assert [str(tuple(v.shape)) for v in a.state_dict().values()] == list(g['shapes_actor'])
    assert list(P.SacCritic(P.critic_configs()).state_dict().keys()) == list(g['keys_q'])


def test_policy_forward_matches_reference():
    check_policy_forward('cpu', 2e-5)


def test_orthogonal_init_distribution():
    """MultiObsEmbedding.orthogonal_init: trunk / embedding weights orthogonal (gain 1), biases 0."""
    torch.manual_seed(0)
    a = P.HopeNet(P.actor_configs(use_img=False))
    w = a.embed_lidar[0].weight                              # [128, 120]: orthonormal columns
    assert torch.allclose(w.t() @ w, torch.eye(120), atol=1e-5)
    w = a.net.output[0].weight                               # [128, 384]: orthonormal rows
    assert torch.allclose(w @ w.t(), torch.eye(128), atol=1e-5)
    assert all(float(p.detach().abs().max()) == 0 for n, p in a.named_parameters() if n.endswith('bias') and 'embed_img' not in n)


def make_ppo(g, dev, mini_batch):
    ag = A.BatchedPPO(device=dev, use_img=False, lr=float(g['lr'][0]), mini_batch=mini_batch, state_norm=False,
                      mini_epoch=len(g['perms']))
    det_fill(ag.actor, 0.1)
    det_fill(ag.critic, 0.2)
    ag.critic_target.load_state_dict(ag.critic.state_dict())
    with torch.no_grad():
        ag.log_std.copy_(torch.tensor([[-0.3, 0.2]], device=dev))
    return ag


def ppo_inputs(g, dev, n_rows):
    """the reference's single 24-step sequence laid out as n_rows scenes x (24 / n_rows) steps (rows end on done = 1)."""
    x = _obs(g, 'x_', dev)
    T = 24 // n_rows
    obs = {k: v[:24].reshape((n_rows, T) + v.shape[1:]) for k, v in x.items()}
    last = {k: torch.stack([v[(r + 1) * T] for r in range(n_rows)]) for k, v in x.items()}
    f = lambda k: torch.from_numpy(g[k]).to(dev)
    return obs, last, f('action').view(n_rows, T, 2), f('reward').view(n_rows, T), f('done').view(n_rows, T), f('log_prob').view(n_rows, T, 2)


def check_ppo_update(dev, n_rows, tol=5e-4):
    g = _load('ppo_update.npz')
    ag = make_ppo(g, dev, 8)
    obs, last, action, reward, done, lp = ppo_inputs(g, dev, n_rows)
    perms = [torch.from_numpy(p).to(dev) for p in g['perms']]
    # collect every mini-batch loss like the reference's actor_loss_list / critic_loss_list
    import torch.nn.functional as F
    real_mse = F.mse_loss
    rec = {'a': [], 'c': []}

    def mse(a, b):
        v = real_mse(a, b)
        rec["c"].append(float(v.detach()))
        return v
    A.F.mse_loss = mse
    try:
        ag.update(obs, action, reward, done, lp, last, perms=perms)
    finally:
        A.F.mse_loss = real_mse
    assert len(rec['c']) == 3 * len(perms) == len(g['critic_losses'])
    assert np.allclose(rec['c'], g['critic_losses'], rtol=tol, atol=1e-6), np.abs(np.array(rec['c']) - g['critic_losses']).max()
    assert abs(ag.last_losses[0] - g['actor_losses'][-1]) < tol * max(1, abs(g['actor_losses'][-1]))
    for name, net in (('probe_actor', ag.actor), ('probe_critic', ag.critic), ('probe_target', ag.critic_target)):
        got, exp = probe(net.cpu()).numpy(), g[name]
        scale = np.maximum(np.abs(exp[:, 1:2]), 1.0)            # abs-sum of the tensor
        assert (np.abs(got - exp) / scale).max() < tol, name
    assert np.allclose(ag.log_std.detach().cpu().numpy(), g['log_std'], atol=tol)


@pytest.mark.parametrize('n_rows', [1, 2])
def test_ppo_update_matches_reference(n_rows):
    """[1, 24] is the reference's own layout; [2, 12] checks that GAE per scene row + done at the row end is the same."""
    check_ppo_update('cpu', n_rows)


def check_sac_update(dev, tol=5e-4):
    g = _load('sac_update.npz')
    ag = A.BatchedSAC(device=dev, use_img=True, lr=2e-4, batch_size=16, state_norm=False)
    det_fill(ag.actor, 0.1)
    det_fill(ag.critic1, 0.3)
    det_fill(ag.critic2, 0.4)
    ag.critic_target1.load_state_dict(ag.critic1.state_dict())
    ag.critic_target2.load_state_dict(ag.critic2.state_dict())
    with torch.no_grad():
        ag.log_std.copy_(torch.tensor([[-0.5, -0.1]], device=dev))
    f = lambda k: torch.from_numpy(g[k]).to(dev)
    batch = {'obs': _obs(g, 'x_', dev), 'next_obs': _obs(g, 'nx_', dev), 'action': f('action'), 'reward': f('reward'), 'done': f('done')}
    eps = f('eps')
    for u in range(3):
        a, b = ag.update(batch, noise=(eps[2 * u], eps[2 * u + 1]))
        assert abs(a - g['losses'][u, 0]) < tol * max(1, abs(g['losses'][u, 0])), (u, a, g['losses'][u])
        assert abs(b - g['losses'][u, 1]) < tol * max(1, abs(g['losses'][u, 1])), (u, b, g['losses'][u])
    for name, net in (('probe_actor', ag.actor), ('probe_q1', ag.critic1), ('probe_q2', ag.critic2),
                      ('probe_t1', ag.critic_target1), ('probe_t2', ag.critic_target2)):
        got, exp = probe(net.cpu()).numpy(), g[name]
        assert (np.abs(got - exp) / np.maximum(np.abs(exp[:, 1:2]), 1.0)).max() < tol, name
    assert abs(float(ag.log_alpha.detach()) - float(g['log_alpha'])) < 1e-5
    assert np.allclose(ag.log_std.detach().cpu().numpy(), g['log_std'], atol=tol)


def test_sac_update_matches_reference():
    check_sac_update('cpu')


def test_act_mask_and_planner_override():
    torch.manual_seed(1)
    ag = A.BatchedPPO(device='cpu', use_img=False)
    n = 64
    obs = {'lidar': torch.rand(n, 120) * 10, 'target': torch.randn(n, 5), 'action_mask': torch.zeros(n, 42)}
    obs['action_mask'][:, 7] = 1.0                                            # only discrete action 7 is allowed
    ag.observe(obs)
    planned = torch.tensor([[1.0, -0.4]]).repeat(n, 1)
    ex = torch.arange(n) % 2 == 0
    a, lp, nobs = ag.act(obs, use_mask=True, planned=planned, executing=ex)
    from hope_amd import tables as T
    a7 = torch.tensor(T.discrete_actions()[7] / [0.75, 1.0], dtype=torch.float32)
    assert torch.allclose(a[~ex], a7.expand(int((~ex).sum()), 2)) and torch.allclose(a[ex], planned[ex])
    mean = ag.policy_mean(nobs)
    assert torch.allclose(lp, torch.distributions.Normal(mean, ag.log_std.exp().expand_as(mean)).log_prob(a), atol=1e-6)


# ---- 2 ranks (gloo): a sharded PPO update equals the single-process update over the same transitions ------------------
def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = _load('ppo_update.npz')
    ag = make_ppo(g, 'cpu', 12)
    obs, last, action, reward, done, lp = ppo_inputs(g, 'cpu', 2)
    sl = slice(rank, rank + 1)
    perms = [torch.arange(12) for _ in range(2)]
    ag.update({k: v[sl] for k, v in obs.items()}, action[sl], reward[sl], done[sl], lp[sl], {k: v[sl] for k, v in last.items()}, perms=perms)
    q.put((rank, probe(ag.actor).numpy(), probe(ag.critic).numpy(), ag.allreduce_bytes, sum(p.numel() for p in ag.trainable() if p.grad is not None)))
    dist.destroy_process_group()


def test_ppo_update_two_ranks_equals_one_process():
    import torch.multiprocessing as mp
    g = _load('ppo_update.npz')
    ag = make_ppo(g, 'cpu', 24)
    obs, last, action, reward, done, lp = ppo_inputs(g, 'cpu', 2)
    ag.update(obs, action, reward, done, lp, last, perms=[torch.arange(24) for _ in range(2)])
    exp_a, exp_c = probe(ag.actor).numpy(), probe(ag.critic).numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in range(2)]
    [p.join(60) for p in ps]
    for rank, pa, pc, nbytes, nparam in res:
        assert nbytes == 2 * nparam * 4                               # one fused bucket per mini-batch, every epoch
        assert (np.abs(pa - exp_a) / np.maximum(np.abs(exp_a[:, 1:2]), 1.0)).max() < 2e-4
        assert (np.abs(pc - exp_c) / np.maximum(np.abs(exp_c[:, 1:2]), 1.0)).max() < 2e-4


# ---- the training loops on a CPU stand-in env (tests/fake_env.py drives the C oracle) --------------------------------
def _small_scenes(n, seed=3):
    from hope_amd.scenes import SceneSource
    src = SceneSource(levels=('Normal', 'Complex', 'Extrem'), seed=seed)
    return [src.draw() for _ in range(n)]


def test_ppo_trainer_loop_cpu():
    from fake_env import OracleEnv
    from hope_amd.rollout import PPOTrainer
    torch.manual_seed(0)
    env = OracleEnv(_small_scenes(12))
    ag = A.BatchedPPO(device='cpu', use_img=False, lr=1e-4, mini_batch=24, mini_epoch=2)
    tr = PPOTrainer(env, ag, horizon=4, seed=1)
    before = probe(ag.actor).numpy().copy()
    out = [tr.step() for _ in range(8)]
    assert [o is not None for o in out] == [False, False, False, True] * 2 and tr.updates == 2
    assert np.isfinite(out[3]).all() and np.isfinite(out[7]).all()
    assert np.abs(probe(ag.actor).numpy() - before).max() > 0           # the actor moved
    assert tr.ring.size == 0 and ag.state_norm.n_state == 1 + 12 * 8 + 11   # first sample + every later observation
    s = tr.stats()
    assert s['steps'] == 8 and np.isfinite(s['mean_reward'])


def test_sac_trainer_loop_cpu():
    from fake_env import OracleEnv
    from hope_amd.rollout import SACTrainer
    torch.manual_seed(0)
    env = OracleEnv(_small_scenes(8))
    ag = A.BatchedSAC(device='cpu', use_img=False, lr=1e-4, batch_size=16)
    tr = SACTrainer(env, ag, horizon=3, update_every=2, seed=1)
    out = [tr.step() for _ in range(8)]
    assert tr.updates == 3 and [o is not None for o in out] == [False, False, False, True, False, True, False, True]
    assert all(np.isfinite(o).all() for o in out if o is not None)
    # ring semantics: newest column's next-observation is the observation the agent acts on next
    b = tr.ring.sample(64, tr.last_obs(), tr.gen)
    assert b['obs']['lidar'].shape == (64, 120) and b['next_obs']['action_mask'].shape == (64, 42)


def _loop_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from fake_env import OracleEnv
    from hope_amd.dist import shard_range
    from hope_amd.rollout import PPOTrainer
    torch.manual_seed(0)                                             # identical initial weights on every rank
    ag = A.BatchedPPO(device='cpu', use_img=False, lr=1e-4, mini_batch=16, mini_epoch=1)
    scenes = _small_scenes(8)
    lo, hi = shard_range(len(scenes), rank, world)
    tr = PPOTrainer(OracleEnv(scenes[lo:hi]), ag, horizon=4, seed=10 + rank)
    for _ in range(4):
        tr.step()
    q.put((rank, probe(ag.actor).numpy(), probe(ag.critic).numpy(), tr.updates))
    dist.destroy_process_group()


def test_ppo_trainer_two_ranks_stay_in_sync():
    """BASELINE config 5's structure on CPU: scenes sharded over 2 ranks, different rollouts, ONE fused gradient
    all-reduce per mini-batch -> both ranks hold identical weights after the update."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_loop_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    [p.join(60) for p in ps]
    assert res[0][3] == res[1][3] == 1
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_advantage_normalisation_survives_a_single_transition():
    """N*T == 1 must not turn the advantages into NaN (unbiased std of one element)."""
    from hope_amd.agents import _global_mean_std
    m, s = _global_mean_std(torch.tensor([0.25]))
    assert torch.isfinite(m) and torch.isfinite(s) and float(s) == 0.0 and float(m) == 0.25
    m, s = _global_mean_std(torch.tensor([1.0, 3.0]))
    assert abs(float(m) - 2.0) < 1e-7 and abs(float(s) - torch.tensor([1.0, 3.0]).std().item()) < 1e-7
