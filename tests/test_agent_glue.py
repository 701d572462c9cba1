"""Batched agent glue against reference-generated vectors (tests/golden/agent_glue.npz:
RsPlanner.set_rs_path, ActionMask.choose_action's probability vector, StateNorm)."""
import numpy as np
import pytest
import torch

from hope_amd import agent_glue as G


def check_rs_planner(g, dev='cpu'):
    n = len(g['words'])
    word = torch.full((n, 8), -1, dtype=torch.int8)
    word[:, :5] = torch.from_numpy(g['words'])
    word[:, 5] = torch.from_numpy((g['words'] >= 0).sum(1).astype(np.int8))
    word[:, 6] = 1
    word, lengths = word.to(dev), torch.from_numpy(g['lengths']).to(dev)
    # (the fixture holds raw calc_all_paths words, some hundreds of metres long: lift the 96-action queue cap here)
    acts, cnt = G.BatchedRsPlanner.expand(word, lengths, max_actions=2048)
    acts, cnt = acts.cpu(), cnt.cpu()
    off = g['action_off']
    assert np.array_equal(cnt.numpy(), np.diff(off))
    for i in range(n):
        assert np.array_equal(acts[i, :cnt[i]].numpy(), g['actions'][off[i]:off[i + 1]]), i     # exact (x - k is exact)
    # queue semantics: adopt only when idle, pop in order, idle again after the last action
    pl = G.BatchedRsPlanner(n, device=dev, max_actions=2048)
    took = pl.set_paths(word, lengths)
    assert np.array_equal(took.cpu().numpy(), np.diff(off) > 0) or took.all()
    first, valid = pl.get_actions()
    has = np.diff(off) > 0
    assert np.array_equal(valid.cpu().numpy(), has)
    assert np.array_equal(first[valid].cpu().numpy(), g['actions'][off[:-1][has]])
    word2 = word.clone(); word2[:, 0] = 0
    assert not pl.set_paths(word2, lengths)[pl.executing].any()       # busy scenes keep their path
    for _ in range(2100):
        pl.get_actions()
        if not pl.executing.any():
            break
    assert not pl.executing.any()


def test_rs_planner_expansion_matches_reference(gold):
    check_rs_planner(gold('agent_glue.npz'))


def check_choose_action(g, dev='cpu'):
    D = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731
    p = G.mask_action_probs(D('pa_mean'), D('pa_std'), D('pa_mask'))
    assert np.abs(p.cpu().numpy() - g['pa_probs']).max() < 1e-13
    assert np.abs(p.sum(1).cpu().numpy() - 1).max() < 1e-12
    gen = torch.Generator(device=dev).manual_seed(0)
    a, idx = G.choose_action(D('pa_mean'), D('pa_std'), D('pa_mask'), gen)
    a, idx = a.cpu(), idx.cpu()
    assert a.shape == (200, 2) and (g['pa_mask'][np.arange(200), idx.numpy()] > 0).all()
    assert set(np.unique(a[:, 1].numpy())) <= {-1.0, 1.0} and a[:, 0].abs().max() <= 1.0 + 1e-12


def test_choose_action_probabilities_match_reference(gold):
    check_choose_action(gold('agent_glue.npz'))


def check_state_norm(g, dev='cpu'):
    lid, tgt = torch.from_numpy(g['sn_lidar']).to(dev), torch.from_numpy(g['sn_target']).to(dev)
    # one at a time == the reference's Welford recurrence
    sn = G.BatchedStateNorm(device=dev)
    outs = []
    for i in range(len(lid)):
        sn.update({'lidar': lid[i:i + 1], 'target': tgt[i:i + 1]})
        outs.append(sn.normalize({'target': tgt[i]})['target'].cpu().numpy())
    assert sn.n_state == int(g['sn_n'])
    assert np.abs(sn.mean['lidar'].cpu().numpy() - g['sn_mean_lidar']).max() < 1e-10
    assert np.abs(sn.std['target'].cpu().numpy() - g['sn_std_target']).max() < 1e-10
    assert np.abs(np.array(outs) - g['sn_norm_target']).max() < 1e-8
    # folded in batches of 37: same statistics (parallel-merge form of the recurrence)
    sb = G.BatchedStateNorm(device=dev)
    for a in range(0, len(lid), 37):
        sb.update({'lidar': lid[a:a + 37], 'target': tgt[a:a + 37]})
    assert sb.n_state == sn.n_state
    assert np.abs(sb.mean['lidar'].cpu().numpy() - g['sn_mean_lidar']).max() < 1e-10
    assert np.abs(sb.std['lidar'].cpu().numpy() - g['sn_std_lidar']).max() < 1e-9
    sb.fix_parameters()
    m = sb.mean['target'].clone()
    sb.update({'lidar': lid[:5], 'target': tgt[:5]})
    assert torch.equal(m, sb.mean['target'])


def test_state_norm_matches_reference_recurrence(gold):
    check_state_norm(gold('agent_glue.npz'))


def test_batched_gae_equals_the_reference_loop_per_scene():
    """ppo_agent.py:258-273 restated literally (one env, time order, Python-float recurrence) vs batched_gae rows"""
    import torch
    from hope_amd.agent_glue import batched_gae
    rng = np.random.default_rng(0)
    N, T, gamma, lam = 7, 33, 0.98, 0.95
    r = rng.normal(size=(N, T)).astype(np.float32)
    v = rng.normal(size=(N, T)).astype(np.float32)
    nv = rng.normal(size=(N, T)).astype(np.float32)
    d = (rng.random((N, T)) < 0.1).astype(np.float32)
    adv, delta = batched_gae(torch.tensor(r), torch.tensor(v), torch.tensor(nv), torch.tensor(d), gamma, lam)
    for i in range(N):
        deltas = torch.tensor(r[i]) + gamma * (1 - torch.tensor(d[i])) * torch.tensor(nv[i]) - torch.tensor(v[i])
        gae, out = 0, []
        for dl, dn in zip(reversed(deltas.flatten().numpy().astype(np.float64)), reversed(d[i].astype(np.float64))):
            gae = dl + gamma * lam * gae * (1.0 - dn)
            out.append(gae)
        out.reverse()
        want = torch.FloatTensor(out)
        assert torch.allclose(adv[i], want, rtol=0, atol=1e-6)
        assert torch.allclose(delta[i], deltas, rtol=0, atol=1e-6)


def test_rollout_storage_order_and_next_state_semantics():
    import torch
    from hope_amd.agent_glue import RolloutStorage
    st = RolloutStorage(3, 4, {'lidar': (5,), 'target': (2,)})
    for t in range(6):                                   # wraps: only the last 4 steps stay
        obs = {'lidar': torch.full((3, 5), float(t)), 'target': torch.full((3, 2), float(t))}
        st.push(obs, torch.full((3, 2), float(t)), torch.full((3,), float(t)), torch.tensor([0, 0, t == 4]),
                log_prob=torch.zeros(3, 2))
    o = st.ordered()
    assert o['reward'][0].tolist() == [2.0, 3.0, 4.0, 5.0] and o['obs']['lidar'][1, :, 0].tolist() == [2.0, 3.0, 4.0, 5.0]
    g = torch.Generator().manual_seed(0)
    b = st.sample(256, generator=g)
    newest = b['reward'] == 5.0
    ended = (b['done'] == 1)
    assert (b['next_valid'] == ~(newest | ended)).all()
    ok = b['next_valid']
    assert (b['next_obs']['lidar'][ok][:, 0] == b['obs']['lidar'][ok][:, 0] + 1).all()
    st.clear()
    assert st.size == 0


def test_load_reference_checkpoint_without_the_reference_on_the_path():
    """tests/golden/ref_checkpoint.pt pickles the reference's own model.state_norm.StateNorm by module path
    (make_checkpoint.py); hope_amd.checkpoint reads it with `model` NOT importable"""
    import os
    import sys
    import torch
    from conftest import GOLD
    from hope_amd.checkpoint import load_hope_checkpoint
    assert 'model' not in sys.modules and not any(p.endswith('reference/src') for p in sys.path)
    ck = load_hope_checkpoint(os.path.join(GOLD, 'ref_checkpoint.pt'))
    e = np.load(os.path.join(GOLD, 'ref_checkpoint_expect.npz'))
    assert set(ck['state_dicts']) == {'actor_net', 'critic_net1'}
    assert np.array_equal(ck['state_dicts']['actor_net']['weight'].numpy(), e['actor_weight'])
    assert ck['log_std'].shape == (1, 2)
    sn = ck['state_norm']
    assert sn.n_state == int(e['n_state']) and sn.modal == ('lidar', 'target')
    assert np.allclose(sn.mean['lidar'].numpy(), e['mean_lidar']) and np.allclose(sn.std['target'].numpy(), e['std_target'])
    obs = {'lidar': torch.ones(2, 120), 'target': torch.zeros(2, 5), 'action_mask': torch.ones(2, 42)}
    out = sn.normalize(obs)
    assert np.allclose(out['lidar'][0].numpy(), (1.0 - e['mean_lidar']) / (e['std_lidar'] + 1e-8), atol=1e-6)
    assert torch.equal(out['action_mask'], obs['action_mask'])


def test_checkpoint_loader_refuses_foreign_globals(tmp_path):
    """a checkpoint is untrusted input -- only tensor / optimizer / container globals and the reference's own
    classes (mapped to inert stand-ins) may be unpickled."""
    import pickle
    import pytest
    import torch
    from hope_amd.checkpoint import load_hope_checkpoint

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo pwned > /dev/null',))
    path = tmp_path / 'evil.pt'
    torch.save({'actor_net': {'weight': torch.zeros(2)}, 'configs': Evil()}, str(path))
    with pytest.raises(pickle.UnpicklingError, match='refusing'):
        load_hope_checkpoint(str(path))


@pytest.mark.parametrize('glob', ['torch._utils._import_dotted_name', 'torch.storage._load_from_bytes', 'torch._utils.importlib',
                                  'torch.optim.lr_scheduler.LambdaLR', 'numpy.core.multiarray.fromfile'])
def test_checkpoint_loader_has_no_module_wide_allowances(glob):
    """the allow-list is a set of names.  A pickle that names a callable living in an otherwise trusted module
    (torch._utils._import_dotted_name returns ANY importable callable; torch.storage._load_from_bytes is an unrestricted
    torch.load) must be refused before anything is imported or called."""
    import io
    import pickle
    from hope_amd.checkpoint import _Unpickler
    mod, name = glob.rsplit('.', 1)
    # GLOBAL mod name ; MARK ; STRING 'os.getcwd' ; TUPLE ; REDUCE ; STOP  -- hand-built, nothing imported on this side
    blob = b'c' + mod.encode() + b'\n' + name.encode() + b"\n(S'os.getcwd'\ntR."
    with pytest.raises(pickle.UnpicklingError, match='refusing'):
        _Unpickler(io.BytesIO(blob)).load()
