"""Batched agent glue against reference-generated vectors (tests/golden/agent_glue.npz:
RsPlanner.set_rs_path, ActionMask.choose_action's probability vector, StateNorm)."""
import numpy as np
import torch

from hope_amd import agent_glue as G


def test_rs_planner_expansion_matches_reference(gold):
    g = gold('agent_glue.npz')
    n = len(g['words'])
    word = torch.full((n, 8), -1, dtype=torch.int8)
    word[:, :5] = torch.from_numpy(g['words'])
    word[:, 5] = torch.from_numpy((g['words'] >= 0).sum(1).astype(np.int8))
    word[:, 6] = 1
    # (the fixture holds raw calc_all_paths words, some hundreds of metres long: lift the 96-action queue cap here)
    acts, cnt = G.BatchedRsPlanner.expand(word, torch.from_numpy(g['lengths']), max_actions=2048)
    off = g['action_off']
    assert np.array_equal(cnt.numpy(), np.diff(off))
    for i in range(n):
        assert np.array_equal(acts[i, :cnt[i]].numpy(), g['actions'][off[i]:off[i + 1]]), i     # exact (x - k is exact)
    # queue semantics: adopt only when idle, pop in order, idle again after the last action
    pl = G.BatchedRsPlanner(n, max_actions=2048)
    took = pl.set_paths(word, torch.from_numpy(g['lengths']))
    assert np.array_equal(took.numpy(), np.diff(off) > 0) or took.all()
    first, valid = pl.get_actions()
    has = np.diff(off) > 0
    assert np.array_equal(valid.numpy(), has)
    assert np.array_equal(first[valid].numpy(), g['actions'][off[:-1][has]])
    word2 = word.clone(); word2[:, 0] = 0
    assert not pl.set_paths(word2, torch.from_numpy(g['lengths']))[pl.executing].any()       # busy scenes keep their path
    for _ in range(2100):
        pl.get_actions()
        if not pl.executing.any():
            break
    assert not pl.executing.any()


def test_choose_action_probabilities_match_reference(gold):
    g = gold('agent_glue.npz')
    p = G.mask_action_probs(torch.from_numpy(g['pa_mean']), torch.from_numpy(g['pa_std']), torch.from_numpy(g['pa_mask']))
    assert np.abs(p.numpy() - g['pa_probs']).max() < 1e-13
    assert np.abs(p.sum(1).numpy() - 1).max() < 1e-12
    gen = torch.Generator().manual_seed(0)
    a, idx = G.choose_action(torch.from_numpy(g['pa_mean']), torch.from_numpy(g['pa_std']), torch.from_numpy(g['pa_mask']), gen)
    assert a.shape == (200, 2) and (g['pa_mask'][np.arange(200), idx.numpy()] > 0).all()
    assert set(np.unique(a[:, 1].numpy())) <= {-1.0, 1.0} and a[:, 0].abs().max() <= 1.0 + 1e-12


def test_state_norm_matches_reference_recurrence(gold):
    g = gold('agent_glue.npz')
    lid, tgt = torch.from_numpy(g['sn_lidar']), torch.from_numpy(g['sn_target'])
    # one at a time == the reference's Welford recurrence
    sn = G.BatchedStateNorm()
    outs = []
    for i in range(len(lid)):
        sn.update({'lidar': lid[i:i + 1], 'target': tgt[i:i + 1]})
        outs.append(sn.normalize({'target': tgt[i]})['target'].numpy())
    assert sn.n_state == int(g['sn_n'])
    assert np.abs(sn.mean['lidar'].numpy() - g['sn_mean_lidar']).max() < 1e-10
    assert np.abs(sn.std['target'].numpy() - g['sn_std_target']).max() < 1e-10
    assert np.abs(np.array(outs) - g['sn_norm_target']).max() < 1e-8
    # folded in batches of 37: same statistics (parallel-merge form of the recurrence)
    sb = G.BatchedStateNorm()
    for a in range(0, len(lid), 37):
        sb.update({'lidar': lid[a:a + 37], 'target': tgt[a:a + 37]})
    assert sb.n_state == sn.n_state
    assert np.abs(sb.mean['lidar'].numpy() - g['sn_mean_lidar']).max() < 1e-10
    assert np.abs(sb.std['lidar'].numpy() - g['sn_std_lidar']).max() < 1e-9
    sb.fix_parameters()
    m = sb.mean['target'].clone()
    sb.update({'lidar': lid[:5], 'target': tgt[:5]})
    assert torch.equal(m, sb.mean['target'])
