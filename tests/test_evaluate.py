"""The batched evaluator (hope_amd/evaluate.py <-> src/evaluation/eval_utils.py:16-84) on the CPU stand-in env, and its
statistics gathered over 2 gloo ranks."""
import os

import numpy as np
import pytest
import torch

from hope_amd import agents as A
from hope_amd import evaluate as E


def _scenes(n, seed=3):
    from hope_amd.scenes import SceneSource
    src = SceneSource(levels=('Normal', 'Complex', 'Extrem'), seed=seed)
    sc = [src.draw() for _ in range(n)]
    rng = np.random.default_rng(seed)
    for k in range(0, n, 2):                        # half start close to the slot: found RS paths park them
        s = sc[k]
        s.start = np.array([s.dest[0] + rng.uniform(3, 6) * np.cos(s.dest[2]), s.dest[1] + rng.uniform(3, 6) * np.sin(s.dest[2]), s.dest[2]])
    return sc


def test_evaluator_bookkeeping_matches_a_scalar_replay_of_eval():
    """The batched loop == the reference's per-episode loop replayed scene by scene with the same actions: step count, summed
    reward, path length from the rear-axle displacements, final status; finished slots stay frozen; the first action of every
    episode is the stuck detector's random one (obs['target'] is compared with itself, eval_utils.py:39,46)."""
    from fake_env import OracleEnv
    torch.manual_seed(0)
    scenes = _scenes(10)
    env = OracleEnv(scenes)
    ag = A.BatchedPPO(device='cpu', use_img=False)
    ev = E.BatchedEvaluator(env, ag, post_proc_action=True, seed=5)
    taken = []
    orig = env.step

    def spy(actions, **kw):                          # record what the loop did
        taken.append((actions.clone(), kw['active'].clone()))
        return orig(actions, **kw)
    env.step = spy
    rec = ev.run(max_steps=40, gather=False).numpy()
    assert rec.shape == (10, 4)
    # scalar replay: one fresh oracle env per scene, the recorded actions of its live steps
    for i, sc in enumerate(scenes):
        e1 = OracleEnv([sc])
        e1.reset_obs()
        steps, total, plen, status = 0, 0.0, 0.0, 1
        last = e1.pose[0, :2].clone()
        for act, active in taken:
            if not active[i]:
                break
            e1.step(act[i:i + 1])
            steps += 1
            total += float(e1.reward[0])
            plen += float((e1.pose[0, :2] - last).norm())
            last = e1.pose[0, :2].clone()
            if e1.done[0]:
                status = int(e1.status[0])
                break
        assert rec[i, 1] == steps and int(rec[i, 0]) == status, i
        assert abs(rec[i, 2] - total) < 1e-4 and abs(rec[i, 3] - plen) < 1e-4, i
    # every episode's first action was the random one, clipped to [-1, 1] like the wrapper does
    a0 = taken[0][0]
    assert float(a0.abs().max()) <= 1.0 and float((a0[:, 1].abs() == 1.0).float().mean()) > 0.2     # U(-2.5, 2.5) saturates often
    s = E.summarize(rec, levels=[sc.level for sc in scenes])
    assert s['all']['episodes'] == 10 and set(s) >= {'all', 'Normal', 'Complex', 'Extrem'} - ({'Normal', 'Complex', 'Extrem'} - {sc.level for sc in scenes})
    assert 0.0 <= s['all']['success_rate'] <= 1.0
    arrived = rec[:, 0] == 2
    assert s['all']['success_rate'] == pytest.approx(arrived.mean())
    if arrived.any():
        assert s['all']['success_step_mean'] == pytest.approx(rec[arrived, 1].mean())


def test_summarize_follows_result_txt_rules():
    # status, steps, reward, path: an OUTBOUND episode counts 200 steps in the per-case step_record only (eval_utils.py:73-76);
    # the per-level "step num" is the raw step_num (step_num_level :71, :143); path length only of episodes shorter than 200
    rec = np.array([[2, 30, 5.0, 12.0], [4, 10, -5.0, 4.0], [5, 201, -1.0, 90.0], [2, 50, 5.0, 20.0]], np.float32)
    s = E.summarize(rec, levels=['a', 'a', 'b', 'b'])
    assert s['all']['success_rate'] == 0.5 and s['all']['success_step_mean'] == 40.0
    assert s['all']['step_num_mean'] == (30 + 200 + 201 + 50) / 4              # step_record: the OUTBOUND episode counts 200
    assert s['a']['step_num_mean'] == (30 + 10) / 2 and s['b']['step_num_mean'] == (201 + 50) / 2   # step_num_level: raw
    assert s['a']['path_length_mean'] == 8.0 and s['b']['path_length_mean'] == 20.0


def _eval_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from fake_env import OracleEnv
    from hope_amd.dist import shard_range
    torch.manual_seed(0)
    scenes = _scenes(9)
    lo, hi = shard_range(len(scenes), rank, world)
    ag = A.BatchedPPO(device='cpu', use_img=False)
    ev = E.BatchedEvaluator(OracleEnv(scenes[lo:hi]), ag, seed=7 + rank)
    rec = ev.run(max_steps=25)                       # gathered: every rank holds all 9 records, in rank order
    q.put((rank, rec.numpy(), hi - lo))
    dist.destroy_process_group()


def test_evaluator_gathers_records_over_two_ranks():
    """eval statistics shard over ranks with the scenes; ONE all-gather brings every rank the full table."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_eval_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    [p.join(60) for p in ps]
    assert res[0][1].shape == (9, 4) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] + res[1][2] == 9
    assert (res[0][1][:, 1] >= 1).all() and set(np.unique(res[0][1][:, 0]).astype(int)) <= {1, 2, 3, 4, 5}
