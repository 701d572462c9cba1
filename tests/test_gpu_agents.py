"""-m gpu: the agent side of BASELINE configs 4 and 5 on the MI355X.

  * the reference-generated vectors of tests/test_agents.py and tests/test_agent_glue.py, evaluated on `cuda`;
  * BatchedRsPlanner fed with rs_word / rs_lengths as the HIP kernels write them;
  * config 4's loop (SAC-style rollout with the transformer policy) at 8 192 scenes and config 5's loop (PPO, a few
    updates) on one GPU, driving the HIP env through the C ABI;
  * the same PPO loop as 2 ranks sharing the one GPU (gloo), gradients all-reduced: ranks stay bit-identical.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_policy_forward_and_updates_match_reference_on_gpu():
    import test_agents as TA
    TA.check_policy_forward('cuda', 2e-4)          # fp32 GEMMs on the GPU re-associate sums
    TA.check_ppo_update('cuda', 1, tol=3e-3)
    TA.check_ppo_update('cuda', 2, tol=3e-3)
    TA.check_sac_update('cuda', tol=3e-3)


def test_agent_glue_reference_vectors_on_gpu(gold):
    import test_agent_glue as TG
    g = gold('agent_glue.npz')
    TG.check_rs_planner(g, 'cuda')
    TG.check_choose_action(g, 'cuda')
    TG.check_state_norm(g, 'cuda')


def make_env(n, seed=5, image=False, levels=('Normal', 'Complex', 'Extrem', 'dlp'), near=True, unique=512, device='cuda:0'):
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    rng = np.random.default_rng(seed)
    src = SceneSource(levels=levels, seed=seed)
    uniq = [src.draw() for _ in range(min(n, unique))]
    if near:                                      # a third of the scenes start within RS range of the slot
        for s in uniq[::3]:
            r, a = rng.uniform(3.0, 9.0), rng.uniform(0, 2 * np.pi)
            s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.4])
    scenes = [uniq[i % len(uniq)] for i in range(n)]
    env = ParkingBatch(n, 128, image=image, device=device)
    for a in range(0, n, 4096):
        env.set_scenes(np.arange(a, min(n, a + 4096)), scenes[a:a + 4096])
    return env, scenes


def test_rs_planner_on_kernel_words():
    """f-4: rs_word / rs_lengths exactly as k_rs_validate writes them -> BatchedRsPlanner.expand == the reference's
    RsPlanner.set_rs_path rule restated per scene in python (parking_agent.py:12-41)."""
    from hope_amd import agent_glue as G
    env, _ = make_env(2048, seed=9)
    g = torch.Generator(device='cuda').manual_seed(1)
    env.reset_obs()
    found_any = 0
    for _ in range(6):
        env.step(torch.rand((env.n, 2), device='cuda', generator=g) * 2 - 1)
        word, lens = env.rs_word.clone(), env.rs_lengths.clone()
        acts, cnt = G.BatchedRsPlanner.expand(word, lens)
        w, l, acts, cnt = word.cpu().numpy(), lens.double().cpu().numpy(), acts.cpu().numpy(), cnt.cpu().numpy()
        for i in np.nonzero(w[:, 6] > 0)[0][:200]:
            exp = []
            for c, x in zip(w[i, :int(w[i, 5])], l[i]):
                steer, v = {1: 1.0, 0: 0.0, 2: -1.0}[int(c)], x / 1.25
                if 1e-3 < abs(v) < 1:
                    exp.append([steer, v])
                elif abs(v) > 1:
                    s = np.sign(v)
                    while abs(v) > 1:
                        exp.append([steer, s]); v -= s
                    if abs(v) > 1e-3:
                        exp.append([steer, v])
            assert cnt[i] == len(exp) and np.allclose(acts[i, :cnt[i]], np.array(exp).reshape(-1, 2), atol=1e-12), i
            found_any += 1
    assert found_any > 50
    env.close()


def test_config4_sac_rollout_8192_scenes():
    """BASELINE config 4 on one GPU's share: 8 192 scenes, transformer policy (909 778-parameter actor incl. the image
    modality), SAC-style rollout with RS replay, replay ring, a few SAC updates."""
    from hope_amd import agents as A, policy as P
    from hope_amd.rollout import SACTrainer
    torch.manual_seed(0)
    env, _ = make_env(8192, image=True)
    ag = A.BatchedSAC(device='cuda', use_img=True, batch_size=4096)
    assert P.count_parameters(ag.actor) == 909778
    tr = SACTrainer(env, ag, horizon=4, update_every=2, seed=3)
    out = [tr.step() for _ in range(10)]
    torch.cuda.synchronize()
    assert tr.updates == 4 and all(np.isfinite(o).all() for o in out if o is not None)
    s = tr.stats()
    assert s['steps'] == 10 and np.isfinite(s['mean_reward']) and s['episodes'] > 0
    assert float(tr.planner.executing.float().mean()) > 0            # some scenes replay a Reeds-Shepp path
    assert ag.state_norm.n_state == 1 + 8192 * 10 + 8191
    env.close()


def test_config5_ppo_loop_one_gpu():
    """BASELINE config 5 on one GPU's share: 16 384 env instances, PPO with the reference-shaped actor/critic,
    horizon 4, two updates (clip, GAE per scene, critic target, fused-bucket all-reduce is a no-op at world 1)."""
    from hope_amd import agents as A
    from hope_amd.rollout import PPOTrainer
    torch.manual_seed(0)
    env, _ = make_env(16384)
    ag = A.BatchedPPO(device='cuda', use_img=False, mini_batch=16384, mini_epoch=2)
    w0 = ag.actor.embed_lidar[0].weight.detach().clone()
    tr = PPOTrainer(env, ag, horizon=4, seed=3)
    out = [tr.step() for _ in range(8)]
    torch.cuda.synchronize()
    assert tr.updates == 2 and np.isfinite(out[3]).all() and np.isfinite(out[7]).all()
    assert float((ag.actor.embed_lidar[0].weight - w0).abs().max()) > 0
    env.close()


def _rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)                       # both ranks share the box's one GPU; gloo carries the collectives
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hope_amd import agents as A
    from hope_amd.dist import shard_range
    from hope_amd.rollout import PPOTrainer
    from test_agents import probe
    torch.manual_seed(0)
    ag = A.BatchedPPO(device='cuda', use_img=False, mini_batch=2048, mini_epoch=1)
    lo, hi = shard_range(2048, rank, world)
    env, _ = make_env(hi - lo, seed=20 + rank)
    tr = PPOTrainer(env, ag, horizon=4, seed=30 + rank)
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    q.put((rank, probe(ag.actor.cpu()).numpy(), probe(ag.critic.cpu()).numpy(), tr.updates, ag.allreduce_bytes))
    env.close()
    dist.destroy_process_group()


def test_ppo_loop_two_ranks_share_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda r: r[0])
    [p.join(120) for p in ps]
    assert res[0][3] == res[1][3] == 1 and res[0][4] > 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_batched_evaluator_on_the_gpu():
    """eval() for 2 048 episodes at once (hope_amd/evaluate.py): one episode per slot, finished slots frozen, per-level table;
    with RS replay a random-init policy parks most cars, as in the rollout loops."""
    from hope_amd import agents as A
    from hope_amd import evaluate as E
    from hope_amd.scene_gen import mixed_arrays
    n = 2048
    from hope_amd import ParkingBatch
    arr = mixed_arrays(n, seed=17, max_obst=128)
    env = ParkingBatch(n, 128)
    env.set_scene_arrays(np.arange(n), *arr[:5])
    torch.manual_seed(0)
    ag = A.BatchedPPO(device='cuda', use_img=False)
    ev = E.BatchedEvaluator(env, ag, post_proc_action=True, seed=3)
    rec = ev.run()
    torch.cuda.synchronize()
    assert rec.shape == (n, 4)
    r = rec.cpu().numpy()
    assert set(np.unique(r[:, 0]).astype(int)) <= {2, 3, 4, 5} and (r[:, 1] >= 1).all() and (r[:, 1] <= 202).all()
    assert (r[r[:, 0] == 5, 1] == 200).all()                                  # OUTTIME fires at t > 200: reset's step + 200 actions
    levels = [('Normal', 'Complex', 'Extrem', 'dlp')[k % 4] for k in range(n)]
    s = E.summarize(rec, levels)
    print('evaluator:', {k: (round(v['success_rate'], 3), round(v['step_num_mean'], 1), round(v['path_length_mean'], 2)) for k, v in s.items()})
    assert s['all']['episodes'] == n and all(s[k]['episodes'] == n // 4 for k in ('Normal', 'Complex', 'Extrem', 'dlp'))
    assert s['all']['success_rate'] > 0.3 and s['all']['path_length_mean'] > 1.0
    # frozen slots: stepping on with everything inactive changes nothing
    pose = env.pose.clone()
    env.step(torch.zeros((n, 2), device='cuda'), active=torch.zeros(n, dtype=torch.uint8, device='cuda'))
    torch.cuda.synchronize()
    assert torch.equal(env.pose, pose)
    env.close()


def _nccl_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    if world == 1:
        os.environ['HOPE_DIST_FORCE'] = '1'                                  # a one-rank group still goes through RCCL
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{rank}'))
    from hope_amd import agents as A
    from hope_amd import dist as D
    from hope_amd import evaluate as E
    from hope_amd.rollout import PPOTrainer
    from test_agents import probe
    torch.manual_seed(0)
    ag = A.BatchedPPO(device=f'cuda:{rank}', use_img=False, mini_batch=2048, mini_epoch=1)
    lo, hi = D.shard_range(4096, rank, world)
    env, _ = make_env(hi - lo, seed=20 + rank, device=f'cuda:{rank}')
    tr = PPOTrainer(env, ag, horizon=4, seed=30 + rank)
    for _ in range(4):
        tr.step()
    rec = E.BatchedEvaluator(env, ag, seed=rank).run(max_steps=30)           # one all-gather of the eval records over RCCL
    torch.cuda.synchronize()
    q.put((rank, probe(ag.actor.cpu()).numpy(), tr.updates, ag.allreduce_bytes, tuple(rec.shape), float(rec[:, 1].sum())))
    env.close()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs: the RCCL path (backend "nccl" over xGMI)')
def test_rccl_gradient_allreduce_and_eval_gather_two_gpus():
    """The two exchange points of a data-parallel run on REAL RCCL: the fused gradient all-reduce keeps two ranks on
    two GPUs bit-identical, the evaluator's all-gather gives both the same table.  Skipped on 1-GPU boxes (the same code runs
    over gloo in tests/test_agents.py / tests/test_evaluate.py and as 2 ranks sharing one GPU above)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    ps = [ctx.Process(target=_nccl_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda r: r[0])
    [p.join(120) for p in ps]
    assert res[0][2] == res[1][2] == 1 and res[0][3] > 0
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][4] == res[1][4] == (4096, 4) and res[0][5] == res[1][5]


def test_rccl_single_rank_group_runs_the_exchange_points():
    """backend "nccl" (= RCCL on ROCm) on the ONE GPU of this box: a process group of a single rank initialises RCCL on the device
    and sends the fused gradient all-reduce of a PPO update and the evaluator's all-gather through it -- the library, its device
    kernels and the HSA_ENABLE_IPC_MODE_LEGACY=0 environment are exercised even where a second GPU is missing (the two-rank
    form of the same function is the test above)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    pr = ctx.Process(target=_nccl_rank, args=(0, 1, port, q))
    pr.start()
    res = q.get(timeout=900)
    pr.join(120)
    assert res[0] == 0 and res[2] == 1 and res[3] > 0            # one update, gradient bytes went through the all-reduce
    assert res[4] == (4096, 4) and np.isfinite(res[1]).all()


def test_rollout_with_deferred_rs_join_takes_the_same_actions():
    """with two completion points per step (HOPE_DEFER_RS) the policy forward is enqueued before the
    Reeds-Shepp outputs are joined and the planner override reads them after ParkingBatch.wait_rs: the actions, log-probs,
    rewards and done flags of the rollout must be bit-identical to the joined form, planner replays included."""
    from hope_amd import agents as A
    from hope_amd.rollout import HopeRollout
    os.environ['HOPE_SPLIT_MIN'] = '1'                 # the two-launch form of the step (the bit only acts there) at this size
    try:
        runs = []
        for defer in (False, True):
            env, _ = make_env(4096, seed=21)
            torch.manual_seed(3)
            agent = A.BatchedSAC(device='cuda', batch_size=1024)
            ro = HopeRollout(env, agent, horizon=16, use_mask=False, seed=4, defer_rs=defer)
            assert ro.defer_rs == defer
            for _ in range(16):
                ro.collect_step()
            torch.cuda.synchronize()
            obs, action, reward, done, log_prob = ro.ring.ordered()
            runs.append((action.clone(), reward.clone(), done.clone(), log_prob.clone(), ro.planner.executing.clone(),
                         {k: v.clone() for k, v in obs.items()}))
            env.close()
        a, b = runs
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y)
        for k in a[5]:
            assert torch.equal(a[5][k], b[5][k]), k
        assert bool(a[4].any())                        # some scenes were replaying a Reeds-Shepp path
    finally:
        del os.environ['HOPE_SPLIT_MIN']


def test_bench_as_eight_ranks_of_8192_scenes_sharing_the_gpu():
    """the multi-GPU run is the driver's to launch; what can be exercised on a 1-GPU box is everything but RCCL
    itself -- `bench.py --gpus 8` under torch.distributed.run with the launch line the driver uses, BASELINE config 4's exact shape
    (65 536 scenes strong-scaled = 8 192 per rank), all ranks on cuda:0 over gloo (HOPE_BENCH_SHARE_GPU=1).  The line must account
    for every rank (ranks, the all-reduce of ones, per-rank step times, the host-core share each rank pinned itself to)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HOPE_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 36500 + os.getpid() % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '8', '--scaling', 'strong', '--scenes', '65536',
           '--steps', '10', '--warmup', '3', '--preroll', '20', '--no-cpu-baseline', '--witness', '0', '--repeat-passes', '0']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.strip().split('\n') if ln.startswith('{')][-1]
    d = json.loads(line)
    print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'ranks', 'rccl_check', 'rank_ms_per_step', 'host')}))
    assert d['n_gpus'] == 8 and d['ranks'] == 8 and d['scaling'] == 'strong'
    assert d['config']['scenes_per_gpu'] == 8192
    assert d['rccl_check']['ok'] and d['rccl_check']['allreduce_of_ones'] == 8.0 and d['rccl_check']['backend'] == 'gloo'
    assert len(d['rank_ms_per_step']['all']) == 8 and d['rank_ms_per_step']['max'] >= d['rank_ms_per_step']['min'] > 0
    assert abs(d['value'] - 65536 * 10 / (d['ms_per_step'] * 10 * 1e-3)) < 1e-3 * d['value']
    # every rank pinned itself to its share of the host's cores; the generator's default fan-out is that share
    assert d['host']['pinned'] and d['host']['cpus_this_rank'] <= max(1, d['host']['cpus_node'] // 8 + 1)
    assert d['host']['generator_threads'] == d['host']['cpus_this_rank']


def test_bench_gpus_n_launches_its_own_ranks():
    """VERDICT r4 #3b: `python bench.py --gpus 2` WITHOUT a launcher (the way the driver starts `--gpus 1`) must not stop at an error
    line: it starts its own ranks under torch.distributed.run and rank 0's JSON line passes through.  On the 1-GPU box the two ranks
    share the GPU over gloo (HOPE_BENCH_SHARE_GPU=1).  The line also carries the round-5 fields: the background pool refresh inside
    the timed loop, the steady-state and joined-form figures next to the driver-timed value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HOPE_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--scenes', '8192', '--steps', '24', '--warmup', '4', '--preroll', '40',
           '--refresh-every', '8', '--no-cpu-baseline', '--witness', '0', '--repeat-passes', '1', '--repeat-steps', '40']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().split('\n') if ln.startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    print(json.dumps({k: d.get(k) for k in ('value', 'value_steady', 'value_joined', 'ms_per_step', 'n_gpus', 'ranks', 'pool_refresh')}))
    assert d['n_gpus'] == 2 and d['ranks'] == 2 and d['rccl_check']['ok'] and d['scaling'] == 'weak'
    assert d['config']['scenes_per_gpu'] == 8192
    assert abs(d['value'] - 2 * 8192 * 24 / (d['ms_per_step'] * 24 * 1e-3)) < 1e-3 * d['value']
    assert d['value_steady'] > 0 and d['value_joined'] > 0
    assert d['pool_refresh']['refresher_commits'] > 0 and d['pool_refresh']['every_steps'] == 8
    # without the share hook a 1-GPU node must refuse loudly instead of putting two ranks on one device
    if torch.cuda.device_count() < 2:
        env.pop('HOPE_BENCH_SHARE_GPU')
        r2 = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert r2.returncode != 0 and 'HIP device' in (r2.stderr + r2.stdout)


def test_graph_captured_policy_forward_equals_the_eager_one():
    """the rollout's inference forward replayed as one captured device graph (agents.enable_fast_policy)
    is the same fp32 arithmetic -- identical means for changing inputs, recapture on a new batch shape; the bf16-autocast variant
    stays within bf16's resolution of the fp32 mean (it is reported by bench.py --policy-fast, never the default)."""
    from hope_amd import agents as A
    torch.manual_seed(1)
    ag = A.BatchedPPO(device='cuda', use_img=False)
    ref = A.BatchedPPO(device='cuda', use_img=False)
    ref.actor.load_state_dict(ag.actor.state_dict())
    ag.enable_fast_policy(graph=True, amp=False)
    g = torch.Generator(device='cuda').manual_seed(2)
    for n in (512, 512, 512, 96):
        nobs = {'lidar': torch.rand((n, 120), device='cuda', generator=g), 'target': torch.rand((n, 5), device='cuda', generator=g),
                'action_mask': torch.rand((n, 42), device='cuda', generator=g)}
        a, b = ag.policy_mean(nobs), ref.policy_mean(nobs)
        assert torch.equal(a, b), n
    assert ag._fast['captures'] == 2 and ag._fast['replays'] == 4
    ag.enable_fast_policy(graph=True, amp=True)
    a = ag.policy_mean(nobs)
    assert float((a - b).abs().max()) < 0.05 and not torch.equal(a, b)


def test_graph_captured_policy_follows_the_weights():
    """the captured graph bakes in parameter ADDRESSES.  In-place updates (an optimizer step, load_state_dict's copy)
    are seen by the replay; weights that are re-materialised (load_state_dict(assign=True)) change the signature and recapture
    instead of replaying against the old storage; a train-mode actor is captured with inference semantics and handed back in
    train mode; disable_fast_policy drops the graph."""
    from hope_amd import agents as A
    torch.manual_seed(3)
    ag = A.BatchedPPO(device='cuda', use_img=False)
    ref = A.BatchedPPO(device='cuda', use_img=False)
    ref.actor.load_state_dict(ag.actor.state_dict())
    g = torch.Generator(device='cuda').manual_seed(4)
    nobs = {'lidar': torch.rand((64, 120), device='cuda', generator=g), 'target': torch.rand((64, 5), device='cuda', generator=g),
            'action_mask': torch.rand((64, 42), device='cuda', generator=g)}
    ag.actor.train()
    ref.actor.eval()
    ag.enable_fast_policy(graph=True, amp=False)
    assert torch.equal(ag.policy_mean(nobs), ref.policy_mean(nobs)) and ag._fast['captures'] == 1
    assert ag.actor.training                                   # handed back as it was
    # in-place change of the weights: same storage, same graph, new values
    with torch.no_grad():
        for p_, q_ in zip(ag.actor.parameters(), ref.actor.parameters()):
            p_.mul_(1.01); q_.mul_(1.01)
    assert torch.equal(ag.policy_mean(nobs), ref.policy_mean(nobs)) and ag._fast['captures'] == 1
    # re-materialised weights: new storage -> recapture
    sd = {k: (v * 0.5).clone() for k, v in ag.actor.state_dict().items()}
    ag.actor.load_state_dict(sd, assign=True)
    ref.actor.load_state_dict(sd)
    assert torch.equal(ag.policy_mean(nobs), ref.policy_mean(nobs)) and ag._fast['captures'] == 2
    ag.disable_fast_policy()
    assert torch.equal(ag.policy_mean(nobs), ref.policy_mean(nobs)) and ag._fast is None
