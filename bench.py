#!/usr/bin/env python3
"""bench.py -- parallel env-steps/s of the HOPE parking-env step hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL).  W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize,
MAX over ranks, rank 0 prints ONE JSON line.

A "step" = one pass of the full hot path (CarParkingWrapper.step: kinematics + collision sub-steps,
lidar, action mask, reward, Reeds-Shepp feasibility search) over a batch of `--scenes` synthetic
scenes PER GPU (weak scaling; default 65 536 = BASELINE.json's "64k scenes"), including the
episode turnover the reference's loop performs: `reset` of finished episodes -- on a NEW map drawn from a
device-resident pool of generated scenes (HOPE_AUTO_REDRAW; `--same-map`: on the same map) -- and its action-less
observation step, fused into the step kernel (HOPE_AUTO_RESET).  Inputs (scene tiles, the scene pool, actions) are
resident in HBM when the timed region starts.  Independent scenes shard over ranks with no data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--scenes', type=int, default=65536, help='scenes per GPU (weak scaling) or in total (strong scaling)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: --scenes per GPU (default); strong: --scenes in total, split over the ranks '
                         "(BASELINE config 4: 65 536 scenes over 8 GPUs = 8 192 per GPU)")
    ap.add_argument('--policy', default='none', choices=['none', 'hope'],
                    help="hope: drive the env with the reference-shaped transformer policy (hope_amd.policy) instead of "
                         'random actions; value = env+agent steps/s (BASELINE configs 4 and 5)')
    ap.add_argument('--algo', default='rollout', choices=['rollout', 'sac', 'ppo'],
                    help='with --policy hope: rollout = inference only (config 4), sac = + SAC updates, ppo = PPO '
                         'collect/update loop with RCCL gradient all-reduce (config 5)')
    ap.add_argument('--horizon', type=int, default=8, help='PPO steps per update / SAC ring depth')
    ap.add_argument('--mini-batch', type=int, default=16384, help='PPO mini-batch / SAC batch (transitions per rank)')
    ap.add_argument('--mini-epoch', type=int, default=2, help='PPO epochs per update (reference: 10)')
    ap.add_argument('--same-map', action='store_true',
                    help='finished episodes restart on the SAME map (HOPE_AUTO_RESET alone).  Default: episode turnover on a '
                         'NEW map, as the reference draws a new case every episode -- finished scenes draw from a '
                         'device-resident pool of --pool generated scenes inside the step kernel (HOPE_AUTO_REDRAW)')
    ap.add_argument('--fresh-scenes', action='store_true', help='(the default now; kept for old command lines)')
    ap.add_argument('--pool', type=int, default=8192)
    ap.add_argument('--refresh-every', type=int, default=-1,
                    help='the device-resident pool of generated lots is REPLACED in the background while the steps run, as a rollout does it '
                         '(scene_gen.PoolRefresher: the native generator fills pinned staging on host threads, an asynchronous upload swaps the '
                         'pool in): the step loop polls it every this many steps, inside the timed region.  -1 (default) = at the rate the '
                         'batch consumes the pool: about 0.45 %% of the scenes draw a generated lot per step, so a pool of P lots is used up '
                         'after P / (0.0045 N) steps (28 steps at 65 536 scenes and 8 192 lots; never below 8); 0 = a static pool')
    ap.add_argument('--overlap', default='auto', choices=['auto', 'on', 'off'],
                    help='launch chains of the two tile classes on two streams (auto = on)')
    ap.add_argument('--refresh-threads', type=int, default=2, help='worker threads of the background pool refill (native generator)')
    ap.add_argument('--max-obst', type=int, default=128)
    ap.add_argument('--mix', default='mixed', choices=['mixed', 'dlp', 'normal'])
    ap.add_argument('--stages', default='all', choices=['all', 'norss', 'motion'])
    ap.add_argument('--unique', type=int, default=2048, help='distinct generated scenes (tiled to --scenes)')
    ap.add_argument('--image', action='store_true', help="also render obs['img'] (USE_IMG, configs.py:100) every step")
    ap.add_argument('--policy-amp', action='store_true',
                    help='--policy hope --image: run the image encoders under bf16 autocast + channels_last (stock PyTorch levers; '
                         'changes the policy\'s numerics, reported next to the fp32 default)')
    ap.add_argument('--policy-fast', default='off', choices=['off', 'graph', 'amp', 'graph+amp'],
                    help='--policy hope: stock levers on the inference forward of the rollout -- graph: captured device graph '
                         '(same fp32 results); amp: bf16 autocast on the token mixer and embedding MLPs (changes the policy\'s '
                         'numerics); reported next to the fp32 eager default')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-scenes', type=int, default=2048)
    ap.add_argument('--cpu-steps', type=int, default=12)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--action-bank', type=int, default=256,
                    help='pre-generated U[-1,1]^2 action tensors, cycled (resident in HBM before the timed region).  256 (default) is longer than an '
                         'episode (<= 200 steps): every scene-step of an episode gets its own draw, as SURVEY 8(d) specifies.  Rounds 1-5 cycled 16: each '
                         'scene then repeats a 16-step pattern with a net drift and the stationary population has twice as many vehicles pressed against '
                         'obstacles (tools/mask_active_census.py, profiles/r06_mask_active_census.txt)')
    ap.add_argument('--rs-join', default='deferred', choices=['deferred', 'joined'],
                    help='deferred (HOPE_DEFER_RS): the caller\'s stream is ordered after the observation / reward outputs of a step; its '
                         'Reeds-Shepp outputs are ordered by a hope_env_wait_rs issued BEFORE the next step -- the next step REPLACES them '
                         '(consecutive steps pipeline on the library\'s streams, also for single-class batches cut into sub-chains); the bench '
                         'reads them after its last step only.  joined: every output ordered on the caller\'s stream before the next step is '
                         'enqueued (reported next to `value` as value_joined)')
    ap.add_argument('--witness', type=int, default=1024,
                    help='after the timed region: this many random scene slots of the TIMED configuration are rebuilt in the CPU '
                         'oracle from the state on the device and one more fused step is compared (parity_check); 0 = off')
    ap.add_argument('--repeat-passes', type=int, default=5,
                    help='extra passes of --repeat-steps steps after the timed region (outside the driver-timed K steps): '
                         'their run-to-run spread and the steady-state episode population are reported')
    ap.add_argument('--repeat-steps', type=int, default=200)
    ap.add_argument('--preroll', type=int, default=300,
                    help='untimed steps of SETUP before the W warm-up steps that bring the batch to its stationary episode '
                         'population: every episode first gets a random age t ~ U[1, 200] (fresh episodes all time out together '
                         'at t = 201, and for their first steps fewer of them pass the Reeds-Shepp gate, so K steps timed right '
                         'after a reset ran ~5 % faster than the loop ever runs again), then this many steps run.  The driver-timed '
                         '`value` is then the steady-state number (`repeat` cross-checks it).  0 = time fresh episodes')
    return ap.parse_args()


def make_scenes(n_unique, mix, rng):
    from hope_amd import scenes as S
    pool = S.DlpScenePool()
    out = []
    levels = {'mixed': ['Normal', 'Complex', 'Extrem', 'dlp'], 'dlp': ['dlp'], 'normal': ['Normal', 'Complex', 'Extrem']}[mix]
    gen = getattr(S, 'generate_scene', None)
    for k in range(n_unique):
        lv = levels[k % len(levels)]
        if lv == 'dlp' or gen is None:
            out.append(pool.sample(rng=rng))
        else:
            out.append(gen(lv, rng))
    return out


def main():
    args = parse()
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's line does
        # (one process per GPU over RCCL; rank 0 of that run prints the JSON line, which passes through)
        import socket
        import subprocess
        if os.environ.get('HOPE_BENCH_SHARE_GPU') != '1' and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f'--gpus {args.gpus}: this node shows {torch.cuda.device_count()} HIP device(s) (HOPE_BENCH_SHARE_GPU=1 runs the '
                             'ranks on one GPU over gloo: control-flow test only)')
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree')
    dist = None
    # test hook (1-GPU boxes): HOPE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the
    # barrier / MAX-reduce, so the multi-rank control flow can be exercised without N GPUs
    share = os.environ.get('HOPE_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    from hope_amd.dist import pin_rank_to_cores, local_world_size
    host_cpus = pin_rank_to_cores(int(os.environ.get('LOCAL_RANK', 0)), local_world_size())    # this rank's share of the host cores
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the product has no CPU path')
    dev = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(dev)

    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import pack_scenes

    N = args.scenes
    if args.scaling == 'strong':
        from hope_amd.dist import shard_range
        lo, hi = shard_range(args.scenes, rank, world)
        N = hi - lo
    from hope_amd.scene_gen import generate_arrays, mixed_arrays
    levels = {'mixed': ('Normal', 'Complex', 'Extrem', 'dlp'), 'dlp': ('dlp',), 'normal': ('Normal', 'Complex', 'Extrem')}[args.mix]
    n_uniq = min(N, args.unique)
    # initial maps: native multi-threaded generator for Normal / Complex / Extrem (hope_scenegen_generate), host DLP sampler
    start, dest, bbox, verts, nob, nvert = mixed_arrays(n_uniq, levels=levels, seed=args.seed + rank, max_obst=args.max_obst)
    dlp_slot = np.array([levels[k % len(levels)] == 'dlp' for k in range(n_uniq)])
    reps = (N + n_uniq - 1) // n_uniq
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:N]  # noqa: E731
    stages = {'all': L.STAGE_ALL, 'norss': L.STAGE_MOTION | L.STAGE_OBS | L.STAGE_REWARD,
              'motion': L.STAGE_MOTION | L.STAGE_REWARD}[args.stages]

    overlap = {'auto': None, 'on': True, 'off': False}[args.overlap]
    env = ParkingBatch(N, args.max_obst, device=str(dev), obs_dtype=torch.float32, action_dtype=torch.float32,
                       profile=True, image=args.image, overlap=overlap)
    chunk = 8192
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        sl = slice(a, b)
        env.set_scene_arrays(np.arange(a, b), tile(start)[sl], tile(dest)[sl], tile(bbox)[sl], tile(verts)[sl],
                             tile(nob)[sl])
    if dlp_slot.any():                                       # Dragon-Lake slots stay Dragon-Lake slots whatever their first map's size
        env.set_draw_class(np.nonzero(tile(dlp_slot))[0], 1)
    n_obst_all = tile(nob)
    edges = 4.0 * n_obst_all
    # algorithmic bytes per scene-step = 808 + 16*E (reads 64 + 16E, writes 744)
    bytes_per_launch = float(np.sum(808.0 + 16.0 * edges))
    if args.image:
        # roofline kernel = k_bev_image: reads pose/constants 64 + obstacle tile 16 E, writes the 3 x 64 x 64 uint8 image
        bytes_per_launch = float(np.sum(64.0 + 16.0 * edges + 3.0 * 64 * 64))

    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + rank)
    act_bank = [torch.rand((N, 2), generator=g, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(max(1, args.action_bank))]

    defer = args.rs_join == 'deferred'

    bench_refresher = None

    def one_step(i, defer_=None):
        # episode turnover is fused into the step (HOPE_AUTO_RESET = step + restart(done) + reset_obs(active=done))
        env.step(act_bank[i % len(act_bank)], stages=stages, auto_reset=True, defer_rs=defer if defer_ is None else defer_)

    fresh = not args.same_map
    gen_rate = None
    gl = []
    if fresh:
        # New maps at episode turnover, drawn inside the step kernel (HOPE_AUTO_REDRAW): generated lots from a device-resident pool
        # (native generator, refilled by the host whenever it likes) and the 248 Dragon-Lake cases drawn on the device with a
        # fresh start candidate / jitter / flips / obstacle cull per episode (ParkingMapDLP.reset).
        gl = [lv for lv in levels if lv != 'dlp']
        if gl:
            tg = time.perf_counter()
            parts = [generate_arrays(lv, args.pool // len(gl), seed=(args.seed + rank) * 7919 + 17 + j, max_obst=args.max_obst) for j, lv in enumerate(gl)]
            gen_rate = sum(len(p_[4]) for p_ in parts) / (time.perf_counter() - tg)
            env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
        if 'dlp' in levels:
            env.set_dlp_cases()
        env.set_redraw_seed(args.seed * 7919 + 1)

        bench_refresher = None
        if args.refresh_every < 0:
            args.refresh_every = max(8, int(args.pool / (0.0045 * N)))
        if gl and args.refresh_every > 0 and args.policy == 'none':
            from hope_amd.scene_gen import PoolRefresher
            bench_refresher = PoolRefresher(env, args.pool, levels=gl, seed=(args.seed + rank) * 31 + 5, relaxed=True, threads=args.refresh_threads)

        def one_step(i, defer_=None):  # noqa: F811
            # the new map is drawn inside the step kernel (HOPE_AUTO_REDRAW = step + redraw(done) + reset_obs(active=done))
            env.step(act_bank[i % len(act_bank)], stages=stages, auto_reset=True, fresh=True, defer_rs=defer if defer_ is None else defer_)
            if bench_refresher is not None and i % args.refresh_every == 0:
                bench_refresher.poll()                        # never blocks: commits a finished batch of new lots, starts the next fill

    trainer = None
    if args.policy == 'hope':
        # BASELINE configs 4 / 5: the transformer policy (stock PyTorch-ROCm) drives the env; gradients of the update
        # are all-reduced in one fused bucket (hope_amd.dist) -- RCCL over xGMI when world > 1
        from hope_amd import agents as A
        from hope_amd.rollout import PPOTrainer, SACTrainer
        torch.manual_seed(args.seed)                                  # identical initial weights on every rank
        refresher = None
        if fresh and gl:                                              # the pool of generated lots is replaced in the background
            from hope_amd.scene_gen import PoolRefresher
            refresher = PoolRefresher(env, args.pool, levels=gl, seed=(args.seed + rank) * 31 + 5, relaxed=True)
        if args.algo == 'ppo':
            agent = A.BatchedPPO(device=dev, use_img=args.image, mini_batch=args.mini_batch, mini_epoch=args.mini_epoch)
            trainer = PPOTrainer(env, agent, horizon=args.horizon, seed=args.seed + rank, fresh_scenes=fresh, pool_refresher=refresher,
                                 defer_rs=defer)
        else:
            agent = A.BatchedSAC(device=dev, use_img=args.image, batch_size=args.mini_batch)
            trainer = SACTrainer(env, agent, horizon=args.horizon, seed=args.seed + rank, learn=args.algo == 'sac',
                                 fresh_scenes=fresh, pool_refresher=refresher, defer_rs=defer)
        if args.policy_fast != 'off':
            agent.enable_fast_policy(graph='graph' in args.policy_fast, amp='amp' in args.policy_fast)
        if args.policy_amp:
            from hope_amd.policy import set_img_amp
            for net in (getattr(agent, 'actor', None), getattr(agent, 'critic', None), getattr(agent, 'critic_target', None)):
                if net is not None:
                    set_img_amp(net, True)
        one_step = lambda i, defer_=None: trainer.step()  # noqa: E731

    # HIP events bracket the launches of the kernel the roofline is stated for, live in the timed region; timing EVERY
    # launch costs ~4 % of the step in launch latency (16 event records per step), so the other kernels are timed in a
    # short separate pass after it (ms_per_bench_step_by_kernel)
    dom = 'k_bev_image' if args.image else 'k_env_step'
    env.profile_kernels([dom])
    if trainer is None:
        env.reset_obs(stages=stages)
        if args.preroll > 0:
            # setup, not measurement: the stationary episode population (mixed ages, cars spread along their episodes)
            env.upload_state(t=np.random.default_rng(args.seed + 99 + rank).integers(1, 201, N).astype(np.int32))
            for i in range(args.preroll):
                one_step(i)
            torch.cuda.synchronize(dev)
    # the maps resident NOW (the pre-roll redrew finished episodes on the device): SURVEY.md §8(d)'s E per scene from them.  (Before the
    # warm-up steps, not between them and the timed region: the download idles the GPU for milliseconds, and a 20-step window that
    # starts on a GPU that has just idled ran 8 % slow.)
    edges = 4.0 * env.n_obst_now()
    bytes_per_launch = float(np.sum(808.0 + 16.0 * edges))
    if args.image:
        bytes_per_launch = float(np.sum(64.0 + 16.0 * edges + 3.0 * 64 * 64))
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize(dev)
    commits0 = bench_refresher.commits if bench_refresher is not None else 0
    env.kernel_union_ms(reset=True)
    env.kernel_ms(reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    commits_timed = (bench_refresher.commits - commits0) if bench_refresher is not None else 0
    n_break = min(args.steps, 10)
    dom_union = env.kernel_union_ms(reset=True)[dom]
    dom_stats = env.kernel_ms(reset=True)[dom]
    env.profile_kernels(None)
    for i in range(n_break):
        one_step(args.warmup + args.steps + i)
    torch.cuda.synchronize(dev)
    env.kernel_union_ms(reset=True)
    kstats = env.kernel_ms(reset=True)
    done_frac = float(env.done.float().mean().item())
    rs_found = float((env.rs_word[:, 6] > 0).float().mean().item())
    # ---- outside the driver-timed K steps: run-to-run spread over longer passes (SURVEY.md §8(d) asks for >= 200 timed steps,
    # median of 5; the driver's command fixes K), at the episode population those passes converge to (OUTTIME needs t > 200)
    repeat = None
    if args.repeat_passes > 0 and trainer is None:
        env.profile_kernels([dom])                           # the configuration of the timed region
        ms = []
        dsum = 0.0
        for r in range(args.repeat_passes):
            torch.cuda.synchronize(dev)
            tr = time.perf_counter()
            for i in range(args.repeat_steps):
                one_step(i)
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - tr) / args.repeat_steps * 1e3)
            dsum += float(env.done.float().mean().item())
            env.kernel_union_ms(reset=True)
            env.kernel_ms(reset=True)
        # the JOINED form of the same step (what a caller with the Reeds-Shepp planner in its loop gets: every output ordered on
        # its stream before the next step is enqueued), same run, same population
        torch.cuda.synchronize(dev)
        tj = time.perf_counter()
        for i in range(args.repeat_steps):
            one_step(i, False)
        torch.cuda.synchronize(dev)
        joined_ms = (time.perf_counter() - tj) / args.repeat_steps * 1e3
        env.kernel_union_ms(reset=True)
        env.kernel_ms(reset=True)
        sm = sorted(ms)
        repeat = {'passes': args.repeat_passes, 'steps_per_pass': args.repeat_steps, 'ms_per_step': ms, 'min': sm[0], 'joined_ms_per_step': joined_ms,
                  'median': sm[len(sm) // 2], 'max': sm[-1], 'spread': (sm[-1] - sm[0]) / sm[len(sm) // 2],
                  'env_steps_per_s_median': N / (sm[len(sm) // 2] * 1e-3), 'done_frac_at_pass_ends': dsum / args.repeat_passes,
                  'note': 'this rank, after the driver-timed region; not part of `value`'}
    if bench_refresher is not None:
        bench_refresher.close()
    # ---- in-run correctness witness of exactly this configuration (after all timing)
    parity = None
    if args.witness > 0 and trainer is None and rank == 0:
        parity = parity_witness(env, stages, fresh, act_bank[3], args.witness, args.seed + 1234)
    rank_ms = [elapsed / args.steps * 1e3]
    rccl = None
    if dist is not None:
        cdev = 'cpu' if share else dev
        mine_ms = elapsed / args.steps * 1e3
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # who took part: an all-reduce of ones must count every rank, each on its own device (over RCCL unless the test hook
        # HOPE_BENCH_SHARE_GPU put all ranks on one GPU with gloo); per-rank step times by all-gather
        ones = torch.ones(1, device=cdev, dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        allms = [torch.zeros(1, device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allms, torch.tensor([mine_ms], device=cdev, dtype=torch.float64))
        rank_ms = [float(x.item()) for x in allms]
        devs = [None] * world
        dist.all_gather_object(devs, f'{os.uname().nodename}:cuda:{local_rank}')
        rccl = {'backend': dist.get_backend(), 'allreduce_of_ones': float(ones.item()), 'ok': float(ones.item()) == float(world),
                'devices': devs, 'distinct_devices': len(set(devs))}

    result = None
    if rank == 0:
        total_scenes = N * world if args.scaling == 'weak' else args.scenes
        value = total_scenes * args.steps / elapsed
        # HIP events recorded by the library on the launch stream: around every launch of the roofline kernel inside the
        # timed region (dom_stats), around every launch of every kernel in the short pass after it (kstats).
        # `kernel_ms` is the roofline kernel's AVERAGE LAUNCH duration over the timed region (directly comparable with
        # rocprofv3's AverageNs: k_env_step is launched per tile class -- from 16 384 scenes on as two launches each, the
        # motion half and the observation half (template instantiations PART 1 / 2, two rows in rocprof's table) -- i.e. 4
        # launches per bench step at the default size; the average is over all of them);
        # `algorithmic bytes per launch` is averaged over the same launches, so achieved = bytes/launch / kernel_ms.
        per_step = {k: v[0] / max(n_break, 1) for k, v in kstats.items()}
        # The roofline is stated for the kernel that moves the algorithmic bytes of §8(d): k_env_step (k_rs_validate
        # takes about the same time but only re-reads obstacle tiles); with --image it is k_bev_image, which is then
        # also the largest by time.  `largest_by_time` is reported next to it.
        largest = max(per_step, key=per_step.get) if per_step else None
        dom_total_ms, dom_launches = dom_stats
        dom_ms = dom_total_ms / max(dom_launches, 1)
        bytes_avg_launch = bytes_per_launch * args.steps / max(dom_launches, 1)
        achieved = bytes_avg_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # Memory-side traffic and instruction counts are NOT measured in this run: they come from the PMC passes committed
        # under profiles/ (tools/collect_profiles.sh: separate rocprofv3 --pmc passes of this same command) and are
        # labelled as such.  They describe the default workload only.
        default_workload = (N == 65536 and args.mix == 'mixed' and args.stages == 'all' and trainer is None and fresh)
        traffic = traffic_source = None
        pmc = next((q for q in (os.path.join(ROOT, 'profiles', f'{r}_pmc_traffic{"_image" if args.image else ""}.json') for r in ('r06', 'r05', 'r04'))
                    if os.path.exists(q)), '')
        if os.path.exists(pmc) and default_workload:
            try:
                traffic = json.load(open(pmc))['kernels'][dom]['hbm_bytes']
                traffic_source = 'static: ' + os.path.relpath(pmc, ROOT) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, per launch)'
            except Exception:
                traffic = None
        # Instruction-issue roofline (every kernel here is VALU-issue / latency bound, none is HBM bound), MIX-WEIGHTED: the issue
        # rate of a wave64 VALU instruction depends on its class -- measured on this GPU type by tools/valu_rate.py
        # (profiles/r03_valu_rates.json): float64 add / mul / fma ~4.4e11 wave-instructions/s chip-wide, float64 rcp / sqrt
        # ~1.5e11, plain 32-bit ALU ops ~7.7e11 -- and the per-kernel instruction counts by class come from the committed
        # rocprofv3 --pmc pass of this command (SQ_INSTS_VALU, SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64).  floor = sum over classes
        # of count / rate; frac = floor / measured time.  Counts are static (labelled), the step time is this run's.
        valu = None
        sq = next((q for q in (os.path.join(ROOT, 'profiles', f'{r}_sq_counters.json') for r in ('r06', 'r05', 'r04')) if os.path.exists(q)), '')
        vr = os.path.join(ROOT, 'profiles', 'r03_valu_rates.json')
        if os.path.exists(sq) and os.path.exists(vr) and default_workload and not args.image:
            try:
                sqd = json.load(open(sq))
                ceil = json.load(open(vr))['ceilings']
                r64, rtr, r32 = ceil['f64_arith_wave_insts_per_s'], ceil['f64_trans_wave_insts_per_s'], ceil['other_valu_wave_insts_per_s']
                per_kernel, floor_s, insts = {}, 0.0, 0.0
                for kname, c in sqd['kernels'].items():
                    v = c.get('SQ_INSTS_VALU', 0.0)
                    f64 = c.get('SQ_INSTS_VALU_ADD_F64', 0.0) + c.get('SQ_INSTS_VALU_MUL_F64', 0.0) + c.get('SQ_INSTS_VALU_FMA_F64', 0.0)
                    tr = c.get('SQ_INSTS_VALU_TRANS_F64', 0.0)
                    fl = f64 / r64 + tr / rtr + max(v - f64 - tr, 0.0) / r32
                    nl = c.get('launches_per_bench_step', 1)
                    per_kernel[kname] = {'valu_insts_per_launch': v, 'f64_share': (f64 + tr) / v if v else None, 'launches_per_bench_step': nl,
                                         'floor_ms_per_bench_step': fl * nl * 1e3}
                    floor_s += fl * nl
                    insts += v * nl
                # (k_obs_pair / k_motion_pair -- the small-tile class's observation / motion launch, two scenes per wave -- are launches of the step kernel to the
                # library's event timers: its floor is compared with k_env_step's event time together with k_env_step's own)
                grp = lambda k: 'k_env_step' if k in ('k_obs_pair', 'k_motion_pair') else k  # noqa: E731
                for kname in per_kernel:
                    fl_g = sum(q['floor_ms_per_bench_step'] for k2, q in per_kernel.items() if grp(k2) == grp(kname))
                    t_meas = per_step.get(grp(kname), 0.0)
                    per_kernel[kname]['frac_of_its_own_event_time'] = fl_g / t_meas if t_meas > 0 else None
                valu = {'valu_insts_per_bench_step': insts, 'mix_weighted_floor_ms_per_step': floor_s * 1e3,
                        'frac': floor_s / (elapsed / args.steps), 'rates_wave_insts_per_s': {'f64_arith': r64, 'f64_trans': rtr, 'other': r32},
                        'per_kernel': per_kernel,
                        'source': 'static instruction counts by class: ' + os.path.relpath(sq, ROOT) + ' (rocprofv3 --pmc passes of this command); '
                                  'issue rates: profiles/r03_valu_rates.json (tools/valu_rate.py on the same GPU type); step and kernel times: this run. '
                                  'A single wave issues at most one instruction per ~8-10 cycles, so a kernel needs >= 4 waves per SIMD to reach these rates.'}
            except Exception:
                valu = None
        # the box's own streaming ceiling next to the 8 TB/s vendor peak: device-to-device copy of 1 GiB
        copy_gbps = None
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize(dev)
            tc = time.perf_counter()
            for _ in range(10):
                dst.copy_(src)
            torch.cuda.synchronize(dev)
            copy_gbps = 10 * 2 * src.numel() * 4 / (time.perf_counter() - tc) / 1e9
            del src, dst
        except Exception:
            pass
        result = {
            'metric': 'parallel env steps/sec (full CarParking step incl. obs + RS search)', 'value': value,
            'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'ranks': world, 'rccl_check': rccl,
            'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms), 'all': rank_ms},
            'host': {'cpus_this_rank': host_cpus, 'cpus_node': os.cpu_count(), 'generator_threads': L.load_library().hope_scenegen_default_threads(),
                     'pinned': world > 1 and os.environ.get('HOPE_NO_PIN') != '1'},
            'config': {'workload': f'{N} scenes/GPU, {args.mix} scene mix, stages={args.stages}{"+img" if args.image else ""}, random actions U[-1,1]^2 per scene-step ({args.action_bank} pre-generated draws, cycled), '
                                   'auto-restart of finished episodes', 'scenes_per_gpu': N, 'preroll_steps': args.preroll if trainer is None else 0, 'mean_edges': float(edges.mean()), 'mean_edges_of': 'the maps resident after the pre-roll (hope_env_download_n_obst)',
                       'parallelism': f'scene-sharded x{world}, no data-path collective', 'obs_dtype': 'f32',
                       'overlap_tile_classes': bool(env.overlap),
                       'rs_join': ('deferred: the caller\'s stream is ordered after each step\'s observation / reward / status outputs, its '
                                   'Reeds-Shepp outputs by the next step on the library\'s streams (HOPE_DEFER_RS); all work of the K steps '
                                   'is complete at the closing synchronize') if defer else 'joined',
                       'episode_turnover': (f'new map per episode, drawn inside the step kernel (HOPE_AUTO_REDRAW): generated lots from a device-resident pool of '
                                            f'{args.pool} scenes, Dragon-Lake cases drawn on the device (start candidate, jitter, flips, cull per episode)'
                                            if fresh else 'restart on the same map, fused into the step (HOPE_AUTO_RESET)'),
                       'host_generator_scenes_per_s': gen_rate,
                       'timed_window': f'{args.steps} steps between two synchronisations: `value` includes one pipeline fill and drain (the last '
                                       'step\'s search chain runs out after its observation); value_steady is the same loop over '
                                       f'{args.repeat_steps}-step passes',
                       'done_frac_last': done_frac, 'rs_found_frac_last': rs_found},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': traffic, 'traffic_source': traffic_source, 'kernel': dom,
                         'largest_by_time': largest,
                         'concurrent_launches': ('the launch chains of the two obstacle-tile classes run on two streams: the '
                                                 'k_env_step launches overlap each other and the other class\'s kernels, so '
                                                 'per-launch durations include that sharing; from 16 384 scenes on k_env_step is '
                                                 'two launches per class (motion half, then the observation half on its own stream '
                                                 'next to the Reeds-Shepp kernels); the small-tile class\'s two halves are the '
                                                 'two-scenes-per-wave forms k_motion_pair / k_obs_pair (their own rows in rocprof\'s table; the library\'s event '
                                                 'timers and this average count them as launches of the step kernel: profiles/*_kernel_stats_top.txt '
                                                 'ends with the combined row)') if env.overlap else None,
                         # the same kernel per step CALL: all its launches' bytes over the time during which at least one of them
                         # ran (the union of the launch intervals) -- what the kernel sustains while its launches overlap
                         'per_call': {'kernel_ms': dom_union[0] / max(dom_union[1], 1), 'calls': dom_union[1],
                                      'achieved': bytes_per_launch * args.steps / max(dom_union[1], 1) / (dom_union[0] / max(dom_union[1], 1) * 1e-3) / 1e9 if dom_union[0] > 0 else None,
                                      'frac': bytes_per_launch * args.steps / max(dom_union[1], 1) / (dom_union[0] / max(dom_union[1], 1) * 1e-3) / 1e9 / 8000.0 if dom_union[0] > 0 else None},
                         'whole_step': {'achieved': bytes_per_launch / (elapsed / args.steps) / 1e9, 'unit': 'GB/s',
                                        'frac': bytes_per_launch / (elapsed / args.steps) / 1e9 / 8000.0,
                                        'note': 'algorithmic bytes of one bench step / driver-timed step'},
                         'valu_issue': valu,
                         'measured_copy_GBps': copy_gbps,
                         'kernel_ms': dom_ms, 'kernel_launches': dom_launches,
                         'algorithmic_bytes_per_launch': bytes_avg_launch,
                         'algorithmic_bytes_per_bench_step': bytes_per_launch,
                         'ms_per_bench_step_by_kernel': per_step, 'breakdown_steps': n_break},
        }
        result['repeat'] = repeat
        if repeat is not None:
            # next to the driver-timed `value`: the steady-state figure (median of the longer passes) and the joined form, this rank x ranks
            result['value_steady'] = total_scenes / (repeat['median'] * 1e-3)
            result['value_joined'] = total_scenes / (repeat['joined_ms_per_step'] * 1e-3)
            result['value_vs_steady'] = value / result['value_steady']
        result['pool_refresh'] = ({'every_steps': args.refresh_every, 'commits_in_timed_region': commits_timed, 'refresher_commits': bench_refresher.commits,
                                   'generator_seconds': bench_refresher.gen_seconds, 'lots_per_commit': args.pool,
                                   'host_generator_lots_per_s': (bench_refresher.commits + 1) * args.pool / max(bench_refresher.gen_seconds, 1e-9)}
                                  if bench_refresher is not None else None)
        result['parity_check'] = parity
        result['queue_check'] = env.queue_check() if env.overlap else None
        if trainer is not None:
            from hope_amd.policy import count_parameters
            result['metric'] = 'env+agent steps/sec (full CarParking step + HOPE transformer policy' + \
                {'rollout': ', inference only)', 'sac': ' + SAC updates)', 'ppo': ' + PPO updates)'}[args.algo]
            result['config']['workload'] = result['config']['workload'].replace(f'random actions U[-1,1]^2 per scene-step ({args.action_bank} pre-generated draws, cycled)', 'actions from the policy / RS replay')
            result['config'].update({'policy': 'HopeNet (MultiObsEmbedding shape), random init', 'algo': args.algo,
                                     'actor_params': count_parameters(trainer.agent.actor), 'use_img': bool(args.image), 'policy_fast': args.policy_fast,
                                     'policy_graph_replays': (getattr(trainer.agent, '_fast', None) or {}).get('replays'),
                                     'horizon': args.horizon, 'mini_batch': args.mini_batch,
                                     'mini_epoch': args.mini_epoch if args.algo == 'ppo' else None,
                                     'updates_in_run': trainer.updates, 'allreduce_bytes_total': trainer.agent.allreduce_bytes,
                                     'gradient_allreduce': __import__('hope_amd.dist', fromlist=['x']).allreduce_stats(),
                                     'pool_refreshes_in_run': (trainer.refresher.commits if trainer.refresher is not None else 0),
                                     'rollout': trainer.stats()})
        if not args.no_cpu_baseline and world == 1 and trainer is None:
            result['cpu_baseline'] = cpu_baseline(args, (start, dest, bbox, verts, nob, nvert), stages)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def parity_witness(env, stages, fresh, act, k, seed):
    """In-run correctness witness of the TIMED configuration (float32 observations and actions, the launch form the library
    picked for this batch size, HOPE_AUTO_RESET [| HOPE_AUTO_REDRAW]): `k` random scene slots are rebuilt in the CPU oracle
    from what is on the device after the timed region -- pose / t / accumulator (hope_env_download_state) and the map each
    slot holds now (hope_env_download_scenes: after device-side draws the maps exist there only) -- then ONE more fused step
    runs on the GPU over the whole batch and in the oracle on those slots, with the same actions.  Scenes whose episode ends
    in that step are followed through the turnover: the oracle resets them on the map the kernel drew.  The oracle is the
    checker here, nothing else (it runs after the timed region).  The GPU computes in float64 and stores float32, so every
    float32 output must equal float32(oracle)."""
    import torch
    from hope_amd import _lib as L
    from hope_amd import tables as T
    from oracle import oracle as O
    N = env.n
    k = min(k, N)
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(N, k, replace=False))
    with_rs = bool(stages & L.STAGE_RS)
    with_obs = bool(stages & L.STAGE_OBS)
    t = T.all_tables()
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    torch.cuda.synchronize(env.device)

    def maps(slots):
        """(start, dest, bbox, verts, n_obst, nvert) of the map each slot holds; a triangle repeats its last vertex"""
        start, dest, bbox, verts, nob = env.download_scenes(slots)
        nvert = np.where((verts[:, :, 3] == verts[:, :, 2]).all(axis=2), 3, 4).astype(np.int32)
        return start, dest, bbox, verts, nob, nvert

    pose, tt, acc = env.download_state()
    start, dest, bbox, verts, nob, nvert = maps(ids)
    orc = O.BatchOracle(k, env.max_obst, omp=True)
    orc.set_scenes(np.arange(k), start, dest, bbox, verts, nvert, nob)
    orc.pose[:], orc.t[:], orc.accum[:] = pose[ids], tt[ids], acc[ids]
    env.step(act, stages=stages, auto_reset=True, fresh=fresh)
    torch.cuda.synchronize(env.device)
    a64 = act[torch.from_numpy(ids).to(act.device)].double().cpu().numpy()
    o = {kk: vv.copy() for kk, vv in orc.step(a64, with_rs=with_rs).items()}
    sel = torch.from_numpy(ids).to(env.device)
    g = {kk: getattr(env, kk)[sel].cpu().numpy() for kk in ('status', 'reward', 'reward_info', 'lidar', 'action_mask', 'target',
                                                            'rs_word', 'rs_lengths', 'done')}
    done = o['status'] != 1
    res = {'scenes': int(k), 'turnovers': int(done.sum()), 'status_mismatch': int((g['status'] != o['status']).sum()),
           'done_mismatch': int((g['done'].astype(bool) != done).sum())}
    err, f32bad = 0.0, 0

    def cont(name, got, want):
        nonlocal err, f32bad
        if got.size:
            err = max(err, float(np.abs(got.astype(np.float64) - want).max()))
            f32bad += int((got != want.astype(got.dtype)).sum())
    cont('reward', g['reward'], o['reward'])
    cont('reward_info', g['reward_info'], o['reward_info'])
    if with_rs:
        res['rs_mismatch'] = int(((g['rs_word'][:, 6] != o['rs_found']) | (g['rs_word'][:, :5] != o['rs_ctypes']).any(axis=1)).sum())
        res['rs_found'] = int(o['rs_found'].sum())
        cont('rs_lengths', g['rs_lengths'], o['rs_lengths'])
    # observation + state: the scenes that go on, then the ones that were turned over (new episode's first observation)
    pose2, tt2, acc2 = env.download_state()
    go = ~done
    state_bad = int((pose2[ids][go] != orc.pose[go]).any(axis=1).sum() + (tt2[ids][go] != orc.t[go]).sum() + (acc2[ids][go] != orc.accum[go]).sum())
    mask_bad = 0
    if with_obs:
        cont('lidar', g['lidar'][go], o['lidar'][go])
        cont('target', g['target'][go], o['target'][go])
        mask_bad += int((g['action_mask'][go] != o['mask'][go].astype(g['action_mask'].dtype)).any(axis=1).sum())
    if done.any():
        dl = np.nonzero(done)[0]
        s2, d2, b2, v2, n2, nv2 = maps(ids[dl])                      # the maps the kernel drew (or kept) for the new episodes
        orc2 = O.BatchOracle(len(dl), env.max_obst, omp=True)
        orc2.set_scenes(np.arange(len(dl)), s2, d2, b2, v2, nv2, n2)
        o2 = orc2.reset_obs(with_rs=False)
        state_bad += int((pose2[ids][dl] != orc2.pose).any(axis=1).sum() + (tt2[ids][dl] != orc2.t).sum() + (acc2[ids][dl] != orc2.accum).sum())
        if with_obs:
            cont('lidar', g['lidar'][dl], o2['lidar'])
            cont('target', g['target'][dl], o2['target'])
            mask_bad += int((g['action_mask'][dl] != o2['mask'].astype(g['action_mask'].dtype)).any(axis=1).sum())
    res.update({'mask_mismatch': mask_bad, 'state_mismatch': state_bad, 'max_abs_err': err, 'f32_value_mismatch': f32bad,
                'checker': 'oracle/hope_oracle.c (CPU, float64) on the device state after the timed region + one more fused step; '
                           'float32 outputs compared with float32(oracle) exactly (f32_value_mismatch) and with the float64 value (max_abs_err)'})
    return res


def cpu_baseline(args, uniq, stages):   # uniq: the packed arrays of the bench's unique scenes
    """The CPU oracle (a C port of the reference algorithm) timed on this box's host cores on a bounded
    sample of the SAME workload: first --cpu-scenes scenes, --cpu-steps steps, single thread (and all
    cores via OpenMP as an extra figure).  Reported, non-target."""
    from hope_amd import _lib as L
    from hope_amd import tables as T
    from hope_amd.scenes import pack_scenes
    from oracle import oracle as O
    n = min(args.cpu_scenes, len(uniq[4]))
    start, dest, bbox, verts, nob, nvert = (a[:n] for a in uniq)
    t = T.all_tables()
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    with_rs = bool(stages & L.STAGE_RS)
    rng = np.random.default_rng(args.seed)
    acts = [rng.uniform(-1, 1, (n, 2)) for _ in range(args.cpu_steps)]
    res = {}
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    omp_default = O.lib(True).orc_num_threads()      # what OpenMP picks by itself (OMP_NUM_THREADS / its own CPU count)
    # the container's CPU-time quota (cgroup v2 cpu.max / v1 cfs quota): the GPU boxes of this pool show 256 CPUs and grant 16 CPUs'
    # worth of time -- with one thread per visible CPU the scheduler throttles the team and the rate FALLS (32 threads 108 k env-steps/s,
    # 128 threads 55 k: profiles/r06_cpu_baseline_threads.txt), which round 5 read as a scaling problem of the oracle
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota = None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            quota = q / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read()) if q > 0 else None
        except (OSError, ValueError):
            quota = None
    can_set = hasattr(O.lib(True), 'orc_set_num_threads')
    # all-core figure: one OpenMP thread per CPU this process may run on AND libgomp's own default (on a 2-way SMT host the
    # default is the core count, and the hardware threads beyond it do not help this float64 code) -- the better one is reported
    cand = {int(omp_default), int(usable)}
    if quota:
        cand |= {max(1, int(round(quota))), max(1, int(round(2 * quota)))}     # the quota's worth of threads, and twice that (SMT-like slack)
        cand = {c for c in cand if c <= 8 * quota}                             # (far beyond the quota: throttled, slower, and minutes of wall time)
    runs = [(False, 1)] + [(True, nt) for nt in (sorted(cand) if can_set else [int(omp_default)])]
    allcore = {}
    for omp, nt in runs:
        rep = 16 if omp else 1                       # the all-core runs get 16x the scenes to keep the threads busy
        nn = n * rep
        if omp and can_set:
            O.lib(True).orc_set_num_threads(nt)
        orc = O.BatchOracle(nn, args.max_obst, omp=omp, track_traj=False)     # (the trajectory list is a serial Python loop over the scenes)
        tl = lambda x: np.concatenate([x] * rep, axis=0)  # noqa: E731
        acts_r = [tl(a) for a in acts]                # (outside the timed region: with 128 threads the C call takes ~0.1 s per step)
        orc.set_scenes(np.arange(nn), tl(start), tl(dest), tl(bbox), tl(verts), tl(nvert), tl(nob))
        orc.reset_obs(with_rs=with_rs)
        t0 = time.perf_counter()
        for a in acts_r:
            o = orc.step(a, with_rs=with_rs)
            done = o['status'] != 1
            if done.any():
                ids = np.nonzero(done)[0]
                orc.pose[ids] = orc.start[ids]
                orc.t[ids] = 0
                orc.accum[ids] = 0
        dt = time.perf_counter() - t0
        if omp:
            allcore[nt] = nn * args.cpu_steps / dt
        else:
            res[False] = nn * args.cpu_steps / dt
    best = max(allcore, key=allcore.get)
    cores = os.cpu_count() or 1
    return {'value': res[False], 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'first {n} scenes of the bench scene set x {args.cpu_steps} steps, same stages/actions, '
                      'oracle/hope_oracle.c (gcc -O2), 1 thread',
            'allcore_value': allcore[best], 'allcore_threads': best, 'allcore_runs': {str(k): v for k, v in allcore.items()},
            'host_cores': cores, 'host_cpus_usable': usable, 'host_cpu_quota': quota, 'openmp_default_threads': omp_default,
            'allcore_note': 'OpenMP over scenes (16x the sample), timed with the container\'s CPU quota worth of threads and twice that (or, without a '
                            'quota, libgomp\'s default and one thread per usable CPU); allcore_value is the best (allcore_runs has all); '
                            'host_cpu_quota = CPUs\' worth of time the cgroup grants this container'}


if __name__ == '__main__':
    main()
