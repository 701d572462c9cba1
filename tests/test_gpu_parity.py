"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): collision/arrival/status flags and action-mask values bit-exact; pose / lidar /
target / reward within a stated tolerance.  Because the kernels and the oracle evaluate sin/cos/atan2/... with the
same deterministic functions (hope_amd/csrc/hope_math.h: IEEE-exact operations only), the float64 observation
mode is required to agree EXACTLY (tolerance 0.0, RS words and lengths included); float32 observation mode
within 2e-5 (the rounding of the stored value).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

TOL64 = 0.0          # exact: every comparison below is `<= TOL64`
TOL32 = 2e-5


def make_pair(n, seed, obs64=True, max_obst=128, level='dlp'):
    from hope_amd import ParkingBatch
    from hope_amd.scenes import DlpScenePool, SceneSource, pack_scenes
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    if level == 'dlp':
        pool = DlpScenePool()
        scenes = [pool.sample(rng=rng) for _ in range(n)]
    else:
        src = SceneSource(levels=('Normal', 'Complex', 'Extrem', 'dlp') if level == 'mixed' else ('Normal', 'Complex', 'Extrem'),
                          seed=seed)
        scenes = [src.draw() for _ in range(n)]
    # half of the scenes start near the destination (tight surroundings, arrival, collisions)
    for k in range(0, n, 2):
        s = scenes[k]
        r, a = rng.uniform(0.0, 6.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    env = ParkingBatch(n, max_obst, obs_dtype=torch.float64 if obs64 else torch.float32, action_dtype=torch.float64)
    assert env.arch.startswith('gfx950'), env.arch
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, max_obst)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, max_obst)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    return env, orc, rng


def compare(env, o, tol, stats, with_rs=False):
    torch.cuda.synchronize()
    st = env.status.cpu().numpy()
    stats['status_mismatch'] += int((st != o['status']).sum())
    m = env.action_mask.cpu().numpy()
    want = o['mask'] if m.dtype == np.float64 else o['mask'].astype(np.float32)      # float32 mode stores the rounded value
    stats['mask_mismatch'] += int((m != want).any(axis=1).sum())
    stats['lidar_err'] = max(stats['lidar_err'], float(np.abs(env.lidar.double().cpu().numpy() - o['lidar']).max()))
    stats['target_err'] = max(stats['target_err'], float(np.abs(env.target.double().cpu().numpy() - o['target']).max()))
    stats['reward_err'] = max(stats['reward_err'], float(np.abs(env.reward.double().cpu().numpy() - o['reward']).max()))
    stats['rinfo_err'] = max(stats['rinfo_err'], float(np.abs(env.reward_info.double().cpu().numpy() - o['reward_info']).max()))
    if with_rs:
        w = env.rs_word.cpu().numpy()
        stats['rs_flag_mismatch'] += int((w[:, 6] != o['rs_found']).sum())
        stats['rs_word_mismatch'] += int((w[:, :5] != o['rs_ctypes']).any(axis=1).sum())
        stats['rs_len_err'] = max(stats['rs_len_err'], float(np.abs(env.rs_lengths.double().cpu().numpy() - o['rs_lengths']).max()))


def new_stats():
    return dict(status_mismatch=0, mask_mismatch=0, lidar_err=0.0, target_err=0.0, reward_err=0.0, rinfo_err=0.0,
                pose_err=0.0, rs_flag_mismatch=0, rs_word_mismatch=0, rs_len_err=0.0)


def rollout(env, orc, rng, steps, tol, with_rs=False, stages=None):
    from hope_amd import _lib as L
    if stages is None:
        stages = L.STAGE_ALL if with_rs else (L.STAGE_MOTION | L.STAGE_OBS | L.STAGE_REWARD)
    stats = new_stats()
    env.reset_obs(stages=stages)
    compare(env, orc.reset_obs(with_rs=with_rs), tol, stats, with_rs)
    seen = set()
    for it in range(steps):
        act = rng.uniform(-1.2, 1.2, (env.n, 2))
        if it % 3 == 0:
            act[:, 1] = np.sign(act[:, 1]) * 1.0          # full speed: more collisions / outbound
        a = torch.from_numpy(act).to(env.device)
        env.step(a, stages=stages)
        o = orc.step(act, with_rs=with_rs)
        compare(env, o, tol, stats, with_rs)
        pose, t, acc = env.download_state()
        stats['pose_err'] = max(stats['pose_err'], float(np.abs(pose - orc.pose).max()))
        assert np.array_equal(t, orc.t.astype(np.int32))
        seen |= set(np.unique(o['status']).tolist())
    stats['seen'] = sorted(seen)
    return stats


def test_library_is_native():
    from hope_amd import load_library
    L = load_library()
    assert L.hope_abi_version() == 8


def test_step_parity_f64_dlp():
    env, orc, rng = make_pair(384, seed=11)
    s = rollout(env, orc, rng, steps=24, tol=TOL64)
    print('parity f64:', s)
    assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0
    assert s['pose_err'] <= TOL64 and s['lidar_err'] <= TOL64 and s['target_err'] <= TOL64
    assert s['reward_err'] <= TOL64 and s['rinfo_err'] <= TOL64
    assert {1, 3}.issubset(set(s['seen'])) or {1, 4}.issubset(set(s['seen']))     # episodes do end
    env.close()


def test_step_parity_with_rs_search():
    """BASELINE config 3 shape (full step incl. Reeds-Shepp feasibility search), flags + words exact."""
    env, orc, rng = make_pair(512, seed=21)
    s = rollout(env, orc, rng, steps=12, tol=TOL64, with_rs=True)
    print('parity f64 + RS:', s)
    assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0
    assert s['rs_flag_mismatch'] == 0 and s['rs_word_mismatch'] == 0 and s['rs_len_err'] <= TOL64
    assert s['pose_err'] <= TOL64 and s['lidar_err'] <= TOL64
    torch.cuda.synchronize()
    env.close()


def test_rs_search_finds_paths_near_goal():
    """scenes started on the slot axis a few metres out: a feasible RS path must be found for some, and the
    (flag, word, lengths) must match the oracle exactly / to 1e-9."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import DlpScenePool, pack_scenes
    from oracle import oracle as O
    n, mo = 256, 128
    pool = DlpScenePool()
    rng = np.random.default_rng(5)
    scenes = []
    for k in range(n):
        s = pool.sample(rng=rng)
        fwd = rng.uniform(3.0, 9.0) * rng.choice([-1, 1])
        s.start = np.array([s.dest[0] + fwd * np.cos(s.dest[2]) + rng.normal() * 0.2,
                            s.dest[1] + fwd * np.sin(s.dest[2]) + rng.normal() * 0.2, s.dest[2] + rng.normal() * 0.1])
        scenes.append(s)
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, mo)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, mo)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    env.reset_obs()
    orc.reset_obs()                      # t = 1: the RS gate needs t > 1
    act = np.zeros((n, 2))
    env.step(torch.from_numpy(act).to(env.device))
    o = orc.step(act)
    torch.cuda.synchronize()
    w = env.rs_word.cpu().numpy()
    assert np.array_equal(w[:, 6], o['rs_found'])
    assert np.array_equal(w[:, :5], o['rs_ctypes'])
    assert np.abs(env.rs_lengths.cpu().numpy() - o['rs_lengths']).max() <= TOL64
    print('rs found', int(o['rs_found'].sum()), 'of', n)
    assert o['rs_found'].sum() >= 10
    env.close()


def test_step_parity_generated_levels_with_rs():
    """Normal / Complex / Extrem scenes from hope_amd.scenes.generate_scene (axis-aligned walls: the
    tolerance-free box tests of the lidar and of is_traj_valid are at their most fragile here) + DLP,
    small and large tile classes in one batch; full step incl. the RS search."""
    env, orc, rng = make_pair(768, seed=31, level='mixed')
    s = rollout(env, orc, rng, steps=16, tol=TOL64, with_rs=True)
    print('parity mixed levels + RS:', s)
    assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0
    assert s['rs_flag_mismatch'] == 0 and s['rs_word_mismatch'] == 0 and s['rs_len_err'] <= TOL64
    assert s['pose_err'] <= TOL64 and s['lidar_err'] <= TOL64 and s['reward_err'] <= TOL64
    env.close()


def test_restart_and_profile_api():
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scenes import SceneSource
    src = SceneSource(seed=3)
    n = 96
    env = ParkingBatch(n, 128, profile=True)
    scenes = [src.draw() for _ in range(n)]
    env.set_scenes(np.arange(n), scenes)
    env.reset_obs()
    g = torch.Generator(device=env.device); g.manual_seed(1)
    for _ in range(5):
        env.step(torch.rand((n, 2), generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize()
    pose, t, acc = env.download_state()
    assert (t == 6).all()
    mask = torch.zeros(n, dtype=torch.uint8, device=env.device)
    mask[::3] = 1
    env.restart(mask)
    env.reset_obs(active=mask)
    torch.cuda.synchronize()
    pose2, t2, acc2 = env.download_state()
    start = np.array([s.start for s in scenes])
    assert np.array_equal(pose2[::3], start[::3]) and (t2[::3] == 1).all() and (acc2[::3] >= 0).all()
    assert np.array_equal(pose2[1::3], pose[1::3]) and (t2[1::3] == 6).all()
    km = env.kernel_ms()
    # 1 reset_obs + 5 steps + 1 masked reset_obs; max_obstacles 128 > 32 -> two tile-class launches each
    # (k_kinematics: 0 launches in the one-launch form of small batches, where the step kernel's waves compute the sub-step poses
    #  themselves; 28: the step kernel's motion and observation halves are launched separately)
    assert km['k_kinematics'][1] in (0, 5, 10) and km['k_env_step'][1] in (7, 14, 28)
    assert km['k_rs_validate'][1] in (7, 14) and km['k_rs_words'][1] in (7, 14) and all(v[0] > 0 for k, v in km.items() if not k.startswith('k_bev') and k not in ('k_rs_screen', 'k_kinematics'))   # (k_rs_screen: only with HOPE_RS_SPLIT=1)
    assert km['k_bev_image'] == (0.0, 0) and km['k_bev_prep'] == (0.0, 0)      # handle created without image=True
    assert km['k_rs_compact'][1] == 14                                         # one queue-compaction launch per call and tile class
    assert all(v == (0.0, 0) for v in env.kernel_ms().values())
    # state upload round trip
    env.upload_state(pose=pose, t=t, accum=acc)
    p3, t3, a3 = env.download_state()
    assert np.array_equal(p3, pose) and np.array_equal(t3, t) and np.array_equal(a3, acc)
    env.close()


def test_step_parity_f32_outputs():
    env, orc, rng = make_pair(256, seed=12, obs64=False)
    s = rollout(env, orc, rng, steps=10, tol=TOL32)
    print('parity f32:', s)
    assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0
    assert s['pose_err'] <= TOL64
    assert s['lidar_err'] < TOL32 and s['target_err'] < TOL32 and s['reward_err'] < TOL32
    env.close()


def test_motion_only_config2():
    """BASELINE config 2: kinematics + OBB collision only (4096 mixed scenes), flags bit-exact."""
    from hope_amd import _lib as L
    env, orc, rng = make_pair(4096, seed=13, level='mixed')       # mixed difficulty, as BASELINE config 2 says
    stages = L.STAGE_MOTION | L.STAGE_REWARD
    stats = new_stats()
    for it in range(6):
        act = rng.uniform(-1, 1, (env.n, 2))
        env.step(torch.from_numpy(act).to(env.device), stages=stages)
        o = orc.step(act, with_rs=False)
        torch.cuda.synchronize()
        stats['status_mismatch'] += int((env.status.cpu().numpy() != o['status']).sum())
        pose, t, acc = env.download_state()
        stats['pose_err'] = max(stats['pose_err'], float(np.abs(pose - orc.pose).max()))
        stats['reward_err'] = max(stats['reward_err'], float(np.abs(env.reward.cpu().numpy() - o['reward']).max()))
    print('config2:', stats)
    assert stats['status_mismatch'] == 0 and stats['pose_err'] <= TOL64 and stats['reward_err'] <= TOL64
    env.close()


def test_active_mask_leaves_scenes_untouched():
    env, orc, rng = make_pair(64, seed=14)
    env.reset_obs()
    torch.cuda.synchronize()
    pose0, t0, _ = env.download_state()
    lid0 = env.lidar.clone()
    active = torch.zeros(64, dtype=torch.uint8, device=env.device)
    active[::2] = 1
    act = torch.from_numpy(rng.uniform(-1, 1, (64, 2))).to(env.device)
    env.step(act, active=active)
    torch.cuda.synchronize()
    pose1, t1, _ = env.download_state()
    assert np.array_equal(pose1[1::2], pose0[1::2]) and np.array_equal(t1[1::2], t0[1::2])
    assert (t1[::2] == t0[::2] + 1).all()
    assert torch.equal(env.lidar[1::2], lid0[1::2])
    env.close()


def test_parity_stress_8k_scenes():
    """larger randomized sweep (8192 mixed scenes x 20 steps, full step incl. RS, ~84 k RS searches): every output
    exact.  (The classifier of the reference's ill-conditioned cases, rs_illcond.py, dates from when the kernels used
    OCML and the oracle glibc; it now only produces the diagnostics if something does differ.)  The oracle runs on all
    host cores (OpenMP build)."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource, pack_scenes
    from oracle import oracle as O
    n, mo = 8192, 128
    src = SceneSource(seed=77)
    uniq = [src.draw() for _ in range(1024)]
    rng = np.random.default_rng(78)
    scenes = []
    for k in range(n):
        s = uniq[k % len(uniq)]
        import copy
        s = copy.copy(s)
        if k % 3 == 0:      # start somewhere around the slot: arrivals, collisions, RS successes
            r, a = rng.uniform(0.0, 8.0), rng.uniform(0, 2 * np.pi)
            s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.6])
        scenes.append(s)
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, mo, omp=True)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, mo)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    from hope_amd import _lib as L
    from rs_illcond import allowed_results
    stats = new_stats()
    env.reset_obs()
    compare(env, orc.reset_obs(), TOL64, stats, True)
    seen, excused, unexplained, evaluated = set(), 0, [], 0
    for it in range(20):
        act = rng.uniform(-1.2, 1.2, (n, 2))
        if it % 3 == 0:
            act[:, 1] = np.sign(act[:, 1])
        env.step(torch.from_numpy(act).to(env.device))
        o = orc.step(act)
        st0 = dict(stats)
        compare(env, o, TOL64, stats, False)
        pose, t, acc = env.download_state()
        stats['pose_err'] = max(stats['pose_err'], float(np.abs(pose - orc.pose).max()))
        seen |= set(np.unique(o['status']).tolist())
        w = env.rs_word.cpu().numpy()
        evaluated += int(((o['status'] == 1) & (np.hypot(*(orc.pose[:, :2] - dest[:, :2]).T) < 10)).sum())
        bad = np.nonzero((w[:, 6] != o['rs_found']) | (w[:, :5] != o['rs_ctypes']).any(axis=1))[0]
        ok = np.setdiff1d(np.arange(n), bad)
        stats['rs_len_err'] = max(stats['rs_len_err'], float(np.abs(env.rs_lengths.cpu().numpy()[ok] - o['rs_lengths'][ok]).max()))
        for i in bad:                     # every disagreement must be one of the reference's ill-conditioned cases
            allowed = allowed_results(orc.pose[i], dest[i], verts[i, :nob[i]], nvert[i, :nob[i]], bbox[i])
            g = tuple(int(c) for c in w[i, :5] if c >= 0)
            r_ = tuple(int(c) for c in o['rs_ctypes'][i] if c >= 0)
            if g in allowed and r_ in allowed:
                excused += 1
            else:
                unexplained.append((it, int(i), g, r_, sorted(allowed)))
    s = stats
    print('stress:', s, 'rs evaluated', evaluated, 'ill-conditioned disagreements', excused, 'unexplained', unexplained)
    assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0
    assert not unexplained
    assert excused == 0      # shared deterministic math: not even the reference's ill-conditioned cases may differ
    assert s['pose_err'] <= TOL64 and s['lidar_err'] <= TOL64 and s['reward_err'] <= TOL64 and s['rs_len_err'] <= TOL64
    s['seen'] = sorted(seen)
    assert set(s['seen']) >= {1, 2, 3}
    env.close()


def test_auto_reset_equals_step_restart_reset_obs():
    """HOPE_AUTO_RESET: one launch sequence == step + restart(done) + reset_obs(active=done), bit for bit, and the
    oracle's step + reset reproduces the new episode's first observation."""
    envA, orc, rng = make_pair(1024, seed=41, level='mixed')
    envB, _, _ = make_pair(1024, seed=41, level='mixed')
    envA.reset_obs(); envB.reset_obs(); orc.reset_obs()
    n_done = 0
    for it in range(40):
        act = rng.uniform(-1.3, 1.3, (envA.n, 2))
        act[:, 1] = np.where(np.arange(envA.n) % 2 == 0, np.sign(act[:, 1] + 1e-9), act[:, 1])   # many leave the map box
        a = torch.from_numpy(act).to(envA.device)
        envA.step(a, auto_reset=True)
        envB.step(a)
        torch.cuda.synchronize()
        fin = {k: getattr(envB, k).clone() for k in ('reward', 'reward_info', 'status', 'done', 'rs_word', 'rs_lengths')}
        envB.restart(envB.done)
        envB.reset_obs(active=fin['done'])
        torch.cuda.synchronize()
        for k, v in fin.items():                       # outputs of the finished step
            assert torch.equal(getattr(envA, k), v), k
        for k in ('lidar', 'action_mask', 'target', 'pose'):   # first observation of the new episode
            assert torch.equal(getattr(envA, k), getattr(envB, k)), k
        pa, ta, aa = envA.download_state()
        pb, tb, ab = envB.download_state()
        assert np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(aa, ab)
        o = orc.step(act)
        assert np.array_equal(envA.status.cpu().numpy(), o['status'])
        done = o['status'] != 1
        n_done += int(done.sum())
        if done.any():
            ids = np.nonzero(done)[0]
            orc.pose[ids] = orc.start[ids]; orc.t[ids] = 0; orc.accum[ids] = 0
            # oracle has no active mask: run the action-less step on a copy of the finished scenes only
            sub = orc.__class__(len(ids), orc.max_obst)
            sub.set_scenes(np.arange(len(ids)), orc.start[ids], orc.dest[ids], orc.bbox[ids], orc.verts[ids], orc.nvert[ids], orc.n_obst[ids])
            so = sub.reset_obs()
            orc.pose[ids] = sub.pose; orc.t[ids] = sub.t; orc.accum[ids] = sub.accum
            assert np.array_equal(envA.lidar.cpu().numpy()[ids], so['lidar']) and np.array_equal(envA.action_mask.cpu().numpy()[ids], so['mask'])
        assert np.array_equal(pa, orc.pose) and np.array_equal(ta, orc.t.astype(np.int32)) and np.array_equal(aa, orc.accum)
    print('auto-reset: episodes finished', n_done)
    assert n_done > 50
    envA.close(); envB.close()


def test_full_size_properties_64k():
    """BASELINE full size (65 536 scenes, full step incl. RS): size-independent properties instead of the oracle.
      * determinism: the same state + actions give bit-identical outputs twice;
      * scenes are independent: permuting the scene slots permutes every output (no cross-scene coupling / races);
      * range invariants of every output; zero speed leaves the pose untouched and only advances t."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    n, mo = 65536, 128
    src = SceneSource(seed=91)
    uniq = [src.draw() for _ in range(512)]
    rng = np.random.default_rng(92)
    order = rng.integers(0, len(uniq), n)
    scenes = [uniq[i] for i in order]
    perm = rng.permutation(n)
    envA = ParkingBatch(n, mo)
    envB = ParkingBatch(n, mo)
    for a in range(0, n, 8192):
        ids = np.arange(a, a + 8192)
        envA.set_scenes(ids, [scenes[i] for i in ids])
        envB.set_scenes(ids, [scenes[perm[i]] for i in ids])         # slot i of B holds scene perm[i] of A
    pt = torch.from_numpy(perm).to(envA.device)
    envA.reset_obs(); envB.reset_obs()
    g = torch.Generator(device=envA.device); g.manual_seed(5)
    names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths')
    hb = torch.from_numpy(envA.tables['hull_base']).to(envA.device).float()
    for it in range(6):
        act = torch.rand((n, 2), generator=g, device=envA.device) * 2.4 - 1.2
        if it == 3:
            act[:, 1] = 0.0                                          # zero speed
        poseA0, tA0, accA0 = envA.download_state()
        envA.step(act)
        envB.step(act[pt].contiguous())
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(getattr(envA, k)[pt], getattr(envB, k)), (it, k)
        # determinism: rewind A and repeat the step
        snap = {k: getattr(envA, k).clone() for k in names}
        poseA1, tA1, accA1 = envA.download_state()
        envA.upload_state(pose=poseA0, t=tA0, accum=accA0)       # the whole episode state: pose, t, accum_arrive_reward
        envA.step(act)
        torch.cuda.synchronize()
        assert np.array_equal(envA.download_state()[2], accA1)
        for k in names:
            assert torch.equal(getattr(envA, k), snap[k]), (it, k)
        # invariants
        lid = envA.lidar + hb
        assert float(lid.min()) >= -1e-5 and float(lid.max()) <= 10.0 + 1e-5
        m = envA.action_mask
        assert bool(((m * 10 - torch.round(m * 10)).abs().max() < 1e-5) | (m == 0.01).all(dim=1).any())
        assert int(envA.status.min()) >= 1 and int(envA.status.max()) <= 5
        assert torch.equal(envA.done.bool(), envA.status != 1)
        w = envA.rs_word
        assert bool(((w[:, 6] == 0) | (w[:, 5] >= 3)).all()) and bool((w[:, 6] <= (envA.status == 1)).all())
        if it == 3:
            assert np.array_equal(poseA1, poseA0) and np.array_equal(tA1, tA0 + 1)
    envA.close(); envB.close()


def test_overlap_modes_equal_the_plain_launch():
    """HOPE_F_OVERLAP (two tile classes on two streams) only changes how the same kernels are launched: every output must be
    bit-identical to the plain stream-ordered launch, including the image, across steps with auto-reset and a changing action buffer.  HOPE_SPLIT_MIN=1 forces the two-launch form of the step
    kernel (motion / observation on separate streams), which the library otherwise uses from 16 384 scenes on."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    os.environ['HOPE_SPLIT_MIN'] = '1'
    try:
        _overlap_modes_body(ParkingBatch, SceneSource)
    finally:
        del os.environ['HOPE_SPLIT_MIN']


def _overlap_modes_body(ParkingBatch, SceneSource):
    n = 3072
    src = SceneSource(seed=5)
    scenes = [src.draw() for _ in range(n)]
    rng = np.random.default_rng(5)
    for s in scenes[::2]:
        r, a = rng.uniform(1.0, 8.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    envs = [ParkingBatch(n, 128, overlap=ov, image=True) for ov in (False, True)]
    for e in envs:
        e.set_scenes(np.arange(n), scenes)
        e.reset_obs()
    g = torch.Generator(device='cuda').manual_seed(3)
    names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths', 'img')
    for it in range(14):
        a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
        for e in envs:
            e.step(a.clone() if it % 2 else a, auto_reset=True)
        torch.cuda.synchronize()
        for e in envs[1:]:
            for k in names:
                assert torch.equal(getattr(e, k), getattr(envs[0], k)), (it, k, e.overlap)
    assert len({int(x) for x in envs[0].status.unique().tolist()}) >= 2
    for e in envs:
        e.close()


@pytest.mark.parametrize('two_launch', [True, False])
def test_deferred_rs_join_equals_the_joined_step(two_launch):
    """HOPE_DEFER_RS (two completion points per step: the caller's stream is ordered after the observation half, the
    Reeds-Shepp outputs by hope_env_wait_rs; consecutive steps PIPELINE across the library's env and search streams) must produce
    the same bits as the joined step: every output, every step, with auto-reset on new maps; entry points that read the state
    while the search of the last step is still unjoined (download_state, restart) join by themselves.  Both forms of the step
    kernel: two launches (motion / observation; forced at this size with HOPE_SPLIT_MIN) and one launch (small batches)."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    if two_launch:
        os.environ['HOPE_SPLIT_MIN'] = '1'
    try:
        n = 6144
        arrs = mixed_arrays(n, seed=77, max_obst=128)
        parts = [generate_arrays(lv, 256, seed=78 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
        pool = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6))
        envs = [ParkingBatch(n, 128, overlap=True) for _ in range(2)]
        for e in envs:
            e.set_scene_arrays(np.arange(n), *arrs[:5])
            e.set_draw_class(np.arange(3, n, 4), 1)          # the Dragon-Lake slots (scene k has level k % 4) keep drawing cases
            e.set_pool(pool)
            e.set_dlp_cases()
            e.set_redraw_seed(5)
            e.reset_obs()
        g = torch.Generator(device='cuda').manual_seed(11)
        names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths')
        side = torch.cuda.Stream()
        found = 0
        for it in range(24):
            a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
            envs[0].step(a, auto_reset=True, fresh=True)
            envs[1].step(a, auto_reset=True, fresh=True, defer_rs=True)
            if it % 3 == 0:                                   # the observation half is ordered on the current stream already
                torch.cuda.current_stream().synchronize()
                for k in names[:8]:
                    assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), (it, k)
            if it % 4 == 1:                                   # a state download joins by itself
                p0, t0, _ = envs[0].download_state()
                p1, t1, _ = envs[1].download_state()
                assert np.array_equal(p0, p1) and np.array_equal(t0, t1)
            if it % 2 == 0:
                envs[1].wait_rs()
                torch.cuda.current_stream().synchronize()
                for k in names:
                    assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), (it, k)
                found += int((envs[1].rs_word[:, 6] > 0).sum())
        envs[1].wait_rs()
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), k
        assert found > 0
        # the `active` mask may be freed / overwritten (in stream order) right after a deferred step: the Reeds-Shepp chain, which
        # runs on after the caller's stream has been released, reads the snapshot the motion launch took, not the caller's buffer
        for it in range(6):
            a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
            keep = (torch.rand(n, device='cuda', generator=g) < 0.7).to(torch.uint8)
            tmp = keep.clone()
            envs[0].step(a, active=keep)
            envs[1].step(a, active=tmp, defer_rs=True)
            tmp.fill_(1 if it % 2 else 0)                     # overwritten at once on the caller's stream ...
            del tmp
            junk = torch.full((n,), 1 - it % 2, dtype=torch.uint8, device='cuda')      # ... and its memory recycled by the allocator
            envs[1].wait_rs()
            torch.cuda.synchronize()
            for k in names:
                assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), (it, k)
            del junk
        for e in envs:
            e.close()
    finally:
        os.environ.pop('HOPE_SPLIT_MIN', None)


def test_config3_exact_size_16384_scenes():
    """BASELINE config 3 at exactly its size: 16 384 parallel scenes, full step (kinematics + collision + lidar +
    action mask + Reeds-Shepp search), every output exact against the oracle (all host cores)."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource, pack_scenes
    from oracle import oracle as O
    import copy
    n, mo = 16384, 128
    src = SceneSource(seed=163)
    uniq = [src.draw() for _ in range(1024)]
    rng = np.random.default_rng(164)
    scenes = []
    for k in range(n):
        s = copy.copy(uniq[k % len(uniq)])
        if k % 2 == 0:
            r, a = rng.uniform(0.5, 9.0), rng.uniform(0, 2 * np.pi)
            s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.6])
        scenes.append(s)
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    orc = O.BatchOracle(n, mo, omp=True)
    start, dest, bbox, verts, nob, nvert = pack_scenes(scenes, mo)
    orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    stats = new_stats()
    env.reset_obs()
    compare(env, orc.reset_obs(), TOL64, stats, True)
    for it in range(3):
        act = rng.uniform(-1.2, 1.2, (n, 2))
        env.step(torch.from_numpy(act).to(env.device))
        compare(env, orc.step(act), TOL64, stats, True)
        pose, tt, acc = env.download_state()
        stats['pose_err'] = max(stats['pose_err'], float(np.abs(pose - orc.pose).max()))
    print('config 3 @16384:', stats)
    assert all(stats[k] == 0 for k in ('status_mismatch', 'mask_mismatch', 'rs_flag_mismatch', 'rs_word_mismatch'))
    assert max(stats[k] for k in ('lidar_err', 'target_err', 'reward_err', 'rinfo_err', 'pose_err', 'rs_len_err')) <= TOL64
    env.close()


def test_independent_math_libm_oracle():
    """The kernels and the default oracle share hope_math.h, so a wrong polynomial there would be wrong on both sides.
    This run compares the HIP path with the oracle's **glibc-libm** build (what Python's `math` gave the reference):
    every step starts from the GPU's state, discrete outputs must agree except for the two documented ill-conditioned
    classes of the Reeds-Shepp search (tests/rs_illcond.py), continuous outputs to 1e-9.  (tests/libm_compare.py; the same
    comparison over 1e6 scene-steps: tools/libm_soak.py -> profiles/r06_libm_soak.txt.)"""
    from libm_compare import run, check
    r = run(n=2048, steps=8, seed=91, omp=False)
    print('libm oracle:', {k: r[k] for k in ('status_bad', 'worst', 'searches', 'excused', 'unexplained')})
    print('  tolerated mask differences (step, scene, distance of the nearest table entry to its scan value):', r['mask_ties'])
    print('  tolerated search differences (step, scene, GPU word, libm-oracle word, class, relative length gap):', r['rs_ties'])
    check(r, min_searches=3000)


def test_scene_pool_turnover_matches_oracle_on_the_drawn_scenes():
    """f-2: episode turnover with a NEW map from the device-resident pool.  After redraw + reset_obs every scene must
    behave exactly like the oracle on the pool scene it drew (identified through pool_index): same class, first
    observation and following steps exact; the finished step's reward / status / done are kept."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource, pack_scenes
    from oracle import oracle as O
    n, mo, P = 1024, 128, 700
    src = SceneSource(seed=31)
    scenes = [src.draw() for _ in range(n)]
    pool = [src.draw() for _ in range(P)]
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), scenes)
    env.set_pool(pool)
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    rng = np.random.default_rng(32)
    env.reset_obs()
    env.upload_state(t=rng.integers(172, 200, n))            # episodes run out of time (t > 200) all through the test
    cur = list(scenes)
    assert (env.pool_index() == -1).all()
    drawn = 0
    for it in range(30):
        act = rng.uniform(-1.2, 1.2, (n, 2))
        act[:, 1] = np.where(np.arange(n) % 2 == 0, np.sign(act[:, 1] + 1e-9), act[:, 1])      # many leave the map / collide
        env.step(torch.from_numpy(act).to(env.device))
        torch.cuda.synchronize()
        fin = {k: getattr(env, k).clone() for k in ('reward', 'status', 'done', 'rs_word')}
        done = env.done.cpu().numpy().astype(bool)
        env.turnover(seed=1000 + it)
        torch.cuda.synchronize()
        for k, v in fin.items():
            assert torch.equal(getattr(env, k), v), k                       # the finished step's outputs survive
        idx = env.pool_index()
        assert (idx[done] >= 0).all()
        for i in np.nonzero(done)[0]:
            new = pool[int(idx[i])]
            assert (new.n_obst > 32) == (cur[i].n_obst > 32)                # same tile class
            cur[i] = new
        drawn += int(done.sum())
        if it % 6 == 5 or it == 29:                                          # full-state check against the oracle
            pose, tt, acc = env.download_state()
            start, dest, bbox, verts, nob, nvert = pack_scenes(cur, mo)
            orc = O.BatchOracle(n, mo, omp=True)
            orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
            orc.pose[:], orc.t[:], orc.accum[:] = pose, tt, acc
            assert np.array_equal(pose[done], start[done]) and (tt[done] == 1).all()
            a2 = rng.uniform(-1, 1, (n, 2))
            env.step(torch.from_numpy(a2).to(env.device))
            st = new_stats()
            compare(env, orc.step(a2), TOL64, st, True)
            assert st['status_mismatch'] == 0 and st['mask_mismatch'] == 0 and st['rs_word_mismatch'] == 0
            assert max(st['lidar_err'], st['target_err'], st['reward_err']) <= TOL64
            env.turnover(seed=5000 + it)
            idx2 = env.pool_index()
            d2 = env.done.cpu().numpy().astype(bool)
            for i in np.nonzero(d2)[0]:
                cur[i] = pool[int(idx2[i])]
    assert drawn > 300
    idx = env.pool_index()
    assert len(np.unique(idx[idx >= 0])) > 150                               # draws spread over the pool
    env.close()


def test_float32_action_rescale_is_the_float32_box_arithmetic():
    """VERDICT r1 missing #6: the reference's action_rescale (env_wrapper.py:46-47) works on a float32 action with gym's
    float32 Box bounds, so the physical action CarParking.step receives is rounded to float32.  HOPE_ACTION_RESCALE_F32
    reproduces that: a float32 [-1, 1] action buffer with the bit == the float64 PHYSICAL path fed numpy's float32 result."""
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd import tables as T
    from hope_amd.scenes import DlpScenePool
    n = 2048
    rng = np.random.default_rng(77)
    pool = DlpScenePool()
    scenes = [pool.sample(rng=rng) for _ in range(n)]
    a = ParkingBatch(n, 128, obs_dtype=torch.float64, action_dtype=torch.float32, rescale_f32=True)
    b = ParkingBatch(n, 128, obs_dtype=torch.float64, action_dtype=torch.float64)
    c = ParkingBatch(n, 128, obs_dtype=torch.float64, action_dtype=torch.float32)          # default: float64 arithmetic
    for e in (a, b, c):
        e.set_scenes(np.arange(n), scenes)
        e.reset_obs()
    low = np.array([T.VALID_STEER[0], T.VALID_SPEED[0]]).astype(np.float32)               # gym.spaces.Box default dtype
    high = np.array([T.VALID_STEER[1], T.VALID_SPEED[1]]).astype(np.float32)
    differs = 0
    for it in range(6):
        act = rng.uniform(-1.1, 1.1, (n, 2)).astype(np.float32)
        phys = np.clip(act, -1, 1)
        phys = phys * (high - low) / 2 + (high + low) / 2
        assert phys.dtype == np.float32
        a.step(torch.from_numpy(act).to(a.device))
        c.step(torch.from_numpy(act).to(c.device))
        b.step(torch.from_numpy(phys.astype(np.float64)).to(b.device), stages=L.STAGE_ALL | L.ACTION_PHYSICAL)
        torch.cuda.synchronize()
        pa, ta, _ = a.download_state()
        pb, tb, _ = b.download_state()
        pc, _, _ = c.download_state()
        assert np.array_equal(pa, pb) and np.array_equal(ta, tb)
        assert torch.equal(a.lidar, b.lidar) and torch.equal(a.reward, b.reward) and torch.equal(a.status, b.status)
        assert torch.equal(a.action_mask, b.action_mask)
        differs += int((pa != pc).any(axis=1).sum())
    assert differs > 0                      # the float64 rescale is a different (documented) arithmetic
    for e in (a, b, c):
        e.close()


def test_two_launch_step_kernel_parity_with_rs_search():
    """The motion / observation halves of the step kernel launched separately (the default from 16 384 scenes on; forced
    here): every output equals the oracle exactly, Reeds-Shepp search included."""
    import os
    os.environ['HOPE_SPLIT_MIN'] = '1'
    try:
        env, orc, rng = make_pair(1536, seed=31, level='mixed')
        s = rollout(env, orc, rng, steps=8, tol=TOL64, with_rs=True)
        print('two-launch parity:', s)
        assert s['status_mismatch'] == 0 and s['mask_mismatch'] == 0 and s['rs_flag_mismatch'] == 0 and s['rs_word_mismatch'] == 0
        assert max(s['pose_err'], s['lidar_err'], s['target_err'], s['reward_err'], s['rinfo_err'], s['rs_len_err']) <= TOL64
        env.close()
    finally:
        del os.environ['HOPE_SPLIT_MIN']


def test_fused_new_map_turnover_equals_step_plus_turnover():
    """HOPE_AUTO_REDRAW (a new map drawn inside the step kernel) == hope_env_step + hope_env_redraw(done, seed) +
    hope_env_reset_obs(active = done) with the same seed: every output, the state and the drawn pool entries, in the
    one-launch and in the two-launch form of the step kernel, image included."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    n, mo, P = 2048, 128, 500
    src = SceneSource(seed=41)
    scenes = [src.draw() for _ in range(n)]
    pool = [src.draw() for _ in range(P)]
    names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths', 'img')
    for split in (False, True):
        if split:
            os.environ['HOPE_SPLIT_MIN'] = '1'
        try:
            a = ParkingBatch(n, mo, obs_dtype=torch.float64, image=True)          # fused
            b = ParkingBatch(n, mo, obs_dtype=torch.float64, image=True)          # step, then turnover
            rng = np.random.default_rng(42)
            t0 = rng.integers(180, 200, n)
            for e in (a, b):
                e.set_scenes(np.arange(n), scenes)
                e.set_pool(pool)
                e.reset_obs()
                e.upload_state(t=t0)                                              # many episodes run out of time
            a.set_redraw_seed(777)
            g = torch.Generator(device='cuda').manual_seed(9)
            turned = 0
            for it in range(24):
                act = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
                a.step(act, auto_reset=True, fresh=True)
                b.step(act)
                turned += int(b.done.sum().item())
                b.turnover(seed=777)
                torch.cuda.synchronize()
                for k in names:
                    assert torch.equal(getattr(a, k), getattr(b, k)), (split, it, k)
                pa, ta, aa = a.download_state()
                pb, tb, ab = b.download_state()
                assert np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(aa, ab)
                assert np.array_equal(a.pool_index(), b.pool_index())
            assert turned > 500
            a.close(); b.close()
        finally:
            os.environ.pop('HOPE_SPLIT_MIN', None)


def test_headline_configuration_has_an_oracle_witness_65536_scenes():
    """the configuration bench.py times -- 65 536 scenes, float32 observations and actions, the step
    kernel in two launches per tile class on concurrent streams, HOPE_AUTO_RESET | HOPE_AUTO_REDRAW from a device pool --
    compared with the oracle DIRECTLY, through bench.py's own witness (parity_witness: 1 024 random slots rebuilt in the
    oracle from the device state, one fused step, turnovers followed onto the drawn map), over 32 steps so that episodes end."""
    import bench
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scene_gen import generate_arrays, mixed_arrays
    n, mo, P, U = 65536, 128, 2046, 1024
    init = mixed_arrays(U, seed=61, max_obst=mo)
    env = ParkingBatch(n, mo, obs_dtype=torch.float32, action_dtype=torch.float32)
    assert env.overlap                                            # the default launch form: two chains, split step kernel
    for a in range(0, n, 8192):
        sl = np.arange(a, a + 8192) % U
        env.set_scene_arrays(np.arange(a, a + 8192), init[0][sl], init[1][sl], init[2][sl], init[3][sl], init[4][sl])
    env.set_draw_class(np.arange(3, n, 4), 1)                    # every fourth slot: Dragon-Lake lots, drawn on the device
    parts = [generate_arrays(lv, P // 3, seed=62 + j, max_obst=mo) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
    env.set_pool(tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6)))
    env.set_dlp_cases()
    env.set_redraw_seed(99)
    env.reset_obs()
    rng = np.random.default_rng(62)
    env.upload_state(t=rng.integers(1, 201, n))                  # a steady-state age mix: OUTTIME fires during the test
    g = torch.Generator(device=env.device).manual_seed(63)
    tot = dict(turnovers=0, status_mismatch=0, done_mismatch=0, mask_mismatch=0, rs_mismatch=0, state_mismatch=0,
               f32_value_mismatch=0, rs_found=0)
    worst = 0.0
    for it in range(32):
        act = torch.rand((n, 2), device=env.device, generator=g) * 2 - 1
        if it % 4 == 3:                                           # plain steps in between, so that states drift apart
            env.step(act, auto_reset=True, fresh=True)
            continue
        r = bench.parity_witness(env, L.STAGE_ALL, True, act, 1024, seed=1000 + it)
        for k in tot:
            tot[k] += r[k]
        worst = max(worst, r['max_abs_err'])
    print('headline witness:', tot, 'max_abs_err', worst)
    assert tot['turnovers'] > 100 and tot['rs_found'] > 50
    assert all(tot[k] == 0 for k in ('status_mismatch', 'done_mismatch', 'mask_mismatch', 'rs_mismatch', 'state_mismatch', 'f32_value_mismatch'))
    assert worst < TOL32
    env.close()


def _rs_filter_stats(reset=True):
    import ctypes as C
    from hope_amd import load_library
    out = (C.c_uint64 * 16)()
    assert load_library().hope_debug_rs_filter_stats(out, int(reset)) == 0
    v = list(out)
    d = dict(zip(('passes', 'hit_passes', 'exact_passes', 'undecided_samples', 'bad_hit', 'bad_clear', 'samples'), v[:7]))
    d['undecided_because'] = dict(zip(('no_certain_crossing', 'axis_parallel_edge', 'axis_parallel_hull', 'corner_near_line', 'shallow_angle'), v[8:13]))
    # the screen pass (round 5): words it condemned, searches it emptied, condemned words the full walk found valid (self-check)
    d['screen_words'], d['screen_dead_searches'], d['screen_bad'] = v[13], v[14], v[7]
    return d


def test_rs_float32_filter_equals_the_float64_kernel_and_never_contradicts_it():
    """The default validation kernel decides most samples in float32 with error margins and falls back to the float64
    arithmetic for the rest (hope_rs.hip, k_rs_validate_f).  (a) Its outputs equal the all-float64 kernel's (HOPE_RS_EXACT)
    on the same states over a mixed 16 384-scene rollout; (b) self-check mode evaluates EVERY sample in float64 too and counts
    float32 verdicts the float64 arithmetic contradicts: none, in either direction; (c) the filter does decide most passes;
    (d) the screen pass (closed-form samples of up to four words per pass) condemns most invalid words, and in self-check mode
    every word it condemned is walked sample by sample anyway and comes out invalid."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    n, mo = 16384, 128
    src = SceneSource(seed=71)
    uniq = [src.draw() for _ in range(1024)]
    rng = np.random.default_rng(72)
    for k in range(0, len(uniq), 2):          # half of them start near the destination: tight surroundings, found paths
        s = uniq[k]
        r, a = rng.uniform(0.0, 7.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    scenes = [uniq[i % len(uniq)] for i in range(n)]
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    for a in range(0, n, 8192):
        env.set_scenes(np.arange(a, a + 8192), scenes[a:a + 8192])
    env.reset_obs()
    names = ('rs_word', 'rs_lengths', 'status', 'reward', 'lidar', 'action_mask')
    g = torch.Generator(device=env.device).manual_seed(73)
    found = searches = 0
    try:
        for it in range(14):
            act = torch.rand((n, 2), device=env.device, generator=g, dtype=torch.float64) * 2.4 - 1.2
            pose, t, acc = env.download_state()
            os.environ['HOPE_RS_EXACT'] = '1'
            env.step(act)
            torch.cuda.synchronize()
            ref = {k: getattr(env, k).clone() for k in names}
            del os.environ['HOPE_RS_EXACT']
            env.upload_state(pose=pose, t=t, accum=acc)
            os.environ['HOPE_RS_DEBUG'] = '0x6000' if it % 2 else '0x4000'          # statistics; odd steps: self-check of every sample
            env.step(act)
            torch.cuda.synchronize()
            for k in names:
                assert torch.equal(getattr(env, k), ref[k]), (it, k)
            found += int((env.rs_word[:, 6] > 0).sum())
            searches += int(((env.status == 1)).sum())
        st = _rs_filter_stats()
    finally:
        os.environ.pop('HOPE_RS_EXACT', None)
        os.environ.pop('HOPE_RS_DEBUG', None)
    print('float32 filter:', st, 'found', found)
    if st['bad_hit'] or st['bad_clear']:
        import ctypes as C
        from hope_amd import load_library
        dump = np.zeros((64, 16))
        load_library().hope_debug_rs_filter_dump(dump.ctypes.data_as(C.c_void_p))
        np.set_printoptions(precision=9, linewidth=250, suppress=True)
        print(dump[:min(64, st['bad_hit'] + st['bad_clear'])])
        np.save('gpurun_out/rs_filter_dump.npy', dump)
    assert found > 2000
    assert st['samples'] > 1_000_000 and st['bad_hit'] == 0 and st['bad_clear'] == 0
    assert st['passes'] > 30_000 and st['exact_passes'] < 0.35 * st['passes']
    assert st['screen_bad'] == 0 and st['screen_words'] > 50_000 and st['screen_dead_searches'] > 10_000, st
    env.close()


def test_two_kernel_validation_equals_the_one_kernel_form():
    """Round 5: with HOPE_RS_SPLIT=1 k_rs_screen (8 waves per SIMD) condemns words and queues the searches that still have a word to
    walk for k_rs_validate_f; the default one-kernel form runs the same screen inside the walk kernel (and is the faster one).  Same
    rs_word / rs_lengths on the same states over a mixed rollout with found paths, whatever the order of the survivor queue's atomics."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scenes import SceneSource
    n, mo = 8192, 128
    src = SceneSource(seed=171)
    uniq = [src.draw() for _ in range(1024)]
    rng = np.random.default_rng(172)
    for k in range(0, len(uniq), 2):
        s = uniq[k]
        r, a = rng.uniform(0.0, 7.0), rng.uniform(0, 2 * np.pi)
        s.start = np.array([s.dest[0] + r * np.cos(a), s.dest[1] + r * np.sin(a), s.dest[2] + rng.normal() * 0.5])
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    env.set_scenes(np.arange(n), [uniq[i % len(uniq)] for i in range(n)])
    env.reset_obs()
    g = torch.Generator(device=env.device).manual_seed(173)
    found = 0
    try:
        for it in range(10):
            act = torch.rand((n, 2), device=env.device, generator=g, dtype=torch.float64) * 2.4 - 1.2
            pose, t, acc = env.download_state()
            os.environ['HOPE_RS_SPLIT'] = '1'
            env.step(act)
            torch.cuda.synchronize()
            ref = {k: getattr(env, k).clone() for k in ('rs_word', 'rs_lengths', 'status', 'reward')}
            del os.environ['HOPE_RS_SPLIT']
            env.upload_state(pose=pose, t=t, accum=acc)
            env.step(act)
            torch.cuda.synchronize()
            for k, v in ref.items():
                assert torch.equal(getattr(env, k), v), (it, k)
            found += int((env.rs_word[:, 6] > 0).sum())
    finally:
        os.environ.pop('HOPE_RS_SPLIT', None)
    assert found > 1000
    env.close()


def _ks2(a, b):
    a, b = np.sort(a), np.sort(b)
    allv = np.concatenate([a, b])
    return float(np.abs(np.searchsorted(a, allv, side='right') / len(a) - np.searchsorted(b, allv, side='right') / len(b)).max())


def test_device_side_dlp_draw_distribution_and_oracle_parity():
    """f-2 (VERDICT r2 #3a): ParkingMapDLP.reset on the device.  (a) > 10^5 turnovers of large-tile scenes draw cases of
    data/dlp.data with a fresh start candidate, jitter, flips, map box and obstacle cull; every marginal of the drawn scenes is
    within KS 0.02 of the host sampler's (`DlpScenePool.sample`, pinned to parking_map_dlp.py:38-101); (b) with a one-candidate
    case the jitter is N(0, 0.05^2) m / N(0, 0.02^2) rad and each flip has probability 1/2; (c) the env steps exactly like the
    oracle on the drawn scenes; (d) the fused turnover (HOPE_AUTO_REDRAW) draws the same scenes as hope_env_redraw; (e) a
    snapshot's maps are restored by repeating the draws."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import DlpScenePool, pack_scenes
    from oracle import oracle as O
    n, mo = 16384, 128
    pool = DlpScenePool()
    rng = np.random.default_rng(81)
    uniq = [pool.sample(rng=rng) for _ in range(256)]
    packed = pack_scenes(uniq, mo)
    env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64)
    for a in range(0, n, 8192):
        sl = np.arange(a, a + 8192) % len(uniq)
        env.set_scene_arrays(np.arange(a, a + 8192), *(x[sl] for x in packed[:5]))
    env.set_draw_class(np.arange(n), 1)                       # Dragon-Lake slots, whatever the first map's obstacle count
    with pytest.raises(Exception, match='pool'):
        env.redraw(torch.ones(n, dtype=torch.uint8, device=env.device), seed=1)          # no pool, no cases yet
    env.set_dlp_cases(pool)
    ones = torch.ones(n, dtype=torch.uint8, device=env.device)
    feats, cases = [], []
    ids = np.arange(n)
    for it in range(7):
        env.redraw(ones, seed=500 + it)
        start, dest, bbox, verts, nob = env.download_scenes(ids)
        idx = env.pool_index()
        assert (idx <= -2).all() and (nob > 0).all() and (nob <= mo).all()
        cases.append(-2 - idx)
        feats.append(np.column_stack([start, dest, bbox[:, 1] - bbox[:, 0], bbox[:, 3] - bbox[:, 2], nob,
                                      np.hypot(start[:, 0] - dest[:, 0], start[:, 1] - dest[:, 1])]))
    assert env.pool_overflow() == 0
    dev = np.concatenate(feats)
    cs = np.concatenate(cases)
    assert len(dev) > 100_000
    hist = np.bincount(cs, minlength=len(pool))
    assert hist.min() > 0.6 * len(cs) / len(pool) and hist.max() < 1.4 * len(cs) / len(pool)     # cases uniform (:50)
    hrng = np.random.default_rng(82)
    host = []
    for _ in range(len(dev) // 4):
        s = pool.sample(rng=hrng)
        host.append(np.concatenate([s.start, s.dest, [s.bbox[1] - s.bbox[0], s.bbox[3] - s.bbox[2], s.n_obst,
                                                      np.hypot(s.start[0] - s.dest[0], s.start[1] - s.dest[1])]]))
    host = np.array(host)
    names = ['start_x', 'start_y', 'start_yaw', 'dest_x', 'dest_y', 'dest_yaw', 'box_w', 'box_h', 'n_obst', 'dist']
    worst = {nm: _ks2(np.cos(dev[:, j]) if 'yaw' in nm else dev[:, j], np.cos(host[:, j]) if 'yaw' in nm else host[:, j]) for j, nm in enumerate(names)}
    print('device DLP draws vs host sampler, KS:', {k: round(v, 4) for k, v in worst.items()})
    assert max(worst.values()) < 0.02, worst
    # (c) the env on the drawn scenes == the oracle on the downloaded scenes
    sub = np.sort(rng.choice(n, 768, replace=False))
    start, dest, bbox, verts, nob = env.download_scenes(sub)
    env.reset_obs()
    t = env.tables
    O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'])
    orc = O.BatchOracle(len(sub), mo, omp=True)
    orc.set_scenes(np.arange(len(sub)), start, dest, bbox, verts, np.full((len(sub), mo), 4, np.int32), nob)
    o = orc.reset_obs()
    st = torch.from_numpy(sub).to(env.device)
    assert np.array_equal(env.lidar[st].cpu().numpy(), o['lidar']) and np.array_equal(env.action_mask[st].cpu().numpy(), o['mask'])
    for it in range(5):
        act = rng.uniform(-1, 1, (n, 2))
        env.step(torch.from_numpy(act).to(env.device))
        o = orc.step(act[sub])
        torch.cuda.synchronize()
        assert np.array_equal(env.status[st].cpu().numpy(), o['status'])
        assert np.array_equal(env.lidar[st].cpu().numpy(), o['lidar']) and np.array_equal(env.reward[st].cpu().numpy(), o['reward'])
        assert np.array_equal(env.rs_word[st][:, 6].cpu().numpy(), o['rs_found'])
    # (e) snapshot: maps + state, then move on, then restore
    pose, tt, acc = env.download_state()
    pidx, ep = env.pool_state()
    gen = env.pool_generation()
    snap = env.download_scenes(sub)
    env.redraw(ones, seed=900)
    assert not np.array_equal(env.download_scenes(sub)[0], snap[0])
    # a slot that now holds a device-drawn lot of > 32 obstacles cannot join the small class (the host's obstacle counts of the
    # uploaded first maps say nothing about what the device drew since)
    big = int(np.argmax(env.download_scenes(np.arange(64))[4]))
    assert env.download_scenes(np.array([big]))[4][0] > 32
    with pytest.raises(Exception, match='32 obstacles'):
        env.set_draw_class(np.array([big]), 0)
    env.restore_maps(pidx, ep, seed=500 + 6, generation=gen)
    env.upload_state(pose=pose, t=tt, accum=acc)
    back = env.download_scenes(sub)
    assert all(np.array_equal(snap[j], back[j]) for j in (0, 1, 2, 4))
    assert all(np.array_equal(snap[3][k, :snap[4][k]], back[3][k, :back[4][k]]) for k in range(len(sub)))     # (slots beyond n_obst are stale)
    assert np.array_equal(env.pool_state()[0], pidx) and np.array_equal(env.pool_state()[1], ep)
    # a snapshot of another pool generation is refused: repeating its draws would silently restore other maps
    env.set_dlp_cases(pool)                                   # (any change of what a draw can return bumps the generation)
    assert env.pool_generation() != gen
    with pytest.raises(Exception, match='generation'):
        env.restore_maps(pidx, ep, seed=500 + 6, generation=gen)
    env.close()

    # (b) one case, one candidate: the residual IS the jitter
    class One:
        dest = np.array([[3.0, 40.0, 1.2]]); starts = np.array([[10.0, 30.0, 0.4]]); start_off = np.array([0, 1]); case_set = np.array([0])
        set_off = np.array([0, 40]); set_nvert = np.full(40, 4)
        set_verts = np.array([[[x, 20.0], [x + 1, 20.0], [x + 1, 21.0], [x, 21.0]] for x in np.linspace(-5, 60, 40)])
    m = 32768
    e2 = ParkingBatch(m, mo, obs_dtype=torch.float64)
    sc = pool.sample(rng=rng)
    for a in range(0, m, 8192):
        e2.set_scenes(np.arange(a, a + 8192), [sc] * 8192)
    e2.set_draw_class(np.arange(m), 1)
    e2.set_dlp_cases(One)
    e2.redraw(torch.ones(m, dtype=torch.uint8, device=e2.device), seed=7)
    start, dest, bbox, verts, nob = e2.download_scenes(np.arange(m))
    flipped_s = np.abs(np.cos(start[:, 2] - 0.4) + 1) < 0.1
    flipped_d = np.abs(np.cos(dest[:, 2] - 1.2) + 1) < 0.1
    assert abs(flipped_s.mean() - 0.5) < 0.02 and abs(flipped_d.mean() - 0.5) < 0.02
    assert abs((flipped_s & flipped_d).mean() - 0.25) < 0.02                         # independent
    r = (start[~flipped_s] - One.starts[0]) / np.array([0.05, 0.05, 0.02])
    from scipy import stats
    for j in range(3):
        d = stats.kstest(r[:, j], 'norm').statistic
        assert d < 0.015, (j, d)
    assert abs(np.corrcoef(r[:, 0], r[:, 1])[0, 1]) < 0.03 and abs(np.corrcoef(r[:, 0], r[:, 2])[0, 1]) < 0.03
    # the unflipped dest is the case's; the flipped one is its mirror about the box centre (heading + pi)
    assert np.allclose(dest[~flipped_d], One.dest[0]) and np.allclose(dest[flipped_d][:, 2], 1.2 + np.pi)
    mid = 0.5 * (3.76 - 0.93)
    assert np.allclose(dest[flipped_d][:, :2], One.dest[0, :2] + 2 * mid * np.array([np.cos(1.2), np.sin(1.2)]))
    # map box and cull rules on the drawn values (parking_map_dlp.py:70-73, :88-101)
    s0 = np.where(flipped_s[:, None], start - np.column_stack([2 * mid * np.cos(start[:, 2] - np.pi), 2 * mid * np.sin(start[:, 2] - np.pi), np.full(m, np.pi)]), start)
    assert np.array_equal(bbox[:, 0], np.floor(np.minimum(s0[:, 0], 3.0) - 20)) and np.array_equal(bbox[:, 3], np.ceil(np.maximum(s0[:, 1], 40.0) + 20))
    keep = ~((One.set_verts[:, :, 0].max(1)[None] <= bbox[:, :1]) | (One.set_verts[:, :, 0].min(1)[None] >= bbox[:, 1:2]))
    assert np.array_equal(nob, keep.sum(1))
    e2.close()


def test_fused_turnover_draws_dlp_cases_like_redraw():
    """(d) of the above: HOPE_AUTO_REDRAW with device-drawn Dragon-Lake cases and generated lots == step + hope_env_redraw(done)
    + reset_obs(done), in both launch forms; a resident size class without pool entries is refused."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    n, mo = 4096, 128
    init = mixed_arrays(n, seed=91, max_obst=mo)
    small = generate_arrays('Complex', 300, seed=92, max_obst=mo)[:6]
    names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths')
    for split in (False, True):
        if split:
            os.environ['HOPE_SPLIT_MIN'] = '1'
        try:
            a = ParkingBatch(n, mo, obs_dtype=torch.float64)
            b = ParkingBatch(n, mo, obs_dtype=torch.float64)
            rng = np.random.default_rng(93)
            t0 = rng.integers(185, 200, n)
            for e in (a, b):
                e.set_scene_arrays(np.arange(n), *init[:5])
                e.set_draw_class(np.arange(3, n, 4), 1)                      # every fourth slot is a Dragon-Lake lot
                e.set_dlp_cases()
                e.reset_obs()
                e.upload_state(t=t0)
            g = torch.Generator(device='cuda').manual_seed(9)
            act = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
            with pytest.raises(Exception, match='size class'):
                a.step(act, auto_reset=True, fresh=True)                   # the <= 32-obstacle lots have nothing to draw from
            for e in (a, b):
                e.set_pool(small)
            a.set_redraw_seed(31)
            turned = cases = 0
            for it in range(20):
                act = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
                a.step(act, auto_reset=True, fresh=True)
                b.step(act)
                turned += int(b.done.sum().item())
                b.turnover(seed=31)
                torch.cuda.synchronize()
                for k in names:
                    assert torch.equal(getattr(a, k), getattr(b, k)), (split, it, k)
                assert np.array_equal(a.pool_index(), b.pool_index())
                assert all(np.array_equal(x, y) for x, y in zip(a.download_state(), b.download_state()))
            ia = a.pool_index()
            cases = int((ia <= -2).sum())
            sub = np.nonzero(ia <= -2)[0][:64]
            da, db = a.download_scenes(sub), b.download_scenes(sub)
            assert all(np.array_equal(da[j], db[j]) for j in (0, 1, 2, 4)) and all(np.array_equal(da[3][k, :da[4][k]], db[3][k, :db[4][k]]) for k in range(len(sub)))
            assert turned > 300 and cases > 50 and a.pool_overflow() == 0
            a.close(); b.close()
        finally:
            os.environ.pop('HOPE_SPLIT_MIN', None)


def test_pool_refresh_in_the_background_keeps_up_and_matches_the_synchronous_upload():
    """f-2 (VERDICT r2 #3b): the pool of generated lots is replaced while the step loop runs -- native generator in a background
    thread into pinned staging, asynchronous upload into the idle pool set, swap by stream order (hope_env_pool_staging /
    hope_env_commit_pool; no hipDeviceSynchronize).  An env refreshed that way behaves exactly like one given the same pools with
    the synchronous hope_env_set_pool at the same steps; committing does not wait for the GPU; scenes turned over after a commit
    hold maps of the NEW pool."""
    import time
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import PoolRefresher, generate_arrays
    n, mo, P = 8192, 32, 1200
    init = generate_arrays('Complex', n, seed=5, max_obst=mo)[:5]
    a = ParkingBatch(n, mo, obs_dtype=torch.float64)
    b = ParkingBatch(n, mo, obs_dtype=torch.float64)
    for e in (a, b):
        e.set_scene_arrays(np.arange(n), *init)
        e.reset_obs()
        e.upload_state(t=np.random.default_rng(6).integers(150, 200, n))
        e.set_redraw_seed(77)
    ref = PoolRefresher(a, P, levels=('Normal', 'Complex', 'Extrem'), seed=9, threads=2)
    assert ref.poll(wait=True)                                        # the first pool
    pools = []

    def snapshot():
        arrs = [x.copy() for x in a.pool_staging(P)]                 # (the staging is being refilled; copies of what was committed are kept below)
        return arrs
    names = ('lidar', 'action_mask', 'reward', 'status', 'done', 'pose', 'rs_word')
    g = torch.Generator(device='cuda').manual_seed(3)
    commit_ms = []
    committed = 1
    cur = None
    for it in range(60):
        if it % 10 == 0:
            # hand b the pool a is about to commit: generate it synchronously with the same (seed, batch) recipe
            batch = (ref.batch - 1 if ref.thread is not None else ref.batch) if it else 0   # the batch being / about to be filled
            per = P // 3
            parts = [generate_arrays(lv, per if j < 2 else P - 2 * per, seed=9 * 1000003 + j, max_obst=mo, first_index=batch * P)
                     for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
            cur = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(5))
            pools.append(cur)
            if it:
                t0 = time.perf_counter()
                assert ref.poll(wait=True)
                commit_ms.append((time.perf_counter() - t0) * 1e3)
                committed += 1
            b.set_pool(cur + (None,))
        act = torch.rand((n, 2), device='cuda', generator=g, dtype=torch.float32) * 2 - 1
        a.step(act, auto_reset=True, fresh=True)
        b.step(act, auto_reset=True, fresh=True)
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(getattr(a, k), getattr(b, k)), (it, k)
        assert np.array_equal(a.pool_index(), b.pool_index())
    # maps of turned-over scenes come from the pool committed last
    idx = a.pool_index()
    turned = np.nonzero(idx >= 0)[0]
    assert len(turned) > 500
    s_, d_, bb_, v_, nob_ = a.download_scenes(turned[:200])
    which = np.array([[np.array_equal(s_[k], pl[0][idx[turned[k]]]) and nob_[k] == pl[4][idx[turned[k]]] for pl in pools] for k in range(200)])
    assert which.any(axis=1).all()                                    # every drawn map is an entry of a pool that was committed
    assert which[:, -1].any() and which[:, 0].sum() < 200             # ... of the newest one too: the swaps took effect
    print('pool commits:', committed, 'ms per commit incl. waiting for the generator thread:', [round(x, 2) for x in commit_ms],
          'generator s total:', round(ref.gen_seconds, 3))
    ref.close()
    a.close(); b.close()


def test_relaxed_pool_commit_followed_by_new_dlp_cases():
    """hope_env_set_dlp_cases (and the snapshot calls) must first apply a hope_env_commit_pool_relaxed swap still
    pending -- its class lists were built with the OLD cases.  Env a: relaxed commit of a new pool, then at once a (smaller) set of
    Dragon-Lake cases; env b: the same pool with the synchronous hope_env_set_pool, then the same cases.  Same generation, same
    draws, same outputs; a snapshot of a restores on b."""
    from hope_amd import ParkingBatch
    from hope_amd.scenes import DlpScenePool
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    n, mo, P = 2048, 128, 300
    init = mixed_arrays(n, seed=191, max_obst=mo)
    first = generate_arrays('Complex', P, seed=192, max_obst=mo)[:5]
    second = generate_arrays('Extrem', P, seed=193, max_obst=mo)[:5]
    full = DlpScenePool()
    few = DlpScenePool()                                            # the first 40 cases only: every list entry <= -2 must stay < 40
    few.case_set, few.dest = full.case_set[:40].copy(), full.dest[:40].copy()
    few.start_off = full.start_off[:41].copy()
    few.starts = full.starts[:int(few.start_off[-1])].copy()
    a = ParkingBatch(n, mo, obs_dtype=torch.float64)
    b = ParkingBatch(n, mo, obs_dtype=torch.float64)
    rng = np.random.default_rng(194)
    t0 = rng.integers(188, 200, n)
    for e in (a, b):
        e.set_scene_arrays(np.arange(n), *init[:5])
        e.set_draw_class(np.arange(3, n, 4), 1)
        e.set_dlp_cases(full)
        e.set_pool(first + (None,))
        e.reset_obs()
        e.upload_state(t=t0)
        e.set_redraw_seed(41)
    stage = a.pool_staging(P)
    for dst, src_ in zip(stage, second):
        dst[...] = src_
    a.commit_pool(P, relaxed=True)
    a.set_dlp_cases(few)                                           # applies the pending swap first
    b.set_pool(second + (None,))
    b.set_dlp_cases(few)
    assert a.pool_generation() == b.pool_generation()
    g = torch.Generator(device='cuda').manual_seed(19)
    names = ('lidar', 'action_mask', 'target', 'reward', 'status', 'done', 'pose', 'rs_word')
    for it in range(14):
        act = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
        a.step(act, auto_reset=True, fresh=True)
        b.step(act, auto_reset=True, fresh=True)
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(getattr(a, k), getattr(b, k)), (it, k)
        assert np.array_equal(a.pool_index(), b.pool_index())
    ia = a.pool_index()
    assert int((ia <= -2).sum()) > 50 and int((ia >= 0).sum()) > 200 and (-2 - ia[ia <= -2]).max() < 40
    assert a.pool_overflow() == 0
    a.close(); b.close()


def test_tie_census_no_decision_of_the_geos_slice_comes_near_a_tie():
    """the three predicates whose arithmetic the reference delegates to GEOS -- LinearRing.intersects
    (car_parking_base.py:153-158), Polygon.intersection().area / area > 0.95 (:164-170), LinearRing.distance(origin) < 10
    (lidar_simulator.py:69) -- are spec-checked, not fixture-pinned (no GEOS in the image).  A conformant-but-different GEOS could
    only flip a bit where the decision is a near tie: this census over > 8 x 10^6 scene-steps of the bench workload (instrumented
    step kernel, tools/tie_census.py; 2 x 10^8 scene-steps in profiles/r04_tie_census.txt) measures how near the workload comes.
    The oracle's own error bands, from tests/test_oracle_exact_geometry.py: area 1e-12 m^2 (1.1e-13 of the dest area), distance
    1e-13 m, orientation signs exact.  Needs its own process: the instrumented build is selected by an environment variable that
    the library reads once."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'tie_census.py'), '--scene-steps', '8.4e6'], capture_output=True, text=True,
                       cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    c = json.loads(r.stdout.strip().split('\n')[-1])
    print(r.stdout)
    assert c['scene_steps'] >= 8_000_000 and c['arrival_evals'] > 10_000 and c['ring_evals'] > 10_000_000
    # arrival: |ratio - 0.95| at least 10^4 x the oracle's band (1.1e-13); statistically ~1e-5 at this sample size
    assert c['arrival_min_abs_ratio_minus_0.95'] > 1e-9
    # ring cull: |distance - 10 m| at least 100 x the band (1e-13 m); ~1e-7 m expected at this sample size (uniform density)
    assert c['ring_min_abs_dist_minus_10'] > 1e-11
    # collision: the orientation filter decides every pair of the workload (exactly collinear / touching input would not)
    assert c['orientation_pairs_undecided_by_filter'] == 0
    # mask: the structural ties exist (a touching obstacle clips the scan to the hull range the table was built from) and are
    # IEEE arithmetic on both sides, not GEOS; they must stay a vanishing part of the workload
    assert c['mask_scene_steps_exact_path'] < 1e-5 * c['scene_steps']


def test_lidar_back_face_cull_changes_no_bit():
    """Round 4: the lidar drops the (beam, edge) pairs of the BACK edges of convex obstacles seen from outside (hope_step_kernel.h:
    the nearer front-edge hit is certain to pass the reference's tests, so the back edge can never be the beam's minimum,
    lidar_simulator.py:116-131).  With the cull switched off (stage bit 0x4000) every lidar value and every mask entry must be the
    same bit, in both launch forms, over mixed and Dragon-Lake scenes with turnover (besides the oracle comparisons of this file,
    which all run with the cull on)."""
    import os
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scene_gen import mixed_arrays
    for split in (False, True):
        if split:
            os.environ['HOPE_SPLIT_MIN'] = '1'
        try:
            n = 8192
            arrs = mixed_arrays(n, seed=91 + split, max_obst=128)
            envs = [ParkingBatch(n, 128, obs_dtype=torch.float64, overlap=True) for _ in range(2)]
            for e in envs:
                e.set_scene_arrays(np.arange(n), *arrs[:5])
                e.reset_obs()
            g = torch.Generator(device='cuda').manual_seed(17)
            for it in range(40):
                a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
                envs[0].step(a, auto_reset=True)
                envs[1].step(a, stages=L.STAGE_ALL | 0x4000, auto_reset=True)
                torch.cuda.synchronize()
                for k in ('lidar', 'action_mask', 'status', 'pose', 'reward', 'rs_word'):
                    assert torch.equal(getattr(envs[0], k), getattr(envs[1], k)), (split, it, k)
            for e in envs:
                e.close()
        finally:
            os.environ.pop('HOPE_SPLIT_MIN', None)


@pytest.mark.parametrize('f64', [True, False])
def test_two_scenes_per_wave_observation_equals_the_one_scene_kernel(f64):
    """Round 6: the motion and the observation launch of the small-tile class run TWO scenes per wavefront (k_motion_pair,
    hope_motion_pair.h; k_obs_pair, hope_obs_pair.h).  Against the one-scene kernels (stage bit 0x8000 selects them) every output
    and the episode state must be the same bit: an odd number
    of small-tile scenes (the last wave's second half is empty), lots of every level incl. Dragon-Lake lots that fit the small
    tile, episode turnover on new maps, a caller's `active` mask that silences one scene of many pairs, float64 and float32
    observations.  (The oracle comparisons of this file run the pair kernel wherever they use the two-launch form.)"""
    import os
    from hope_amd import ParkingBatch, _lib as L
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    os.environ['HOPE_SPLIT_MIN'] = '1'
    try:
        n = 6001
        arrs = mixed_arrays(n, seed=61, max_obst=128)
        small = int((arrs[4] <= 32).sum())
        parts = [generate_arrays(lv, 256, seed=63 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
        pool = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6))
        dt = torch.float64 if f64 else torch.float32
        envs = [ParkingBatch(n, 128, obs_dtype=dt, overlap=True) for _ in range(2)]
        for e in envs:
            e.set_scene_arrays(np.arange(n), *arrs[:5])
            e.set_pool(pool)
            e.set_dlp_cases()
            e.set_redraw_seed(5)
            e.reset_obs()
            e.upload_state(t=np.random.default_rng(62).integers(150, 200, n))
        g = torch.Generator(device='cuda').manual_seed(23)
        rng = np.random.default_rng(64)
        masked = 0
        for it in range(36):
            a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
            act = None
            if it % 3 == 2:
                m = rng.random(n) < 0.7
                act = torch.from_numpy(m.astype(np.uint8)).cuda()
                masked += int((~m).sum())
            envs[0].step(a, active=act, auto_reset=True, fresh=True, defer_rs=bool(it % 2))
            envs[1].step(a, active=act, stages=L.STAGE_ALL | 0x8000, auto_reset=True, fresh=True, defer_rs=bool(it % 2))
            for e in envs:
                e.wait_rs()
            torch.cuda.synchronize()
            for k in ('lidar', 'action_mask', 'status', 'done', 'pose', 'reward', 'reward_info', 'target', 'rs_word', 'rs_lengths'):
                assert torch.equal(getattr(envs[0], k), getattr(envs[1], k)), (f64, it, k)
            if it % 6 == 5:                                   # the episode state the motion launch keeps: pose, t, accum_arrive_reward
                for u, v in zip(envs[0].download_state(), envs[1].download_state()):
                    assert np.array_equal(u, v), (f64, it)
                assert np.array_equal(envs[0].n_obst_now(), envs[1].n_obst_now())
        lid = envs[0].lidar.cpu().numpy()
        assert small % 2 == 1
        assert masked > 5000 and float(lid.min()) < 1.0 and float(lid.max()) > 5.0
        print('pair kernel vs one-scene kernel: small-tile scenes', small, 'steps 36, masked scene-steps', masked)
        for e in envs:
            e.close()
    finally:
        os.environ.pop('HOPE_SPLIT_MIN', None)


@pytest.mark.parametrize('which', ['generated', 'dragon_lake'])
def test_single_class_batch_in_two_sub_chains_equals_the_single_chain(which):
    """A batch whose scenes all sit in one obstacle-tile class is cut into two sub-chains for deferred steps (hope_env_step: automatic
    from 32 768 small-tile scenes / up to 32 768 large-tile scenes; forced here at a small size with HOPE_AUTO_CHAINS) so that the
    two-stream pipelined form applies.  Same bits as the joined single-chain step on every output, every step, with episode
    turnover on new maps; and switching between the forms from step to step (deferred <-> joined) is safe."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays, generate_arrays
    n = 4096
    if which == 'generated':
        arrs = mixed_arrays(n, levels=('Normal', 'Complex', 'Extrem'), seed=31, max_obst=128)
    else:
        arrs = mixed_arrays(n, levels=('dlp',), seed=32, max_obst=128)
    parts = [generate_arrays(lv, 256, seed=33 + j, max_obst=128) for j, lv in enumerate(('Normal', 'Complex', 'Extrem'))]
    pool = tuple(np.concatenate([p_[j] for p_ in parts]) for j in range(6))
    os.environ['HOPE_AUTO_CHAINS'] = '1:1073741824:1:1073741824'
    os.environ['HOPE_SPLIT_MIN'] = '1'                       # the two-launch form of the step kernel at this size, as at 32 768 scenes
    try:
        envs = [ParkingBatch(n, 128, overlap=True) for _ in range(2)]
        for e in envs:
            e.set_scene_arrays(np.arange(n), *arrs[:5])
            if which == 'dragon_lake':
                e.set_draw_class(np.arange(n), 1)
            e.set_pool(pool)
            e.set_dlp_cases()
            e.set_redraw_seed(9)
            e.reset_obs()
        g = torch.Generator(device='cuda').manual_seed(13)
        names = ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths')
        found = turnovers = 0
        for it in range(30):
            a = torch.rand((n, 2), device='cuda', generator=g) * 2 - 1
            envs[0].step(a, auto_reset=True, fresh=True)                                      # joined: one chain
            envs[1].step(a, auto_reset=True, fresh=True, defer_rs=(it % 7 != 5))              # deferred: two sub-chains, pipelined; now and then a joined step in between
            if it % 3 == 0:
                envs[1].wait_rs()
                torch.cuda.synchronize()
                for k in names:
                    assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), (it, k)
                found += int((envs[1].rs_word[:, 6] > 0).sum())
                turnovers += int(envs[1].done.sum())
        envs[1].wait_rs()
        torch.cuda.synchronize()
        for k in names:
            assert torch.equal(getattr(envs[1], k), getattr(envs[0], k)), k
        assert found > 0 and turnovers > 0
        for e in envs:
            e.close()
    finally:
        del os.environ['HOPE_AUTO_CHAINS']
        del os.environ['HOPE_SPLIT_MIN']


@pytest.mark.parametrize('form', ['two_launch_pair_kernels', 'one_launch'])
def test_mask_exact_path_on_obstacles_that_touch_the_hull(form):
    """Round 6: the mask stage reads the count-interval table; a table entry within 1e-9 of a scan value must still send the scene to
    the exact 1200-beam evaluation (action_mask.py:166-177).  Such ties are structural -- an obstacle that reaches the hull clips the
    scan to the hull range the table was built from -- but rare in a rollout (125 in 2 x 10^8 scene-steps), so this test makes them:
    every scene gets an obstacle pushed onto / against its vehicle.  Masks (float64) must equal the oracle's on the reset observation
    and on the steps after it, in the pair kernel of the two-launch form and in the one-scene kernel of the one-launch form, and the
    premise is checked: hundreds of scenes have a coarse-beam scan within 1e-9 of a table entry."""
    import os
    from hope_amd import ParkingBatch
    from hope_amd.scene_gen import mixed_arrays
    from oracle import oracle as O
    os.environ['HOPE_SPLIT_MIN'] = '1' if form == 'two_launch_pair_kernels' else '1000000000'
    try:
        n, mo = 3000, 128
        start, dest, bbox, verts, nob, nvert = mixed_arrays(n, levels=('Normal', 'Complex', 'Extrem', 'dlp'), seed=71, max_obst=mo)
        rng = np.random.default_rng(72)
        mid = 0.5 * (3.76 - 0.93)
        for k in range(n):
            if nob[k] == 0:
                continue
            j = int(rng.integers(nob[k]))
            c, s_ = np.cos(start[k, 2]), np.sin(start[k, 2])
            # where the obstacle's centroid goes: anywhere from deep inside the hull to just beyond its outline
            ahead, side = rng.uniform(-3.0, 3.0), rng.uniform(-1.6, 1.6)
            tx = start[k, 0] + (mid + ahead) * c - side * s_
            ty = start[k, 1] + (mid + ahead) * s_ + side * c
            cen = verts[k, j].mean(axis=0)
            verts[k, j] += np.array([tx, ty]) - cen
        env = ParkingBatch(n, mo, obs_dtype=torch.float64, action_dtype=torch.float64, overlap=True)
        env.set_scene_arrays(np.arange(n), start, dest, bbox, verts, nob)
        orc = O.BatchOracle(n, mo, omp=True, track_traj=False)
        orc.set_scenes(np.arange(n), start, dest, bbox, verts, nvert, nob)
        t = env.tables
        O.set_tables(hull_base=t['hull_base'], beam_a=t['beam_ab'][:, 0], beam_b=t['beam_ab'][:, 1], dist_star=t['dist_star'], omp=True)
        tab = np.maximum.accumulate(t['dist_star'][::10], axis=2)            # coarse rows [120][42][10]
        ties = 0

        def check(o, tag):
            nonlocal ties
            torch.cuda.synchronize()
            m = env.action_mask.cpu().numpy()
            lid = env.lidar.cpu().numpy()
            assert np.array_equal(lid, o['lidar']), tag
            bad = np.nonzero((m != o['mask']).any(axis=1))[0]
            assert len(bad) == 0, (tag, bad[:8], m[bad[:2]], o['mask'][bad[:2]])
            x = np.clip(lid, 0, 10) + t['hull_base'][None]
            near = np.abs(tab[None] - x[:, :, None, None]) <= 1e-9            # [n, 120, 42, 10]
            ties += int(near.any(axis=(1, 2, 3)).sum())

        env.reset_obs()
        check(orc.reset_obs(with_rs=False), 'reset')
        for it in range(4):
            act = rng.uniform(-1, 1, (n, 2))
            env.step(torch.from_numpy(act).to(env.device))
            check(orc.step(act, with_rs=False), it)
        print(f'mask exact path ({form}): scene-observations with a table entry within 1e-9 of a coarse-beam scan: {ties}')
        assert ties > 300
        env.close()
    finally:
        os.environ.pop('HOPE_SPLIT_MIN', None)
