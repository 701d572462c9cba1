"""Error paths of the C ABI (include/hope_env.h), driven through ctypes the way a foreign binding would: every documented
misuse returns its code (never a crash, never an exception across the ABI), hope_last_error() says why, and the handle is
still usable afterwards -- the abused handle's next step equals a fresh twin's bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from hope_amd import _lib as L

pytestmark = pytest.mark.gpu

OK, EINVAL, ENODEV, ENOMEM, EHIP, ESTATE = 0, -1, -2, -3, -4, -5
N, MO = 256, 64


def _lib():
    return L.load_library()


def _err():
    return _lib().hope_last_error().decode(errors='replace')


def _create(flags=L.F_OVERLAP, n=N, mo=MO):
    h = C.c_void_p()
    rc = _lib().hope_env_create(C.byref(h), n, mo, 0, flags)
    assert rc == OK, _err()
    return h


def _tables(h):
    from hope_amd import tables as T
    t = T.all_tables()
    ds = np.ascontiguousarray(t['dist_star'], np.float64)
    hb = np.ascontiguousarray(t['hull_base'], np.float64)
    ab = np.ascontiguousarray(t['beam_ab'], np.float64)
    return _lib().hope_env_upload_tables(h, ds.ctypes.data, hb.ctypes.data, ab.ctypes.data)


def _scene_arrays(n=N, mo=MO, seed=3):
    from hope_amd.scene_gen import mixed_arrays
    start, dest, bbox, verts, nob, nvert = mixed_arrays(n, seed=seed, max_obst=mo, levels=('Normal', 'Complex', 'Extrem'))
    return [np.ascontiguousarray(a) for a in (start, dest, bbox, verts, nob.astype(np.int32))]


def _set_scenes(h, arrays, ids=None, nob=None, n=None):
    start, dest, bbox, verts, nob0 = arrays
    ids = np.arange(len(nob0), dtype=np.int32) if ids is None else np.asarray(ids, np.int32)
    nob = nob0 if nob is None else np.asarray(nob, np.int32)
    return _lib().hope_env_set_scenes(h, ids.ctypes.data, len(ids) if n is None else n, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data,
                                      verts.ctypes.data, nob.ctypes.data)


class Outs:
    def __init__(self, n=N, image=False):
        dev = 'cuda:0'
        self.t = dict(lidar=torch.zeros((n, 120), device=dev), action_mask=torch.zeros((n, 42), device=dev), target=torch.zeros((n, 5), device=dev),
                      reward=torch.zeros(n, device=dev), reward_info=torch.zeros((n, 5), device=dev), status=torch.zeros(n, dtype=torch.int32, device=dev),
                      done=torch.zeros(n, dtype=torch.uint8, device=dev), pose=torch.zeros((n, 3), dtype=torch.float64, device=dev),
                      rs_word=torch.zeros((n, 8), dtype=torch.int8, device=dev), rs_lengths=torch.zeros((n, 5), device=dev),
                      img=torch.zeros((n, 3, 64, 64), dtype=torch.uint8, device=dev) if image else None)
        self.s = L.StepOut(*[C.c_void_p(self.t[k].data_ptr()) if self.t[k] is not None else None
                             for k in ('lidar', 'action_mask', 'target', 'reward', 'reward_info', 'status', 'done', 'pose', 'rs_word', 'rs_lengths', 'img')])


def _step(h, outs, actions, stages=L.STAGE_ALL, active=None):
    return _lib().hope_env_step(h, C.c_void_p(actions.data_ptr()) if actions is not None else None, active, stages,
                                C.byref(outs.s) if outs is not None else None, None)


def test_create_rejects_bad_arguments_and_removed_flags():
    lib = _lib()
    h = C.c_void_p()
    assert lib.hope_env_create(None, 16, 16, 0, 0) == EINVAL
    assert lib.hope_env_create(C.byref(h), 0, 16, 0, 0) == EINVAL
    assert lib.hope_env_create(C.byref(h), 16, 0, 0, 0) == EINVAL
    assert lib.hope_env_create(C.byref(h), 16, 16, 99, 0) == EINVAL and 'device_id' in _err()
    assert lib.hope_env_create(C.byref(h), 16, 16, 0, 0x20) == EINVAL and 'ABI 7' in _err()          # hipGraph replay: removed
    # each limit is named on its own: the 8-bit obstacle count of the Reeds-Shepp queue entry, not the LDS tile,
    # is what bounds max_obstacles
    assert lib.hope_env_create(C.byref(h), 16, 100000, 0, 0) == EINVAL and 'HOPE_MAX_OBSTACLES (255)' in _err()
    assert lib.hope_env_create(C.byref(h), 16, 300, 0, 0) == EINVAL and 'HOPE_MAX_OBSTACLES' in _err() and 'LDS' not in _err()
    assert lib.hope_env_create(C.byref(h), 1 << 24, 16, 0, 0) == EINVAL and '2^24' in _err()
    assert not h.value


@pytest.mark.parametrize('case', ['step_before_tables', 'step_before_scenes', 'n_obst_too_large', 'scene_id_out_of_range', 'null_arrays',
                                  'img_without_flag', 'redraw_without_reset', 'redraw_without_pool', 'pool_without_class', 'null_out',
                                  'null_actions', 'profile_without_flag', 'stale_wait_rs_step', 'null_masks', 'commit_beyond_staging',
                                  'bad_draw_class', 'restore_without_pool'])
def test_misuse_returns_its_code_and_leaves_the_handle_usable(case):
    lib = _lib()
    arrays = _scene_arrays()
    g = torch.Generator(device='cuda:0').manual_seed(5)
    acts = [torch.rand((N, 2), device='cuda:0', generator=g) * 2 - 1 for _ in range(3)]
    h = _create()
    outs = Outs()
    ready = case not in ('step_before_tables', 'step_before_scenes')
    if ready:
        assert _tables(h) == OK and _set_scenes(h, arrays) == OK

    if case == 'step_before_tables':
        assert _step(h, outs, acts[0]) == ESTATE and 'upload_tables' in _err()
        assert lib.hope_env_reset_obs(h, None, L.STAGE_ALL, C.byref(outs.s), None) == ESTATE
        assert _tables(h) == OK
        assert _step(h, outs, acts[0]) == ESTATE and 'set_scenes' in _err()
        assert _set_scenes(h, arrays) == OK
    elif case == 'step_before_scenes':
        assert _tables(h) == OK
        assert _step(h, outs, acts[0]) == ESTATE and 'set_scenes' in _err()
        mask = torch.ones(N, dtype=torch.uint8, device='cuda:0')
        assert lib.hope_env_restart(h, C.c_void_p(mask.data_ptr()), None) == ESTATE
        assert _set_scenes(h, arrays) == OK
    elif case == 'n_obst_too_large':
        nob = arrays[4].copy()
        nob[7] = MO + 1
        assert _set_scenes(h, arrays, nob=nob) == EINVAL and 'max_obstacles' in _err()
        nob[7] = -1
        assert _set_scenes(h, arrays, nob=nob) == EINVAL
    elif case == 'scene_id_out_of_range':
        ids = np.arange(N, dtype=np.int32)
        ids[3] = N
        assert _set_scenes(h, arrays, ids=ids) == EINVAL and 'out of range' in _err()
        ids[3] = -1
        assert _set_scenes(h, arrays, ids=ids) == EINVAL
        buf = np.zeros(3)
        assert lib.hope_env_download_scenes(h, np.array([N], np.int32).ctypes.data, 1, buf.ctypes.data, None, None, None, None) == EINVAL
    elif case == 'null_arrays':
        ids = np.arange(N, dtype=np.int32)
        start, dest, bbox, verts, nob = arrays
        assert lib.hope_env_set_scenes(h, ids.ctypes.data, N, None, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data) == EINVAL
        assert lib.hope_env_set_scenes(h, ids.ctypes.data, N, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, None, nob.ctypes.data) == EINVAL
        assert lib.hope_env_set_scenes(h, ids.ctypes.data, -1, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data) == EINVAL
        assert lib.hope_env_upload_tables(h, None, None, None) == EINVAL
        assert lib.hope_env_download_n_obst(h, None) == EINVAL
        assert lib.hope_env_last_step(h, None) == EINVAL
    elif case == 'img_without_flag':
        oi = Outs(image=True)
        assert _step(h, oi, acts[0], stages=L.STAGE_ALL | L.STAGE_IMG) == ESTATE and 'HOPE_F_IMAGE' in _err()
        h2 = _create(flags=L.F_OVERLAP | L.F_IMAGE)
        assert _tables(h2) == OK and _set_scenes(h2, arrays) == OK
        assert _step(h2, outs, acts[0], stages=L.STAGE_ALL | L.STAGE_IMG) == EINVAL and 'img' in _err()      # out->img is NULL
        assert _step(h2, oi, acts[0], stages=L.STAGE_ALL | L.STAGE_IMG) == OK
        assert lib.hope_env_destroy(h2) == OK
    elif case == 'redraw_without_reset':
        assert _step(h, outs, acts[0], stages=L.STAGE_ALL | L.AUTO_REDRAW) == EINVAL and 'HOPE_AUTO_RESET' in _err()
    elif case == 'redraw_without_pool':
        assert _step(h, outs, acts[0], stages=L.STAGE_ALL | L.AUTO_RESET | L.AUTO_REDRAW) == ESTATE and 'pool' in _err()
        mask = torch.ones(N, dtype=torch.uint8, device='cuda:0')
        assert lib.hope_env_redraw(h, C.c_void_p(mask.data_ptr()), 1, None) == ESTATE
    elif case == 'pool_without_class':
        # every resident slot is of the small class; a pool of large lots only cannot serve them
        big = _scene_arrays(n=8, seed=9)
        start, dest, bbox, verts, nob = big
        nob = np.full(8, 40, np.int32)                         # (> 32 obstacles: the large class; the vertex data is irrelevant here)
        assert lib.hope_env_set_pool(h, 8, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data) == OK
        assert _step(h, outs, acts[0], stages=L.STAGE_ALL | L.AUTO_RESET | L.AUTO_REDRAW) == ESTATE and 'size class' in _err()
        nob[:] = MO + 5
        assert lib.hope_env_set_pool(h, 8, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data) == EINVAL
        assert lib.hope_env_set_pool(h, 0, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data) == EINVAL
    elif case == 'null_out':
        assert _step(h, None, acts[0]) == EINVAL
        assert lib.hope_env_reset_obs(h, None, L.STAGE_ALL, None, None) == EINVAL
        assert lib.hope_env_step(None, C.c_void_p(acts[0].data_ptr()), None, L.STAGE_ALL, C.byref(outs.s), None) == EINVAL
    elif case == 'null_actions':
        assert _step(h, outs, None) == EINVAL and 'actions' in _err()
    elif case == 'profile_without_flag':
        ms = np.zeros(len(L.KERNELS))
        cnt = np.zeros(len(L.KERNELS), np.int64)
        assert lib.hope_env_kernel_ms(h, ms.ctypes.data, cnt.ctypes.data, 1) == ESTATE and 'HOPE_F_PROFILE' in _err()
        assert lib.hope_env_kernel_union_ms(h, ms.ctypes.data, cnt.ctypes.data, 1) == ESTATE
        assert lib.hope_env_profile_kernels(h, 1) == ESTATE
    elif case == 'stale_wait_rs_step':
        seq = C.c_uint64(0)
        assert _step(h, outs, acts[0], stages=L.STAGE_ALL | L.DEFER_RS) == OK
        assert lib.hope_env_last_step(h, C.byref(seq)) == OK and seq.value >= 1
        first = seq.value
        assert _step(h, outs, acts[1], stages=L.STAGE_ALL | L.DEFER_RS) == OK
        assert lib.hope_env_wait_rs_step(h, first, None) == ESTATE and 'replaced' in _err()       # step k's words are gone
        assert lib.hope_env_wait_rs_step(h, first + 5, None) == EINVAL
        assert lib.hope_env_wait_rs_step(h, 0, None) == EINVAL
        assert lib.hope_env_wait_rs_step(h, first + 1, None) == OK
        assert lib.hope_env_wait_rs(h, None) == OK                                               # (nothing outstanding any more: no-op)
    elif case == 'null_masks':
        assert lib.hope_env_restart(h, None, None) == EINVAL
        assert lib.hope_env_redraw(h, None, 0, None) == EINVAL
        assert lib.hope_env_set_draw_class(h, None, 3, None) == EINVAL
        assert lib.hope_env_pool_overflow(h, None) == EINVAL
        assert lib.hope_env_pool_generation(h, None) == EINVAL
    elif case == 'commit_beyond_staging':
        ptrs = [C.c_void_p() for _ in range(5)]
        assert lib.hope_env_pool_staging(h, 4, *[C.byref(p_) for p_ in ptrs]) == OK
        assert lib.hope_env_commit_pool(h, 64, None) == ESTATE and 'staging' in _err()
        assert lib.hope_env_commit_pool(h, 0, None) == EINVAL
        assert lib.hope_env_pool_staging(h, 0, *[C.byref(p_) for p_ in ptrs]) == EINVAL
    elif case == 'bad_draw_class':
        ids = np.array([N + 3], np.int32)
        cls = np.array([1], np.uint8)
        assert lib.hope_env_set_draw_class(h, ids.ctypes.data, 1, cls.ctypes.data) == EINVAL
    elif case == 'restore_without_pool':
        drawn = np.zeros(N, np.uint8)
        ep = np.zeros(N, np.uint32)
        assert lib.hope_env_restore_maps(h, drawn.ctypes.data, ep.ctypes.data, 0, 0) == ESTATE
        assert lib.hope_env_restore_maps(h, None, ep.ctypes.data, 0, 0) == EINVAL

    # ---- the handle still works: its next steps equal a fresh twin's -------------------------------------------------------------
    if case == 'pool_without_class':
        pass                                                   # (the pool stays; plain steps do not draw from it)
    twin = _create()
    o2 = Outs()
    assert _tables(twin) == OK and _set_scenes(twin, arrays) == OK
    # bring both to the same episode state: the abused handle may have stepped (stale_wait_rs_step): restart both
    mask = torch.ones(N, dtype=torch.uint8, device='cuda:0')
    for hh, oo in ((h, outs), (twin, o2)):
        assert lib.hope_env_restart(hh, C.c_void_p(mask.data_ptr()), None) == OK
        assert lib.hope_env_reset_obs(hh, None, L.STAGE_ALL, C.byref(oo.s), None) == OK
        for a in acts:
            assert _step(hh, oo, a) == OK
    torch.cuda.synchronize()
    for k, v in outs.t.items():
        if v is not None:
            assert torch.equal(v, o2.t[k]), (case, k)
    assert lib.hope_env_destroy(twin) == OK
    assert lib.hope_env_destroy(h) == OK


def test_destroy_is_idempotent_for_null_and_refuses_a_dead_handle():
    lib = _lib()
    assert lib.hope_env_destroy(None) == OK
    h = _create(n=16, mo=16)
    raw = C.c_void_p(h.value)
    assert lib.hope_env_destroy(h) == OK
    assert lib.hope_env_destroy(raw) == EINVAL and 'live handle' in _err()          # double destroy: refused, nothing touched
    outs = Outs(n=16)
    a = torch.zeros((16, 2), device='cuda:0')
    assert _step(raw, outs, a) == EINVAL and 'live handle' in _err()               # use after destroy: refused


def test_n_obst_download_matches_the_per_scene_download():
    lib = _lib()
    arrays = _scene_arrays()
    h = _create()
    assert _tables(h) == OK and _set_scenes(h, arrays) == OK
    out = np.zeros(N, np.int32)
    assert lib.hope_env_download_n_obst(h, out.ctypes.data) == OK
    assert np.array_equal(out, arrays[4])
    assert lib.hope_env_destroy(h) == OK
