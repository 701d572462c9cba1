"""ParkingBatch -- N independent parking scenes stepped on one MI355X through libhope_env.so.

Tensor-in / tensor-out counterpart of `CarParkingWrapper.step` (src/env/env_wrapper.py:73-81) for N >> 1.
PyTorch is used for device memory and streams only; all arithmetic is in the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import tables as T
from .scenes import pack_scenes


class ParkingBatch:
    def __init__(self, n_scenes, max_obstacles=128, device='cuda:0', obs_dtype=torch.float32,
                 action_dtype=torch.float32, tables=None, profile=False, image=False, overlap=None,
                 rescale_f32=False):
        """overlap: run the launch chains of the two obstacle-tile classes on two streams (default on: +15 % at 65 536
        scenes, +45 % at 4 096-8 192, measured).  rescale_f32: with float32 actions, evaluate action_rescale in float32 as the reference does with gym's float32 Box
        (env_wrapper.py:46-47); default: float64 arithmetic whatever the action dtype."""
        if not torch.cuda.is_available():
            raise L.HopeError('ParkingBatch needs a HIP device (torch.cuda.is_available() is False); no CPU fallback')
        self.lib = L.load_library()
        self.device = torch.device(device)
        if self.device.index is None:              # 'cuda' -> the current device, so that tensor.device == self.device
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.n, self.max_obst = int(n_scenes), int(max_obstacles)
        assert obs_dtype in (torch.float32, torch.float64) and action_dtype in (torch.float32, torch.float64)
        self.obs_dtype, self.action_dtype = obs_dtype, action_dtype
        if overlap is None:
            overlap = True
        flags = (L.F_OBS_F64 if obs_dtype == torch.float64 else 0) | (L.F_ACTION_F64 if action_dtype == torch.float64 else 0) | (L.F_PROFILE if profile else 0) | (L.F_IMAGE if image else 0) | (L.F_OVERLAP if overlap else 0)
        self.overlap = bool(overlap)
        self.image = bool(image)
        self._action_bits = L.ACTION_RESCALE_F32 if (rescale_f32 and action_dtype == torch.float32) else 0
        h = C.c_void_p()
        L.check(self.lib.hope_env_create(C.byref(h), self.n, self.max_obst, self.device.index or 0, flags),
                'hope_env_create')
        self.h = h
        self.arch = self.lib.hope_env_device_arch(self.h).decode()
        t = T.all_tables() if tables is None else tables
        self.tables = t
        ds, hb, ab = (np.ascontiguousarray(t[k], dtype=np.float64) for k in ('dist_star', 'hull_base', 'beam_ab'))
        assert ds.shape == (1200, 42, 10) and hb.shape == (120,) and ab.shape == (120, 2)
        L.check(self.lib.hope_env_upload_tables(self.h, ds.ctypes.data, hb.ctypes.data, ab.ctypes.data),
                'hope_env_upload_tables')
        n, dev, od = self.n, self.device, obs_dtype
        # every output lives in ONE device arena (typed views at 256-byte aligned offsets): `download_outputs()` brings all of a
        # step's results to the host with a single copy -- what the N = 1 look-alike classes (hope_amd/env.py) need per step
        spec = [('lidar', (n, L.LIDAR_NUM), od), ('action_mask', (n, L.N_ACTION), od), ('target', (n, L.TARGET_DIM), od),
                ('reward', (n,), od), ('reward_info', (n, 5), od), ('status', (n,), torch.int32), ('done', (n,), torch.uint8),
                ('pose', (n, 3), torch.float64), ('rs_word', (n, 8), torch.int8), ('rs_lengths', (n, L.RS_MAX_SEG), od)]
        if image:       # obs['img'] * 255 as uint8, channel-first (env_wrapper.py:53-54); the reference's float image is img / 255
            spec.append(('img', (n, L.IMG_CHANNELS, L.IMG_SIZE, L.IMG_SIZE), torch.uint8))
        off, self._layout = 0, {}
        for name, shape, dt in spec:
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._layout[name] = (off, nbytes, shape, dt)
            off = (off + nbytes + 255) & ~255
        self._arena = torch.zeros(off, dtype=torch.uint8, device=dev)
        self._arena_host = None
        for name, (o, nbytes, shape, dt) in self._layout.items():
            setattr(self, name, self._arena[o:o + nbytes].view(dt).view(shape))
        self.rs_word.fill_(-1)
        if not image:
            self.img = None
        self._done_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        # observation-only view for turnover(): the finished step's reward / status / done / RS outputs are kept
        self._out_obs = L.StepOut(self.lidar.data_ptr(), self.action_mask.data_ptr(), self.target.data_ptr(), None, None, None,
                                  None, self.pose.data_ptr(), None, None, self.img.data_ptr() if image else None)
        self._out = L.StepOut(self.lidar.data_ptr(), self.action_mask.data_ptr(), self.target.data_ptr(),
                              self.reward.data_ptr(), self.reward_info.data_ptr(), self.status.data_ptr(),
                              self.done.data_ptr(), self.pose.data_ptr(), self.rs_word.data_ptr(),
                              self.rs_lengths.data_ptr(), self.img.data_ptr() if image else None)

    # -- scenes ------------------------------------------------------------------------------------
    def set_scenes(self, ids, scenes):
        """map.reset + vehicle.reset for the listed scene slots (host -> device; synchronous)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        start, dest, bbox, verts, nob, _nv = pack_scenes(scenes, self.max_obst)
        self.set_scene_arrays(ids, start, dest, bbox, verts, nob)

    def set_draw_class(self, ids, cls):
        """size class of scene slots (0: lots of <= 32 obstacles, 1: larger lots / Dragon-Lake cases): which launch chain steps a
        slot and which pool entries it draws at turnover; set_scenes derives it from the obstacle count of the uploaded map"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(cls, dtype=np.uint8), ids.shape))
        torch.cuda.synchronize(self.device)
        L.check(self.lib.hope_env_set_draw_class(self.h, ids.ctypes.data, len(ids), c.ctypes.data), 'hope_env_set_draw_class')
        return self

    def set_scene_arrays(self, ids, start, dest, bbox, verts, n_obst):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (start, dest, bbox, verts)]
        nob = np.ascontiguousarray(n_obst, dtype=np.int32)
        assert a[3].shape == (len(ids), self.max_obst, 4, 2)
        torch.cuda.synchronize(self.device)
        L.check(self.lib.hope_env_set_scenes(self.h, ids.ctypes.data, len(ids), a[0].ctypes.data, a[1].ctypes.data,
                                             a[2].ctypes.data, a[3].ctypes.data, nob.ctypes.data),
                'hope_env_set_scenes')

    def set_pool(self, scenes):
        """upload a pool of complete scenes (list[Scene] or the tuple pack_scenes returns) that redraw() draws from"""
        start, dest, bbox, verts, nob, _nv = pack_scenes(scenes, self.max_obst) if isinstance(scenes, list) else scenes
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (start, dest, bbox, verts)]
        nob = np.ascontiguousarray(nob, dtype=np.int32)
        if getattr(self, '_refresher_filling', False):
            raise L.HopeError('set_pool while a PoolRefresher fill is running: the pinned staging arrays are being written by its '
                              'thread (close() / poll(wait=True) the refresher first)')
        torch.cuda.synchronize(self.device)
        L.check(self.lib.hope_env_set_pool(self.h, len(nob), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                           a[3].ctypes.data, nob.ctypes.data), 'hope_env_set_pool')
        self.pool_size = len(nob)

    def pool_staging(self, n_pool):
        """pinned host arrays (start[n,3], dest[n,3], bbox[n,4], verts[n,max_obst,4,2], n_obst[n]) of the handle that the next
        commit_pool(n_pool) uploads; any host thread may fill them (e.g. the native generator, GIL released)"""
        ptr = [C.c_void_p() for _ in range(5)]
        L.check(self.lib.hope_env_pool_staging(self.h, int(n_pool), *[C.byref(q) for q in ptr]), 'hope_env_pool_staging')
        shapes = [(n_pool, 3), (n_pool, 3), (n_pool, 4), (n_pool, self.max_obst, 4, 2), (n_pool,)]
        out = []
        for q, shp, ct in zip(ptr, shapes, (C.c_double,) * 4 + (C.c_int32,)):
            out.append(np.ctypeslib.as_array(C.cast(q, C.POINTER(ct)), shape=(int(np.prod(shp)),)).reshape(shp))
        return tuple(out)

    def pool_staging_ready(self):
        """True when pool_staging() would not wait (the previous commit's copies have left the pinned arrays)"""
        rc = self.lib.hope_env_pool_staging_ready(self.h)
        if rc < 0:
            L.check(rc, 'hope_env_pool_staging_ready')
        return bool(rc)

    def pool_generation(self):
        """changes whenever the set of maps a draw can return changes (set_pool / commit_pool / set_dlp_cases)"""
        g = C.c_uint64(0)
        L.check(self.lib.hope_env_pool_generation(self.h, C.byref(g)), 'hope_env_pool_generation')
        return int(g.value)

    def commit_pool(self, n_pool, relaxed=False):
        """asynchronous upload of the staged pool + swap: steps enqueued afterwards draw from it (no host synchronisation; the next
        step's launch waits on its stream for the upload).  relaxed=True (hope_env_commit_pool_relaxed): the swap happens at the first
        step enqueued after the upload has completed -- no step waits at all"""
        if relaxed:
            L.check(self.lib.hope_env_commit_pool_relaxed(self.h, int(n_pool)), 'hope_env_commit_pool_relaxed')
        else:
            L.check(self.lib.hope_env_commit_pool(self.h, int(n_pool), self._stream()), 'hope_env_commit_pool')
        self.pool_size = int(n_pool)
        return self

    def set_dlp_cases(self, pool=None):
        """make the Dragon-Lake-Parking cases (a `DlpScenePool`, default data/dlp_scenes.npz) drawable on the device: at episode
        turnover a large-tile scene then gets a case with a freshly drawn start candidate, jitter, flips and obstacle cull
        (ParkingMapDLP.reset) instead of a frozen host-side sample.  pool=False removes them."""
        if pool is False:
            L.check(self.lib.hope_env_set_dlp_cases(self.h, 0, None, None, None, None, 0, None, None), 'hope_env_set_dlp_cases')
            return self
        from .scenes import DlpScenePool
        pool = pool or DlpScenePool()
        dest = np.ascontiguousarray(pool.dest, dtype=np.float64)
        cand_off = np.ascontiguousarray(pool.start_off, dtype=np.int32)
        cand = np.ascontiguousarray(pool.starts, dtype=np.float64)
        case_set = np.ascontiguousarray(pool.case_set, dtype=np.int32)
        set_off = np.ascontiguousarray(pool.set_off, dtype=np.int32)
        sv = np.array(pool.set_verts, dtype=np.float64)          # [total][4][2]; triangles repeat their last vertex
        tri = np.asarray(pool.set_nvert) == 3
        sv[tri, 3] = sv[tri, 2]
        sv = np.ascontiguousarray(sv)
        torch.cuda.synchronize(self.device)
        L.check(self.lib.hope_env_set_dlp_cases(self.h, len(dest), dest.ctypes.data, cand_off.ctypes.data, cand.ctypes.data,
                                                case_set.ctypes.data, len(set_off) - 1, set_off.ctypes.data, sv.ctypes.data),
                'hope_env_set_dlp_cases')
        self.n_dlp_cases = len(dest)
        return self

    def pool_overflow(self):
        out = np.zeros(1, np.int32)
        L.check(self.lib.hope_env_pool_overflow(self.h, out.ctypes.data), 'hope_env_pool_overflow')
        return int(out[0])

    def download_scenes(self, ids):
        """the maps the listed scene slots hold now -> (start, dest, bbox, verts, n_obst) (host arrays)"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        n = len(ids)
        start, dest, bbox = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 4))
        verts, nob = np.zeros((n, self.max_obst, 4, 2)), np.zeros(n, np.int32)
        L.check(self.lib.hope_env_download_scenes(self.h, ids.ctypes.data, n, start.ctypes.data, dest.ctypes.data, bbox.ctypes.data,
                                                  verts.ctypes.data, nob.ctypes.data), 'hope_env_download_scenes')
        return start, dest, bbox, verts, nob

    def pool_state(self):
        """(pool index, episode counter) of every scene: with download_state() and pool_generation() the whole snapshot of a
        run that draws maps from a pool that is NOT replaced in the meantime (refreshed runs: download_scenes / set_scene_arrays)"""
        idx, ep = np.zeros(self.n, np.int32), np.zeros(self.n, np.uint32)
        L.check(self.lib.hope_env_download_pool_state(self.h, idx.ctypes.data, ep.ctypes.data), 'hope_env_download_pool_state')
        return idx, ep

    def restore_maps(self, pool_index, episode, seed, generation=0):
        """repeat the draws of a snapshot (the seed in use then); `generation` = pool_generation() saved with the snapshot: the
        call fails (HOPE_ESTATE) if the pool has been replaced since -- the draws would return other maps; 0 skips the check.
        Follow with upload_state()"""
        drawn = np.ascontiguousarray(np.asarray(pool_index) != -1, dtype=np.uint8)
        ep = np.ascontiguousarray(episode, dtype=np.uint32)
        L.check(self.lib.hope_env_restore_maps(self.h, drawn.ctypes.data, ep.ctypes.data, C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                               C.c_uint64(int(generation))), 'hope_env_restore_maps')
        return self

    def redraw(self, mask, seed=0):
        """map.reset for the flagged scenes: each takes a pool scene of its tile class and restarts (asynchronous)"""
        assert mask.dtype == torch.uint8 and mask.shape == (self.n,) and mask.device == self.device
        L.check(self.lib.hope_env_redraw(self.h, C.c_void_p(mask.data_ptr()), C.c_uint64(int(seed) & (2 ** 64 - 1)), self._stream()),
                'hope_env_redraw')
        return self

    def turnover(self, seed=0):
        """episode turnover with a NEW map, as the reference's training loop does (`env.reset(...)` after `done`):
        finished scenes draw a pool scene and get their first observation; reward / reward_info / status / done / RS
        outputs keep the values of the finished step (like auto_reset)."""
        self._done_mask.copy_(self.done)
        self.redraw(self._done_mask, seed)
        stages = L.STAGE_ALL | (L.STAGE_IMG if self.image else 0)
        L.check(self.lib.hope_env_reset_obs(self.h, C.c_void_p(self._done_mask.data_ptr()), stages, C.byref(self._out_obs),
                                            self._stream()), 'hope_env_reset_obs')
        return self

    def pool_index(self):
        out = np.zeros(self.n, np.int32)
        L.check(self.lib.hope_env_download_pool_index(self.h, out.ctypes.data), 'hope_env_download_pool_index')
        return out

    # -- the hot path ------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset_obs(self, active=None, stages=L.STAGE_ALL):
        """the action-less step of CarParking.reset (car_parking_base.py:138)."""
        if self.image and (stages & L.STAGE_ALL) == L.STAGE_ALL:
            stages |= L.STAGE_IMG
        ap = C.c_void_p(active.data_ptr()) if active is not None else None
        L.check(self.lib.hope_env_reset_obs(self.h, ap, stages, C.byref(self._out), self._stream()),
                'hope_env_reset_obs')
        return self

    def set_redraw_seed(self, seed):
        """seed of the draws of step(..., auto_reset=True, fresh=True)"""
        L.check(self.lib.hope_env_set_redraw_seed(self.h, C.c_uint64(int(seed) & (2 ** 64 - 1))), 'hope_env_set_redraw_seed')
        return self

    def step(self, actions, active=None, stages=L.STAGE_ALL, auto_reset=False, fresh=False, defer_rs=False):
        """actions: [N, 2] (steer, speed) in [-1, 1] on this device.  auto_reset=True: finished scenes restart inside
        the step (their lidar / action_mask / target are the new episode's first observation) -- on the same map, or with
        fresh=True on a NEW map drawn from the device-resident scene pool (set_pool), like step() + turnover() in one call.
        defer_rs=True (HOPE_DEFER_RS): the current stream is ordered after the observation / reward / status outputs only;
        call wait_rs() before reading rs_word / rs_lengths (the planner override), the policy forward in between overlaps the
        Reeds-Shepp kernels."""
        if defer_rs:
            stages |= L.DEFER_RS
        if self.image and (stages & L.STAGE_ALL) == L.STAGE_ALL:
            stages |= L.STAGE_IMG                    # USE_IMG (configs.py:100): the image is part of the observation
        if auto_reset:
            stages |= L.AUTO_RESET | (L.AUTO_REDRAW if fresh else 0)
        stages |= self._action_bits
        assert actions.shape == (self.n, 2) and actions.dtype == self.action_dtype and actions.is_contiguous()
        assert actions.device == self.device
        ap = C.c_void_p(active.data_ptr()) if active is not None else None
        L.check(self.lib.hope_env_step(self.h, C.c_void_p(actions.data_ptr()), ap, stages, C.byref(self._out),
                                       self._stream()), 'hope_env_step')
        return self

    def last_step(self):
        """sequence number of the most recent step() / reset_obs() of this handle (hope_env_last_step)"""
        g = C.c_uint64(0)
        L.check(self.lib.hope_env_last_step(self.h, C.byref(g)), 'hope_env_last_step')
        return int(g.value)

    def wait_rs(self, step=None):
        """orders the current stream after the Reeds-Shepp outputs of the last step(defer_rs=True); no-op otherwise.
        step = a value of last_step(): the wait is for THAT step's search and raises (HOPE_ESTATE) when a newer step has been
        enqueued since -- consecutive deferred steps pipeline and the newer one replaces rs_word / rs_lengths."""
        if step is None:
            L.check(self.lib.hope_env_wait_rs(self.h, self._stream()), 'hope_env_wait_rs')
        else:
            L.check(self.lib.hope_env_wait_rs_step(self.h, C.c_uint64(int(step)), self._stream()), 'hope_env_wait_rs_step')
        return self

    def queue_check(self):
        """hope_env_create's measurement of which library streams share a hardware queue (hope_env_queue_check):
        {'queue_of_role': [8 ints, [0] = the NULL stream], 'distinct_queues': n, 'roles_shared': pairs of roles busy in the same step
        form that sit on one queue, 'ms': cost of the measurement}"""
        q = (C.c_int32 * 8)()
        n, ms = C.c_int32(0), C.c_double(0)
        L.check(self.lib.hope_env_queue_check(self.h, q, C.byref(n), C.byref(ms)), 'hope_env_queue_check')
        q = [int(v) for v in q]
        shared = []
        for group in ((0, 5, 1, 3), (0, 1, 3, 4), (5, 1, 3, 7)):
            for i, a in enumerate(group):
                for b in group[i + 1:]:
                    if q[a] >= 0 and q[a] == q[b] and [a, b] not in shared:
                        shared.append([a, b])
        return {'queue_of_role': q, 'distinct_queues': int(n.value), 'roles_shared': shared, 'ms': float(ms.value)}

    def n_obst_now(self):
        """obstacle count of every scene as it is on the device now (device-side draws change it): numpy int32 [N]"""
        out = np.zeros(self.n, np.int32)
        L.check(self.lib.hope_env_download_n_obst(self.h, out.ctypes.data), 'hope_env_download_n_obst')
        return out

    def restart(self, mask):
        """episodes flagged in mask (u8 [N]) go back to their start pose, t = 0 (same map)."""
        assert mask.dtype == torch.uint8 and mask.shape == (self.n,) and mask.device == self.device
        L.check(self.lib.hope_env_restart(self.h, C.c_void_p(mask.data_ptr()), self._stream()), 'hope_env_restart')
        return self

    def kernel_ms(self, reset=True):
        """{kernel name: (accumulated ms, launches)} from the library's per-launch HIP events (profile=True)."""
        ms = np.zeros(len(L.KERNELS))
        cnt = np.zeros(len(L.KERNELS), np.int64)
        L.check(self.lib.hope_env_kernel_ms(self.h, ms.ctypes.data, cnt.ctypes.data, int(reset)), 'hope_env_kernel_ms')
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(L.KERNELS)}

    def kernel_union_ms(self, reset=True):
        """{kernel name: (accumulated ms during which >= 1 launch of the kernel ran, calls)}: per step call the UNION of the
        kernel's launch intervals (the two tile classes overlap on two streams).  Read it before kernel_ms(reset=True)."""
        ms = np.zeros(len(L.KERNELS))
        cnt = np.zeros(len(L.KERNELS), np.int64)
        L.check(self.lib.hope_env_kernel_union_ms(self.h, ms.ctypes.data, cnt.ctypes.data, int(reset)), 'hope_env_kernel_union_ms')
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(L.KERNELS)}

    def profile_kernels(self, names=None):
        """time only the listed kernels (names from _lib.KERNELS; None = all) -- every event pair costs launch latency"""
        mask = 0xffffffff if names is None else sum(1 << L.KERNELS.index(k) for k in names)
        L.check(self.lib.hope_env_profile_kernels(self.h, mask), 'hope_env_profile_kernels')

    def obs(self, clone=False):
        """the observation dict in the reference's key order (car_parking_base.py:399-407).  The tensors are this
        object's persistent output buffers: the next step()/reset_obs() overwrites them IN PLACE (asynchronously), so
        `obs = env.obs(); env.step(a); next_obs = env.obs()` aliases obs and next_obs -- pass clone=True (or copy what
        you keep before stepping, as hope_amd.rollout does) when both are needed."""
        o = {'img': self.img, 'lidar': self.lidar, 'target': self.target, 'action_mask': self.action_mask}
        return {k: (v.clone() if (clone and v is not None) else v) for k, v in o.items()}

    def download_outputs(self):
        """every output of the last step on the host, through ONE device-to-host copy of the output arena into pinned memory (and one
        stream synchronisation): {name: numpy view}.  The views alias the pinned buffer: valid until the next call."""
        if self._arena_host is None:
            self._arena_host = torch.empty(self._arena.shape, dtype=torch.uint8, pin_memory=True)
        self._arena_host.copy_(self._arena, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        hb = self._arena_host.numpy()
        np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.uint8: np.uint8, torch.int8: np.int8}
        return {name: hb[o:o + nbytes].view(np_dt[dt]).reshape(shape) for name, (o, nbytes, shape, dt) in self._layout.items()}

    # -- state -------------------------------------------------------------------------------------
    def download_state(self):
        pose, t, acc = np.zeros((self.n, 3)), np.zeros(self.n, np.int32), np.zeros(self.n)
        L.check(self.lib.hope_env_download_state(self.h, pose.ctypes.data, t.ctypes.data, acc.ctypes.data),
                'hope_env_download_state')
        return pose, t, acc

    def upload_state(self, pose=None, t=None, accum=None):
        p = np.ascontiguousarray(pose, dtype=np.float64) if pose is not None else None
        tt = np.ascontiguousarray(t, dtype=np.int32) if t is not None else None
        a = np.ascontiguousarray(accum, dtype=np.float64) if accum is not None else None
        L.check(self.lib.hope_env_upload_state(self.h, p.ctypes.data if p is not None else None,
                                               tt.ctypes.data if tt is not None else None,
                                               a.ctypes.data if a is not None else None), 'hope_env_upload_state')

    def close(self):
        if getattr(self, 'h', None):
            self.lib.hope_env_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
