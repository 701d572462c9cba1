"""Batch scene sources for N >> 1: packed arrays straight from the native generator.

`generate_arrays` wraps hope_scenegen_generate (hope_amd/csrc/hope_scenegen.cpp): Normal / Complex / Extrem lots drawn as
ParkingMapNormal.reset does (src/env/parking_map_normal.py:474-494), ~3 x 10^5 scenes/s per host core, multi-threaded, no
device involved.  `mixed_arrays` interleaves difficulty levels the way the reference's scene chooser mixes them before its
curriculum kicks in (uniform over the levels, src/train/train_HOPE_sac.py:23-29); DLP scenes come from `DlpScenePool.sample`
(host) here, and at episode turnover from the device-side draw (hope_env_set_dlp_cases)."""
import threading
import time

import numpy as np

from . import _lib as L
from .scenes import DlpScenePool, pack_scenes

LEVEL_ID = {'Normal': 0, 'Complex': 1, 'Extrem': 2}


def generate_arrays(level, n, seed=0, max_obst=128, first_index=0, bay_mode=-1, threads=0):
    """-> (start[n,3], dest[n,3], bbox[n,4], verts[n,max_obst,4,2], n_obst[n], nvert[n,max_obst], case_id[n])"""
    lib = L.load_library()
    start, dest, bbox = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 4))
    verts = np.zeros((n, max_obst, 4, 2))
    nob, cid = np.zeros(n, np.int32), np.zeros(n, np.int32)
    rc = lib.hope_scenegen_generate(LEVEL_ID[level], int(bay_mode), int(n), int(seed) & (2 ** 64 - 1), int(first_index), int(max_obst),
                                    start.ctypes.data, dest.ctypes.data, bbox.ctypes.data, verts.ctypes.data, nob.ctypes.data,
                                    cid.ctypes.data, int(threads))
    if rc != 0:
        raise L.HopeError(f'hope_scenegen_generate failed (code {rc})')
    return start, dest, bbox, verts, nob, np.full((n, max_obst), 4, np.int32), cid


def mixed_arrays(n, levels=('Normal', 'Complex', 'Extrem', 'dlp'), seed=0, max_obst=128, threads=0, dlp_pool=None):
    """n scenes, scene k of level levels[k % len(levels)] -> (start, dest, bbox, verts, n_obst, nvert) like pack_scenes"""
    start, dest, bbox = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 4))
    verts = np.zeros((n, max_obst, 4, 2))
    nob = np.zeros(n, np.int32)
    nvert = np.full((n, max_obst), 4, np.int32)
    nl = len(levels)
    for j, lv in enumerate(levels):
        ids = np.arange(j, n, nl)
        if len(ids) == 0:
            continue
        if lv == 'dlp':
            pool = dlp_pool or DlpScenePool()
            rng = np.random.default_rng([int(seed), 977 + j])
            part = pack_scenes([pool.sample(rng=rng) for _ in ids], max_obst)
        else:
            part = generate_arrays(lv, len(ids), seed=int(seed) * 1000003 + j, max_obst=max_obst, threads=threads)[:6]
        start[ids], dest[ids], bbox[ids], verts[ids], nob[ids], nvert[ids] = part
    return start, dest, bbox, verts, nob, nvert


class PoolRefresher:
    """Keeps the device-resident pool of generated lots fresh without stopping the step loop: a background thread runs the
    native generator (ctypes releases the GIL) straight into the handle's pinned staging arrays; `poll()` -- called from the
    thread that owns the env, e.g. once per policy update -- commits a finished batch (asynchronous upload + swap,
    hope_env_commit_pool) and starts the next one.  Every batch is new: batch b of the run uses first_index = b * n_pool."""

    def __init__(self, env, n_pool, levels=('Normal', 'Complex', 'Extrem'), seed=0, threads=2, relaxed=False):
        # threads: 2 workers of the native generator refill 8 192 lots in ~7 ms (1.2 M lots/s, profiles/r05_host_generator_threads.txt):
        # 65 536 scenes consume 0.5 M lots/s; more threads only add ways to get in the way of the thread that enqueues the steps.  0 = all CPUs
        self.env, self.n, self.levels, self.seed, self.threads = env, int(n_pool), tuple(levels), int(seed), int(threads)
        # relaxed: commit with hope_env_commit_pool_relaxed -- the new pool takes over once its upload has finished, no step waits for it
        # (the strict commit makes the next step wait 2-3 ms for the 67 MB of an 8 192-lot pool); which step sees it depends on timing
        self.relaxed = bool(relaxed)
        self.batch = 0
        self.thread = None
        self.commits = 0
        self.gen_seconds = 0.0
        self.error = None

    def _fill(self, arrays, batch):
        try:
            t0 = time.perf_counter()
            lib = L.load_library()
            start, dest, bbox, verts, nob = arrays
            per = self.n // len(self.levels)
            for j, lv in enumerate(self.levels):
                a = j * per
                b = self.n if j == len(self.levels) - 1 else a + per
                rc = lib.hope_scenegen_generate(LEVEL_ID[lv], -1, b - a, (self.seed * 1000003 + j) & (2 ** 64 - 1), batch * self.n,
                                                self.env.max_obst, start[a:].ctypes.data, dest[a:].ctypes.data, bbox[a:].ctypes.data,
                                                verts[a:].ctypes.data, nob[a:].ctypes.data, None, self.threads)
                if rc != 0:
                    raise L.HopeError(f'hope_scenegen_generate failed (code {rc})')
            self.gen_seconds += time.perf_counter() - t0
        except Exception as e:                      # surfaced by poll()
            self.error = e

    def start_fill(self, block=True):
        """start the next fill; block=False: only if the pinned staging is free already (the previous commit's copies are gated
        on the last enqueued step, so waiting for them would stall the host until the GPU has drained its queue)"""
        assert self.thread is None
        if not block and not self.env.pool_staging_ready():
            return False
        arrays = self.env.pool_staging(self.n)
        self.env._refresher_filling = True           # (env.set_pool refuses while the fill thread writes the pinned arrays)
        self.thread = threading.Thread(target=self._fill, args=(arrays, self.batch), daemon=True)
        self.thread.start()
        self.batch += 1
        return True

    def poll(self, wait=False):
        """commit a finished batch and start the next (as soon as the staging is free again: at a later poll() if the upload of
        the batch just committed is still reading it); returns True when a new pool was committed.  Never blocks unless wait."""
        if self.thread is None:
            if not self.start_fill(block=wait):
                return False
            if not wait:
                return False
        if wait:
            self.thread.join()
        if self.thread.is_alive():
            return False
        self.thread = None
        self.env._refresher_filling = False
        if self.error is not None:
            raise self.error
        self.env.commit_pool(self.n, relaxed=self.relaxed)
        self.commits += 1
        self.start_fill(block=False)
        return True

    def close(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        self.env._refresher_filling = False
