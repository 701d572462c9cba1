/*
 * hope_env.h -- C ABI of libhope_env.so, the MI355X-native batched parking simulator for HOPE.
 *
 * The reference (jiamiya/HOPE) has NO FFI / plugin layer: its boundary is the Python class surface
 * `CarParking` (src/env/car_parking_base.py:39) wrapped by `CarParkingWrapper`
 * (src/env/env_wrapper.py:58).  This header is therefore the interface a maintainer would bind
 * with ctypes from a `CarParking` look-alike (INTEGRATION.md shows the stub); each entry point
 * names the reference method(s) whose work it replaces.
 *
 * Conventions
 *   - plain C, no HIP/torch types: device buffers are `void*` device addresses (e.g.
 *     tensor.data_ptr()), streams are the raw hipStream_t passed as `void*` (NULL = default stream);
 *   - every call returns 0 on success or a negative HOPE_E* code; hope_last_error() gives the text.
 *     Nothing throws across the ABI;
 *   - one handle per GPU process; a handle is not thread-safe, distinct handles are independent;
 *   - the library owns the tables, the per-scene obstacle tiles and the per-scene episode state
 *     (pose, t, accum_arrive_reward); the caller owns every observation/reward buffer;
 *   - all launches are asynchronous on the caller's stream; no hidden synchronisation except in
 *     functions documented as host-synchronous (create/destroy/upload/set_scenes/download).
 *   - the library fails loudly when no HIP device is usable: there is NO CPU fallback.
 *
 * Obstacle tile: every obstacle is a ring of 3 or 4 vertices stored in a fixed 4-vertex slot
 * (64 B, float64 x,y pairs); a triangle repeats its last vertex in slot 3.  The reference's scenes
 * (data/dlp.data, parking_map_normal.py) contain only 3- and 4-vertex rings.
 */
#ifndef HOPE_ENV_H
#define HOPE_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOPE_ABI_VERSION 8

#define HOPE_LIDAR_NUM 120   /* configs.py:96  */
#define HOPE_N_ACTION 42     /* configs.py:108-115 */
#define HOPE_N_ITER 10       /* action_mask.py:9 n_iter */
#define HOPE_UPSAMPLE 10     /* action_mask.py:18 */
#define HOPE_TARGET_DIM 5    /* car_parking_base.py:72 */
#define HOPE_RS_MAX_SEG 5    /* longest Reeds-Shepp word (CCSCC) */
#define HOPE_IMG_SIZE 64      /* OBS_W // downsample_rate (configs.py:88, observation_processor.py:8) */
#define HOPE_IMG_CHANNELS 3
#define HOPE_TRAJ_RENDER_LEN 20 /* configs.py:86 */

/* error codes */
#define HOPE_OK 0
#define HOPE_EINVAL (-1)     /* bad argument                       */
#define HOPE_ENODEV (-2)     /* no usable HIP device               */
#define HOPE_ENOMEM (-3)     /* device/host allocation failed      */
#define HOPE_EHIP (-4)       /* a HIP runtime call failed          */
#define HOPE_ESTATE (-5)     /* call order violated (e.g. step before tables/scenes) */

/* vehicle.py:13-18 Status */
#define HOPE_STATUS_CONTINUE 1
#define HOPE_STATUS_ARRIVED 2
#define HOPE_STATUS_COLLIDED 3
#define HOPE_STATUS_OUTBOUND 4
#define HOPE_STATUS_OUTTIME 5

/* Reeds-Shepp segment types in rs_word[] */
#define HOPE_RS_NONE (-1)
#define HOPE_RS_S 0
#define HOPE_RS_L 1
#define HOPE_RS_R 2

/* hope_env_create flags */
#define HOPE_F_OBS_F64 0x1      /* observation/reward buffers are float64 (parity mode); default float32 */
#define HOPE_F_ACTION_F64 0x2   /* action buffer is float64; default float32 */
#define HOPE_F_PROFILE 0x4      /* record HIP events around every kernel launch (hope_env_kernel_ms) */
#define HOPE_F_IMAGE 0x8        /* keep vehicle.trajectory (last 20 poses/scene) so that HOPE_STAGE_IMG can be used */
#define HOPE_F_OVERLAP 0x10     /* launch the two obstacle-tile classes of a step on two streams (fork / join with events),
                                   and from 16 k scenes on the observation half of the step kernel on further streams
                                   next to the Reeds-Shepp kernels: one chain alone leaves issue slots idle */
/* 0x20 was HOPE_F_GRAPH (hipGraph replay of a step's launches, ABI <= 6).  Removed in ABI 7: measured slower than plain asynchronous
 * launches at every batch size (4 096 scenes 0.269 vs 0.237 ms, 8 192 0.293 vs 0.263, 16 384 0.511 vs 0.352; profiles/r04_bench_modes.txt
 * of the commit before the removal) -- the launches are not host-bound, and the runtime runs a captured multi-stream graph on fewer
 * queues than the streams it was captured from.  hope_env_create rejects the bit. */

/* Environment variables read by the library (tuning / diagnostics only; results never depend on them):
 *   HOPE_SPLIT_MIN   scenes per handle from which HOPE_F_OVERLAP launches k_env_step as a motion and an observation
 *                    launch (default 16384); HOPE_NO_SPLIT: never
 *   HOPE_CLS1_FRAC   share of the scenes the large-tile launch chain should hold after the small-tile class has handed
 *                    scenes over (default 0.42 below 32768 scenes, else 0 = no hand-over); read by hope_env_set_scenes
 *   HOPE_RS_EXACT    validate Reeds-Shepp words with the all-float64 kernel (the reference of the default kernel's float32 filter)
 *   HOPE_AUTO_CHAINS lo0:hi0:lo1:hi1 -- scene-count ranges in which a batch whose scenes all sit in ONE tile class (0: <= 32 obstacles,
 *                    1: larger) is stepped as two sub-chains of that class, so that the two-stream pipelined form of deferred steps
 *                    applies (default 32768:2^30:4096:32768, measured: profiles/r04_single_class_chains.txt); HOPE_CHAINS=n fixes
 *                    the number of sub-chains per class instead (1 = never)
 *   HOPE_BALANCE, HOPE_PRIO, HOPE_RS_OCC   rejected launch / build variants kept for experiments
 *   HOPE_STEP_TIMING, HOPE_RS_TIMING, HOPE_RS_DEBUG      instrumented kernel builds and profiling switches (tools/) */

/* hope_env_step stage mask */
#define HOPE_STAGE_MOTION 0x1   /* kinematics + arrival + collision sub-step loop (CarParking.step :255-277) */
#define HOPE_STAGE_OBS 0x2      /* lidar + action mask + target (CarParking.render :399-407)               */
#define HOPE_STAGE_REWARD 0x4   /* status + reward (:279-289) + wrapper reward_shaping                     */
#define HOPE_STAGE_RS 0x8       /* Reeds-Shepp feasibility search (:293-297, find_rs_path :413)           */
#define HOPE_STAGE_ALL 0xF
/* bird's-eye image observation obs['img'] (_render :301-320, _get_img_observation :322-350, Obs_Processor
 * observation_processor.py:11-23, transpose env_wrapper.py:53-54).  Not part of HOPE_STAGE_ALL (the reference's
 * USE_IMG switch, configs.py:100); needs a handle created with HOPE_F_IMAGE and out->img. */
#define HOPE_STAGE_IMG 0x40
/* modifier bit: actions are already physical (steer [rad], speed [m/s]) as CarParking.step receives them
 * (car_parking_base.py:235); without it they are the wrapper's [-1,1] actions and action_rescale
 * (env_wrapper.py:37-50) is applied first.  KSModel's own clip (vehicle.py:85-86) always applies. */
#define HOPE_ACTION_PHYSICAL 0x10
/* modifier bit: with a float32 action buffer, action_rescale is evaluated in float32 -- what the reference's
 * action_rescale (env_wrapper.py:37-50) does when the agent hands it a float32 array: gym's Box bounds are float32, so
 * clip, scale and shift all round to float32 before CarParking.step sees the action.  Everything after the rescale
 * (KSModel.step) is float64 as without the bit.  No effect with a float64 buffer or with HOPE_ACTION_PHYSICAL. */
#define HOPE_ACTION_RESCALE_F32 0x80
/* modifier bit: fused episode turnover (gym VectorEnv "autoreset").  A scene whose step ends with status !=
 * CONTINUE is reset in the same kernel exactly as CarParking.reset (:127-138) does on the same map: pose = start,
 * accum_arrive_reward = 0, t = 0 and the action-less step (t = 1, incl. its _get_reward bookkeeping).  reward /
 * reward_info / status / done / RS outputs are those of the FINISHED step; lidar / action_mask / target / pose are
 * the new episode's first observation.  Equivalent to hope_env_step + hope_env_restart(done) +
 * hope_env_reset_obs(active = done) without the extra launches. */
#define HOPE_AUTO_RESET 0x20
/* modifier bit, with HOPE_AUTO_RESET: the finished scene continues on a NEW map, as CarParking.reset does
 * (map.reset, car_parking_base.py:134): a pool entry of its obstacle-tile class (hope_env_set_pool), chosen exactly as
 * hope_env_redraw chooses it -- by a counter-based hash of (the seed of hope_env_set_redraw_seed, scene, episodes drawn
 * so far) -- is copied into the scene inside the step kernel.  Equivalent to hope_env_step + hope_env_redraw(done, seed) +
 * hope_env_reset_obs(active = done) without the extra launches.  HOPE_ESTATE without a pool. */
#define HOPE_AUTO_REDRAW 0x100
/* modifier bit: TWO completion points per step.  Without it `stream` is ordered after every output of the step when
 * hope_env_step returns.  With it `stream` is ordered after the observation half only -- lidar / action_mask / target / img,
 * reward / reward_info / status / done / pose -- so that the caller's policy forward (which reads the observation,
 * parking_agent.py:78-99) runs while the Reeds-Shepp kernels of the step finish on the library's own streams; rs_word /
 * rs_lengths are ordered by hope_env_wait_rs(h, stream) (the planner override just before the next step).  A caller that does
 * not read them enqueues the next hope_env_step without waiting: the search of step k then runs on the library's search streams
 * NEXT TO the kinematics / motion / observation launches of step k + 1 (consecutive steps pipeline; ABI 7), and its rs_word /
 * rs_lengths are replaced by step k + 1's -- they can only be read after a hope_env_wait_rs issued BEFORE the next step.  The outputs are the same bits either way.  The bit takes effect on handles with HOPE_F_OVERLAP whose scenes
 * span both obstacle-tile classes, at every batch size (the one-launch form of the step kernel below 32 768 scenes, the two-launch
 * form from there on); elsewhere the step is joined as without it and hope_env_wait_rs is a no-op.  Every other entry point that reads or writes the handle's state joins first by itself.
 * Lifetime of the inputs: `actions` and `active` are only read by launches `stream` is ordered after when hope_env_step returns
 * (the Reeds-Shepp chain reads a snapshot of `active` that the motion launch takes into a buffer of the handle), so the caller
 * may free or overwrite both right after the call, in stream order, as without the bit. */
#define HOPE_DEFER_RS 0x200

typedef struct hope_env hope_env_t;

/* Caller-owned DEVICE buffers filled by hope_env_step / hope_env_reset_obs.  Any pointer may be NULL
 * (that output is skipped).  "real" = float32, or float64 when the handle has HOPE_F_OBS_F64. */
typedef struct hope_step_out {
    void *lidar;        /* real [N][120]  obs['lidar']  (range minus hull range; may be slightly < 0) */
    void *action_mask;  /* real [N][42]   obs['action_mask'] in {0,.1,..,1} or all .01               */
    void *target;       /* real [N][5]    obs['target']                                               */
    void *reward;       /* real [N]       wrapper reward (env_wrapper.py:10-35)                       */
    void *reward_info;  /* real [N][5]    time_cost, rs_dist, dist, angle, box_union                  */
    int32_t *status;    /* i32  [N]       HOPE_STATUS_*                                               */
    uint8_t *done;      /* u8   [N]       status != CONTINUE                                          */
    double *pose;       /* f64  [N][3]    x, y, heading after the step (always float64)               */
    int8_t *rs_word;    /* i8   [N][8]    [0..4] segment types (HOPE_RS_*), [5] n_seg, [6] found, [7] 0 */
    void *rs_lengths;   /* real [N][5]    signed segment lengths in metres (PATH.lengths)             */
    uint8_t *img;       /* u8   [N][3][64][64]  obs['img'] * 255 (channel-first as the wrapper returns it); the
                           reference's float image is this / 255.0                                    */
} hope_step_out;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Replaces CarParking.__init__ (car_parking_base.py:48-115) for n_scenes parallel environments.
 * max_obstacles bounds the obstacle count of any scene (LDS tile = 64 B x max_obstacles per wave).  It is limited to
 * HOPE_MAX_OBSTACLES: the Reeds-Shepp queue entry carries a scene's obstacle count in 8 bits (the LDS formulas alone would admit
 * over a thousand; the reference's largest map, a Dragon-Lake case after its +-20 m cull, holds 125).  A larger value -- or one
 * whose tile does not fit the 160 KiB LDS of a CU, or n_scenes >= 2^24 -- fails with HOPE_EINVAL and a message naming the limit. */
#define HOPE_MAX_OBSTACLES 255
int hope_env_create(hope_env_t **out, int n_scenes, int max_obstacles, int device_id, uint32_t flags);
int hope_env_destroy(hope_env_t *h);                       /* CarParking.close (:536) */
const char *hope_last_error(void);
int hope_abi_version(void);

/* ---- tables: ActionMask.__init__ / LidarSimlator.__init__ products (host pointers, host-sync) -- */
/* dist_star  [1200][42][10] float64, reference order (action_mask.py:114-143 `precompute`);
 * hull_base  [120]  range from the rear axle to the hull per beam (lidar_simulator.py:48-53);
 * beam_ab    [120][2]  (sin theta_i, -cos theta_i) of lidar_simulator.py:86-88.
 * The library derives its device tables from them on the host: dist_star prefix-maxed over k and transposed, its per-beam maximum,
 * and (round 6) the count-interval table of the mask stage -- per coarse beam 128 scan-value bins, per bin and action the interval that
 * brackets ActionMask.get_steps' first-exceed count (action_mask.py:166-177) for every scan value of the bin (hope_debug_mask_lut). */
int hope_env_upload_tables(hope_env_t *h, const double *dist_star, const double *hull_base,
                           const double *beam_ab);

/* ---- scenes: map.reset + vehicle.reset (car_parking_base.py:127-137), host pointers, host-sync -- */
/* For k in [0,n): scene scene_ids[k] gets start[k][3], dest[k][3] (x,y,heading), bbox[k][4]
 * (xmin,xmax,ymin,ymax; parking_map_dlp.py:70-73 / parking_map_normal.py:486-489), n_obst[k]
 * obstacles whose vertices are verts[k][max_obstacles][4][2].  Episode state is reset: pose=start,
 * t=0, accum_arrive_reward=0.  The caller then runs hope_env_reset_obs (reset's `self.step()`). */
int hope_env_set_scenes(hope_env_t *h, const int32_t *scene_ids, int n, const double *start,
                        const double *dest, const double *bbox, const double *verts,
                        const int32_t *n_obst);

/* ---- the hot path (asynchronous on `stream`) ------------------------------------------------- */
/* CarParkingWrapper.step(action) (env_wrapper.py:73-81) for all scenes:
 * actions = DEVICE [N][2] (steer, speed) in [-1,1] (float32, or float64 with HOPE_F_ACTION_F64).
 * active  = optional DEVICE u8 [N]; scenes with active[i]==0 are left untouched (outputs too). */
int hope_env_step(hope_env_t *h, const void *actions, const uint8_t *active, uint32_t stages,
                  const hope_step_out *out, void *stream);

/* Orders `stream` after the Reeds-Shepp outputs of the LAST hope_env_step that carried HOPE_DEFER_RS (no-op if none is
 * outstanding).  "Last" is literal: since ABI 7 consecutive deferred steps pipeline, and step k + 1 REPLACES step k's rs_word /
 * rs_lengths instead of ordering them -- a hope_env_wait_rs issued after step k + 1 was enqueued joins step k + 1's search. */
int hope_env_wait_rs(hope_env_t *h, void *stream);
/* (ABI 8) The same wait with the caller saying WHICH step's search it means: hope_env_last_step returns the sequence number of
 * the most recent hope_env_step / hope_env_reset_obs call of the handle (1, 2, ...); hope_env_wait_rs_step(h, step, stream)
 * orders `stream` after that step's Reeds-Shepp outputs and fails with HOPE_ESTATE -- touching nothing -- when a newer step has
 * been enqueued since (its outputs have been, or are being, replaced).  What a planner-in-the-loop caller should use: an
 * accidental "step, step, wait" then fails loudly instead of handing it the other step's (or a mix of both steps') words. */
int hope_env_last_step(hope_env_t *h, uint64_t *step);
int hope_env_wait_rs_step(hope_env_t *h, uint64_t step, void *stream);

/* The action-less step of CarParking.reset (:138) / CarParkingWrapper.step(None) (:74-75):
 * t += 1, observation, status, reward; no motion. */
int hope_env_reset_obs(hope_env_t *h, const uint8_t *active, uint32_t stages, const hope_step_out *out,
                       void *stream);

/* vehicle.reset(initial_state) + accumulator reset of CarParking.reset (:128-136) WITHOUT a new map:
 * scenes with mask[i] != 0 go back to pose = start, t = 0, accum_arrive_reward = 0 (asynchronous).
 * Follow with hope_env_reset_obs(active = mask) to obtain the first observation. */
int hope_env_restart(hope_env_t *h, const uint8_t *mask, void *stream);

/* ---- scene pool: a new map at episode turnover without the host in the loop -------------------------------------- */
/* The reference draws a new case for EVERY episode (ParkingMapNormal.reset parking_map_normal.py:474-494, ParkingMapDLP.reset
 * parking_map_dlp.py:38-86, called from CarParking.reset car_parking_base.py:134).  hope_env_set_pool uploads n_pool complete
 * scenes (host pointers, layout as hope_env_set_scenes, host-synchronous) into device memory owned by the handle;
 * hope_env_redraw (asynchronous on `stream`) gives every scene with mask[i] != 0 a pool entry of ITS obstacle-tile class
 * (n_obst <= 32 or larger: the dense per-class launch lists stay valid), chosen by a counter-based hash of
 * (seed, scene, episodes drawn so far), and restarts it (pose = start, t = 0, accum_arrive_reward = 0).  Follow with
 * hope_env_reset_obs(active = mask) for the first observation.  The host refills / replaces the pool whenever it likes.
 * The size-class mix of the resident scenes therefore stays what hope_env_set_scenes made it (a slot that holds a <= 32-obstacle
 * lot keeps drawing such lots); a resident class without pool entries is an error (HOPE_ESTATE), not a silent restart. */
int hope_env_set_pool(hope_env_t *h, int n_pool, const double *start, const double *dest, const double *bbox,
                      const double *verts, const int32_t *n_obst);
/* The same without blocking the step loop: hope_env_pool_staging hands out PINNED host arrays of the handle (layout as above, for
 * n_pool entries; valid until the next hope_env_pool_staging call) that any host thread may fill -- e.g. hope_scenegen_generate
 * writing into them directly -- and hope_env_commit_pool uploads them asynchronously into the pool set the kernels are not reading
 * and swaps the sets: launches enqueued afterwards wait (on the device, in stream order) for the upload and draw from the new pool;
 * launches already enqueued keep the old one.  Neither call synchronises with the step kernels (hope_env_pool_staging waits at most
 * for the previous upload's copy to leave the staging).  Both must be called from the thread that owns the handle. */
int hope_env_pool_staging(hope_env_t *h, int n_pool, double **start, double **dest, double **bbox, double **verts, int32_t **n_obst);
int hope_env_commit_pool(hope_env_t *h, int n_pool, void *stream);
/* (ABI 8) The same upload, with the swap RELAXED: the new pool takes over at the first step (or hope_env_redraw) enqueued after the
 * upload has COMPLETED -- no step ever waits for an upload.  (hope_env_commit_pool swaps at once, so the next step's launch waits on
 * its stream for the 67 MB of an 8 192-lot pool: 2-3 ms, five 65 536-scene steps.)  Which step that is depends on timing, like the
 * moment a background refresher's fill finishes; a later commit of either kind, hope_env_set_pool and hope_env_pool_generation first
 * wait for and apply a swap still pending.  What a rollout's background refresher should use (hope_amd.scene_gen.PoolRefresher). */
int hope_env_commit_pool_relaxed(hope_env_t *h, int n_pool);
/* 1 when hope_env_pool_staging would return without waiting (the previous commit's copies have left the pinned arrays), 0 when not
 * yet, < 0 on error: a background refresher polls this instead of blocking the thread that drives the step loop */
int hope_env_pool_staging_ready(hope_env_t *h);
/* An identity of the set of maps a draw can return: changes whenever that set changes (hope_env_set_pool / hope_env_commit_pool /
 * hope_env_set_dlp_cases).  Part of a snapshot of drawn maps: hope_env_restore_maps refuses a snapshot of another generation.
 * ABI 8: the value is a HASH CHAIN over the contents of every upload since hope_env_create (pool entries' start / dest / box /
 * obstacle counts and a strided sample of their vertices; the Dragon-Lake case table), not a per-handle counter -- a snapshot taken
 * in one process is refused by another process whose pool holds other maps even if both uploaded "one pool".  0 is never
 * returned for a handle with a pool (0 = "no pool yet" / "skip the check" in hope_env_restore_maps). */
int hope_env_pool_generation(hope_env_t *h, uint64_t *generation);
/* seed of HOPE_AUTO_REDRAW's draws (default 0) */
int hope_env_set_redraw_seed(hope_env_t *h, uint64_t seed);
int hope_env_redraw(hope_env_t *h, const uint8_t *mask, uint64_t seed, void *stream);
/* pool entry each scene currently holds: >= 0 a complete pool scene, -1 the map hope_env_set_scenes uploaded;
 * values <= -2 name Dragon-Lake-Parking case c = -2 - value drawn on the device (hope_env_set_dlp_cases); host-synchronous */
int hope_env_download_pool_index(hope_env_t *h, int32_t *out /*[N]*/);

/* Dragon-Lake-Parking cases drawn ON THE DEVICE at episode turnover (ParkingMapDLP.reset, src/env/parking_map_dlp.py:38-86): per
 * episode a start candidate of the case chosen uniformly (:58-59) with N(0, 0.05^2) m / N(0, 0.02^2) rad jitter (:60-65), the map
 * box floor / ceil(min / max(start, dest) -/+ 20 m) (:70-73), the case's obstacles culled by that box (:88-101), independent 50 %
 * flips of dest and start about their box centres (:80-83).  The data of data/dlp.data (host pointers, host-synchronous):
 * dest [n_cases][3]; cand_off [n_cases + 1] + cand [][3]: the start candidates of each case; case_set [n_cases]: the obstacle set
 * a case uses; set_off [n_sets + 1] + set_verts [][4][2]: the rings of each set (a triangle repeats its last vertex).  The cases
 * join the draw list of the large-tile class (every culled lot has 37 .. 125 obstacles): hope_env_redraw / HOPE_AUTO_REDRAW pick
 * uniformly among that class's complete pool scenes and the cases.  n_cases = 0 removes them. */
int hope_env_set_dlp_cases(hope_env_t *h, int n_cases, const double *dest, const int32_t *cand_off, const double *cand,
                           const int32_t *case_set, int n_sets, const int32_t *set_off, const double *set_verts);
/* The size class of scene slots: 0 = lots of at most 32 obstacles (small LDS tile, draws from the pool's small maps), 1 = larger
 * lots (draws from the large maps and the Dragon-Lake cases).  hope_env_set_scenes sets it from the uploaded map's obstacle count;
 * this call overrides it, e.g. to keep a slot among the Dragon-Lake lots although its first map kept only 30 obstacles (class 1 is
 * always allowed, class 0 only for maps of <= 32 obstacles).  cls: host u8 [n].  Host-synchronous. */
int hope_env_set_draw_class(hope_env_t *h, const int32_t *scene_ids, int n, const uint8_t *cls);
/* draws whose culled obstacle set did not fit max_obstacles and was truncated (must stay 0); host-synchronous */
int hope_env_pool_overflow(hope_env_t *h, int32_t *count);
/* the maps the listed scenes hold NOW (after device-side draws they exist on the device only); any output may be NULL; layout as
 * hope_env_set_scenes; host-synchronous */
int hope_env_download_scenes(hope_env_t *h, const int32_t *scene_ids, int n, double *start, double *dest, double *bbox, double *verts,
                             int32_t *n_obst);
/* (ABI 8) the obstacle count of EVERY scene as it is now (device-side draws change it behind the host): one copy; host-synchronous */
int hope_env_download_n_obst(hope_env_t *h, int32_t *n_obst /*[N]*/);
/* Snapshot / restore of drawn maps.  A draw is a pure function of (seed, scene, episode counter) and of the pool / case lists:
 * hope_env_download_pool_state returns the pool index and the episode counter of every scene; hope_env_restore_maps (host arrays:
 * drawn[i] != 0 where the scene held a drawn map, the saved counters, the seed in use when the maps were drawn) repeats those draws
 * with the SAME pool / cases resident: pass the hope_env_pool_generation value saved with the snapshot and the call fails with
 * HOPE_ESTATE when the pool has been replaced since (a background refresher, hope_env_commit_pool) instead of silently restoring
 * other maps; 0 skips the check.  Runs whose pool is refreshed snapshot the maps themselves (hope_env_download_scenes ->
 * hope_env_set_scenes).  Restore pose / t / accumulator afterwards with hope_env_upload_state.  Host-synchronous. */
int hope_env_download_pool_state(hope_env_t *h, int32_t *pool_index /*[N]*/, uint32_t *episode /*[N]*/);
int hope_env_restore_maps(hope_env_t *h, const uint8_t *drawn /*[N]*/, const uint32_t *episode /*[N]*/, uint64_t seed, uint64_t pool_generation);

/* (ABI 8) Which of the library's streams share a HARDWARE queue.  The runtime spreads HIP streams over a few hardware queues (4 by
 * default) in creation order, and launches of streams that share one serialise: which library stream plays which role of the step's
 * launch structure decides 0.52 vs 0.6+ ms per 65 536-scene step.  Up to round 4 the assignment was a table found by search on one
 * box; since round 5 hope_env_create MEASURES it (pairs of ~100 us spin kernels: two streams on one queue take twice as long; < 5 ms
 * once, HOPE_QUEUE_CHECK=0 skips it) and assigns the roles so that the streams a deferred / joined / sub-chain step keeps busy at
 * the same time sit on different queues, whatever other streams the process created first.  queue_of_role[r]: hardware-queue class
 * (0, 1, ...) of role r's stream, r = 1 .. 7 ([0]: the NULL stream, the usual caller's stream); -1 = not measured.
 * Roles: 1 search chain of the small-tile class, 2 image side, 3 / 4 observation half of the large- / small-tile class, 5 chain of
 * the large-tile class (deferred), 6 spare, 7 second sub-chain's observation.  *n_queues = distinct classes seen; *ms = what the
 * measurement took.  Any pointer may be NULL. */
int hope_env_queue_check(hope_env_t *h, int32_t *queue_of_role /*[8]*/, int32_t *n_queues, double *ms);

/* With HOPE_F_PROFILE every kernel launch is bracketed by its own HIP event pair on the launch stream.
 * Returns the accumulated time (ms) and launch count per kernel since the last call with reset != 0:
 * arrays of HOPE_N_KERNELS entries indexed by HOPE_K_*.  Host-synchronous. */
#define HOPE_K_KINEMATICS 0    /* k_kinematics   (four lanes per scene)                      */
#define HOPE_K_STEP 1          /* k_env_step     (wave per scene; one launch per tile class) */
#define HOPE_K_RS_WORDS 2      /* k_rs_words     (wave per queued scene)                     */
#define HOPE_K_RS_VALIDATE 3   /* k_rs_validate  (wave per queued scene; per tile class)     */
#define HOPE_K_IMAGE 4         /* k_bev_image    (wave per scene tile: 16 per scene)         */
#define HOPE_K_IMAGE_PREP 5    /* k_bev_prep     (wave per scene: map + new boxes' spans)    */
#define HOPE_K_RS_COMPACT 6    /* k_rs_compact   (Reeds-Shepp work queues from per-scene flags) */
#define HOPE_K_POST 7           /* k_post         (reward + target arithmetic, one lane per scene; per tile class) */
#define HOPE_K_RS_SEGS 8       /* k_rs_segs      (segment origins of the words to test, one lane per word; per tile class) */
#define HOPE_K_RS_SCREEN 9     /* k_rs_screen    (ABI 8: coarse look at up to four words per pass, wave per queued scene; per tile class) */
#define HOPE_N_KERNELS 10
int hope_env_kernel_ms(hope_env_t *h, double *ms /*[HOPE_N_KERNELS]*/, int64_t *launches /*[HOPE_N_KERNELS]*/,
                       int reset);
/* Same bookkeeping, per CALL instead of per launch: for every hope_env_step / hope_env_reset_obs call and kernel, the time
 * during which at least one launch of that kernel was running (with HOPE_F_OVERLAP the launches of the two tile classes run
 * concurrently on two streams, so this union is shorter than the sum of the per-launch durations).  ms / calls: arrays of
 * HOPE_N_KERNELS entries.  Host-synchronous; call it BEFORE hope_env_kernel_ms with reset (both drain the same events). */
int hope_env_kernel_union_ms(hope_env_t *h, double *ms /*[HOPE_N_KERNELS]*/, int64_t *calls /*[HOPE_N_KERNELS]*/, int reset);
/* Restricts the event bracketing to the kernels whose bit (1 << HOPE_K_*) is set; default all.  An event pair costs
 * a few microseconds of launch latency, so a throughput measurement times only the kernel it reports on. */
int hope_env_profile_kernels(hope_env_t *h, uint32_t kernel_mask);

/* ---- state access (host-sync; tests, checkpointing) ------------------------------------------ */
int hope_env_download_state(hope_env_t *h, double *pose /*[N][3]*/, int32_t *t /*[N]*/,
                            double *accum /*[N]*/);
int hope_env_upload_state(hope_env_t *h, const double *pose, const int32_t *t, const double *accum);

/* ---- test hook ------------------------------------------------------------------------------------------ */
/* Evaluates one elementary function of hope_amd/csrc/hope_math.h (plus sqrt and division) on the DEVICE for n
 * float64 inputs: fn 0 sin, 1 cos, 2 tan, 3 atan2(a,b), 4 asin, 5 acos, 6 hypot(a,b), 7 fmod(a,b), 8 tanh, 9 exp,
 * 10 sqrt, 11 a/b, 12 the kinematics kernel's three-instruction a/20.  a, b, out are DEVICE pointers (b may be NULL for unary functions).  The functions are built
 * from IEEE-exact operations only, so the results must equal the host evaluation bit for bit. */
int hope_debug_math(int fn, int n, const double *a, const double *b, double *out, void *stream);

/* Test hook (pure host code, no device needed): the count-interval table hope_env_upload_tables derives from dist_star / hull_base for
 * the action-mask stage (hope_amd/csrc/hope_step_kernel.h MASK_LUT_*): lut_out [(120 * 128 + 1)][32] uint16, scale_out [120] bins per
 * metre.  tests/test_mask_lut.py checks its bracketing property against the float64 table (action_mask.py:166-177). */
int hope_debug_mask_lut(const double *dist_star, const double *hull_base, uint16_t *lut_out, double *scale_out);

/* Profiling hook: one sweep over `bytes` of the DEVICE buffer `buf` with a known byte count in one of the library's access widths --
 * mode 0 / 1 / 2: reads of 16 / 8 / 4 bytes per lane, 3: one 8-byte word per 64-byte line; 4 / 5 / 6 / 7: the same as writes.  Run
 * under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` it calibrates those counters (tools/pmc_calib.py; the MI355X guide calibrates the
 * 16-byte read only).  `bytes` should exceed the 256 MiB Infinity Cache several times. */
int hope_debug_traffic(int mode, size_t bytes, void *buf, void *stream);

/* Cycle accounting of the Reeds-Shepp validation kernel: 16 counters accumulated by the instrumented build that the library
 * launches when the environment variable HOPE_RS_TIMING is set (hope_amd/csrc/hope_rs.hip lists the sections); zeros
 * otherwise.  Host-synchronous.  tools/rs_timing.py prints the breakdown. */
int hope_debug_rs_prof(uint64_t *out /*[16]*/, int reset);
/* Statistics of k_rs_validate's float32 filter, accumulated while the environment variable HOPE_RS_DEBUG has bit 0x4000 set:
 * [0] passes [1] passes decided by a certain float32 hit [2] passes that re-evaluated undecided samples in float64 [3] those
 * samples; with bit 0x2000 (self-check: float64 for EVERY sample) also [4] float32 "hit" verdicts float64 contradicts, [5] float32
 * "clear" verdicts float64 contradicts (both must stay 0), [6] samples checked; [8..12] passes left undecided because of: no certain crossing / an
 * axis-parallel obstacle edge / an axis-parallel hull / a hull corner near the edge line / a shallow crossing angle.  Host-synchronous. */
int hope_debug_rs_filter_stats(uint64_t *out /*[16]*/, int reset);
/* details of the first 64 contradicted verdicts of the self-check (16 doubles each; tools/rs_filter_stats.py --check) */
int hope_debug_rs_filter_dump(double *out /*[64][16]*/);
/* Per-search log of the same instrumented build: up to cap records of 4 int32 {wave cycles, words tested | words allowed << 8 |
 * large-tile class << 16, passes, found}; *n = records logged since the last reset.  Host-synchronous.  tools/rs_tail.py. */
int hope_debug_rs_log(int32_t *out /*[cap][4]*/, int cap, int32_t *n, int reset);
/* the same for k_env_step (environment variable HOPE_STEP_TIMING; float32 observation / action handles); tools/step_timing.py */
int hope_debug_step_prof(uint64_t *out /*[16]*/, int reset);
/* Tie census of the same instrumented build (HOPE_STEP_TIMING): how close the workload comes to the decisions whose arithmetic the
 * reference delegates to GEOS (rows a-3, a-4, the ring cull of a-8) and to the mask compares.  out: [0] arrival-ratio evaluations,
 * [1] min |overlap / dest area - 0.95| (bit pattern of a double), [2] lidar ring-keep evaluations, [3] min |ring distance - 10 m|
 * (double), [4] (hull edge, obstacle edge) pairs the orientation filter left undecided (decided by the exact expansion), [6] mask table
 * compares, [7] min non-zero |table entry - scan value| (double), [8] scene-steps whose mask took the exact 1200-beam evaluation,
 * [9] scene-steps, [10] compares whose entry equals the scan value bit for bit.
 * Call once with reset != 0 before counting (the minima start at +inf then).  tools/tie_census.py. */
int hope_debug_census(uint64_t *out /*[16]*/, int reset);

/* ---- host-side scene generator (no device involved; any thread) ------------------------------------------------------ */
/* n scenes of `level` (0 Normal, 1 Complex, 2 Extrem) drawn as ParkingMapNormal.reset does (src/env/parking_map_normal.py:474-494
 * over generate_bay_parking_case :40-246 / generate_parallel_parking_case :248-457): bay lot with probability 1/2 for Normal /
 * Complex, parallel otherwise; map box = floor / ceil of min / max(start, dest) -/+ 10 m.  bay_mode: -1 as the reference, 0
 * parallel only, 1 bay only.  Outputs are HOST arrays in the layout of hope_env_set_scenes / hope_env_set_pool: start [n][3],
 * dest [n][3], bbox [n][4], verts [n][max_obstacles][4][2] (slots beyond n_obst[i] untouched), n_obst [n], case_id [n] (0 bay,
 * 1 parallel; may be NULL).  Scene i depends only on (seed, first_index + i), not on n_threads (<= 0: the environment variable
 * HOPE_HOST_THREADS, else the CPUs in the calling process's affinity mask -- one rank's share of the node once it has pinned
 * itself, hope_amd.dist.pin_rank_to_cores -- never more).
 * Returns 0, HOPE_EINVAL, or -100 - i when scene i has more than max_obstacles obstacles (impossible for max_obstacles >= 18). */
int hope_scenegen_generate(int level, int bay_mode, int n, uint64_t seed, int64_t first_index, int max_obstacles, double *start,
                           double *dest, double *bbox, double *verts, int32_t *n_obst, int32_t *case_id, int n_threads);
/* the fan-out hope_scenegen_generate uses for n_threads <= 0 in this process, now */
int hope_scenegen_default_threads(void);

/* ---- introspection ---------------------------------------------------------------------------- */
int hope_env_num_scenes(const hope_env_t *h);
int hope_env_max_obstacles(const hope_env_t *h);
/* name of the device the handle runs on (e.g. "gfx950"), static storage */
const char *hope_env_device_arch(const hope_env_t *h);

#ifdef __cplusplus
}
#endif
#endif /* HOPE_ENV_H */
