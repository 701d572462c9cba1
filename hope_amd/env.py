"""N = 1 look-alikes of the reference's env classes, backed by libhope_env.so on the GPU.

  CarParking         <-> src/env/car_parking_base.py:39   (reset / step / set_level / close, .map .vehicle
                                                           .action_space .observation_space)
  CarParkingWrapper  <-> src/env/env_wrapper.py:58         (action_rescale, reward_shaping, done)
  Status             <-> src/env/vehicle.py:13

so that the reference's callers (train_HOPE_sac.py:114-118,183-213; train_HOPE_ppo.py; eval_utils.py:16-84)
can drive it.  Old-gym API: reset -> obs ; step -> (obs, reward, done, info).

`use_img_observation=True` adds the bird's-eye image obs['img'] ((64, 64, 3) float64 in [0, 1] from the raw env,
(3, 64, 64) after the wrapper; rendered on the GPU by k_bev_image).  It defaults to True like the
reference's USE_IMG (configs.py:100); its pixel parity with pygame / OpenCV is unpinned.  `map.map_level` is the generator's level for Normal / Complex / Extrem maps and, for DLP maps, the
label `get_map_level` gives (hope_amd/map_level.py <-> src/env/map_level.py, as ParkingMapDLP.reset does :84).
Not provided: the pygame window (`render` returns None).
"""
import math
from collections import OrderedDict
from enum import Enum

import numpy as np

from . import _lib as L
from . import scenes as S
from . import tables as T
from .rs_path import PATH, TYPE_NAMES

# src/configs.py constants the callers read
NUM_STEP, STEP_LENGTH, LIDAR_NUM, LIDAR_RANGE, N_DISCRETE_ACTION = 10, 5e-2, 120, 10.0, 42
VALID_SPEED, VALID_STEER = [-2.5, 2.5], [-0.75, 0.75]
MAX_DIST_TO_DEST, TOLERANT_TIME = 20, 200
REWARD_WEIGHT = OrderedDict({'time_cost': 1, 'rs_dist_reward': 0, 'dist_reward': 5, 'angle_reward': 0,
                             'box_union_reward': 10})


class Status(Enum):       # vehicle.py:13-18
    CONTINUE = 1
    ARRIVED = 2
    COLLIDED = 3
    OUTBOUND = 4
    OUTTIME = 5


class Box:
    """the slice of gym.spaces.Box the reference's scripts use: shape, low, high, sample(), seed()."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype) if shape is None else np.full(shape, low, dtype=dtype) \
            if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype) if shape is None else np.full(shape, high, dtype=dtype) \
            if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.shape = tuple(shape) if shape is not None else self.low.shape
        self.dtype = dtype
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class _Loc:
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)

    @property
    def coords(self):
        return [(self.x, self.y)]

    def distance(self, other):
        return math.sqrt((self.x - other.x) ** 2 + (self.y - other.y) ** 2)


class State:              # vehicle.py:21-39
    def __init__(self, raw_state):
        self.loc = _Loc(raw_state[0], raw_state[1])
        self.heading = float(raw_state[2])
        self.speed = float(raw_state[3]) if len(raw_state) > 3 else 0
        self.steering = float(raw_state[4]) if len(raw_state) > 3 else 0

    def create_box(self):
        return S.create_box((self.loc.x, self.loc.y, self.heading))

    def get_pos(self):
        return (self.loc.x, self.loc.y, self.heading)


class _KineticModel:
    def __init__(self):
        self.wheel_base, self.step_len, self.n_step = T.WHEEL_BASE, STEP_LENGTH, NUM_STEP
        self.speed_range, self.angle_range, self.mini_iter = VALID_SPEED, VALID_STEER, 20


class _Vehicle:
    def __init__(self):
        self.kinetic_model = _KineticModel()
        self.initial_state = self.state = None
        self.box = None
        self.trajectory = []

    def reset(self, state):
        self.initial_state = self.state = state
        self.box = state.create_box()
        self.trajectory = [state]


class _Map:
    def __init__(self, level):
        self.map_level = level
        self.case_id = None
        self.start = self.dest = self.start_box = self.dest_box = None
        self.xmin = self.xmax = self.ymin = self.ymax = 0
        self.obstacles, self.n_obstacle = [], 0
        self.scene = None

    def load(self, scene):
        self.scene = scene
        self.case_id = scene.case_id
        self.start, self.dest = State(list(scene.start) + [0, 0]), State(list(scene.dest) + [0, 0])
        self.start_box, self.dest_box = self.start.create_box(), self.dest.create_box()
        self.xmin, self.xmax, self.ymin, self.ymax = (float(v) for v in scene.bbox)
        self.obstacles = [scene.verts[i, :scene.nvert[i]].copy() for i in range(scene.n_obst)]
        self.n_obstacle = len(self.obstacles)
        return self.start


class CarParking:
    metadata = {'render_mode': ['human', 'rgb_array']}

    def __init__(self, render_mode=None, fps=100, verbose=True, use_lidar_observation=True,
                 use_img_observation=True, use_action_mask=True, device='cuda:0', max_obstacles=128, seed=None,
                 level='Normal'):
        import torch
        from .batch_env import ParkingBatch
        self.verbose, self.fps = verbose, fps
        self.render_mode = 'human' if render_mode is None else render_mode
        self.use_lidar_observation, self.use_img_observation, self.use_action_mask = \
            use_lidar_observation, bool(use_img_observation), use_action_mask
        self.level = level
        self.t = 0.0
        self.tgt_repr_size = 5
        self.rng = np.random.default_rng(seed)
        self._torch = torch
        self._batch = ParkingBatch(1, max_obstacles, device=device, obs_dtype=torch.float64,
                                   action_dtype=torch.float64, image=self.use_img_observation)
        self._pool = None
        self._dlp_path = None
        self.map = _Map(level)
        self.vehicle = _Vehicle()
        self.reward = self.prev_reward = self.accum_arrive_reward = 0.0
        self.action_space = Box(np.array([VALID_STEER[0], VALID_SPEED[0]]), np.array([VALID_STEER[1], VALID_SPEED[1]]))
        self.observation_space = {}
        if use_action_mask:
            self.observation_space['action_mask'] = Box(0, 1, shape=(N_DISCRETE_ACTION,), dtype=np.float64)
        if self.use_img_observation:                       # car_parking_base.py:87-93
            self.observation_space['img'] = Box(0, 255, shape=(L.IMG_SIZE, L.IMG_SIZE, L.IMG_CHANNELS), dtype=np.uint8)
            self.raw_img_shape = (256, 256, 3)
        if use_lidar_observation:
            self.observation_space['lidar'] = Box(0, LIDAR_RANGE, shape=(LIDAR_NUM,), dtype=np.float64)
        self.observation_space['target'] = Box(np.array([0, -1, -1, -1, -1]), np.array([MAX_DIST_TO_DEST, 1, 1, 1, 1]),
                                               dtype=np.float64)
        self.is_open = True

    # -- scene handling ------------------------------------------------------------------------------
    def set_level(self, level=None):
        self.level = 'Normal' if level is None else level
        self.map = _Map(self.level)

    def _draw_scene(self, case_id, data_dir):
        if self.level == 'dlp':
            if self._pool is None or data_dir != self._dlp_path:
                if data_dir is None:
                    self._pool = S.DlpScenePool()
                else:
                    from .dlp_io import pool_from_pickle
                    self._pool = pool_from_pickle(data_dir)
                self._dlp_path = data_dir
            return self._pool.sample(case=case_id, rng=self.rng)
        return S.generate_scene(self.level, self.rng, case_id=case_id)

    def reset(self, case_id=None, data_dir=None, level=None):
        self.reward = self.prev_reward = self.accum_arrive_reward = 0.0
        self.t = 0.0
        if level is not None:
            self.set_level(level)
        scene = self._draw_scene(case_id, data_dir)
        return self.reset_to_scene(scene)

    def reset_to_scene(self, scene):
        """reset onto an explicit `hope_amd.scenes.Scene` (deterministic tests, replay)."""
        self.reward = self.prev_reward = self.accum_arrive_reward = 0.0
        self.t = 0.0
        initial_state = self.map.load(scene)
        self.map.map_level = scene.map_level
        self.vehicle.reset(initial_state)
        self._batch.set_scenes([0], [scene])
        return self.step()[0]

    def _obs_from(self, o, img_chw):
        """the observation dict from one `ParkingBatch.download_outputs()` snapshot (copies: the snapshot is reused)"""
        obs = {'img': None, 'lidar': None, 'target': None, 'action_mask': None}
        if self.use_img_observation:                       # processed_img / 255.0, (W, H, C)  observation_processor.py:14
            im = o['img'][0] if img_chw else o['img'][0].transpose(1, 2, 0)
            obs['img'] = im.astype(np.float64) / 255.0
        if self.use_lidar_observation:
            obs['lidar'] = o['lidar'][0].copy()
        if self.use_action_mask:
            obs['action_mask'] = o['action_mask'][0].copy()
        obs['target'] = o['target'][0].copy()
        return obs

    def _last_reset_obs(self, img_chw):
        """the observation of the reset just done, in the wrapper's layout (buffers still hold it: no second launch)"""
        return self._obs_from(self._batch.download_outputs(), img_chw)

    # -- the step ---------------------------------------------------------------------------------------
    def step(self, action=None):
        """action: physical (steer [rad], speed [m/s]) or None (the reset observation)."""
        return self._step(action, physical=True)

    def _step(self, action=None, physical=True, img_chw=False):
        """physical=False: `action` is the wrapper's [-1, 1] action and the kernel applies action_rescale itself
        (k_kinematics); img_chw=True: obs['img'] stays channel-first as k_bev_image writes it (what the wrapper returns)."""
        torch, b = self._torch, self._batch
        assert self.vehicle.state is not None
        speed = steer = 0.0
        if action is not None:
            act = np.asarray(action, dtype=np.float64).reshape(2)
            a = torch.as_tensor(act.reshape(1, 2), device=b.device)
            b.step(a, stages=L.STAGE_ALL | (L.ACTION_PHYSICAL if physical else 0))
            if not physical:                                # State.steering / State.speed hold the physical values
                act = np.clip(act, -1, 1) * np.array([VALID_STEER[1], VALID_SPEED[1]])
            steer, speed = float(np.clip(act[0], *VALID_STEER)), float(np.clip(act[1], *VALID_SPEED))
        else:
            b.reset_obs()
        o = b.download_outputs()                            # ONE packed device-to-host copy + one synchronisation per step
        pose = o['pose'][0]
        self.t += 1
        prev = self.vehicle.state.get_pos()
        self.vehicle.state = State([pose[0], pose[1], pose[2], speed, steer])
        self.vehicle.box = self.vehicle.state.create_box()
        # vehicle.trajectory (vehicle.py:144,158; car_parking_base.py:274-276): of a step's sub-step states only the last
        # kept one stays; the action-less reset step and a step blocked at its first sub-step (collision -> retreat) add
        # nothing -- the rule the device-side trajectory ring follows.  (A zero-speed action keeps its sub-steps.)
        if action is not None and (self.vehicle.state.get_pos() != prev or speed == 0):
            self.vehicle.trajectory.append(self.vehicle.state)
        observation = self._obs_from(o, img_chw)
        status = Status(int(o['status'][0]))
        reward_info = OrderedDict(zip(REWARD_WEIGHT.keys(), (float(v) for v in o['reward_info'][0])))
        info = OrderedDict({'reward_info': reward_info, 'path_to_dest': None})
        w = o['rs_word'][0]
        if w[6]:
            n = int(w[5])
            info['path_to_dest'] = PATH(o['rs_lengths'][0, :n].copy(), [TYPE_NAMES[int(c)] for c in w[:n]],
                                        self.vehicle.state.get_pos())
        # what the wrapper adds on top (env_wrapper.py:10-35,80) leaves the kernels ready-made: k_post's shaped reward and done
        self._wrapped = (float(o['reward'][0]), bool(o['done'][0]))
        return observation, reward_info, status, info

    def render(self, mode='human'):
        return None

    def close(self):
        if self.is_open:
            self._batch.close()
            self.is_open = False


# ---- env_wrapper.py ----------------------------------------------------------------------------------
class CarParkingWrapper:
    """CarParkingWrapper (env_wrapper.py:58-85; gym.Wrapper attribute pass-through).  Its three functions are NOT re-derived
    on the host: handed the wrapper's [-1, 1] action the step kernels apply `action_rescale` themselves (k_kinematics;
    env_wrapper.py:37-50), k_post emits the shaped reward of `reward_shaping` (:10-35) and `done` (:80), and the image
    leaves k_bev_image channel-first, which is `observation_rescale`'s transpose (:52-55).  A caller that passes its own
    `action_func(action, action_space)` / `reward_func(obs, reward_info, status, info)` / `observation_func(obs)` gets
    them applied on the host instead (the reference's constructor signature)."""

    def __init__(self, env, action_func=None, reward_func=None, observation_func=None):
        self.env = env
        self.reward_func, self.action_func, self.obs_func = reward_func, action_func, observation_func
        self.observation_shape = {k: env.observation_space[k].shape for k in env.observation_space}
        if 'img' in self.observation_shape:                # env_wrapper.py:69-71
            w, h, c = self.observation_shape['img']
            self.observation_shape['img'] = (c, w, h)

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.env, name)

    def _obs(self, obs):
        return obs if self.obs_func is None else self.obs_func(obs)

    def step(self, action=None):
        chw = self.obs_func is None
        if action is None:                                 # env_wrapper.py:74-75
            return self._obs(self.env._step(None, img_chw=chw)[0])
        if self.action_func is None:
            np.random.random()                             # the reference's rescale draws from the global RNG every call (:48)
            obs, reward_info, status, info = self.env._step(action, physical=False, img_chw=chw)
        else:
            obs, reward_info, status, info = self.env._step(self.action_func(action, self.env.action_space), img_chw=chw)
        if self.reward_func is None:
            reward, done = self.env._wrapped
            info['status'] = status
        else:
            obs, reward, status, info = self.reward_func(obs, reward_info, status, info)
            done = status != Status.CONTINUE
        return self._obs(obs), reward, done, info

    def reset(self, *args):
        self.env.reset(*args)                              # (its own observation is rebuilt below in the wrapper's layout)
        return self._obs(self.env._last_reset_obs(self.obs_func is None))
