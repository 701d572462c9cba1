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
REWARD_RATIO = 0.1
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

    # -- the step ---------------------------------------------------------------------------------------
    def step(self, action=None):
        """action: physical (steer [rad], speed [m/s]) or None (the reset observation)."""
        torch, b = self._torch, self._batch
        assert self.vehicle.state is not None
        if action is not None:
            a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 2), device=b.device)
            b.step(a, stages=L.STAGE_ALL | L.ACTION_PHYSICAL)
        else:
            b.reset_obs()
        torch.cuda.synchronize(b.device)
        pose = b.pose[0].cpu().numpy()
        self.t += 1
        speed = float(np.clip(action[1], *VALID_SPEED)) if action is not None else 0
        steer = float(np.clip(action[0], *VALID_STEER)) if action is not None else 0
        prev = self.vehicle.state.get_pos()
        self.vehicle.state = State([pose[0], pose[1], pose[2], speed, steer])
        self.vehicle.box = self.vehicle.state.create_box()
        # vehicle.trajectory (vehicle.py:144,158; car_parking_base.py:274-276): of a step's sub-step states only the last
        # kept one stays; the action-less reset step and a step blocked at its first sub-step (collision -> retreat) add
        # nothing -- the rule the device-side trajectory ring follows.  (A zero-speed action keeps its sub-steps.)
        if action is not None and (self.vehicle.state.get_pos() != prev or speed == 0):
            self.vehicle.trajectory.append(self.vehicle.state)
        observation = {'img': None, 'lidar': None, 'target': None, 'action_mask': None}
        if self.use_img_observation:                       # processed_img / 255.0, (W, H, C)  observation_processor.py:14
            observation['img'] = b.img[0].permute(1, 2, 0).cpu().numpy().astype(np.float64) / 255.0
        if self.use_lidar_observation:
            observation['lidar'] = b.lidar[0].cpu().numpy()
        if self.use_action_mask:
            observation['action_mask'] = b.action_mask[0].cpu().numpy()
        observation['target'] = b.target[0].cpu().numpy()
        status = Status(int(b.status[0].item()))
        ri = b.reward_info[0].cpu().numpy()
        reward_info = OrderedDict(zip(REWARD_WEIGHT.keys(), (float(v) for v in ri)))
        info = OrderedDict({'reward_info': reward_info, 'path_to_dest': None})
        w = b.rs_word[0].cpu().numpy()
        if w[6]:
            n = int(w[5])
            info['path_to_dest'] = PATH(b.rs_lengths[0, :n].cpu().numpy(), [TYPE_NAMES[int(c)] for c in w[:n]],
                                        self.vehicle.state.get_pos())
        return observation, reward_info, status, info

    def render(self, mode='human'):
        return None

    def close(self):
        if self.is_open:
            self._batch.close()
            self.is_open = False


# ---- env_wrapper.py ----------------------------------------------------------------------------------
def reward_shaping(*args):                                # env_wrapper.py:10-35
    obs, reward_info, status, info = args
    if status == Status.CONTINUE:
        reward = 0
        for k in REWARD_WEIGHT.keys():
            reward += REWARD_WEIGHT[k] * reward_info[k]
    elif status == Status.OUTBOUND:
        reward = -50
    elif status == Status.OUTTIME:
        reward = -1
    elif status == Status.ARRIVED:
        reward = 50
    elif status == Status.COLLIDED:
        reward = -50
    reward *= REWARD_RATIO
    info['status'] = status
    return obs, reward, status, info


def action_rescale(action, action_space, raw_action_range=(-1, 1), explore=True, epsilon=0.0):   # :37-50
    action = np.clip(action, *raw_action_range)
    action = action * (action_space.high - action_space.low) / 2 + (action_space.high + action_space.low) / 2
    if explore and np.random.random() < epsilon:
        action = action_space.sample()
    return action


def observation_rescale(obs):                             # :52-55
    if obs['img'] is not None:
        obs['img'] = obs['img'].transpose((2, 0, 1))
    return obs


class CarParkingWrapper:                                  # :58-85 (gym.Wrapper attribute pass-through)
    def __init__(self, env, action_func=action_rescale, reward_func=reward_shaping,
                 observation_func=observation_rescale):
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

    def step(self, action=None):
        if action is None:
            return self.obs_func(self.env.step()[0])
        action = self.action_func(action, self.env.action_space)
        returns = self.env.step(action)
        obs, reward, status, info = self.reward_func(*returns)
        obs = self.obs_func(obs)
        done = False if status == Status.CONTINUE else True
        return obs, reward, done, info

    def reset(self, *args):
        return self.obs_func(self.env.reset(*args))
