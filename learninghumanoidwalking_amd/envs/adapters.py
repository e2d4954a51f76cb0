"""N=1 views of the batched environments with the reference's single-env surface
(reference envs/common/base_humanoid_env.py, tests/test_environments.py:39-266): ``reset() -> ndarray``,
``step(a) -> (ndarray, float, bool, dict)``, ``observation_space`` / ``action_space`` zero arrays,
``obs_mean`` / ``obs_std``, ``robot.{iteration_count, mirrored_obs, mirrored_acts, clock_inds}``."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .cartpole import CartpoleSpec
from .h1 import H1Spec
from .h1_walk import H1WalkSpec
from .jvrc_step import JvrcStepSpec
from .jvrc_walk import JvrcWalkSpec


class _InterfaceView:
    """Read-only subset of the reference's RobotInterface (envs/common/robot_interface.py:60-185) over the device state
    of the single env: what tests and evaluation scripts query after reset()/step().  Mutators (set_pd_gains,
    set_motor_torque, step ...) are the kernel's business and are not offered."""

    def __init__(self, owner):
        self._o = owner

    def nq(self): return self._o._env.nq
    def nv(self): return self._o._env.nv
    def nu(self): return self._o.spec.act_dim
    def sim_dt(self): return self._o.spec.sim_dt
    def get_robot_mass(self): return float(self._o.spec.model().totalmass)
    def get_qpos(self): return self._o._env.get_state()[0][0].copy()
    def get_qvel(self): return self._o._env.get_state()[1][0].copy()
    def get_gear_ratios(self): return np.asarray(self._o.spec.model().actuator_gear, dtype=float).copy()
    def get_motor_names(self): return list(self._o.spec.model().actuator_names)
    def get_act_joint_positions(self): return self._o._env.get_actuator_state()[0][0]
    def get_act_joint_velocities(self): return self._o._env.get_actuator_state()[1][0]
    def get_act_joint_torques(self): return self._o._env.get_actuator_state()[2][0]


class _DataView:
    """`env.data.qpos` / `.qvel` as read-only snapshots (the reference exposes mjData)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def qpos(self): return self._o._env.get_state()[0][0].copy()

    @property
    def qvel(self): return self._o._env.get_state()[1][0].copy()


class _SingleEnv:
    TERMS: list = []

    def __init__(self, spec, seed=0, device=0):
        self.spec = spec
        self._env = spec.make_batched(1, seed=seed, device=device, max_traj_len=0)
        self.observation_space = np.zeros(spec.obs_dim)
        self.action_space = np.zeros(spec.act_dim)
        self.base_obs_len, self.history_len = getattr(spec, "base_obs_dim", spec.obs_dim), getattr(spec, "history_len", 1)
        if spec.obs_mean is not None:
            self.obs_mean, self.obs_std = np.asarray(spec.obs_mean), np.asarray(spec.obs_std)
        self.robot = SimpleNamespace(iteration_count=np.inf)
        # reference attribute names (tests/test_environments.py:232-246): the task runs inside the kernel, so `task` only
        # describes it; `interface` / `data` are read-only views of the device state, `model` is the compiled model
        self.task = SimpleNamespace(name=type(spec).__name__.replace("Spec", ""), reward_terms=list(self.TERMS))
        self.model = spec.model()
        self.interface = _InterfaceView(self)      # (the get_act_joint_* getters raise for cartpole: its kernel keeps no such fields)
        self.data = _DataView(self)
        self._mask = torch.ones(1, dtype=torch.uint8, device=self._env.device)
        self._act = torch.zeros(1, spec.act_dim, dtype=torch.float32, device=self._env.device)

    def reset(self):
        return self._env.reset(self._mask).cpu().numpy()[0].astype(np.float64)

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, -1)
        if a.shape[1] != self.spec.act_dim:
            raise AssertionError(f"Action vector length expected to be: {self.spec.act_dim} but is {a.shape[1]}")
        self._act.copy_(torch.from_numpy(a))
        obs, rew, done, _ = self._env.step(self._act)
        terms = self._env.rew_terms.cpu().numpy()[0]
        info = {k: float(v) for k, v in zip(self.TERMS, terms)}
        return obs.cpu().numpy()[0].astype(np.float64), float(rew.cpu()[0]), bool(int(done.cpu()[0]) & 1), info

    def get_state(self):
        q, v = self._env.get_state()
        return q[0], v[0]

    def close(self):
        self._env.close()


class CartpoleEnv(_SingleEnv):
    TERMS = ["upright", "center", "velocity", "action"]          # cartpole_env.py:182-187

    def __init__(self, path_to_yaml=None, seed=0, device=0):
        super().__init__(CartpoleSpec(), seed=seed, device=device)
        self.robot.iteration_count = 0


class JvrcWalkEnv(_SingleEnv):
    TERMS = ["foot_frc_score", "foot_vel_score", "root_accel", "height_error", "com_vel_error", "yaw_vel_error",
             "upper_body_reward", "posture_error", "torque_penalty", "action_penalty"]   # walking_task.py:131-146

    def __init__(self, path_to_yaml=None, seed=0, device=0):
        spec = JvrcWalkSpec(yaml_path=path_to_yaml) if path_to_yaml else JvrcWalkSpec()
        super().__init__(spec, seed=seed, device=device)
        mo, ma, clock = spec.mirror_inds()
        self.robot.mirrored_obs, self.robot.mirrored_acts, self.robot.clock_inds = mo, ma, clock


class JvrcStepEnv(_SingleEnv):
    TERMS = ["foot_frc_score", "foot_vel_score", "orient_cost", "height_error", "step_reward", "upper_body_reward"]  # stepping_task.py:109-122

    def __init__(self, path_to_yaml=None, seed=0, device=0):
        spec = JvrcStepSpec(yaml_path=path_to_yaml) if path_to_yaml else JvrcStepSpec()
        super().__init__(spec, seed=seed, device=device)
        mo, ma, clock = spec.mirror_inds()
        self.robot.mirrored_obs, self.robot.mirrored_acts, self.robot.clock_inds = mo, ma, clock
        self.robot.iteration_count = 0

    def reset(self):
        self._env.set_iteration(int(min(self.robot.iteration_count, 1 << 30)))   # rollout_worker.py:95 curriculum input
        return super().reset()


class H1Env(_SingleEnv):
    TERMS = ["com_vel_error", "yaw_vel_error", "height", "upperbody", "joint_torque_reward", "posture"]   # standing_task.py:99-106

    def __init__(self, path_to_yaml=None, seed=0, device=0):
        super().__init__(H1Spec(yaml_path=path_to_yaml) if path_to_yaml else H1Spec(), seed=seed, device=device)


class H1WalkEnv(_SingleEnv):
    TERMS = JvrcWalkEnv.TERMS          # same WalkingTask reward dictionary

    def __init__(self, path_to_yaml=None, seed=0, device=0):
        spec = H1WalkSpec(yaml_path=path_to_yaml) if path_to_yaml else H1WalkSpec()
        super().__init__(spec, seed=seed, device=device)
        mo, ma, clock = spec.mirror_inds()
        self.robot.mirrored_obs, self.robot.mirrored_acts, self.robot.clock_inds = mo, ma, clock
