"""Unitree H1 walking task (`h1_walk`): batched factory mirroring reference envs/h1/h1_walk.py:20-148 -- the H1 robot
state, observation noise and domain randomisation of envs/h1/h1_base.py / base_humanoid_env.py combined with
tasks/walking_task.py (three walk modes, gait clock) -- and envs/h1/configs/walk.yaml.  Stand-in model as for ``h1``."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from ..batched_env import TASK_H1_WALK, BatchedEnv
from .h1 import _ASSETS, H1Spec
from .jvrc_walk import phase_clock_lut

H1_WALK_YAML = os.path.join(_ASSETS, "h1_walk.yaml")

# h1_walk.py:68-107: 35-D robot state (joint blocks are left(5) then right(5): yaw, roll, pitch, knee, ankle)
BASE_MIRROR_OBS = [-0.1, 1, -2, 3, -4,
                   -10, -11, 12, 13, 14, -5, -6, 7, 8, 9,
                   -20, -21, 22, 23, 24, -15, -16, 17, 18, 19,
                   -30, -31, 32, 33, 34, -25, -26, 27, 28, 29]
MIRROR_ACTS = [-5, -6, 7, 8, 9, -0.1, -1, 2, 3, 4]   # h1_walk.py:113


@dataclass
class H1WalkSpec(H1Spec):
    yaml_path: str = H1_WALK_YAML
    name: str = "h1_walk"
    obs_dim: int = 43
    step_kernel_name: str = "humanoid_kernel<0, 4, 32>"

    def __post_init__(self):
        super().__post_init__()
        t = self.cfg["task"]
        self.goal_height = float(t["goal_height"])
        self.total_duration, self.swing_duration, self.stance_duration = (
            float(t["total_duration"]), float(t["swing_duration"]), float(t["stance_duration"]))
        self.period = int(np.floor(2 * self.total_duration * (1 / self.control_dt)))   # walking_task.py:204
        # h1_walk.py:125-148
        self.obs_mean = np.concatenate([np.zeros(5), self.half_sitting_pose, np.zeros(10), np.zeros(10), [0, 0], [0.5, 0.5, 0.5, 0, 0, 0]])
        self.obs_std = np.concatenate([[0.2, 0.2, 1, 1, 1], 0.5 * np.ones(10), 4 * np.ones(10), 100 * np.ones(10), [1, 1],
                                       [1, 1, 1, 0.5, 0.5, 0.5]])
        self._apply_history()

    def clock_lut(self):
        return phase_clock_lut(self.swing_duration, self.stance_duration, 0.1, 1 / self.control_dt, self.period)

    def mirror_inds(self):
        ext = [len(BASE_MIRROR_OBS) + i for i in range(self.base_obs_dim - 35)]
        return BASE_MIRROR_OBS + ext, MIRROR_ACTS, ext[0:2]

    def mirror_tables(self):
        from .jvrc_walk import JvrcWalkSpec
        return JvrcWalkSpec.mirror_tables(self)       # same signed-permutation construction over this env's index lists

    def task_params(self):
        tp = super().task_params()
        tp[0] = self.goal_height
        return tp

    def make_batched(self, n_envs, seed=0, device=0, max_traj_len=0, env_id_base=0) -> BatchedEnv:
        return BatchedEnv(self.model(), TASK_H1_WALK, n_envs, frame_skip=self.frame_skip, kp=self.kp, kd=self.kd, seed=seed,
                          device=device, max_traj_len=max_traj_len, env_id_base=env_id_base,
                          action_smoothing=self.action_smoothing, nominal_qpos=self.nominal_pose,
                          action_offset=self.action_offset(), task_params=self.task_params(), task_iparams=self.task_iparams(),
                          clock_lut=self.clock_lut(), history_len=self.history_len)

    def algorithmic_bytes_per_env_step(self) -> int:
        return 2 * (168 + 128) * 8 + 10 * 4 + 2 * 43 * 4 + 4 + 1 + 10 * 4
