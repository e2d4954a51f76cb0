"""Cartpole swing-up: batched factory mirroring reference envs/cartpole/cartpole_env.py:54-107."""
from __future__ import annotations

import os
from dataclasses import dataclass

from .. import mjcf
from ..batched_env import TASK_CARTPOLE, BatchedEnv

CARTPOLE_XML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "cartpole.xml")


@dataclass
class CartpoleSpec:
    sim_dt: float = 0.005      # cartpole_env.py:84
    control_dt: float = 0.02   # cartpole_env.py:85
    kp: float = 100.0          # cartpole_env.py:93
    kd: float = 10.0
    obs_dim: int = 5
    act_dim: int = 1
    name: str = "cartpole"
    obs_mean = None            # no fixed normalisation: PPO warms up a RunningMeanStd (ppo.py:99-103)
    obs_std = None

    @property
    def frame_skip(self) -> int:
        return int(self.control_dt / self.sim_dt)

    def model(self):
        return mjcf.compile_file(CARTPOLE_XML, self.sim_dt)

    step_kernel_name = "cartpole_step_kernel"

    def algorithmic_bytes_per_env_step(self) -> int:
        """State read + written once per control step (9 f64 + 2 i32 words), action in, obs/term_obs/reward/done/terms out."""
        return 2 * (9 * 8 + 2 * 4) + 4 + 2 * 5 * 4 + 4 + 1 + 4 * 4

    def algorithmic_flops_per_env_step(self) -> int:
        """~0.3 kFLOP per sim sub-step x frame_skip + reward/obs."""
        return 300 * self.frame_skip + 60

    def mirror_tables(self):
        return None            # cartpole has no mirror symmetry (run_experiment.py:127-128 falls back)

    def make_batched(self, n_envs, seed=0, device=0, max_traj_len=0, env_id_base=0):
        return make_cartpole(n_envs, seed=seed, device=device, max_traj_len=max_traj_len, env_id_base=env_id_base, spec=self)


def make_cartpole(n_envs: int, seed: int = 0, device=0, max_traj_len: int = 0, env_id_base: int = 0,
                  spec: CartpoleSpec | None = None) -> BatchedEnv:
    spec = spec or CartpoleSpec()
    return BatchedEnv(spec.model(), TASK_CARTPOLE, n_envs, frame_skip=spec.frame_skip, kp=[spec.kp], kd=[spec.kd],
                      seed=seed, device=device, max_traj_len=max_traj_len, env_id_base=env_id_base)
