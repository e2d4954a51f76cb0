"""Oracle cartpole env: the reference's CartpoleEnv logic over the C oracle physics.
TEST INFRASTRUCTURE ONLY.

Follows reference envs/cartpole/cartpole_env.py:36-51 (CartpoleRobot.step), :109-121
(reset_model), :123-133 (_get_obs), :135-153 (step), :155-187 (_compute_reward), :189-192
(_check_termination); envs/common/robot_interface.py:493-508 (step_pd);
envs/common/mujoco_env.py:113-127 (reset / set_state).  Random draws come from oracle/rng.py
(the reference uses the global np.random stream).
"""
import numpy as np

from . import rng
from .physics import OracleSim


class OracleCartpoleEnv:
    def __init__(self, model, seed=0, env_id=0, kp=100.0, kd=10.0, frame_skip=4, max_traj_len=0):
        self.sim = OracleSim(model)
        self.seed, self.env_id = seed, env_id
        self.kp, self.kd, self.frame_skip, self.max_traj_len = kp, kd, frame_skip, max_traj_len
        self.gear = float(model.actuator_gear[0])
        self.reset_count = 0
        self.traj_len = 0

    def _obs(self):
        q, v = self.sim.qpos, self.sim.qvel
        return np.array([q[0], np.cos(q[1]), np.sin(q[1]), v[0], v[1]])

    def set_state(self, qpos, qvel):
        self.sim.qpos[:] = qpos
        self.sim.qvel[:] = qvel
        self.sim.forward(actuation=False)

    def reset(self):
        s, e, c = self.seed, self.env_id, self.reset_count
        self.sim.reset_data()
        pole = rng.uniform(s, e, rng.STREAM_RESET, c, 0, -np.pi, np.pi)
        qpos = np.array([0.0, pole])
        qpos[0] += rng.uniform(s, e, rng.STREAM_RESET, c, 1, -0.1, 0.1)
        qpos[1] += rng.uniform(s, e, rng.STREAM_RESET, c, 2, -0.1, 0.1)
        qvel = np.array([rng.uniform(s, e, rng.STREAM_RESET, c, 3, -0.1, 0.1),
                         rng.uniform(s, e, rng.STREAM_RESET, c, 4, -0.1, 0.1)])
        self.reset_count += 1
        self.traj_len = 0
        self.set_state(qpos, qvel)
        return self._obs()

    def step(self, action):
        """action: float32 scalar/array as produced by the policy."""
        a = np.clip(np.asarray(action, dtype=np.float32).reshape(-1), -0.8, 0.8)
        target = float(a[0])
        sim = self.sim
        for _ in range(self.frame_skip):
            q = sim.actuator_length[0] / self.gear
            w = sim.actuator_velocity[0] / self.gear
            tau = self.kp * (target - q) + self.kd * (0.0 - w)
            sim.ctrl[0] = tau
            sim.step()
        obs = self._obs()
        cos = obs[1]
        terms = dict(
            upright=0.35 * (1.0 + cos) / 2.0 + 0.35 * np.exp(-2.0 * (1.0 - cos) ** 2),
            center=0.1 * np.exp(-2.0 * obs[0] ** 2),
            velocity=0.1 * np.exp(-0.05 * obs[4] ** 2),
            action=0.1 * np.exp(-1.0 * float(a[0] * a[0])),
        )
        done = bool(np.abs(obs[0]) > 0.99)
        self.traj_len += 1
        return obs, sum(terms.values()), done, terms

    def step_auto(self, action):
        """env.step + RolloutWorker's truncation / reset bookkeeping (rollout_worker.py:142-181)."""
        obs, r, done, terms = self.step(action)
        truncated = self.max_traj_len > 0 and self.traj_len >= self.max_traj_len
        flags = int(done) | (2 if truncated else 0)
        term_obs = obs
        if self.max_traj_len > 0 and (done or truncated):
            obs = self.reset()
        return obs, r, flags, term_obs, terms
