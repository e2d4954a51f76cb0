"""Task code OUTSIDE the fused kernels: the consumer of the batched sim facade (include/lhw.h: LhwTaskInput).

The reference's robots call an exchangeable task once per control step (robots/robot_base.py:88-96: ``task.step()``,
``task.calc_reward(prev_torque, prev_action, action)``, ``task.done()``; tasks/base_task.py:41-70) and the task reads the robot
through ``RobotInterface``.  The stepper kernels fuse the reference's own tasks; for task code that is not compiled into them -- an
edited ``tasks/rewards.py``, a user's ``BaseTask`` -- the kernels export, per env and control step, everything those reads return
(``lhw_env_task_inputs_device``), and ``Rollout(task=...)`` / ``PPO(..., task=...)`` hand that record to a batched task object after
every control step and train on ITS reward and termination instead of the fused ones:

    class MyTask(VectorTask):
        def evaluate(self, ti):            # ti: TaskInputs (named float64 device tensors, [N] or [N, k])
            return reward, done            # float tensor [N], bool tensor [N]

Two tasks ship: ``VectorWalkingTask`` -- WalkingTask.calc_reward / done (tasks/walking_task.py:85-147,184-192) on torch tensors,
term weights adjustable -- and ``PerEnvRewards`` -- the slow path proper: it calls the functions of a ``tasks/rewards.py`` MODULE
(the reference's own file, or an edited copy) env by env on the host, unchanged.  The task's state machine that feeds the
OBSERVATION (gait phase, walk mode, mode_ref) stays in the kernel; the record carries it (``phase``, ``mode``, ``mode_ref``).
With a task plugged in, the env never resets itself (it is created with ``max_traj_len = 0``): the rollout truncates and resets on
the host through ``lhw_env_reset(mask)``, which runs the same reset code with the same random draws as the in-kernel auto-reset.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


class _DevArray:
    """a raw device pointer as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


class TaskInputs:
    """Named views of the [N, LHW_TASK_INPUT_DIM] float64 record the last control step exported (include/lhw.h: enum LhwTaskInput):
    grf_r grf_l contact_z foot_contact self_collision phase mode mode_ref rfoot_vel lfoot_vel root_vel_local root_xpos head_xpos
    rfoot_xpos lfoot_xpos qpos qvel qacc act_pos act_vel act_tau prev_torque prev_action action root_xmat."""

    def __init__(self, rec: torch.Tensor, nq: int, nv: int, nu: int):
        self.rec, self.n_envs = rec, rec.shape[0]
        self._cut = dict(qpos=nq, qvel=nv, qacc=nv, act_pos=nu, act_vel=nu, act_tau=nu, prev_torque=nu, prev_action=nu, action=nu)

    def __getattr__(self, name):
        f = _lib.TASK_INPUT_FIELDS.get(name)
        if f is None:
            raise AttributeError(name)
        o, n = f
        n = self._cut.get(name, n)
        return self.rec[:, o] if n == 1 else self.rec[:, o:o + n]

    def numpy(self) -> dict:
        """host copy, as BatchedEnv.get_task_inputs() returns it"""
        return _lib.split_task_inputs(self.rec.cpu().numpy(), self._cut["qpos"], self._cut["qvel"], self._cut["action"])


def device_task_inputs(env) -> TaskInputs:
    """The env's task-input record as device tensors (no copy; arms the export on first use).  The view stays valid until the env
    is destroyed; its contents are those of the env's last control step on the stream that ran it."""
    view = getattr(env, "_task_inputs_view", None)
    if view is None:
        env.enable_task_inputs(True)
        p = ctypes.c_void_p()
        _lib.check(env._L.lhw_env_task_inputs_device(env._h, ctypes.byref(p)))
        if not p.value:
            raise _lib.LhwError(-4, "this env exports no task inputs (humanoid tasks only)")
        rec = torch.as_tensor(_DevArray(p.value, (env.n_envs, _lib.TASK_INPUT_DIM)), device=env.device)
        view = env._task_inputs_view = TaskInputs(rec, env.nq, env.nv, env.act_dim)
    return view


class VectorTask:
    """Batched counterpart of the reference's BaseTask (tasks/base_task.py:12-83) for one control step of N envs.

    `reward_only = True` declares that the task changes the REWARD only -- its termination rule is the fused task's (the common
    case: an edited tasks/rewards.py, other term weights).  The rollout then stays resident (one launch per rollout, the kernel's
    own termination / truncation / resets), exports the record of every control step, and `evaluate` is called ONCE over the whole
    [T * N] batch after the launch; the `done` it returns is ignored (Rollout._collect_resident_hooked).  With False (default)
    the task is consulted after every control step and decides terminations itself (Rollout._collect_hooked)."""
    reward_only = False

    def evaluate(self, ti: TaskInputs):
        """-> (reward [N] float tensor, done [N] bool tensor); called once per control step, after the env step"""
        raise NotImplementedError

    def reset(self, mask: torch.Tensor):
        """episodes of the envs in `mask` ([N] bool) start over (BaseTask.reset); optional"""


class VectorWalkingTask(VectorTask):
    """WalkingTask.calc_reward + done (reference tasks/walking_task.py:85-147, 184-192; tasks/rewards.py:9-194) on the exported
    inputs, vectorised over the batch in float64 torch.  `weights` overrides the reference's term weights by name."""

    reward_only = True      # done() below IS the fused WalkingTask.done: the resident rollout may keep the kernel's own flags
    TERMS = ("foot_frc_score", "foot_vel_score", "root_accel", "height_error", "com_vel_error", "yaw_vel_error", "upper_body_reward",
             "posture_error", "torque_penalty", "action_penalty")
    WEIGHTS = dict(foot_frc_score=0.225, foot_vel_score=0.225, root_accel=0.050, height_error=0.050, com_vel_error=0.150,
                   yaw_vel_error=0.150, upper_body_reward=0.050, posture_error=0.050, torque_penalty=0.025, action_penalty=0.025)
    STANDING, INPLACE, FORWARD = 0, 1, 2      # the kernels' mode codes (tasks/walking_task.py: WalkModes)

    def __init__(self, spec, device, weights: dict | None = None, height_limits=(0.6, 1.4)):
        if tuple(height_limits) != (0.6, 1.4):
            self.reward_only = False      # another termination rule than the fused one: consulted step by step
        self.w = dict(self.WEIGHTS, **(weights or {}))
        unknown = set(self.w) - set(self.TERMS)
        if unknown:
            raise KeyError(f"unknown reward terms {sorted(unknown)}")
        self.lut = torch.as_tensor(np.asarray(spec.clock_lut(), dtype=np.float64), device=device)     # [4][period]: r_frc r_vel l_frc l_vel
        self.mass = float(spec.model().body_mass.sum())
        self.goal_height = float(spec.goal_height)
        self.neutral = torch.as_tensor(np.asarray(spec.half_sitting_pose, dtype=np.float64), device=device)
        self.zlim = height_limits
        self.last_terms = None

    @staticmethod
    def _clock(a, b, ca, cb, cap):
        na = torch.clamp(a, max=cap) / cap * 2 - 1
        nb = torch.clamp(b, max=cap) / cap * 2 - 1
        return (torch.tan(np.pi / 4 * ca * na) + torch.tan(np.pi / 4 * cb * nb)) / 2

    def evaluate(self, ti: TaskInputs):
        ph, mode = ti.phase.long(), ti.mode.long()
        standing, inplace = mode == self.STANDING, mode == self.INPLACE
        one = torch.ones_like(ti.grf_r)
        r_frc = torch.where(standing, one, self.lut[0][ph]); r_vel = torch.where(standing, -one, self.lut[1][ph])
        l_frc = torch.where(standing, one, self.lut[2][ph]); l_vel = torch.where(standing, -one, self.lut[3][ph])
        ref = ti.mode_ref
        zero = torch.zeros_like(one)
        yaw_ref = torch.where(standing | (mode == self.FORWARD), zero, ref[:, 0])
        vx = torch.where(standing | inplace, zero, ref[:, 1]); vy = torch.where(standing | inplace, zero, ref[:, 2])
        gs = torch.sqrt(vx * vx + vy * vy)
        t = {}
        t["foot_frc_score"] = self._clock(ti.grf_l, ti.grf_r, l_frc, r_frc, self.mass * 9.8 * 0.5)
        t["foot_vel_score"] = self._clock(ti.lfoot_vel.norm(dim=1), ti.rfoot_vel.norm(dim=1), l_vel, r_vel, 0.2)
        t["root_accel"] = torch.exp(-0.25 * (ti.qvel[:, 3:6].abs().sum(1) + ti.qacc[:, 0:3].abs().sum(1)))
        cz = torch.where(ti.foot_contact != 0, ti.contact_z, zero)
        herr = (ti.root_xpos[:, 2] - cz - self.goal_height).abs()
        herr = torch.where(herr < 0.01 + 0.05 * gs, zero, herr)
        t["height_error"] = torch.exp(-40 * herr * herr)
        ex, ey = ti.root_vel_local[:, 0] - vx, ti.root_vel_local[:, 1] - vy
        t["com_vel_error"] = torch.exp(-10 * (ex * ex + ey * ey))
        t["yaw_vel_error"] = torch.exp(-10 * (ti.qvel[:, 5] - yaw_ref).abs() ** 3)
        t["upper_body_reward"] = torch.exp(-10 * (ti.head_xpos[:, :2] - ti.root_xpos[:, :2]).norm(dim=1))
        t["posture_error"] = torch.exp(-(self.neutral - ti.act_pos).norm(dim=1))
        nu = ti.action.shape[1]
        t["torque_penalty"] = torch.exp(-0.25 * ((ti.prev_torque - ti.act_tau).abs().sum(1) / nu))
        t["action_penalty"] = torch.exp(-5 * (ti.prev_action - ti.action).abs().sum(1) / nu)
        self.last_terms = {k: self.w[k] * t[k] for k in self.TERMS}
        reward = sum(self.last_terms[k] for k in self.TERMS)      # python sum() over the dict, left to right
        z = ti.qpos[:, 2]
        done = (z < self.zlim[0]) | (z > self.zlim[1]) | (ti.self_collision != 0)     # walking_task.py:184-192
        return reward, done


class VectorStandingTask(VectorTask):
    """StandingTask.calc_reward + done (reference tasks/standing_task.py:49-131) on the exported inputs -- the H1 standing env's task as a
    plug-in (`head_xpos` of that env is the torso link, `root_xmat` the pelvis frame).  `weights` overrides term weights by name."""

    reward_only = True
    TERMS = ("com_vel_error", "yaw_vel_error", "height", "upperbody", "joint_torque_reward", "posture")
    WEIGHTS = dict(com_vel_error=0.3, yaw_vel_error=0.3, height=0.1, upperbody=0.1, joint_torque_reward=0.1, posture=0.1)

    def __init__(self, spec, device, weights: dict | None = None, height_limits=(0.9, 1.4)):
        if tuple(height_limits) != (0.9, 1.4):
            self.reward_only = False
        self.w = dict(self.WEIGHTS, **(weights or {}))
        unknown = set(self.w) - set(self.TERMS)
        if unknown:
            raise KeyError(f"unknown reward terms {sorted(unknown)}")
        self.goal_height = float(getattr(spec, "goal_height", 0.98))          # standing_task.py:78 target_root_h
        self.neutral = torch.as_tensor(np.asarray(spec.action_offset(), dtype=np.float64), device=device)   # standing_task.py:33 (the nominal leg pose)
        self.zlim = height_limits
        self.last_terms = None

    def evaluate(self, ti: TaskInputs):
        R = ti.root_xmat.reshape(-1, 3, 3)
        dh = ti.head_xpos - ti.root_xpos
        hloc = torch.einsum("nji,nj->ni", R, dh)            # inv(root_pose) . head_pose: R^T (head - root)
        t = {}
        t["com_vel_error"] = torch.exp(-4 * (ti.root_vel_local[:, 0] ** 2 + ti.root_vel_local[:, 1] ** 2))
        t["yaw_vel_error"] = torch.exp(-4 * ti.qvel[:, 5] ** 2)
        t["height"] = torch.exp(-0.5 * (ti.root_xpos[:, 2] - self.goal_height) ** 2)
        t["upperbody"] = torch.exp(-40 * (hloc[:, 0] ** 2 + hloc[:, 1] ** 2))
        t["joint_torque_reward"] = torch.exp(-5e-5 * (ti.act_tau ** 2).sum(1))
        t["posture"] = torch.exp(-((ti.act_pos - self.neutral) ** 2).sum(1))
        self.last_terms = {k: self.w[k] * t[k] for k in self.TERMS}
        reward = sum(self.last_terms[k] for k in self.TERMS)
        z = ti.qpos[:, 2]
        done = (z < self.zlim[0]) | (z > self.zlim[1]) | (ti.self_collision != 0)     # standing_task.py:111-131
        return reward, done


class PerEnvRewards(VectorTask):
    """The slow path: WalkingTask.calc_reward assembled, env by env on the host, from the functions of a `tasks/rewards.py` MODULE
    -- the reference's own file loaded by path, or an edited copy: whatever its functions return is what the policy is trained
    on.  (Float64 numpy per env and control step: for small batches and for checking; use a VectorTask to train at scale.)"""

    def __init__(self, rewards_module, spec, weights: dict | None = None):
        self.rw, self.spec = rewards_module, spec
        self.w = dict(VectorWalkingTask.WEIGHTS, **(weights or {}))
        self.lut = np.asarray(spec.clock_lut(), dtype=np.float64)
        self.mass = float(spec.model().body_mass.sum())

    def evaluate(self, ti: TaskInputs):
        rw, w, lut = self.rw, self.w, self.lut
        h = ti.numpy()
        N = ti.n_envs
        rew = np.zeros(N)
        done = np.zeros(N, dtype=bool)
        for i in range(N):
            ph, mode = int(h["phase"][i]), int(h["mode"][i])
            r_frc, r_vel, l_frc, l_vel = (lambda p, k=k: lut[k, int(p)] for k in range(4))
            if mode == 0:
                r_frc = l_frc = lambda _: 1
                r_vel = l_vel = lambda _: -1
            yaw_ref, vx, vy = h["mode_ref"][i]
            if mode == 0:
                yaw_ref, vx, vy = 0.0, 0.0, 0.0
            elif mode == 1:
                vx, vy = 0.0, 0.0
            else:
                yaw_ref = 0.0
            goal = np.array([vx, vy])
            cz = h["contact_z"][i] if h["foot_contact"][i] else 0
            terms = dict(
                foot_frc_score=rw.calc_foot_frc_clock_reward(h["grf_l"][i], h["grf_r"][i], ph, l_frc, r_frc, self.mass),
                foot_vel_score=rw.calc_foot_vel_clock_reward(h["lfoot_vel"][i], h["rfoot_vel"][i], ph, l_vel, r_vel),
                root_accel=rw.calc_root_accel_reward(h["qvel"][i], h["qacc"][i]),
                height_error=rw.calc_height_reward(h["root_xpos"][i][2], self.spec.goal_height, float(np.linalg.norm(goal)), cz),
                com_vel_error=rw.calc_fwd_vel_reward(h["root_vel_local"][i][:2], goal),
                yaw_vel_error=rw.calc_yaw_vel_reward(h["qvel"][i][5], yaw_ref),
                upper_body_reward=np.exp(-10 * np.linalg.norm(h["head_xpos"][i][:2] - h["root_xpos"][i][:2])),
                posture_error=np.exp(-np.linalg.norm(np.asarray(self.spec.half_sitting_pose) - h["act_pos"][i])),
                torque_penalty=rw.calc_torque_reward(h["act_tau"][i], h["prev_torque"][i]),
                action_penalty=rw.calc_action_reward(h["action"][i], h["prev_action"][i]))
            rew[i] = sum(w[k] * terms[k] for k in VectorWalkingTask.TERMS)
            z = h["qpos"][i][2]
            done[i] = z < 0.6 or z > 1.4 or bool(h["self_collision"][i])
        dev = ti.rec.device
        return torch.as_tensor(rew, device=dev), torch.as_tensor(done, device=dev)
