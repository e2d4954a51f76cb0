"""Unitree H1 standing task: batched factory mirroring reference envs/h1/h1_env.py:11-55, envs/h1/h1_base.py:20-125,
tasks/standing_task.py:12-131, envs/common/domain_randomization.py:10-56 and envs/h1/configs/base.yaml.

The robot model is the hand-authored stand-in ``assets/h1_standin.xml`` (mujoco_menagerie is an empty submodule in the
reference checkout)."""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import yaml

from .. import mjcf
from ..batched_env import TASK_H1_STAND, BatchedEnv
from ..model import fit_stepper_limits

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
H1_STANDIN_XML = os.path.join(_ASSETS, "h1_standin.xml")
H1_BASE_YAML = os.path.join(_ASSETS, "h1_base.yaml")

LEG_JOINTS = ["left_hip_yaw", "left_hip_roll", "left_hip_pitch", "left_knee", "left_ankle",
              "right_hip_yaw", "right_hip_roll", "right_hip_pitch", "right_knee", "right_ankle"]  # gen_xml.py:9-20


def load_config(path):
    """YAML config with an optional ``inherits: <file in the same directory>`` key (values of the child win)."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg.pop("timing", None)
    parent = cfg.pop("inherits", None)
    if parent:
        base = load_config(os.path.join(os.path.dirname(os.path.abspath(path)), parent))
        base.update(cfg)
        cfg = base
    return cfg


@dataclass
class H1Spec:
    yaml_path: str = H1_BASE_YAML
    xml_path: str = H1_STANDIN_XML
    name: str = "h1"
    obs_dim: int = 35
    act_dim: int = 10
    step_kernel_name: str = "humanoid_kernel<0, 2, 32>"     # rocprof name of the control-step kernel (MODE 0, TASK_STAND)
    cfg: dict = field(default_factory=dict)

    def __post_init__(self):
        self.cfg = load_config(self.yaml_path)
        c = self.cfg
        self.sim_dt, self.control_dt = float(c["sim_dt"]), float(c["control_dt"])
        self.history_len = int(c.get("obs_history_len", 1))     # base_humanoid_env.py:53,177-197 (kept above the kernels: BatchedEnv)
        if self.history_len < 1:
            raise ValueError("obs_history_len must be >= 1")
        self.action_smoothing = float(c["action_smoothing"])
        g = c["pdgains"]
        self.kp = np.array([g[j][0] for j in LEG_JOINTS], dtype=float)   # h1_base.py:48-51
        self.kd = np.array([g[j][1] for j in LEG_JOINTS], dtype=float)
        self.half_sitting_pose = np.array(c["half_sitting_pose"], dtype=float)
        self.nominal_pose = np.concatenate([[0, 0, 0.98], [1, 0, 0, 0], self.half_sitting_pose])   # h1_base.py:57-59
        self.init_noise_deg = float(c.get("init_noise") or 0.0)
        on = c.get("observation_noise") or {}
        self.obs_noise_enabled = bool(on.get("enabled", False))
        self.obs_noise_type = str(on.get("type", "uniform"))
        if self.obs_noise_enabled and self.obs_noise_type not in ("uniform", "gaussian"):
            raise ValueError("Observation noise type must be 'uniform' or 'gaussian'")     # base_humanoid_env.py:328
        sc, mult = on.get("scales", {}), float(on.get("multiplier", 1.0))
        # per-observation-entry noise half-widths (base_humanoid_env.py:307-338; groups h1_base.py:107-113)
        self.obs_noise_scale = np.concatenate([
            np.full(2, sc.get("root_orient", 0.0)), np.full(3, sc.get("root_ang_vel", 0.0)), np.full(10, sc.get("motor_pos", 0.0)),
            np.full(10, sc.get("motor_vel", 0.0)), np.full(10, sc.get("motor_tau", 0.0))]) * mult * float(self.obs_noise_enabled)
        # the kernel reads the type from the sign: scale > 0 uniform in [-scale, scale], scale < 0 Gaussian with std -scale
        self.obs_noise_param = -self.obs_noise_scale if self.obs_noise_type == "gaussian" else self.obs_noise_scale
        pc = c.get("perturbation") or {}
        self.perturb_interval = int(pc["interval"] / self.control_dt) if pc.get("enable") else 0   # base_humanoid_env.py:86-92
        self.perturb_bodies = list(pc.get("bodies", []))
        self.force_magnitude, self.torque_magnitude = float(pc.get("force_magnitude", 0)), float(pc.get("torque_magnitude", 0))
        dc = c.get("dynamics_randomization") or {}
        self.dynrand_interval = int(dc["interval"] / self.control_dt) if dc.get("enable") else 0     # base_humanoid_env.py:78-84
        # h1_env.py:41-55
        self.obs_mean = np.concatenate([np.zeros(5), self.half_sitting_pose, np.zeros(10), np.zeros(10)])
        self.obs_std = np.concatenate([[0.2, 0.2, 1, 1, 1], 0.5 * np.ones(10), 4 * np.ones(10), 100 * np.ones(10)])
        self._model = None
        self._apply_history()

    def _apply_history(self):
        from .jvrc_walk import JvrcWalkSpec
        JvrcWalkSpec._apply_history(self)    # (h1_env.py:54-55: the same np.tile over the history)

    @property
    def frame_skip(self) -> int:
        return int(self.control_dt / self.sim_dt)

    def model(self):
        if self._model is None:
            m = mjcf.compile_file(self.xml_path, self.sim_dt)
            if [m.jnt_names[j] for j in m.actuator_trnid] != LEG_JOINTS or m.nq != 17 or m.nv != 16:
                raise ValueError("model does not have the H1 leg actuator layout (free root + 10 leg hinges)")
            # h1_base.py:44-45: masses edited after compilation (no mj_setConst: invweight0 / meaninertia stay as compiled)
            m.arrays["body_mass"][m.body_id("pelvis")] = 8.89
            m.arrays["body_mass"][m.body_id("torso_link")] = 21.289
            m.totalmass = float(m.arrays["body_mass"].sum())
            # (a full menagerie H1 keeps its arm links welded to the torso: folded if there are too many; the torso and the
            # perturbed bodies stay bodies)
            # randomize_dynamics (domain_randomization.py:44-49) rescales the mass and shifts the inertial offset of the pelvis AND of the
            # body of every leg joint relative to the default model: none of them may absorb a welded link (a sole plate under an ankle
            # link would silently change the randomisation base) -- such a link stays a body, and the fit raises if that is one too many
            rand = ("pelvis",) + tuple(m.body_names[int(m.jnt_bodyid[m.jnt_id(j)])] for j in LEG_JOINTS)
            self._model = fit_stepper_limits(m, 15, keep=("torso_link",) + tuple(getattr(self, "perturb_bodies", ())), protect=rand)
        return self._model

    def mirror_tables(self):
        return None    # the reference's H1Env defines no mirror indices (run_experiment.py:127-128 falls back to no mirror)

    def rand_bodies(self):
        """pelvis + the body of each leg joint (domain_randomization.py:44-49)."""
        m = self.model()
        return [m.body_id("pelvis")] + [int(m.jnt_bodyid[m.jnt_id(j)]) for j in LEG_JOINTS]

    def rand_dofs(self):
        m = self.model()
        return [int(m.jnt_dofadr[m.jnt_id(j)]) for j in LEG_JOINTS]

    def body_ids(self):
        m = self.model()
        return [m.body_id("pelvis"), m.body_id("torso_link"), m.body_id("right_ankle_link"), m.body_id("left_ankle_link")]

    def action_offset(self):
        m = self.model()
        return np.array([self.nominal_pose[m.jnt_qposadr[m.jnt_id(j)]] for j in LEG_JOINTS])

    def task_params(self):
        """LHW_TP_* layout for LHW_TASK_H1_STAND."""
        return np.concatenate([[0.98, np.deg2rad(self.init_noise_deg), self.force_magnitude, self.torque_magnitude],
                               self.obs_noise_param])

    def task_iparams(self):
        m = self.model()
        pb = [m.body_id(b) for b in self.perturb_bodies] + [0, 0]
        return self.body_ids() + [self.dynrand_interval, self.perturb_interval, len(self.perturb_bodies), pb[0], pb[1]] + \
            self.rand_dofs() + self.rand_bodies()

    def make_batched(self, n_envs, seed=0, device=0, max_traj_len=0, env_id_base=0) -> BatchedEnv:
        return BatchedEnv(self.model(), TASK_H1_STAND, n_envs, frame_skip=self.frame_skip, kp=self.kp, kd=self.kd, seed=seed,
                          device=device, max_traj_len=max_traj_len, env_id_base=env_id_base,
                          action_smoothing=self.action_smoothing, nominal_qpos=self.nominal_pose,
                          action_offset=self.action_offset(), task_params=self.task_params(), task_iparams=self.task_iparams(),
                          history_len=self.history_len)

    def algorithmic_bytes_per_env_step(self) -> int:
        """State record (168 f64) + per-env randomised model parameters (128 f64) read + written, action in, obs x2, reward, flags, 6 terms."""
        return 2 * (168 + 128) * 8 + 10 * 4 + 2 * 35 * 4 + 4 + 1 + 6 * 4

    def algorithmic_flops_per_env_step(self) -> int:
        return 65_000 * self.frame_skip
