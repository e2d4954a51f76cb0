"""JVRC-1 walking task: batched factory mirroring reference envs/jvrc/jvrc_walk.py:13-67,
envs/jvrc/jvrc_base.py:20-145 and envs/jvrc/configs/base.yaml.

The robot model is the hand-authored stand-in ``assets/jvrc_standin.xml`` (the real
jvrc_mj_description submodule is absent from the reference checkout);
pass ``xml_path`` to use a real export of reference envs/jvrc/gen_xml.py instead.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import yaml

from .. import mjcf
from ..model import fit_stepper_limits
from ..batched_env import TASK_JVRC_WALK, BatchedEnv

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
JVRC_STANDIN_XML = os.path.join(_ASSETS, "jvrc_standin.xml")
JVRC_BASE_YAML = os.path.join(_ASSETS, "jvrc_base.yaml")

LEG_JOINTS = ["R_HIP_P", "R_HIP_R", "R_HIP_Y", "R_KNEE", "R_ANKLE_R", "R_ANKLE_P",
              "L_HIP_P", "L_HIP_R", "L_HIP_Y", "L_KNEE", "L_ANKLE_R", "L_ANKLE_P"]  # gen_xml.py:44-57

# jvrc_base.py:73-110 (29 robot-state entries) + identity on the external obs; mirrored_acts :110
BASE_MIRROR_OBS = [-0.1, 1, -2, 3, -4, 11, -12, -13, 14, -15, 16, 5, -6, -7, 8, -9, 10,
                   23, -24, -25, 26, -27, 28, 17, -18, -19, 20, -21, 22]
MIRROR_ACTS = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]


def phase_clock_lut(swing_duration, stance_duration, strict_relaxer, freq, period):
    """[4][period] table r_frc, r_vel, l_frc, l_vel of the "grounded" gait clocks at integer phases.

    Restates the knot construction of reference tasks/rewards.py:196-300 (create_phase_reward):
    8 knots per cycle (relaxed ends of right swing, first double stance, left swing, second double
    stance), repeated over three cycles, interpolated with scipy's PchipInterpolator exactly as the
    reference does.  The reference only ever evaluates the splines at integer phases, so the table
    is the whole function.
    """
    from scipy.interpolate import PchipInterpolator
    sw, st = swing_duration * freq, stance_duration * freq
    segs = [(0.0, sw), (sw, sw + st), (sw + st, 2 * sw + st), (2 * sw + st, 2 * (sw + st))]
    x = []
    for a, b in segs:
        off = (b - a) * strict_relaxer
        x += [a + off, b - off]
    x = np.array(x)
    last_off = (segs[3][1] - segs[3][0]) * strict_relaxer
    # right foot force clock: -1 in right swing, +1 otherwise; velocity clocks are the negation; left is the mirror
    r_frc = np.array([-1, -1, 1, 1, 1, 1, 1, 1], dtype=float)
    l_frc = np.array([1, 1, 1, 1, -1, -1, 1, 1], dtype=float)
    r_vel = np.array([1, 1, -1, -1, -1, -1, -1, -1], dtype=float)
    l_vel = np.array([-1, -1, -1, -1, 1, 1, -1, -1], dtype=float)
    xs = np.concatenate([x - x[-1] - last_off, x, x + x[-1] + last_off])
    ph = np.arange(int(period))
    return np.stack([PchipInterpolator(xs, np.tile(y, 3))(ph) for y in (r_frc, r_vel, l_frc, l_vel)])


@dataclass
class JvrcWalkSpec:
    yaml_path: str = JVRC_BASE_YAML
    xml_path: str = JVRC_STANDIN_XML
    name: str = "jvrc_walk"
    obs_dim: int = 37
    act_dim: int = 12
    step_kernel_name: str = "humanoid_kernel<0, 1, 32>"     # rocprof name of the control-step kernel (MODE 0, TASK_WALK)
    cfg: dict = field(default_factory=dict)

    def __post_init__(self):
        with open(self.yaml_path) as f:
            self.cfg = yaml.safe_load(f)
        c = self.cfg
        self.sim_dt, self.control_dt = float(c["sim_dt"]), float(c["control_dt"])
        self.history_len = int(c.get("obs_history_len", 1))     # base_humanoid_env.py:53,177-197 (kept above the kernels: BatchedEnv)
        if self.history_len < 1:
            raise ValueError("obs_history_len must be >= 1")
        # BaseHumanoidEnv applies these keys to any env that configures them (base_humanoid_env.py:76-92,221-225,247-305) -- but what
        # the reference DOES with them on a JVRC env differs by key (round 6, read off the reference's code):
        #   observation_noise        ignored: only H1BaseEnv._get_robot_state calls _apply_observation_noise (h1_base.py:107-115);
        #                            JvrcBaseEnv._get_robot_state (jvrc_base.py:133-138) never does -> accepted and ignored here too
        #   dynamics_randomization   raises: randomize_dynamics looks up the body "pelvis" (domain_randomization.py:44), which a JVRC
        #                            model does not have (its root is "PELVIS_S", jvrc_base.py:32) -> KeyError in the reference, refused here
        #   init_noise               runs there (base_humanoid_env.py:260-263, 278-305) and here: root z / roll / pitch / joint noise at every
        #                            reset, in every humanoid kernel (the JVRC auto-reset then computes the reset instead of copying a template)
        #   perturbation             runs there (base_humanoid_env.py:86-92, 224-225; domain_randomization.py:10-26) and here: the wrenches live
        #                            in the per-env HBM record (LhwEnvConfig.perturb_*; at most two bodies, as the H1 kernels)
        self.init_noise_deg = float(c.get("init_noise") or 0.0)
        pc = c.get("perturbation") or {}
        self.perturb_interval = int(pc["interval"] / self.control_dt) if pc.get("enable") else 0
        self.perturb_bodies = list(pc.get("bodies", [])) if self.perturb_interval > 0 else []
        self.force_magnitude, self.torque_magnitude = float(pc.get("force_magnitude", 0)), float(pc.get("torque_magnitude", 0))
        if len(self.perturb_bodies) > 2:
            raise NotImplementedError(f"perturbation of more than two bodies ({self.perturb_bodies}) in {self.yaml_path}")
        v = c.get("dynamics_randomization")
        if isinstance(v, dict) and v.get("enable", v.get("enabled", False)):
            raise NotImplementedError(f"dynamics_randomization is configured in {self.yaml_path}: the reference itself fails on it for a JVRC model "
                                      "(randomize_dynamics needs a body named 'pelvis', envs/common/domain_randomization.py:44)")
        self.action_smoothing = float(c["action_smoothing"])
        self.kp, self.kd = np.array(c["kp"], dtype=float), np.array(c["kd"], dtype=float)
        self.half_sitting_pose = np.deg2rad(np.array(c["half_sitting_pose"], dtype=float))
        # jvrc_base.py:52-54
        self.nominal_pose = np.concatenate([[0, 0, 0.81], [1, 0, 0, 0], self.half_sitting_pose])
        t = c["task"]
        self.goal_height = float(t["goal_height"])
        self.total_duration, self.swing_duration, self.stance_duration = (
            float(t["total_duration"]), float(t["swing_duration"]), float(t["stance_duration"]))
        self.period = int(np.floor(2 * self.total_duration * (1 / self.control_dt)))  # walking_task.py:204
        # jvrc_walk.py:43-63
        self.obs_mean = np.concatenate([np.zeros(5), self.half_sitting_pose, np.zeros(12), [0, 0, 0.5, 0.5, 0.5, 0, 0, 0]])
        self.obs_std = np.concatenate([[0.2, 0.2, 1, 1, 1], 0.5 * np.ones(12), 4 * np.ones(12), [1, 1, 1, 1, 1, 0.5, 0.5, 0.5]])
        self._model = None
        self._apply_history()

    def _apply_history(self):
        """obs_dim / obs_mean / obs_std for obs_history_len > 1 (jvrc_walk.py:62-63: np.tile over the history); called at the end
        of every __post_init__ of the Spec hierarchy (a parent's call leaves a child's longer base observation alone)."""
        base = type(self).__dataclass_fields__["obs_dim"].default
        if not hasattr(self, "base_obs_dim") and self.obs_dim != base:
            # the base observation is produced by the task's kernel: its width is not a free parameter of the Spec
            raise ValueError(f"{type(self).__name__}: obs_dim is fixed by the task ({base}); got obs_dim={self.obs_dim}")
        self.base_obs_dim, self.obs_dim = base, base * self.history_len
        # (a parent's __post_init__ runs this while obs_mean is still the parent's: only a vector of the base length is tiled here,
        # and PPO checks the final length against obs_dim before it hands the vectors to the kernels)
        if self.obs_mean is not None and len(self.obs_mean) == base and self.history_len > 1:
            self.obs_mean, self.obs_std = np.tile(self.obs_mean, self.history_len), np.tile(self.obs_std, self.history_len)

    @property
    def frame_skip(self) -> int:
        if np.around(self.control_dt % self.sim_dt, 6):  # robot_base.py:37-38
            raise Exception("Control dt should be an integer multiple of Simulation dt.")
        return int(self.control_dt / self.sim_dt)

    def model(self):
        if self._model is None:
            m = mjcf.compile_file(self.xml_path, self.sim_dt)
            names = [m.jnt_names[j] for j in m.actuator_trnid]
            if names != LEG_JOINTS or m.nq != 19 or m.nv != 18:
                raise ValueError("model does not have the JVRC leg actuator layout (free root + 12 leg hinges)")
            # a real JVRC export keeps ~30 arm / head / finger links welded to the torso after gen_xml.py:84-87 deleted their
            # joints: they are folded into the bodies they move with (exact; the head stays a body, the task reads its position)
            self._model = fit_stepper_limits(m, 18, keep=("NECK_P_S",) + tuple(getattr(self, "perturb_bodies", ())))
        return self._model

    def clock_lut(self):
        return phase_clock_lut(self.swing_duration, self.stance_duration, 0.1, 1 / self.control_dt, self.period)

    def mirror_inds(self):
        n_ext = self.base_obs_dim - 29
        ext = [len(BASE_MIRROR_OBS) + i for i in range(n_ext)]
        return BASE_MIRROR_OBS + ext, MIRROR_ACTS, ext[0:2]

    def mirror_tables(self):
        """((obs_src, obs_sign), (act_src, act_sign)): signed permutations of rl/envs/wrappers.py:78-85 as gathers."""
        if self.history_len > 1:
            # the reference's mirrored_obs lists base_obs_len indices only (jvrc_walk.py, h1_walk.py): its SymmetricEnv cannot
            # mirror a history observation either
            raise NotImplementedError("mirror loss with obs_history_len > 1: the reference defines mirror indices for the base observation only; train with --no-mirror")
        mo, ma, clock = self.mirror_inds()

        def tab(mirrored, clock_inds=()):
            n = len(mirrored)
            src, sign = np.zeros(n, np.int32), np.zeros(n, np.float32)
            for i, v in enumerate(mirrored):
                j = int(abs(v))
                src[j], sign[j] = i, np.sign(v)
            for c in clock_inds:
                sign[c] = -sign[c]  # sin(arcsin(c) + pi) == -c (wrappers.py:69-74)
            return src, sign

        return tab(mo, clock), tab(ma)

    def body_ids(self):
        m = self.model()
        return [m.body_id("PELVIS_S"), m.body_id("NECK_P_S"), m.body_id("R_ANKLE_P_S"), m.body_id("L_ANKLE_P_S")]

    def action_offset(self):
        m = self.model()
        return np.array([self.nominal_pose[m.jnt_qposadr[m.jnt_id(j)]] for j in LEG_JOINTS])  # base_humanoid_env.py:238-245

    def make_batched(self, n_envs, seed=0, device=0, max_traj_len=0, env_id_base=0) -> BatchedEnv:
        return BatchedEnv(self.model(), TASK_JVRC_WALK, n_envs, frame_skip=self.frame_skip, kp=self.kp, kd=self.kd, seed=seed,
                          device=device, max_traj_len=max_traj_len, env_id_base=env_id_base,
                          action_smoothing=self.action_smoothing, nominal_qpos=self.nominal_pose,
                          action_offset=self.action_offset(), task_params=[self.goal_height],
                          task_iparams=self.body_ids(), clock_lut=self.clock_lut(), history_len=self.history_len,
                          init_noise=np.deg2rad(self.init_noise_deg), perturbation=self.perturbation_config())

    def perturbation_config(self):
        """BatchedEnv(perturbation=...) for this YAML (None: off): interval in control steps, packed-model body ids, magnitudes"""
        if self.perturb_interval <= 0:
            return None
        m = self.model()
        return dict(interval=self.perturb_interval, bodies=[m.body_id(b) for b in self.perturb_bodies], force=self.force_magnitude, torque=self.torque_magnitude)

    def algorithmic_bytes_per_env_step(self) -> int:
        """Persistent state record read + written once per control step (168 f64 words) plus
        action in, obs / terminal obs / reward / flags / 10 reward terms out (~2.2 KB)."""
        return 2 * 168 * 8 + 12 * 4 + 2 * 37 * 4 + 4 + 1 + 10 * 4

    def algorithmic_flops_per_env_step(self) -> int:
        """~75 kFLOP per sim sub-step x 25."""
        return 75_000 * self.frame_skip
