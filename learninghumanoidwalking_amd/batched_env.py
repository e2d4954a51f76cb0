"""Batched environments on one MI355X: thin Python host over the C ABI (include/lhw.h).

`BatchedEnv` is what the trainer uses (N envs advanced by one kernel launch per control
step); the N=1 `BaseHumanoidEnv`-shaped adapters live in ``envs/``.  torch is used only to
own device buffers and streams.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .model import Model

TASK_CARTPOLE, TASK_JVRC_WALK, TASK_H1_STAND, TASK_JVRC_STEP, TASK_H1_WALK = 0, 1, 2, 3, 4
DONE_TERMINATED, DONE_TRUNCATED = 1, 2


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def history_update(prev_full: torch.Tensor, base_obs: torch.Tensor, term_base: torch.Tensor, done: torch.Tensor, base: int):
    """Observation history of the reference (envs/common/base_humanoid_env.py:177-197, 274): the observation is a deque of the
    last `history_len` base observations, newest first, flattened; a reset empties it and zero-fills before the first push.
    prev_full [n, H*base]: the full observation before this control step; base_obs: what the kernel returns (for an env whose
    episode ended: the first observation after the auto-reset); term_base: the base observation of the state the step reached;
    done [n] uint8.  Returns (full observation, full terminal observation), both [n, H*base]."""
    tail = prev_full[:, :-base]
    keep = (done == 0).to(prev_full.dtype).unsqueeze(1)
    return torch.cat([base_obs, tail * keep], dim=1), torch.cat([term_base, tail], dim=1)


class BatchedEnv:
    """N copies of one task, state resident in HBM.

    Mirrors, per env, the reference's ``env.reset()`` / ``env.step(a)`` contract (reference
    envs/common/base_humanoid_env.py:199-276) plus the episode bookkeeping of
    ``RolloutWorker.sample`` (reference rl/workers/rollout_worker.py:142-181) when
    ``max_traj_len > 0``.
    """

    def __init__(self, model: Model, task: int, n_envs: int, *, frame_skip: int, kp, kd, seed: int = 0,
                 device: int | torch.device = 0, max_traj_len: int = 0, env_id_base: int = 0,
                 action_smoothing: float = 1.0, nominal_qpos=None, action_offset=None, task_params=None,
                 task_iparams=None, clock_lut=None, history_len: int = 1, init_noise: float = 0.0, perturbation: dict | None = None):
        if not torch.cuda.is_available():
            raise _lib.LhwError(-5, "no GPU visible: BatchedEnv has no CPU fallback")
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self.model = model
        self.n_envs = int(n_envs)
        self.task = task
        self._ib, self._db = model.pack()
        keep = []

        def arr(x, dt):
            if x is None:
                return None, 0
            a = np.ascontiguousarray(x, dtype=dt)
            keep.append(a)
            return a.ctypes.data, a.size

        cfg = _lib.LhwEnvConfig()
        cfg.task, cfg.n_envs, cfg.device = task, self.n_envs, self.device.index or 0
        cfg.frame_skip, cfg.max_traj_len, cfg.env_id_base = int(frame_skip), int(max_traj_len), int(env_id_base)
        self.env_id_base = int(env_id_base)      # global index of env 0: every RNG key of the kernels uses env_id_base + n
        cfg.seed, cfg.action_smoothing = int(seed) & (2**64 - 1), float(action_smoothing)
        cfg.kp, _ = arr(np.atleast_1d(kp), np.float64)
        cfg.kd, _ = arr(np.atleast_1d(kd), np.float64)
        cfg.nominal_qpos, _ = arr(nominal_qpos, np.float64)
        cfg.action_offset, _ = arr(action_offset, np.float64)
        cfg.task_params, cfg.n_task_params = arr(task_params, np.float64)
        cfg.task_iparams, cfg.n_task_iparams = arr(task_iparams, np.int32)
        lut, _ = arr(clock_lut, np.float64)
        cfg.clock_lut = lut
        cfg.period = 0 if clock_lut is None else int(np.asarray(clock_lut).shape[-1])
        cfg.init_noise = float(init_noise)      # radians (base_humanoid_env.py:287: cfg.init_noise degrees * pi / 180)
        if perturbation:                        # JVRC tasks: dict(interval=<control steps>, bodies=[ids], force=, torque=)
            bodies = [int(b) for b in perturbation.get("bodies", [])]
            if len(bodies) > 2:
                raise ValueError("perturbation: at most two bodies")
            cfg.perturb_interval, cfg.n_perturb_bodies = int(perturbation["interval"]), len(bodies)
            for i, b in enumerate(bodies):
                cfg.perturb_bodies[i] = b
            cfg.perturb_force, cfg.perturb_torque = float(perturbation.get("force", 0.0)), float(perturbation.get("torque", 0.0))
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        _lib.check(L.lhw_env_create(self._ib.ctypes.data, self._ib.size, self._db.ctypes.data, self._db.size,
                                    ctypes.byref(cfg), ctypes.byref(self._h)))
        self._L = L
        # obs_history_len > 1 (base_humanoid_env.py:177-197) is kept above the kernels: they emit the base observation, the
        # history rows are shifted on the device by a few torch ops per step (history_update); obs_dim is the full length
        self.base_obs_dim = L.lhw_env_obs_dim(self._h)
        self.history_len = int(history_len)
        if self.history_len < 1:
            raise ValueError("history_len must be >= 1")
        self.obs_dim = self.base_obs_dim * self.history_len
        self.act_dim = L.lhw_env_act_dim(self._h)
        self.n_terms = L.lhw_env_num_reward_terms(self._h)
        self.nq, self.nv = L.lhw_env_nq(self._h), L.lhw_env_nv(self._h)
        N, dev = self.n_envs, self.device
        self.obs = torch.zeros(N, self.obs_dim, dtype=torch.float32, device=dev)
        self.term_obs = torch.zeros(N, self.obs_dim, dtype=torch.float32, device=dev)
        self.rew = torch.zeros(N, dtype=torch.float32, device=dev)
        self.done = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.rew_terms = torch.zeros(N, self.n_terms, dtype=torch.float32, device=dev)
        if self.history_len > 1:
            self._base = torch.zeros(N, self.base_obs_dim, dtype=torch.float32, device=dev)    # kernel outputs
            self._tbase = torch.zeros(N, self.base_obs_dim, dtype=torch.float32, device=dev)
            self._full = torch.zeros(N, self.obs_dim, dtype=torch.float32, device=dev)         # current full observation

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.lhw_env_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask: torch.Tensor | None = None, obs_out: torch.Tensor | None = None) -> torch.Tensor:
        """Reset the envs whose mask byte is non-zero (None: all).  The first observation of the reset envs goes to `obs_out`
        ([N, obs_dim], rows of the other envs untouched) or, by default, to self.obs."""
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_cuda and mask.numel() == self.n_envs
        if obs_out is not None and self.history_len == 1:
            assert obs_out.is_cuda and obs_out.dtype == torch.float32 and obs_out.is_contiguous() and obs_out.shape == self.obs.shape
            _lib.check(self._L.lhw_env_reset(self._h, _ptr(mask), _ptr(obs_out), _stream_ptr(self.device)))
            return obs_out
        if self.history_len > 1:
            _lib.check(self._L.lhw_env_reset(self._h, _ptr(mask), _ptr(self._base), _stream_ptr(self.device)))
            sel = slice(None) if mask is None else mask.bool()
            self._full[sel] = 0
            self._full[sel, :self.base_obs_dim] = self._base[sel]
            self.obs.copy_(self._full)
            return self.obs
        _lib.check(self._L.lhw_env_reset(self._h, _ptr(mask), _ptr(self.obs), _stream_ptr(self.device)))
        return self.obs

    def _history(self, a, b, obs, tob, done):
        """full observation / terminal observation of envs [a, b) from the kernel's base outputs"""
        full, term = history_update(self._full[a:b], self._base[a:b], self._tbase[a:b], done[a:b], self.base_obs_dim)
        self._full[a:b] = full
        obs[a:b] = full
        tob[a:b] = term

    def step(self, act: torch.Tensor, obs_out: torch.Tensor | None = None, term_obs_out: torch.Tensor | None = None,
             rew_out: torch.Tensor | None = None, done_out: torch.Tensor | None = None):
        """act [N, act_dim] float32 on device.  Returns (obs, rew, done, term_obs) device tensors."""
        assert act.is_cuda and act.dtype == torch.float32 and act.is_contiguous() and act.numel() == self.n_envs * self.act_dim
        obs = self.obs if obs_out is None else obs_out
        tob = self.term_obs if term_obs_out is None else term_obs_out
        rew = self.rew if rew_out is None else rew_out
        done = self.done if done_out is None else done_out
        if self.history_len > 1:
            _lib.check(self._L.lhw_env_step(self._h, _ptr(act), _ptr(self._base), _ptr(self._tbase), _ptr(rew), _ptr(done),
                                            _ptr(self.rew_terms), _stream_ptr(self.device)))
            self._history(0, self.n_envs, obs, tob, done)
            return obs, rew, done, tob
        _lib.check(self._L.lhw_env_step(self._h, _ptr(act), _ptr(obs), _ptr(tob), _ptr(rew), _ptr(done),
                                        _ptr(self.rew_terms), _stream_ptr(self.device)))
        return obs, rew, done, tob

    def step_range(self, first: int, count: int, act: torch.Tensor, obs: torch.Tensor, term_obs: torch.Tensor, rew: torch.Tensor,
                   done: torch.Tensor):
        """Advance envs [first, first + count) only, on the current stream.  All tensors are the FULL-batch buffers ([N, ...]);
        independent groups issued on different streams overlap on the GPU (no batch-wide barrier per control step)."""
        assert act.is_cuda and act.dtype == torch.float32 and act.is_contiguous() and act.numel() == self.n_envs * self.act_dim
        if self.history_len > 1:
            _lib.check(self._L.lhw_env_step_range(self._h, int(first), int(count), _ptr(act), _ptr(self._base), _ptr(self._tbase), _ptr(rew),
                                                  _ptr(done), _ptr(self.rew_terms), _stream_ptr(self.device)))
            self._history(int(first), int(first) + int(count), obs, term_obs, done)
            return
        _lib.check(self._L.lhw_env_step_range(self._h, int(first), int(count), _ptr(act), _ptr(obs), _ptr(term_obs), _ptr(rew),
                                              _ptr(done), _ptr(self.rew_terms), _stream_ptr(self.device)))

    def rollout(self, policy, T: int, obs: torch.Tensor, act: torch.Tensor, logp: torch.Tensor, term_obs: torch.Tensor, rew: torch.Tensor,
                done: torch.Tensor, first: int = 0, count: int | None = None, task_inputs: torch.Tensor | None = None) -> bool:
        """The resident rollout (lhw_env_rollout): T control steps of envs [first, first + count) in ONE launch on the current
        stream, the actor (`policy`: PpoKernels.rollout_policy()) evaluated inside the stepper's wavefronts -- the body of
        RolloutWorker.sample's loop (reference rl/workers/rollout_worker.py:142-181) with no wavefront waiting for another env.
        Buffers are time-major over the full batch: obs [T + 1, N, D] (slice 0 in), act [T, N, A], logp / rew / done [T, N],
        term_obs [T, N, D].  `task_inputs` [T, N, TASK_INPUT_DIM] float64 (optional): the sim-facade record of EVERY control step
        (lhw_env_rollout_task_inputs), for reward-only task plug-ins.  Returns False (nothing launched) where the library has no
        resident kernel for this env / policy; a HIP failure raises."""
        N = self.n_envs
        if self.history_len > 1 or policy is None or not hasattr(self._L, "lhw_env_rollout"):
            return False
        assert obs.shape == (T + 1, N, self.obs_dim) and act.shape == (T, N, self.act_dim) and term_obs.shape == (T, N, self.obs_dim)
        assert logp.shape == (T, N) and rew.shape == (T, N) and done.shape == (T, N) and done.dtype == torch.uint8
        for x in (obs, act, logp, term_obs, rew, done):
            assert x.is_cuda and x.is_contiguous()
        args = (self._h, ctypes.byref(policy), int(first), int(N - first if count is None else count), int(T), _ptr(obs),
                _ptr(act), _ptr(logp), _ptr(term_obs), _ptr(rew), _ptr(done), _ptr(self.rew_terms))
        if task_inputs is not None:
            assert task_inputs.shape == (T, N, _lib.TASK_INPUT_DIM) and task_inputs.dtype == torch.float64 and task_inputs.is_cuda and task_inputs.is_contiguous()
            rc = self._L.lhw_env_rollout_task_inputs(*args, _ptr(task_inputs), _stream_ptr(self.device))
        else:
            rc = self._L.lhw_env_rollout(*args, _stream_ptr(self.device))
        if rc == -4:      # LHW_ERR_UNSUPPORTED: the caller keeps the launch-per-step pipeline (anything else -- LHW_ERR_HIP ... -- raises)
            return False
        _lib.check(rc)
        return True

    def last_rollout_queued(self) -> bool:
        """the most recent resident rollout drained the job queue (stepping task with more envs than wave slots)"""
        return bool(getattr(self._L, "lhw_env_last_rollout_queued", lambda h: 0)(self._h) == 1)

    def get_state(self):
        qpos = np.zeros((self.n_envs, self.nq))
        qvel = np.zeros((self.n_envs, self.nv))
        _lib.check(self._L.lhw_env_get_state(self._h, qpos.ctypes.data, qvel.ctypes.data))
        return qpos, qvel

    def set_state(self, qpos, qvel):
        qpos = np.ascontiguousarray(qpos, dtype=np.float64).reshape(self.n_envs, self.nq)
        qvel = np.ascontiguousarray(qvel, dtype=np.float64).reshape(self.n_envs, self.nv)
        _lib.check(self._L.lhw_env_set_state(self._h, qpos.ctypes.data, qvel.ctypes.data))

    def pop_episode_stats(self):
        r, l, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._L.lhw_env_pop_episode_stats(self._h, ctypes.byref(r), ctypes.byref(l), ctypes.byref(c)))
        return r.value, l.value, c.value

    def phase_cycles(self, enable=True):
        """Per-phase shader-clock cycles of env 0 since the last call (diagnostic, wave-per-env stepper only)."""
        out = np.zeros(16, dtype=np.int64)
        _lib.check(self._L.lhw_env_phase_cycles(self._h, int(enable), out.ctypes.data))
        return out

    def pop_fault_stats(self):
        """(contact-overflow steps, diverged-env steps) since the last call."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.lhw_env_pop_fault_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def get_actuator_state(self):
        """(positions, velocities, torques) [N, nu] of the actuated joints as of the last forward pass
        (reference robot_interface.py:163-185)."""
        nu = self.act_dim
        out = [np.zeros((self.n_envs, nu)) for _ in range(3)]
        _lib.check(self._L.lhw_env_get_actuator_state(self._h, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return tuple(out)

    def enable_task_inputs(self, enable: bool = True):
        """Arm / disarm the batched sim facade (include/lhw.h: LhwTaskInput): from the next control step on the kernel exports,
        per env, what the reference's tasks read through RobotInterface and what RobotBase.step passes to calc_reward."""
        _lib.check(self._L.lhw_env_enable_task_inputs(self._h, int(bool(enable))))

    def get_task_inputs(self) -> dict:
        """Named float64 arrays of the last control step's task inputs (host copy, synchronous)."""
        rec = np.zeros((self.n_envs, _lib.TASK_INPUT_DIM))
        _lib.check(self._L.lhw_env_get_task_inputs(self._h, rec.ctypes.data))
        return _lib.split_task_inputs(rec, self.nq, self.nv, self.act_dim)

    def wave_cycles(self):
        """Per-env shader-clock cycles of the last control-step launch (diagnostic; the first call only arms the recording)."""
        out = np.zeros(self.n_envs, dtype=np.int64)
        _lib.check(self._L.lhw_env_debug_wave_cycles(self._h, out.ctypes.data))
        return out

    def pop_rerun_count(self):
        """Control steps since the last call that needed the one-env-per-wave re-run (more than 8 contacts)."""
        a = ctypes.c_int64()
        _lib.check(self._L.lhw_env_pop_rerun_count(self._h, ctypes.byref(a)))
        return a.value

    def debug_step_record(self):
        """Stepping task test hook: (sequence [N,20,6] = x y z theta cos sin, floor_z [N], istate [N,5] = t1 t2 reached frames nseq)."""
        seq = np.zeros((self.n_envs, 20, 6))
        fz = np.zeros(self.n_envs)
        ist = np.zeros((self.n_envs, 5), np.int32)
        _lib.check(self._L.lhw_env_debug_step_record(self._h, seq.ctypes.data, fz.ctypes.data, ist.ctypes.data))
        return seq, fz, ist

    def set_iteration(self, it: int):
        _lib.check(self._L.lhw_env_set_iteration(self._h, int(it)))
