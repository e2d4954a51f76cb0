"""TEST INFRASTRUCTURE: host-side SIMT emulation of the HIP stepper kernels.

`build()` compiles the product's HIP sources (csrc/lhw_humanoid.hip, lhw_cartpole.hip, lhw_api.hip) with g++
against the emulator header in this directory into tests/emu/_build/liblhw_emu.so; `EmuBatchedEnv` drives that
library through the same C ABI (include/lhw.h) with numpy buffers.  It exists so that the `-m "not gpu"` suite can
check the *kernel source* against the CPU oracle (lane mappings, cross-lane reductions, LDS hand-offs, sub-wave
groups) before a GPU is involved.  Nothing under learninghumanoidwalking_amd/ imports this package.  Of the PPO
kernels only the LDS-resident MLP strip kernels (lhw_mlp_strip.hip, f32 MFMA emulated as its fmaf chain) are built here.
"""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "learninghumanoidwalking_amd", "csrc")
_BUILD = os.path.join(_HERE, "_build")
LIB_PATH = os.path.join(_BUILD, "liblhw_emu.so")
SOURCES = ["lhw_humanoid.hip", "lhw_humanoid_rollout.hip", "lhw_humanoid_rollout_step.hip", "lhw_cartpole.hip", "lhw_api.hip", "lhw_mlp_strip.hip"]
_LIB = None


def build(force: bool = False, opt: str = "-O1") -> str:
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    deps = srcs + glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_ROOT, "include", "*.h")) + [
        os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "emu_runtime.cpp")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    os.makedirs(_BUILD, exist_ok=True)
    objs = []
    procs = []
    for s in srcs + [os.path.join(_HERE, "emu_runtime.cpp")]:
        o = os.path.join(_BUILD, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-x", "c++", "-std=c++17", opt, "-g0", "-fPIC", "-fno-strict-aliasing", "-Wno-unused-value"] + os.environ.get("LHW_EMU_DEFS", "").split() + ["-DLHW_EMU_BUILD", "-I", _HERE,
               "-I", os.path.join(_ROOT, "include"), "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emulator build failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", LIB_PATH] + objs)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        from learninghumanoidwalking_amd import _lib as product
        L = ctypes.CDLL(build())
        product.declare(L)
        _LIB = L
    return _LIB


class EmuBatchedEnv:
    """numpy twin of learninghumanoidwalking_amd.batched_env.BatchedEnv on the emulated library."""

    def __init__(self, model, task, n_envs, *, frame_skip, kp, kd, seed=0, max_traj_len=0, env_id_base=0,
                 action_smoothing=1.0, nominal_qpos=None, action_offset=None, task_params=None, task_iparams=None,
                 clock_lut=None, device=0, history_len=1, init_noise=0.0, perturbation=None):
        from learninghumanoidwalking_amd import _lib as product
        if int(history_len) != 1:
            raise NotImplementedError("the emulated env returns base observations (the history is kept by BatchedEnv, above the kernels)")
        self.n_envs, self.task, self.model = int(n_envs), task, model
        self._ib, self._db = model.pack()
        self._keep = []

        def arr(x, dt):
            if x is None:
                return None, 0
            a = np.ascontiguousarray(x, dtype=dt)
            self._keep.append(a)
            return a.ctypes.data, a.size

        cfg = product.LhwEnvConfig()
        cfg.task, cfg.n_envs, cfg.device = task, self.n_envs, 0
        cfg.frame_skip, cfg.max_traj_len, cfg.env_id_base = int(frame_skip), int(max_traj_len), int(env_id_base)
        cfg.seed, cfg.action_smoothing = int(seed) & (2**64 - 1), float(action_smoothing)
        cfg.kp, _ = arr(np.atleast_1d(kp), np.float64)
        cfg.kd, _ = arr(np.atleast_1d(kd), np.float64)
        cfg.nominal_qpos, _ = arr(nominal_qpos, np.float64)
        cfg.action_offset, _ = arr(action_offset, np.float64)
        cfg.task_params, cfg.n_task_params = arr(task_params, np.float64)
        cfg.task_iparams, cfg.n_task_iparams = arr(task_iparams, np.int32)
        cfg.clock_lut, _ = arr(clock_lut, np.float64)
        cfg.period = 0 if clock_lut is None else int(np.asarray(clock_lut).shape[-1])
        cfg.init_noise = float(init_noise)
        if perturbation:
            bodies = [int(b) for b in perturbation.get("bodies", [])]
            cfg.perturb_interval, cfg.n_perturb_bodies = int(perturbation["interval"]), len(bodies)
            for i, b in enumerate(bodies):
                cfg.perturb_bodies[i] = b
            cfg.perturb_force, cfg.perturb_torque = float(perturbation.get("force", 0.0)), float(perturbation.get("torque", 0.0))
        self._L = lib()
        self._h = ctypes.c_void_p()
        self._check(self._L.lhw_env_create(self._ib.ctypes.data, self._ib.size, self._db.ctypes.data, self._db.size,
                                           ctypes.byref(cfg), ctypes.byref(self._h)))
        L = self._L
        self.obs_dim, self.act_dim = L.lhw_env_obs_dim(self._h), L.lhw_env_act_dim(self._h)
        self.n_terms = L.lhw_env_num_reward_terms(self._h)
        self.nq, self.nv = L.lhw_env_nq(self._h), L.lhw_env_nv(self._h)
        N = self.n_envs
        self.obs = np.zeros((N, self.obs_dim), np.float32)
        self.term_obs = np.zeros((N, self.obs_dim), np.float32)
        self.rew = np.zeros(N, np.float32)
        self.done = np.zeros(N, np.uint8)
        self.rew_terms = np.zeros((N, self.n_terms), np.float32)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"liblhw_emu error {rc}: {self._L.lhw_last_error().decode(errors='replace')}")

    def close(self):
        if self._h.value:
            self._L.lhw_env_destroy(self._h)
            self._h = ctypes.c_void_p()

    def reset(self, mask=None):
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            mp = mask.ctypes.data
        self._check(self._L.lhw_env_reset(self._h, mp, self.obs.ctypes.data, None))
        return self.obs

    def step(self, act):
        act = np.ascontiguousarray(act, np.float32).reshape(self.n_envs, self.act_dim)
        self._check(self._L.lhw_env_step(self._h, act.ctypes.data, self.obs.ctypes.data, self.term_obs.ctypes.data,
                                         self.rew.ctypes.data, self.done.ctypes.data, self.rew_terms.ctypes.data, None))
        return self.obs, self.rew, self.done, self.term_obs

    def rollout(self, policy, T, obs, act, logp, tob, rew, done, first=0, count=None, task_inputs=None):
        """lhw_env_rollout on numpy buffers (time-major over the full batch; obs[0] is the input); task_inputs [T][N][TASK_INPUT_DIM]
        float64: lhw_env_rollout_task_inputs (the sim-facade record of every control step)."""
        N = self.n_envs
        assert obs.shape == (T + 1, N, self.obs_dim) and act.shape == (T, N, self.act_dim) and tob.shape == (T, N, self.obs_dim)
        assert all(a.flags.c_contiguous for a in (obs, act, logp, tob, rew, done))
        args = (self._h, ctypes.byref(policy), int(first), int(N - first if count is None else count), int(T),
                obs.ctypes.data, act.ctypes.data, logp.ctypes.data, tob.ctypes.data, rew.ctypes.data, done.ctypes.data, self.rew_terms.ctypes.data)
        if task_inputs is not None:
            assert task_inputs.dtype == np.float64 and task_inputs.shape[:2] == (T, N) and task_inputs.flags.c_contiguous
            self._check(self._L.lhw_env_rollout_task_inputs(*args, task_inputs.ctypes.data, None))
        else:
            self._check(self._L.lhw_env_rollout(*args, None))

    def get_state(self):
        q, v = np.zeros((self.n_envs, self.nq)), np.zeros((self.n_envs, self.nv))
        self._check(self._L.lhw_env_get_state(self._h, q.ctypes.data, v.ctypes.data))
        return q, v

    def set_state(self, qpos, qvel):
        q = np.ascontiguousarray(qpos, np.float64).reshape(self.n_envs, self.nq)
        v = np.ascontiguousarray(qvel, np.float64).reshape(self.n_envs, self.nv)
        self._check(self._L.lhw_env_set_state(self._h, q.ctypes.data, v.ctypes.data))

    def pop_fault_stats(self):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._L.lhw_env_pop_fault_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def enable_task_inputs(self, enable=True):
        self._check(self._L.lhw_env_enable_task_inputs(self._h, int(bool(enable))))

    def get_task_inputs(self):
        from learninghumanoidwalking_amd import _lib as product
        rec = np.zeros((self.n_envs, product.TASK_INPUT_DIM))
        self._check(self._L.lhw_env_get_task_inputs(self._h, rec.ctypes.data))
        return product.split_task_inputs(rec, self.nq, self.nv, self.act_dim)

    def get_actuator_state(self):
        out = [np.zeros((self.n_envs, self.act_dim)) for _ in range(3)]
        self._check(self._L.lhw_env_get_actuator_state(self._h, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return tuple(out)

    def pop_rerun_count(self):
        a = ctypes.c_int64()
        self._check(self._L.lhw_env_pop_rerun_count(self._h, ctypes.byref(a)))
        return a.value

    def pop_episode_stats(self):
        r, l, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        self._check(self._L.lhw_env_pop_episode_stats(self._h, ctypes.byref(r), ctypes.byref(l), ctypes.byref(c)))
        return r.value, l.value, c.value

    def set_iteration(self, it):
        self._check(self._L.lhw_env_set_iteration(self._h, int(it)))

    def debug_step_record(self):
        seq, fz, ist = np.zeros((self.n_envs, 20, 6)), np.zeros(self.n_envs), np.zeros((self.n_envs, 5), np.int32)
        self._check(self._L.lhw_env_debug_step_record(self._h, seq.ctypes.data, fz.ctypes.data, ist.ctypes.data))
        return seq, fz, ist


def make_emulated(spec, n_envs, **kw):
    """Build the emulated twin of `spec.make_batched(...)` by intercepting the BatchedEnv constructor arguments."""
    import learninghumanoidwalking_amd.batched_env as be
    captured = {}

    class _Capture:
        def __init__(self, model, task, n, **k):
            captured.update(model=model, task=task, n=n, k=k)

    import sys
    mod = sys.modules[type(spec).make_batched.__module__]      # the module whose BatchedEnv name make_batched resolves (a subclass
                                                               # defined elsewhere, e.g. in a test, inherits the method)
    orig = mod.BatchedEnv
    mod.BatchedEnv = _Capture
    try:
        spec.make_batched(n_envs, **kw)
    finally:
        mod.BatchedEnv = orig
    k = captured["k"]
    k.pop("device", None)
    return EmuBatchedEnv(captured["model"], captured["task"], captured["n"], **k)


class TorchEmuBatchedEnv(EmuBatchedEnv):
    """EmuBatchedEnv with the torch-tensor surface of BatchedEnv (CPU tensors sharing the numpy buffers), so that the
    `-m gpu` stepper tests can be replayed on the emulator: `LHW_EMU=1 python -m pytest tests/test_jvrc_gpu.py -m gpu`."""

    def __init__(self, model, task, n_envs, **kw):
        import torch
        kw.pop("device", None)
        super().__init__(model, task, n_envs, **kw)
        self._np = dict(obs=self.obs, term_obs=self.term_obs, rew=self.rew, done=self.done, rew_terms=self.rew_terms)
        for k, v in self._np.items():
            setattr(self, k, torch.from_numpy(v))
        self.device = torch.device("cpu")

    def reset(self, mask=None):
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask.numpy() if hasattr(mask, "numpy") else mask, np.uint8)
            mp = mask.ctypes.data
        self._check(self._L.lhw_env_reset(self._h, mp, self._np["obs"].ctypes.data, None))
        return self.obs

    def step(self, act, obs_out=None, term_obs_out=None, rew_out=None, done_out=None):
        a = np.ascontiguousarray(act.numpy() if hasattr(act, "numpy") else act, np.float32).reshape(self.n_envs, self.act_dim)
        n = self._np
        self._check(self._L.lhw_env_step(self._h, a.ctypes.data, n["obs"].ctypes.data, n["term_obs"].ctypes.data, n["rew"].ctypes.data,
                                         n["done"].ctypes.data, n["rew_terms"].ctypes.data, None))
        return self.obs, self.rew, self.done, self.term_obs


def install_as_backend():
    """Route BatchedEnv to the emulator and make `.cuda()` a no-op (debugging aid for replaying GPU tests on a CPU)."""
    import torch
    import learninghumanoidwalking_amd.batched_env as be
    import learninghumanoidwalking_amd.envs as envs_pkg
    import importlib, pkgutil
    be.BatchedEnv = TorchEmuBatchedEnv
    for mi in pkgutil.iter_modules(envs_pkg.__path__):
        mod = importlib.import_module(f"{envs_pkg.__name__}.{mi.name}")
        if hasattr(mod, "BatchedEnv"):
            mod.BatchedEnv = TorchEmuBatchedEnv
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.is_available = lambda: True
