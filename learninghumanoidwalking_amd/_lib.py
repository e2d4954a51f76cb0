"""ctypes loader for liblhw.so (the C ABI in include/lhw.h).  No CPU fallback: if the
library is missing or no GPU is visible, calls fail loudly."""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("LHW_LIB") or os.path.join(_HERE, "liblhw.so")   # LHW_LIB: kernel-variant experiments (scripts/)
_LIB = None


class LhwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"liblhw error {code}: {msg}")
        self.code = code


class LhwEnvConfig(ctypes.Structure):
    _fields_ = [
        ("task", ctypes.c_int32), ("n_envs", ctypes.c_int32), ("device", ctypes.c_int32),
        ("frame_skip", ctypes.c_int32), ("max_traj_len", ctypes.c_int32), ("env_id_base", ctypes.c_int32),
        ("seed", ctypes.c_uint64), ("action_smoothing", ctypes.c_double),
        ("kp", ctypes.c_void_p), ("kd", ctypes.c_void_p), ("nominal_qpos", ctypes.c_void_p),
        ("action_offset", ctypes.c_void_p), ("task_params", ctypes.c_void_p), ("n_task_params", ctypes.c_int32),
        ("task_iparams", ctypes.c_void_p), ("n_task_iparams", ctypes.c_int32),
        ("clock_lut", ctypes.c_void_p), ("period", ctypes.c_int32), ("init_noise", ctypes.c_double),
        ("perturb_interval", ctypes.c_int32), ("n_perturb_bodies", ctypes.c_int32), ("perturb_bodies", ctypes.c_int32 * 2),
        ("perturb_force", ctypes.c_double), ("perturb_torque", ctypes.c_double),
    ]


class LhwRolloutPolicy(ctypes.Structure):      # include/lhw.h: the frozen actor as lhw_env_rollout reads it
    _fields_ = [(n, ctypes.c_void_p) for n in ("w1t", "b1", "w2t", "b2", "w3t", "b3", "stdv", "obs_mean", "obs_std")] + [
        (n, ctypes.c_int32) for n in ("obs_dim", "obs_pad", "act_dim", "act_pad", "hidden", "deterministic", "fp16_operands")] + [
        ("seed", ctypes.c_uint64), ("counter", ctypes.c_uint32)]


# include/lhw.h: enum LhwTaskInput (offsets into one env's record, length)
TASK_INPUT_DIM = 176
TASK_INPUT_FIELDS = dict(grf_r=(0, 1), grf_l=(1, 1), contact_z=(2, 1), foot_contact=(3, 1), self_collision=(4, 1), phase=(5, 1), mode=(6, 1),
                         mode_ref=(7, 3), rfoot_vel=(10, 3), lfoot_vel=(13, 3), root_vel_local=(16, 3), root_xpos=(19, 3), head_xpos=(22, 3),
                         rfoot_xpos=(25, 3), lfoot_xpos=(28, 3), qpos=(32, 19), qvel=(51, 18), qacc=(69, 18), act_pos=(87, 12), act_vel=(99, 12),
                         act_tau=(111, 12), prev_torque=(123, 12), prev_action=(135, 12), action=(147, 12), root_xmat=(160, 9))


def split_task_inputs(rec, nq, nv, nu):
    """[N][TASK_INPUT_DIM] records -> dict of named arrays (vectors cut to the model's nq / nv / nu)."""
    cut = dict(qpos=nq, qvel=nv, qacc=nv, act_pos=nu, act_vel=nu, act_tau=nu, prev_torque=nu, prev_action=nu, action=nu)
    out = {}
    for k, (o, n) in TASK_INPUT_FIELDS.items():
        n = cut.get(k, n)
        out[k] = rec[:, o] if n == 1 else rec[:, o:o + n]
    return out


def sources():
    return sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")))


# Per-source compile flags of the stepper's translation units.
# -disable-machine-licm: LLVM's machine LICM hoists the materialisation of 64-bit constants (1.0, polynomial coefficients) and of
# lane-derived addresses out of the 25-sub-step loop, the register allocator then parks them in scratch and every use becomes a
# scratch reload (one slot holding the constant 1.0 was reloaded 42 times in the round-3 ISA).  Without it: scratch 480 -> 352 B
# per lane, SGPR spills 438 -> 206, VGPR spills 154 -> 100, control step -5 %.
# -ffp-contract=on (round 5): hipcc's default, fast, lets the BACKEND fuse a multiply with an add wherever the two end up in one
# basic block with the right use counts -- a decision that depends on the code around them, so the same control_step source
# compiled into two kernels (the launch-per-step kernel and the resident rollout kernel) rounded a handful of box-contact
# expressions differently and the two rollouts differed by 1e-15 in the state after a reset.  With `on` the FRONT END fuses, within a
# source expression only: the same source gives the same arithmetic in every kernel (measured: same speed, both kernels bitwise
# equal on all four humanoid tasks).  The GEMM / strip / cartpole kernels keep the default pipeline.
_STEPPER_FLAGS = ["-mllvm", "-disable-machine-licm", "-ffp-contract=on"]
# -amdgpu-sched-strategy=iterative-ilp (round 6, the stepping task's rollout kernels only -- a translation unit of their own): one env per
# wave leaves a third of the lanes idle and the instruction stream latency-bound; LLVM's iterative ILP scheduler makes jvrc_step's rollout
# 3 % faster than the default strategy (max-ilp: 1 %, iterative-maxocc: 2 %, iterative-minreg: 4 % slower).  The two-envs-per-wave kernels
# lose 2.6 % under max-ilp, and this ROCm's clang crashes on their translation unit under iterative-ilp, so they keep the default;
# with machine LICM back on the stepping kernels lose 50 % (profiles/r06_stepper_compiler_flags.txt).
# -amdgpu-sched-strategy=iterative-maxocc (round 6, the other stepper translation units): LLVM's iterative scheduler aiming at the kernel's
# occupancy target -- the same 254 VGPRs and no spills, jvrc_walk's rollout 1.8 % faster than under the default strategy.
_MAXOCC = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]
EXTRA_FLAGS = {"lhw_humanoid.hip": _STEPPER_FLAGS + _MAXOCC, "lhw_humanoid_rollout.hip": _STEPPER_FLAGS + _MAXOCC,
               "lhw_humanoid_rollout_step.hip": _STEPPER_FLAGS + ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}


def _stale(deps) -> bool:
    return not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile liblhw.so for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per source, in parallel, then link.
    Safe under concurrent callers (the ranks of a torch.distributed.run job that all find the library stale): an exclusive file lock
    serialises them and the staleness test is repeated once the lock is held; objects and the library are written under per-process
    temporary names and renamed into place, so nobody can dlopen a half-written file."""
    import fcntl
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_ROOT, "include", "*.h")) + [os.path.abspath(__file__)]
    if not force and not _stale(deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale(deps):      # another process built it while this one waited
            return LIB_PATH
        base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
        tag = f".{os.getpid()}.tmp"
        procs, objs, tmps = [], [], []
        try:
            for s in srcs:
                o = os.path.join(objdir, os.path.basename(s) + ".o")
                objs.append(o)
                tmps.append(o + tag)
                cmd = base + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o + tag]
                if verbose:
                    print(" ".join(cmd))
                procs.append((cmd, subprocess.Popen(cmd)))
            failed = None
            for cmd, p in procs:
                if p.wait() != 0 and failed is None:
                    failed = (p.returncode, cmd)
                    for _, q in procs:          # one source failed: stop the others instead of leaving them running
                        if q.poll() is None:
                            q.terminate()
            if failed:
                raise subprocess.CalledProcessError(*failed)
            for o in objs:
                os.replace(o + tag, o)
            cmd = base + ["-shared", "-o", LIB_PATH + tag] + objs
            tmps.append(LIB_PATH + tag)
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(LIB_PATH + tag, LIB_PATH)
        finally:
            for _, q in procs:
                if q.poll() is None:
                    q.kill()
                q.wait()
            for t in tmps:
                if os.path.exists(t):
                    os.remove(t)
    return LIB_PATH


def declare(L):
    """Attach the argument types of include/lhw.h's entry points to a loaded library (entry points the library
    does not export are skipped: the emulated test build has no PPO kernels)."""
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64

    def sig(name, argtypes=None, restype=None):
        f = getattr(L, name, None)
        if f is None:
            return
        if argtypes is not None:
            f.argtypes = argtypes
        if restype is not None:
            f.restype = restype

    sig("lhw_version", restype=ctypes.c_int)
    sig("lhw_last_error", restype=ctypes.c_char_p)
    sig("lhw_env_create", [vp, i64, vp, i64, ctypes.POINTER(LhwEnvConfig), ctypes.POINTER(vp)])
    sig("lhw_env_destroy", [vp])
    for f in ("lhw_env_obs_dim", "lhw_env_act_dim", "lhw_env_num_reward_terms", "lhw_env_nq", "lhw_env_nv"):
        sig(f, [vp])
    sig("lhw_env_reset", [vp, vp, vp, vp])
    sig("lhw_env_step", [vp, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_env_get_state", [vp, vp, vp])
    sig("lhw_env_set_state", [vp, vp, vp])
    sig("lhw_env_pop_episode_stats", [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)])
    sig("lhw_env_set_iteration", [vp, i64])
    sig("lhw_env_pop_fault_stats", [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)])
    sig("lhw_env_pop_rerun_count", [vp, ctypes.POINTER(i64)])
    sig("lhw_env_get_actuator_state", [vp, vp, vp, vp])
    sig("lhw_env_enable_task_inputs", [vp, i32])
    sig("lhw_env_get_task_inputs", [vp, vp])
    sig("lhw_env_task_inputs_device", [vp, ctypes.POINTER(vp)])
    sig("lhw_env_debug_wave_cycles", [vp, vp])
    sig("lhw_debug_gemm", [i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp])
    sig("lhw_debug_mlp_strip_forward", [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp])
    sig("lhw_debug_mlp_strip_backward", [i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp])
    sig("lhw_debug_mlp_strip_forward_bits", [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_debug_mlp_strip_backward_bits", [i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_debug_wgrad_wide", [vp, vp, i32, i32, vp, vp, vp])
    sig("lhw_debug_wgrad_skinny", [i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp])
    sig("lhw_env_phase_cycles", [vp, ctypes.c_int, vp])
    sig("lhw_env_step_range", [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_ppo_set_imitation", [vp, vp, vp, ctypes.c_float, i64])
    sig("lhw_ppo_step", [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i64, ctypes.c_float, vp])
    sig("lhw_env_debug_step_record", [vp, vp, vp, vp])
    sig("lhw_env_rollout", [vp, ctypes.POINTER(LhwRolloutPolicy), i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_env_last_rollout_queued", [vp])
    sig("lhw_env_rollout_task_inputs", [vp, ctypes.POINTER(LhwRolloutPolicy), i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp])
    sig("lhw_ppo_rollout_policy", [vp, vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(LhwRolloutPolicy)])
    sig("lhw_debug_policy_step", [ctypes.POINTER(LhwRolloutPolicy), vp, i32, ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp, vp])
    return L


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LhwError(-5, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the stepper / PPO kernels)")
    L = declare(ctypes.CDLL(LIB_PATH))
    _LIB = L
    return L


_POISON = os.environ.get("LHW_POISON") == "1"


def empty(*shape, dtype, device):
    """torch.empty for the buffers the kernels fill -- NaN-filled under LHW_POISON=1 (debug: together with the library's 0xFF-filled
    allocations and a -DLHW_POISON build, a kernel that reads what nobody wrote shows up as a NaN instead of as a result that
    depends on what the allocator handed out)."""
    import torch
    t = torch.empty(*shape, dtype=dtype, device=device)
    if _POISON:
        t.fill_(float("nan"))
    return t


def check(rc: int):
    if rc != 0:
        raise LhwError(rc, lib().lhw_last_error().decode(errors="replace"))
