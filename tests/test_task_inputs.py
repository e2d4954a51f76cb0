"""The batched sim-facade seam (include/lhw.h: lhw_env_enable_task_inputs / lhw_env_get_task_inputs): the kernel exports what
the reference's tasks read through RobotInterface for the control step it just rewarded; recomputing the ten walking reward
terms OUTSIDE the kernel from that export -- with the reference's own tasks/rewards.py, loaded by file path, when /root/reference
is present (build container), else with the restatement pinned to it (oracle/env_jvrc_walk.py, tests/test_specs.py) -- must give
the kernel's fused `rew_terms`.  This is the slow-path BaseTask hook of INTEGRATION.md, exercised: it ties the fused reward to
reference code directly.  Runs on the SIMT emulator here; tests/test_task_inputs_gpu.py is the GPU twin."""
import importlib.util
import os

import numpy as np

REF_REWARDS = "/root/reference/tasks/rewards.py"


def reward_functions():
    if os.path.exists(REF_REWARDS):
        spec = importlib.util.spec_from_file_location("ref_rewards_seam", REF_REWARDS)
        rw = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rw)
        return rw, True
    from oracle import env_jvrc_walk as o

    class RW:      # same call signatures as the reference module
        calc_fwd_vel_reward = staticmethod(o.r_fwd_vel)
        calc_yaw_vel_reward = staticmethod(o.r_yaw_vel)
        calc_action_reward = staticmethod(o.r_action)
        calc_torque_reward = staticmethod(o.r_torque)
        calc_height_reward = staticmethod(o.r_height)
        calc_root_accel_reward = staticmethod(o.r_root_accel)

        @staticmethod
        def calc_foot_frc_clock_reward(l, r, phase, lf, rf, mass):
            return o.r_clock(l, r, lf(phase), rf(phase), mass * 9.8 * 0.5)

        @staticmethod
        def calc_foot_vel_clock_reward(lv, rv, phase, lf, rf):
            return o.r_clock(np.linalg.norm(lv), np.linalg.norm(rv), lf(phase), rf(phase), 0.2)
    return RW, False


def walking_terms(rw, ti, i, spec, lut, mass):
    """WalkingTask.calc_reward (reference tasks/walking_task.py:85-147) on the exported inputs of env i."""
    ph, mode = int(ti["phase"][i]), int(ti["mode"][i])          # kernel modes: 0 STANDING, 1 INPLACE, 2 FORWARD
    r_frc, r_vel, l_frc, l_vel = (lambda p, k=k: lut[k, int(p)] for k in range(4))
    if mode == 0:
        r_frc = l_frc = lambda _: 1
        r_vel = l_vel = lambda _: -1
    yaw_ref, vx, vy = ti["mode_ref"][i]
    if mode == 0:
        yaw_ref, vx, vy = 0.0, 0.0, 0.0
    elif mode == 1:
        vx, vy = 0.0, 0.0
    else:
        yaw_ref = 0.0
    goal = np.array([vx, vy])
    cz = ti["contact_z"][i] if ti["foot_contact"][i] else 0
    return [
        0.225 * rw.calc_foot_frc_clock_reward(ti["grf_l"][i], ti["grf_r"][i], ph, l_frc, r_frc, mass),
        0.225 * rw.calc_foot_vel_clock_reward(ti["lfoot_vel"][i], ti["rfoot_vel"][i], ph, l_vel, r_vel),
        0.050 * rw.calc_root_accel_reward(ti["qvel"][i], ti["qacc"][i]),
        0.050 * rw.calc_height_reward(ti["root_xpos"][i][2], spec.goal_height, float(np.linalg.norm(goal)), cz),
        0.150 * rw.calc_fwd_vel_reward(ti["root_vel_local"][i][:2], goal),
        0.150 * rw.calc_yaw_vel_reward(ti["qvel"][i][5], yaw_ref),
        0.050 * np.exp(-10 * np.linalg.norm(ti["head_xpos"][i][:2] - ti["root_xpos"][i][:2])),
        0.050 * np.exp(-np.linalg.norm(spec.half_sitting_pose - ti["act_pos"][i])),
        0.025 * rw.calc_torque_reward(ti["act_tau"][i], ti["prev_torque"][i]),
        0.025 * rw.calc_action_reward(ti["action"][i], ti["prev_action"][i]),
    ]


def check_env(env, spec, steps, rs):
    rw, is_ref = reward_functions()
    lut, mass = spec.clock_lut(), float(spec.model().body_mass.sum())
    env.reset()
    env.enable_task_inputs(True)
    worst = 0.0
    for t in range(steps):
        act = (rs.normal(size=(env.n_envs, 12)) * 0.3).astype(np.float32)
        if hasattr(env.rew_terms, "cpu"):      # the product env takes device tensors
            import torch
            act = torch.from_numpy(act).cuda()
        env.step(act)
        ti = env.get_task_inputs()
        terms = np.array(env.rew_terms.cpu() if hasattr(env.rew_terms, "cpu") else env.rew_terms, dtype=np.float64)
        for i in range(env.n_envs):
            mine = walking_terms(rw, ti, i, spec, lut, mass)
            np.testing.assert_allclose(terms[i], mine, rtol=0, atol=1e-6, err_msg=f"t={t} env={i} (reference rewards.py: {is_ref})")
            worst = max(worst, float(np.abs(terms[i] - mine).max()))
    return worst, is_ref


def test_fused_walking_reward_equals_reference_rewards_on_exported_task_inputs():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from tests import emu
    spec = JvrcWalkSpec()
    env = emu.make_emulated(spec, 3, seed=5)
    worst, is_ref = check_env(env, spec, 6, np.random.default_rng(2))
    print(f"fused reward terms vs tasks/rewards.py on the exported inputs: max |diff| {worst:.2e} (reference module: {is_ref})")
