"""The BaseTask seam with a consumer (task_hook.py, Rollout(task=...) / PPO(..., task=...)): task code that is NOT compiled into the
kernels -- the reference's own tasks/rewards.py called env by env, or a torch-vectorised task -- is evaluated on the exported task
inputs after every control step and the policy is trained on ITS reward / termination.  With the reference's reward code plugged
in, training must reproduce the fused path (rewards 1e-6, weights 3e-6); with one weight changed in the plugged task, the rewards
must change.  Reference: robots/robot_base.py:88-96 (task.step / calc_reward / done per control step), tasks/base_task.py:41-70,
tasks/walking_task.py:85-147."""
import importlib.util
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REF_REWARDS = "/root/reference/tasks/rewards.py"


def _args(N, T):
    return SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=N * T // 2, epochs=2,
                           max_traj_len=T, num_procs=N, num_envs=N, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                           recurrent=False, imitate=None, learn_std=False, std_dev=0.3, no_mirror=False, continued=None,
                           logdir="/tmp/lhw_test_hook", device_index=0)


def _rewards_module():
    """the reference's tasks/rewards.py by file path where the reference checkout exists (build container); on the GPU box the
    restatement that tests/test_specs.py pins to that module, behind the same function names"""
    if os.path.exists(REF_REWARDS):
        spec = importlib.util.spec_from_file_location("ref_rewards_hook", REF_REWARDS)
        rw = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rw)
        return rw
    from tests.test_task_inputs import reward_functions
    return reward_functions()[0]


def _train(task, iters=2, N=24, T=10, env="jvrc_walk"):
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    algo = PPO(ENVIRONMENTS[env], _args(N, T), seed=5, task=task)
    rews, dones, stats = [], [], []
    for itr in range(iters):
        algo.iterate(itr)
        rews.append(algo.rollout.rew.clone())
        dones.append(algo.rollout.done.clone())
        stats.append(algo._ep_stats)
    return algo, rews, dones, stats


def test_training_through_the_task_hook_reproduces_the_fused_task(monkeypatch):
    """(the rewards module is the reference's own file where /root/reference exists -- the build container -- and the restatement
    tests/test_specs.py pins to it on the GPU box)"""
    from learninghumanoidwalking_amd.task_hook import PerEnvRewards, VectorWalkingTask
    fused, rf, df, sf = _train(None)
    assert fused.rollout.last_mode in ("resident", "steps")
    slow, rs, ds, ss = _train(lambda spec, dev: PerEnvRewards(_rewards_module(), spec))          # tasks/rewards.py, env by env
    vect, rv, dv, sv = _train(lambda spec, dev: VectorWalkingTask(spec, dev))                   # the same task, torch-vectorised, reward-only
    own, ro, do, so = _train(lambda spec, dev: VectorWalkingTask(spec, dev, height_limits=(0.6, 1.4000001)))   # ... deciding terminations itself
    monkeypatch.setenv("LHW_ROLLOUT_MODE", "steps")
    stp, rp, dp, sp = _train(lambda spec, dev: VectorWalkingTask(spec, dev))                    # reward-only on the launch-per-step pipeline
    monkeypatch.delenv("LHW_ROLLOUT_MODE")
    assert slow.rollout.last_mode == own.rollout.last_mode == stp.rollout.last_mode == "hooked"
    # a reward-only task keeps the resident rollout: one launch, the record of every control step, one evaluation behind it
    assert vect.rollout.last_mode == fused.rollout.last_mode and vect.rollout.reward_only and not own.rollout.reward_only
    for other_r, other_d, other_s, other in ((rs, ds, ss, slow), (rv, dv, sv, vect), (ro, do, so, own), (rp, dp, sp, stp)):
        for a, b in zip(rf, other_r):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-6)
        for a, b in zip(df, other_d):
            assert torch.equal(a, b)
        for a, b in zip(sf, other_s):          # finished-episode statistics: the rollout's own count equals the kernel's
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(other.kernels.theta.cpu().numpy(), fused.kernels.theta.cpu().numpy(), rtol=0, atol=3e-6)
    assert any((d != 0).any() for d in df), "no episode ended: the host-side truncation / reset path was not exercised"


def test_standing_task_as_a_plug_in_reproduces_the_fused_h1_task():
    """VectorStandingTask (tasks/standing_task.py:49-131 on the exported record, incl. the pelvis frame root_xmat) on the H1 env with
    its observation noise, dynamics randomisation and perturbations on: rewards 1e-6, the kernel's own flags, weights 3e-6."""
    from learninghumanoidwalking_amd.task_hook import VectorStandingTask
    fused, rf, df, sf = _train(None, env="h1", N=16, T=12)
    vect, rv, dv, sv = _train(lambda spec, dev: VectorStandingTask(spec, dev), env="h1", N=16, T=12)
    assert vect.rollout.last_mode == fused.rollout.last_mode and vect.rollout.reward_only
    for a, b in zip(rf, rv):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-6)
    for a, b in zip(df, dv):
        assert torch.equal(a, b)
    for a, b in zip(sf, sv):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(vect.kernels.theta.cpu().numpy(), fused.kernels.theta.cpu().numpy(), rtol=0, atol=3e-6)


def test_a_changed_task_changes_what_the_policy_is_trained_on():
    from learninghumanoidwalking_amd.task_hook import VectorTask, VectorWalkingTask
    base, rb, _, _ = _train(lambda spec, dev: VectorWalkingTask(spec, dev), iters=1)
    heavy, rh, _, _ = _train(lambda spec, dev: VectorWalkingTask(spec, dev, weights=dict(posture_error=0.5)), iters=1)
    diff = (rh[0] - rb[0]).cpu().numpy()
    assert (diff > 1e-3).all() and diff.max() <= 0.45 + 1e-6          # 0.45 x a term in (0, 1] more, everywhere
    assert not torch.equal(base.kernels.theta, heavy.kernels.theta)

    class Lazy(VectorTask):                    # a user's task: reward for low joint speeds, episodes end when the root tilts
        def evaluate(self, ti):
            w = ti.qpos[:, 3]
            return torch.exp(-ti.act_vel.abs().sum(1)), w.abs() < 0.9

    lazy, rl, dl, _ = _train(lambda spec, dev: Lazy(), iters=1)
    assert lazy.rollout.last_mode == "hooked" and torch.isfinite(rl[0]).all() and (rl[0] > 0).all()
    assert not torch.equal(rl[0], rb[0])


def test_task_inputs_device_view_is_the_host_copy():
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from learninghumanoidwalking_amd.task_hook import device_task_inputs
    spec = JvrcWalkSpec()
    env = spec.make_batched(6, seed=3, device=0)
    env.reset()
    ti = device_task_inputs(env)
    env.step(torch.randn(6, 12, device="cuda") * 0.2)
    host = env.get_task_inputs()
    for k, v in host.items():
        np.testing.assert_array_equal(getattr(ti, k).cpu().numpy(), v, err_msg=k)
