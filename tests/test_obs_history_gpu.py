"""obs_history_len = 3 on the GPU path: a BatchedEnv with history against a twin without, same seed and actions -- the full
observation must be the reference's deque (newest first, zero-filled after an auto-reset; tests/test_obs_history.py pins that
rule to the executed reference), through step(), through step_range() on two env groups, and through a PPO iteration."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spec_h3(tmp_path):
    from learninghumanoidwalking_amd.envs.jvrc_walk import JVRC_BASE_YAML, JvrcWalkSpec
    y = tmp_path / "h3.yaml"
    y.write_text(open(JVRC_BASE_YAML).read().replace("obs_history_len: 1", "obs_history_len: 3"))
    return JvrcWalkSpec(yaml_path=str(y)), JvrcWalkSpec()


def test_history_observation_is_the_deque_of_base_observations(tmp_path):
    import torch
    s3, s1 = _spec_h3(tmp_path)
    N, T, B = 6, 150, 37
    e3, e1 = s3.make_batched(N, seed=2, device=0, max_traj_len=60), s1.make_batched(N, seed=2, device=0, max_traj_len=60)
    assert e3.obs_dim == 111 and e1.obs_dim == 37
    o3, o1 = e3.reset().clone(), e1.reset().clone()
    assert torch.equal(o3[:, :B], o1) and (o3[:, B:] == 0).all()
    rs = np.random.default_rng(0)
    hist = o3.clone()
    ends = 0
    obs3 = torch.zeros(N, 111, device="cuda"); tob3 = torch.zeros(N, 111, device="cuda"); rew = torch.zeros(N, device="cuda"); done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    for t in range(T):
        a = torch.from_numpy((rs.normal(size=(N, 12)) * 0.3).astype(np.float32)).cuda()
        b1, r1, d1, t1 = e1.step(a)
        if t % 2 == 0:
            b3, r3, d3, t3 = e3.step(a)
        else:      # the rollout's path: two env groups, full-batch buffers
            e3.step_range(0, 3, a, obs3, tob3, rew, done); e3.step_range(3, 3, a, obs3, tob3, rew, done)
            b3, r3, d3, t3 = obs3, rew, done, tob3
        assert torch.equal(d3, d1) and torch.equal(r3, r1)
        keep = (d1 == 0).float().unsqueeze(1)
        want = torch.cat([b1, hist[:, :-B] * keep], dim=1)
        want_term = torch.cat([t1, hist[:, :-B]], dim=1)
        assert torch.equal(b3, want), f"step {t}"
        assert torch.equal(t3[d1 != 0], want_term[d1 != 0])
        hist = want.clone()
        ends += int(d1.sum())
    assert ends >= N        # truncations at 60 steps and falls: every env went through at least one auto-reset


def test_ppo_iteration_runs_with_history(tmp_path):
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.ppo import PPO
    import functools
    s3, _ = _spec_h3(tmp_path)
    args = SimpleNamespace(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=2, num_procs=64,
                           max_grad_norm=0.05, max_traj_len=16, use_gae=True, mirror_coeff=0.4, eval_freq=100, continued=None, recurrent=False,
                           imitate=None, imitate_coeff=0.3, logdir=str(tmp_path / "log"), std_dev=0.223, learn_std=False, n_itr=1, seed=0,
                           no_mirror=True, num_envs=64)
    algo = PPO(functools.partial(type(s3), yaml_path=s3.yaml_path), args, seed=0)
    algo.train(functools.partial(type(s3), yaml_path=s3.yaml_path), n_itr=1)
    assert algo.kernels.obs_dim == 111 if hasattr(algo.kernels, "obs_dim") else True
