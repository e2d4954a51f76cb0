"""One full PPO iteration on the GPU (rollout -> GAE -> advantage normalisation -> minibatch update) against the CPU
oracle chain (oracle env + reference-pinned oracle learner) on identical seeds / weights / sampled actions:
BASELINE config 2 "cartpole ... PPO update on-device, parity vs CPU returns", at a size the oracle finishes in seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NAMES = ["w1", "b1", "w2", "b2", "w3", "b3"]


@pytest.mark.parametrize("env_name", ["cartpole", "jvrc_walk", "jvrc_step", "h1", "h1_walk"])
def test_full_iteration_matches_oracle_chain(env_name):
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    from oracle import make_oracle_env, ppo_oracle as po

    N, T = (16, 24) if env_name == "cartpole" else (6, 10) if env_name.startswith("jvrc") else (6, 8)
    mirror = env_name.startswith("jvrc") or env_name == "h1_walk"
    args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=N * T, epochs=1,
                           max_traj_len=T, num_procs=N, num_envs=N, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                           recurrent=False, imitate=None, learn_std=False, std_dev=0.223, no_mirror=not mirror, continued=None,
                           logdir="/tmp/lhw_test_iter", device_index=0)
    algo = PPO(ENVIRONMENTS[env_name], args, seed=4)
    k = algo.kernels
    if algo.obs_rms is not None:   # cartpole: freeze some non-trivial normalisation
        k.set_obs_norm(np.array([0.0, 0.1, -0.1, 0.0, 0.2]), np.array([0.5, 0.7, 0.7, 1.5, 3.0]))
    w0 = k.get_tensors()
    obs_mean, obs_std = k.obs_mean.cpu().numpy(), k.obs_std.cpu().numpy()

    # ---- GPU: rollout + GAE
    batch = algo.sample_parallel_with_workers()
    ro = algo.rollout
    g_obs, g_act = ro.obs[:T].cpu().numpy(), ro.act.cpu().numpy()
    g_rew, g_val, g_done = ro.rew.cpu().numpy(), ro.val.cpu().numpy(), ro.done.cpu().numpy()
    g_vterm, g_vfinal, g_logp = ro.vterm.cpu().numpy(), ro.vfinal.cpu().numpy(), ro.logp.cpu().numpy()
    g_ret = batch.returns.view(T, N).cpu().numpy()

    # ---- oracle chain driven by the actions the device sampled
    env_seed = algo.env_seed
    envs = [make_oracle_env(env_name, seed=env_seed, env_id=i, max_traj_len=T)[0] for i in range(N)]
    mo = ma = None
    if mirror:
        (os_, og), (as_, ag) = algo.spec.mirror_tables()
        mo, ma = (os_, og), (as_, ag)
    orc = po.OraclePPO([w0[f"a_{n}"] for n in NAMES], [w0[f"c_{n}"] for n in NAMES], w0["stds"], obs_mean, obs_std,
                       mirror_obs=mo, mirror_act=ma)
    o_obs = np.zeros_like(g_obs)
    o_rew, o_done = np.zeros((T, N), np.float32), np.zeros((T, N), np.uint8)
    o_tobs = np.zeros_like(g_obs)
    cur = np.array([e.reset() for e in envs])
    for t in range(T):
        o_obs[t] = cur
        for i, e in enumerate(envs):
            nxt, r, fl, tob, _ = e.step_auto(g_act[t, i])
            cur[i], o_rew[t, i], o_done[t, i], o_tobs[t, i] = nxt, r, fl, tob
    tol = dict(rtol=2e-4, atol=2e-4 if not env_name.startswith("h1") else 2e-3)   # float32 obs; h1 obs include torques of O(100)
    np.testing.assert_allclose(g_obs, o_obs, **tol)
    np.testing.assert_allclose(g_rew, o_rew, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(g_done, o_done)
    with torch.no_grad():
        T_ = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
        o_val = orc.value(T_(o_obs.reshape(T * N, -1))).numpy().reshape(T, N)
        o_vterm = orc.value(T_(o_tobs.reshape(T * N, -1))).numpy().reshape(T, N)
        o_vfinal = orc.value(T_(cur)).numpy().reshape(N)
        o_logp = orc.log_prob(T_(o_obs.reshape(T * N, -1)), T_(g_act.reshape(T * N, -1))).numpy().reshape(T, N)
    np.testing.assert_allclose(g_val, o_val, rtol=1e-3, atol=2e-4)
    used = ((g_done & 2) != 0) & ((g_done & 1) == 0)       # V(terminal obs) is evaluated where GAE bootstraps from it (truncations) ...
    o_vterm = np.where(used, o_vterm, 0.0)                  # ... and is 0 (unread) everywhere else
    np.testing.assert_allclose(g_vterm, o_vterm, rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(g_vfinal, o_vfinal, rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(g_logp, o_logp, rtol=1e-3, atol=2e-3)
    o_ret = po.gae_batch(g_rew, g_val, g_done, g_vterm, g_vfinal, 0.99, 0.95)     # GAE on identical inputs
    np.testing.assert_allclose(g_ret, o_ret, rtol=0, atol=1e-5)
    o_ret_chain = po.gae_batch(o_rew, o_val, o_done, o_vterm, o_vfinal, 0.99, 0.95)  # ... and along the whole oracle chain
    np.testing.assert_allclose(g_ret, o_ret_chain, rtol=2e-3, atol=2e-3)

    # ---- update: one minibatch = the whole batch, so the shuffle does not matter
    adv = T_(g_ret - g_val).reshape(-1, 1)
    adv = (adv - adv.mean()) / (adv.std() + args.eps)                               # ppo.py:484-485
    res = orc.update(T_(g_obs.reshape(T * N, -1)), T_(g_act.reshape(T * N, -1)), T_(g_ret.reshape(-1, 1)), adv, T_(g_logp.reshape(-1, 1)))
    algo.optimize(0)
    L = algo.last_losses
    np.testing.assert_allclose([L["actor"], L["critic"], L["mirror"]], [res[0], res[2], res[4]], rtol=2e-3, atol=2e-5)
    w1 = k.get_tensors()
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(w1[f"a_{n}"].numpy(), orc.actor[i].detach().numpy(), rtol=0, atol=5e-6, err_msg=f"actor {n}")
        np.testing.assert_allclose(w1[f"c_{n}"].numpy(), orc.critic[i].detach().numpy(), rtol=0, atol=5e-6, err_msg=f"critic {n}")


def test_two_trainings_same_seed_identical_weights():
    """reference tests/test_determinism.py: two same-seed 2-iteration trainings end with identical weights."""
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run(seed):
        args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=2,
                               max_traj_len=16, num_procs=64, num_envs=64, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                               recurrent=False, imitate=None, learn_std=False, std_dev=0.223, no_mirror=False, continued=None,
                               logdir="/tmp/lhw_test_det", device_index=0)
        algo = PPO(ENVIRONMENTS["jvrc_walk"], args, seed=seed)
        for itr in range(2):
            algo.iterate(itr)
        return algo.kernels.theta.clone()

    a, b, c = run(3), run(3), run(4)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)


@pytest.mark.parametrize("env_name", ["jvrc_walk", "h1"])
def test_grouped_rollout_is_bitwise_identical_to_single_stream(env_name, monkeypatch):
    """Rollout.collect with the batch split into independent env groups on separate HIP streams (lhw_env_step_range +
    lhw_ppo_forward_at) stores exactly the same observations, actions, log-probs, rewards, flags and values as the
    single-stream rollout: scheduling does not enter the RNG keys or the arithmetic."""
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run(groups):
        monkeypatch.setenv("LHW_ROLLOUT_MODE", "steps")      # (the resident rollout, tests/test_rollout_resident_gpu.py, has no groups)
        monkeypatch.setenv("LHW_ROLLOUT_GROUPS", str(groups))
        args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=1,
                               max_traj_len=12, num_procs=96, num_envs=96, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                               recurrent=False, imitate=None, learn_std=False, std_dev=0.4, no_mirror=True, continued=None,
                               logdir="/tmp/lhw_test_groups", device_index=0)
        algo = PPO(ENVIRONMENTS[env_name], args, seed=9)
        assert algo.rollout.groups == groups
        for _ in range(2):
            algo.sample_parallel_with_workers()
        ro = algo.rollout
        assert ro.last_mode == "steps"
        return [x.clone() for x in (ro.obs, ro.act, ro.logp, ro.rew, ro.done, ro.val, ro.vterm, ro.vfinal)]

    a, b, c = run(1), run(2), run(3)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.parametrize("variant", ["mirror", "learn_std", "fp16"])
def test_the_optimiser_step_as_one_graph_launch_is_bitwise_the_two_call_path(variant, monkeypatch):
    """lhw_ppo_step (round 6): lhw_ppo_grad + lhw_ppo_apply captured once as a hipGraph and replayed, the minibatch's index pointer and
    Adam's bias corrections patched into its kernel nodes per step -- the same kernels in the same order: after two iterations (2 epochs x
    2 minibatches each, so the graph is replayed with four different index pointers and eight Adam steps) the weights, the Adam state and
    the loss statistics equal those of the eager two-call path bit for bit.  Reference: rl/algos/ppo.py:387-396."""
    from types import SimpleNamespace
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run(graph):
        monkeypatch.setenv("LHW_PPO_GRAPH", "1" if graph else "0")
        args = SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.01 if variant == "learn_std" else 0.0, clip=0.2, minibatch_size=512,
                               epochs=2, max_traj_len=16, num_procs=64, num_envs=64, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                               recurrent=False, imitate=None, learn_std=variant == "learn_std", std_dev=0.223, no_mirror=variant != "mirror",
                               fp16=variant == "fp16", continued=None, logdir="/tmp/lhw_test_graph", device_index=0)
        algo = PPO(ENVIRONMENTS["jvrc_walk"], args, seed=11)
        losses = []
        for itr in range(2):
            algo.iterate(itr)
            losses.append(dict(algo.last_losses))
        k = algo.kernels
        return k.theta.clone(), k.adam_m.clone(), k.adam_v.clone(), losses, getattr(algo, "_update_stream", None) is not None

    ta, ma, va, la, used_a = run(True)
    tb, mb, vb, lb, used_b = run(False)
    assert used_a and not used_b
    assert torch.equal(ta, tb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert la == lb and la[0]["n_updates"] == 4
