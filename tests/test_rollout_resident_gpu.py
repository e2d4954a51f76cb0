"""The resident rollout on the GPU (lhw_env_rollout: all T control steps of a rollout in one launch, the actor evaluated inside the
stepper's wavefronts; csrc/lhw_humanoid_rollout.hip) against the launch-per-step pipeline (T x { lhw_ppo_forward_at ;
lhw_env_step_range }): every stored value BITWISE equal -- which also holds the in-wave v_fma_f32 policy step to the MFMA strip
kernel's bits.  Reference: the body of RolloutWorker.sample's loop, /root/reference/rl/workers/rollout_worker.py:142-181.
GPU twin of tests/test_rollout_resident.py (SIMT emulator)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(N, T, std=0.4):
    return SimpleNamespace(gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=512, epochs=1,
                           max_traj_len=T, num_procs=N, num_envs=N, max_grad_norm=0.5, mirror_coeff=0.4, eval_freq=10**9,
                           recurrent=False, imitate=None, learn_std=False, std_dev=std, no_mirror=True, continued=None,
                           logdir="/tmp/lhw_test_resident", device_index=0)


def _buffers(ro):
    return [x.clone() for x in (ro.obs, ro.act, ro.logp, ro.tob_all, ro.rew, ro.done, ro.val, ro.vterm, ro.vfinal)]


@pytest.mark.parametrize("env_name", ["jvrc_walk", "h1", "h1_walk", "jvrc_step"])
def test_resident_rollout_is_bitwise_the_launch_per_step_rollout(env_name, monkeypatch):
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run(mode):
        monkeypatch.setenv("LHW_ROLLOUT_MODE", mode)
        algo = PPO(ENVIRONMENTS[env_name], _args(97, 12), seed=9)      # odd batch: the last wavefront holds one env
        out = []
        for _ in range(3):                                            # episodes end (falls, truncation at 12) and carry over
            algo.sample_parallel_with_workers()
            assert algo.rollout.last_mode == mode
            out.append(_buffers(algo.rollout))
        q, v = algo.env.get_state()
        return out, q, v, algo.env.pop_fault_stats()

    (a, qa, va, fa), (b, qb, vb, fb) = run("steps"), run("resident")
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)
    np.testing.assert_array_equal(qa, qb)
    np.testing.assert_array_equal(va, vb)
    assert fa == fb == (0, 0)
    assert (a[0][5] != 0).any(), "no episode ended inside the rollouts"


def test_resident_rollout_repeats_overflowing_envs_inside_the_wave(monkeypatch):
    """envs lying on the floor with 10 .. 13 contacts (beyond the two-envs-per-wave layout's 8): a second launch in the
    launch-per-step pipeline, the same wavefront with its LDS re-interpreted in the resident rollout -- same bits, same count"""
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    poses = ([0.0, 0.0, 0.2571, -0.7309, -0.1371, 0.232, 0.627, -0.9859, -0.3243, -0.4729, 0.119, 0.609, 0.1118, -1.4146, 0.1458, 0.4932, 2.1903, 0.42, -0.5225],
             [0.0, 0.0, 0.1223, -0.4502, 0.0287, 0.3794, 0.8078, -1.2999, -0.2333, 0.4393, 0.4916, 0.2839, -0.8672, -1.533, 0.0192, -0.4235, 2.2823, -0.1645, -1.051],
             [0.0, 0.0, 0.142, -0.7055, -0.3568, 0.5444, 0.2804, 0.2372, -0.0299, -0.0106, 1.5965, -0.2912, -0.7387, -0.6059, -0.272, 0.0915, 0.445, -0.3003, 0.7158])

    def run(mode):
        monkeypatch.setenv("LHW_ROLLOUT_MODE", mode)
        algo = PPO(ENVIRONMENTS["jvrc_walk"], _args(16, 6, std=0.1), seed=2)
        env = algo.env
        algo.rollout.obs[algo.rollout.T].copy_(env.reset())      # (collect() continues from the last observation of the previous rollout)
        algo.rollout.started = True
        q, v = env.get_state()
        for i, pose in zip((0, 5, 15), poses):      # first / second env of a wavefront
            q[i] = pose
            q[i, 3:7] /= np.linalg.norm(q[i, 3:7])
            v[i] = 0
        env.set_state(q, v)
        env.pop_rerun_count()
        algo.rollout.collect()
        assert algo.rollout.last_mode == mode
        return _buffers(algo.rollout), env.get_state(), env.pop_rerun_count(), env.pop_fault_stats()

    (a, sa, ra, fa), (b, sb, rb, fb) = run("steps"), run("resident")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    np.testing.assert_array_equal(sa[0], sb[0])
    assert ra > 0 and ra == rb, (ra, rb)
    assert fa == fb == (0, 0)


def test_resident_rollout_through_the_job_queue_is_bitwise_the_same(monkeypatch):
    """Stepping task with more env groups than wave slots (jvrc_step @ 4096; forced here with LHW_ROLLOUT_SLOTS = 48): the resident
    waves pop (group, chunk of control steps) jobs from a device queue, a group's chunks run on whichever wave is free -- on other
    CUs and XCDs than the chunk before (the HBM record and the rollout rows travel under agent-scope fences).  Bitwise the
    one-wave-per-group rollout, twice in a row, resets and in-wave re-runs included."""
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    monkeypatch.setenv("LHW_ROLLOUT_MODE", "resident")
    env_name, N = "jvrc_step", 301

    def run(chunk):
        monkeypatch.setenv("LHW_ROLLOUT_CHUNK", str(chunk))
        algo = PPO(ENVIRONMENTS[env_name], _args(N, 14), seed=4)
        out = []
        for _ in range(2):
            algo.sample_parallel_with_workers()
            assert algo.rollout.last_mode == "resident"
            out.append(_buffers(algo.rollout))
        q, v = algo.env.get_state()
        return out, q, v, algo.env.pop_fault_stats(), algo.env.pop_episode_stats()

    monkeypatch.setenv("LHW_ROLLOUT_SLOTS", "48")
    (a, qa, va, fa, ea), (b, qb, vb, fb, eb), (c, qc, vc, fc, ec) = run(0), run(3), run(5)
    for other in (b, c):
        for ra, rb in zip(a, other):
            for x, y in zip(ra, rb):
                assert torch.equal(x, y)
    np.testing.assert_array_equal(qa, qb)
    np.testing.assert_array_equal(va, vc)
    assert fa == fb == fc == (0, 0) and ea == eb == ec
    assert (a[0][5] != 0).any()


def test_training_with_the_resident_rollout_ends_with_the_same_weights(monkeypatch):
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    def run(mode):
        monkeypatch.setenv("LHW_ROLLOUT_MODE", mode)
        a = _args(64, 16, std=0.223)
        a.no_mirror = False
        algo = PPO(ENVIRONMENTS["jvrc_walk"], a, seed=3)
        for itr in range(2):
            algo.iterate(itr)
        return algo.kernels.theta.clone()

    assert torch.equal(run("steps"), run("resident"))


def test_fp16_operand_policy_in_the_resident_rollout_matches_the_fp16_mfma_forward(monkeypatch):
    """BASELINE config 5 (fp16 actor): with fp16 inference selected the in-wave policy step rounds weights and activations to fp16 and
    accumulates in float32 -- what the launch-per-step path's fp16 MFMA GEMMs do, up to the order in which an MFMA adds its 16
    products: the means agree to float32 rounding, and the rollout runs resident."""
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO
    monkeypatch.setenv("LHW_ROLLOUT_MODE", "auto")
    a = _args(64, 8, std=0.3)
    a.infer_fp16 = True
    algo = PPO(ENVIRONMENTS["h1"], a, seed=4)
    algo.sample_parallel_with_workers(deterministic=True)
    ro = algo.rollout
    assert ro.last_mode == "resident"
    mu, _, _, _ = algo.kernels.forward(ro.obs[0], deterministic=True, want_value=False)       # fp16-operand MFMA GEMMs
    assert torch.isfinite(ro.act).all()
    np.testing.assert_allclose(ro.act[0].cpu().numpy(), mu.cpu().numpy(), rtol=0, atol=2e-5)
    algo.kernels.set_inference_fp16(False)
    mu32, _, _, _ = algo.kernels.forward(ro.obs[0], deterministic=True, want_value=False)
    assert not torch.equal(ro.act[0], mu32)               # and they are not the float32 means


def test_resident_rollout_against_the_oracle_directly(monkeypatch):
    """The resident rollout held to the float64 ORACLE without the launch-per-step pipeline in between: 288 jvrc_walk envs, one
    launch of T = 48 control steps with the trained-size actor sampling in the wavefronts, episodes truncated at 12 (so every env
    is reset three times inside the launch), three envs started lying on the floor (the in-wave one-env-per-wave re-run).  The oracle
    envs replay the actions the launch stored (an action tape: the policy's float32 arithmetic is not the subject here -- its bits
    are held to the strip kernel by the tests above) and must reproduce every stored observation (2e-5: float32 rows), reward (2e-6),
    flag (exactly) and the final state (1e-9 / 1e-7 after 1200 free-running sub-steps with resets).  Reference:
    rl/workers/rollout_worker.py:142-181 over robots/robot_base.py:64-98.  Physics parity is UNPINNED against MuJoCo."""
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from learninghumanoidwalking_amd.ppo import PPO
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    monkeypatch.setenv("LHW_ROLLOUT_MODE", "resident")
    N, T, L = 288, 48, 12
    a = _args(N, T, std=0.223)
    a.max_traj_len = L
    algo = PPO(ENVIRONMENTS["jvrc_walk"], a, seed=6)
    env, spec = algo.env, JvrcWalkSpec()
    # a rollout of T control steps with episodes of L: Rollout is built with T = max_traj_len, so lhw_env_rollout is driven directly
    obs0 = env.reset()
    orc = [OracleJvrcWalkEnv(spec, seed=algo.env_seed, env_id=i, max_traj_len=L) for i in range(N)]
    ref0 = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs0.cpu().numpy(), ref0, rtol=1e-6, atol=2e-6)
    poses = ([0.0, 0.0, 0.2571, -0.7309, -0.1371, 0.232, 0.627, -0.9859, -0.3243, -0.4729, 0.119, 0.609, 0.1118, -1.4146, 0.1458, 0.4932, 2.1903, 0.42, -0.5225],
             [0.0, 0.0, 0.1223, -0.4502, 0.0287, 0.3794, 0.8078, -1.2999, -0.2333, 0.4393, 0.4916, 0.2839, -0.8672, -1.533, 0.0192, -0.4235, 2.2823, -0.1645, -1.051],
             [0.0, 0.0, 0.142, -0.7055, -0.3568, 0.5444, 0.2804, 0.2372, -0.0299, -0.0106, 1.5965, -0.2912, -0.7387, -0.6059, -0.272, 0.0915, 0.445, -0.3003, 0.7158])
    q, v = env.get_state()
    for i, pose in zip((0, 5, 287), poses):
        q[i] = pose
        q[i, 3:7] /= np.linalg.norm(q[i, 3:7])
        v[i] = 0
    env.set_state(q, v)
    for i, o in enumerate(orc):      # (set_state re-runs the forward pass: on both sides, for every env)
        o.set_state(q[i], v[i])
    env.pop_rerun_count()
    dev = obs0.device
    D, A = env.obs_dim, env.act_dim
    obs = torch.zeros(T + 1, N, D, device=dev); act = torch.zeros(T, N, A, device=dev); logp = torch.zeros(T, N, device=dev)
    tob = torch.zeros(T, N, D, device=dev); rew = torch.zeros(T, N, device=dev); done = torch.zeros(T, N, dtype=torch.uint8, device=dev)
    obs[0].copy_(obs0)
    k = algo.kernels
    k.begin_rollout()
    try:
        pol = k.rollout_policy(seed=123, counter=0, deterministic=False)
        assert pol is not None and env.rollout(pol, T, obs, act, logp, tob, rew, done)
    finally:
        k.end_rollout()
    torch.cuda.synchronize()
    A_, O_, R_, F_, TB_ = (x.cpu().numpy() for x in (act, obs, rew, done, tob))
    for t in range(T):
        res = [o.step_auto(A_[t, i]) for i, o in enumerate(orc)]
        np.testing.assert_array_equal(F_[t], np.array([r[2] for r in res], dtype=np.uint8), err_msg=f"flags t={t}")
        np.testing.assert_allclose(R_[t], np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
        np.testing.assert_allclose(O_[t + 1], np.array([r[0] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"obs t={t}")
        np.testing.assert_allclose(TB_[t], np.array([r[3] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"terminal obs t={t}")
    qg, vg = env.get_state()
    np.testing.assert_allclose(qg, np.array([o.sim.qpos for o in orc]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(vg, np.array([o.sim.qvel for o in orc]), rtol=0, atol=1e-7)
    assert (F_ & 2).sum() >= 3 * N * 0.8 and (F_ & 1).any(), "truncations / falls missing"
    assert env.pop_rerun_count() > 0, "no env took the in-wave one-env-per-wave re-run"
    assert env.pop_fault_stats() == (0, 0)
