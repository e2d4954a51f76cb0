"""HIP cartpole stepper vs the float64 CPU oracle, through the C ABI (GPU parity tests)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(n, seed, max_traj_len=0):
    import torch
    from learninghumanoidwalking_amd.envs import make_cartpole
    assert torch.cuda.is_available()
    return make_cartpole(n, seed=seed, device=0, max_traj_len=max_traj_len)


def _oracle(n, seed, max_traj_len=0):
    from learninghumanoidwalking_amd.envs import CartpoleSpec
    from oracle.env_cartpole import OracleCartpoleEnv
    spec = CartpoleSpec()
    m = spec.model()
    return [OracleCartpoleEnv(m, seed=seed, env_id=i, kp=spec.kp, kd=spec.kd, frame_skip=spec.frame_skip,
                              max_traj_len=max_traj_len) for i in range(n)]


def test_reset_matches_oracle():
    import torch
    env = _mk(8, seed=7)
    obs = env.reset().cpu().numpy()
    orc = _oracle(8, seed=7)
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref.astype(np.float32), rtol=0, atol=1e-7)
    q, v = env.get_state()
    np.testing.assert_allclose(q, np.array([o.sim.qpos for o in orc]), rtol=0, atol=1e-14)
    np.testing.assert_allclose(v, np.array([o.sim.qvel for o in orc]), rtol=0, atol=1e-14)


def _run_tape(tape, seed, resync_every=0, check_every=25, atol=1e-9):
    """Drive HIP and oracle with the same action tape; returns max |dq|,|dv| seen at the check points."""
    import torch
    T, N = tape.shape
    env = _mk(N, seed=seed)
    orc = _oracle(N, seed=seed)
    env.reset()
    for o in orc:
        o.reset()
    hits, worst = 0, 0.0
    for t in range(T):
        act = torch.from_numpy(tape[t].reshape(N, 1)).cuda()
        obs, rew, done, _ = env.step(act)
        res = [o.step(tape[t, i]) for i, o in enumerate(orc)]
        hits += sum(o.sim.nefc > 0 for o in orc)
        if t % check_every == check_every - 1 or t == T - 1:
            q, v = env.get_state()
            oq, ov = np.array([o.sim.qpos for o in orc]), np.array([o.sim.qvel for o in orc])
            worst = max(worst, np.abs(q - oq).max(), np.abs(v - ov).max())
            np.testing.assert_allclose(q, oq, rtol=0, atol=atol, err_msg=f"qpos t={t}")
            np.testing.assert_allclose(v, ov, rtol=0, atol=atol, err_msg=f"qvel t={t}")
            np.testing.assert_allclose(obs.cpu().numpy(), np.array([r[0] for r in res]), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(rew.cpu().numpy(), np.array([r[1] for r in res]), rtol=0, atol=1e-6)
            np.testing.assert_array_equal(done.cpu().numpy() & 1, np.array([int(r[2]) for r in res]))
        if resync_every and t % resync_every == resync_every - 1:
            # put the HIP state back on the oracle trajectory (set_state hook == MujocoEnv.set_state)
            oq, ov = np.array([o.sim.qpos for o in orc]), np.array([o.sim.qvel for o in orc])
            env.set_state(oq, ov)
            for o in orc:
                o.set_state(o.sim.qpos.copy(), o.sim.qvel.copy())
    return worst, hits


def test_random_tape_1000_steps_resynchronised():
    """Action tape a[t,n] ~ U(-1,1), seed 1234, 1000 control steps.

    A whirling pole under random pushes is chaotic (measured error growth ~x1000 per 100 control
    steps, identical at solver tolerance 1e-8 and 1e-14: scripts/cartpole_divergence.py), so a
    free-running 1000-step comparison only measures the Lyapunov exponent.  The size-independent
    property checked instead: over all 1000 steps, 25-step segments started from the oracle's state
    stay within 1e-10 (float64 both sides, different operation order)."""
    tape = np.random.default_rng(1234).uniform(-1, 1, size=(1000, 6)).astype(np.float32)
    _run_tape(tape, seed=3, resync_every=25, check_every=25, atol=1e-10)


def test_joint_limit_row_parity():
    """Carts thrown at both slider limits: soft limit row (impedance, aref, Newton) vs oracle."""
    import torch
    N = 8
    env = _mk(N, seed=1)
    orc = _oracle(N, seed=1)
    env.reset()
    rs = np.random.default_rng(5)
    q = np.stack([np.where(np.arange(N) % 2 == 0, 0.96, -0.96), rs.uniform(-3, 3, N)], axis=1)
    v = np.stack([np.where(np.arange(N) % 2 == 0, 1.0, -1.0) * rs.uniform(2, 6, N), rs.uniform(-5, 5, N)], axis=1)
    env.set_state(q, v)
    for i, o in enumerate(orc):
        o.sim.reset_data()
        o.set_state(q[i], v[i])
    hits = 0
    for t in range(12):
        a = np.where(np.arange(N) % 2 == 0, 0.8, -0.8).astype(np.float32)
        env.step(torch.from_numpy(a.reshape(N, 1)).cuda())
        for i, o in enumerate(orc):
            for _ in range(1):
                o.step(a[i])
            hits += o.sim.nefc > 0
        gq, gv = env.get_state()
        np.testing.assert_allclose(gq, np.array([o.sim.qpos for o in orc]), rtol=0, atol=1e-11, err_msg=f"t={t}")
        np.testing.assert_allclose(gv, np.array([o.sim.qvel for o in orc]), rtol=0, atol=1e-10, err_msg=f"t={t}")
    assert hits > 0, "limit row never active"
    assert np.all(np.abs(gq[:, 0]) < 1.2)


def test_random_tape_free_running_100_steps():
    tape = np.random.default_rng(1234).uniform(-1, 1, size=(100, 6)).astype(np.float32)
    _run_tape(tape, seed=3, check_every=10, atol=1e-9)


def test_smooth_tape_free_running_1000_steps():
    """Non-chaotic regime (gentle sinusoidal targets): free-running 1000 steps within 1e-9."""
    t = np.arange(1000)[:, None]
    tape = (0.3 * np.sin(0.05 * t + np.arange(6)[None, :])).astype(np.float32)
    _run_tape(tape, seed=5, check_every=50, atol=1e-9)


def test_auto_reset_and_truncation_bit_exact_flags():
    """With max_traj_len the step fuses RolloutWorker's bookkeeping: flags/reset timing must be identical."""
    import torch
    N, T, L = 16, 300, 37
    env = _mk(N, seed=11, max_traj_len=L)
    orc = _oracle(N, seed=11, max_traj_len=L)
    env.reset()
    for o in orc:
        o.reset()
    tape = np.random.default_rng(99).uniform(-2, 2, size=(T, N)).astype(np.float32)
    n_term = 0
    for t in range(T):
        obs, rew, done, tob = env.step(torch.from_numpy(tape[t].reshape(N, 1)).cuda())
        res = [o.step_auto(tape[t, i]) for i, o in enumerate(orc)]
        flags = np.array([r[2] for r in res], dtype=np.uint8)
        np.testing.assert_array_equal(done.cpu().numpy(), flags, err_msg=f"flags t={t}")
        np.testing.assert_allclose(obs.cpu().numpy(), np.array([r[0] for r in res]), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tob.cpu().numpy(), np.array([r[3] for r in res]), rtol=1e-6, atol=1e-6)
        n_term += int((flags & 2).sum() // 2)
    assert n_term >= N * (T // L)
    ret, length, count = env.pop_episode_stats()
    assert count >= N * (T // L)
    assert length > 0 and np.isfinite(ret)


def test_reward_terms_sum():
    import torch
    env = _mk(32, seed=5)
    env.reset()
    act = torch.rand(32, 1, device="cuda") * 2 - 1
    obs, rew, done, _ = env.step(act)
    assert abs(float((env.rew_terms.sum(1) - rew).abs().max())) < 1e-6  # reference tests/test_environments.py:174-188
