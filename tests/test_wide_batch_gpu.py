"""Oracle parity on wide batches.  The tape tests (tests/test_jvrc_gpu.py, test_h1_gpu.py, test_jvrc_step_gpu.py) follow a handful
of envs for thousands of sub-steps; this one follows hundreds of envs for a few control steps, so that many more initial states,
walking modes, terrains and randomised dynamics meet the kernel -- and so that envs sit in every position of a wavefront (first /
second half, next to a neighbour that needs the one-env-per-wave re-run, last wave of a partly filled launch).  Free running, no
resynchronisation, episodes of four control steps so that every env is reset once inside the run (the reset draws of hundreds of
env ids): positions within 1e-11, velocities within 1e-9, flags exact.
Physics parity is UNPINNED against MuJoCo (see tests/test_jvrc_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _classes(name):
    if name == "jvrc_walk":
        from learninghumanoidwalking_amd.envs.jvrc_walk import JvrcWalkSpec as S
        from oracle.env_jvrc_walk import OracleJvrcWalkEnv as O
    elif name == "jvrc_step":
        from learninghumanoidwalking_amd.envs.jvrc_step import JvrcStepSpec as S
        from oracle.env_jvrc_step import OracleJvrcStepEnv as O
    elif name == "h1":
        from learninghumanoidwalking_amd.envs.h1 import H1Spec as S
        from oracle.env_h1 import OracleH1Env as O
    else:
        from learninghumanoidwalking_amd.envs.h1_walk import H1WalkSpec as S
        from oracle.env_h1_walk import OracleH1WalkEnv as O
    return S, O


@pytest.mark.parametrize("name,N,scale", [("jvrc_walk", 509, 0.223), ("h1", 387, 0.05), ("h1_walk", 250, 0.1), ("jvrc_step", 131, 0.223)])
def test_hundreds_of_envs_free_running_match_oracle(name, N, scale):
    """(odd batch sizes: the last wavefront of the launch is half empty)"""
    import torch
    S, O = _classes(name)
    spec = S()
    T = 6
    env = spec.make_batched(N, seed=77, device=0, max_traj_len=4)
    orc = [O(spec, seed=77, env_id=i, max_traj_len=4) for i in range(N)]
    obs = env.reset().cpu().numpy()
    ref = np.array([o.reset() for o in orc])
    np.testing.assert_allclose(obs, ref, rtol=1e-6, atol=2e-6)
    tape = (np.random.default_rng(N).normal(size=(T, N, spec.act_dim)) * scale).astype(np.float32)
    max_ncon = n_trunc = 0
    for t in range(T):
        o_dev, rew, done, tob = env.step(torch.from_numpy(tape[t]).cuda())
        max_ncon = max(max_ncon, max(o.sim.ncon for o in orc))
        res = [o.step_auto(tape[t, i]) for i, o in enumerate(orc)]
        q, v = env.get_state()
        oq = np.array([o.sim.qpos.copy() for o in orc])
        ov = np.array([o.sim.qvel.copy() for o in orc])
        flags = np.array([r[2] for r in res], dtype=np.uint8)
        np.testing.assert_array_equal(done.cpu().numpy(), flags, err_msg=f"flags t={t}")
        np.testing.assert_allclose(q, oq, rtol=0, atol=1e-11, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(v, ov, rtol=0, atol=1e-9, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(o_dev.cpu().numpy(), np.array([r[0] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"obs t={t}")
        np.testing.assert_allclose(tob.cpu().numpy(), np.array([r[3] for r in res]), rtol=1e-5, atol=2e-5, err_msg=f"terminal obs t={t}")
        np.testing.assert_allclose(rew.cpu().numpy(), np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
        n_trunc += int((flags & 2 != 0).sum())
    assert n_trunc >= N * 0.9
    assert env.pop_fault_stats() == (0, 0)
    if name == "jvrc_step":
        assert max_ncon > 8, "no env needed the one-env-per-wave re-run"
    env.close()
