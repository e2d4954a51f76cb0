"""GPU twin of tests/test_fuse_static.py: the synthetic full-body-count JVRC model, folded (`Model.fuse_static`), in the HIP stepper
through the C ABI, against the float64 oracle on the UNFOLDED model."""
import numpy as np
import pytest
import torch

from tests.test_fuse_static import _BigSpec, _BigSpecUnfused

pytestmark = pytest.mark.gpu


def test_unfolded_model_is_refused_with_the_remedy_named():
    with pytest.raises(RuntimeError, match="fuse_static"):      # (LhwError is a RuntimeError)
        _BigSpecUnfused().make_batched(2, seed=0, device=0)


def test_folded_model_on_the_gpu_matches_the_oracle_on_the_unfolded_model():
    from oracle.env_jvrc_walk import OracleJvrcWalkEnv
    N = 6
    env = _BigSpec().make_batched(N, seed=4, device=0)
    orc = [OracleJvrcWalkEnv(_BigSpecUnfused(), seed=4, env_id=i) for i in range(N)]
    obs = env.reset().cpu().numpy()
    np.testing.assert_allclose(obs, np.array([o.reset() for o in orc]), rtol=1e-6, atol=1e-6)
    tape = (np.random.default_rng(7).normal(size=(5, N, 12)) * 0.223).astype(np.float32)
    for t in range(tape.shape[0]):
        obs, rew, done, _ = env.step(torch.from_numpy(tape[t]).cuda())
        res = [o.step(tape[t, i]) for i, o in enumerate(orc)]
        q, v = env.get_state()
        np.testing.assert_allclose(q, np.array([o.sim.qpos for o in orc]), rtol=0, atol=1e-10, err_msg=f"qpos t={t}")
        np.testing.assert_allclose(v, np.array([o.sim.qvel for o in orc]), rtol=0, atol=1e-8, err_msg=f"qvel t={t}")
        np.testing.assert_allclose(rew.cpu().numpy(), np.array([r[1] for r in res]), rtol=0, atol=2e-6, err_msg=f"rew t={t}")
    # fallen onto the right side: the hand sphere -- a geom of a folded link -- carries the contact
    q0 = np.array([o.sim.qpos.copy() for o in orc])
    q0[:, 2] = 0.25
    q0[:, 3:7] = [np.cos(0.9), np.sin(0.9), 0, 0]
    env.set_state(q0, np.zeros((N, 18)))
    for o, qq in zip(orc, q0):
        o.set_state(qq, np.zeros(18))
    zero = torch.zeros(N, 12).cuda()
    for t in range(3):
        env.step(zero)
        for o in orc:
            o.step(np.zeros(12, np.float32))
        q, v = env.get_state()
        np.testing.assert_allclose(q, np.array([o.sim.qpos for o in orc]), rtol=0, atol=1e-9, err_msg=f"fallen qpos t={t}")
    over, div = env.pop_fault_stats()
    assert over == 0 and div == 0
